#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz.

These vectors are produced by the CPU oracle (oracle/panda_oracle.c), NOT by a live reference
build: the reference cannot be built in this image (see DESIGN.md "Oracle"), so they pin the
oracle against regressions and give the GPU tests a second, committed set of expected outputs.
The one reference-derived known answer is checked in test_oracle_cpu.py.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po  # noqa: E402
from pandaseq_amd import synth  # noqa: E402
from util import adversarial_pairs  # noqa: E402

CASES = {
    # name: (config index, n, oracle kwargs)
    "sb_2x250": (1, 600, dict(algo="simple_bayesian")),
    "pear_2x250": (1, 600, dict(algo="pear")),
    "rdp_2x250_modules": (1, 600, dict(algo="rdp_mle", modules=[("min_phred", 2), ("min_overlapbits", 15)])),
    "sb_2x150_novaseq": (4, 600, dict(algo="simple_bayesian")),
    "pear_2x300_primers": (2, 400, dict(algo="pear", primers=True)),
    "sb_2x150_ov40": (0, 400, dict(algo="simple_bayesian")),
}


def checksum(seq_nt, seq_p, res):
    """position-weighted checksums of every accepted pair's merged bases and log-probabilities"""
    n = len(res)
    cs_nt = np.zeros(n, dtype=np.uint64)
    cs_p = np.zeros(n, dtype=np.float64)
    for i in range(n):
        if res["status"][i] == 0:
            m = int(res["seq_len"][i])
            w = np.arange(1, m + 1, dtype=np.uint64)
            cs_nt[i] = (seq_nt[i, :m].astype(np.uint64) * w).sum()
            cs_p[i] = float((seq_p[i, :m] * np.arange(1, m + 1)).sum())
    return cs_nt, cs_p


def build_case(name):
    cfg_index, n, kw = CASES[name]
    kw = dict(kw)
    spec = synth.config_spec(cfg_index)
    primers = kw.pop("primers", False)
    if name.startswith("sb_2x250") or name.startswith("pear_2x250") or name.startswith("rdp"):
        fwd, rev, flen, rlen = adversarial_pairs(spec, n, seed=len(name))
        if kw["algo"] == "pear":
            rlen = np.minimum(rlen, flen)
    else:
        fwd, rev, flen, rlen = synth.generate_numpy(spec, 0, n)
    if primers:
        kw["fprimer"] = synth.nt_codes(spec.fprimer)
        kw["rprimer"] = synth.nt_codes(spec.rprimer, complement=True)
    algo = kw.pop("algo")
    res, nt, p, ctr = po.assemble_batch(po.make_config(algo, **kw), fwd, rev, flen, rlen)
    cs_nt, cs_p = checksum(nt, p, res)
    return dict(fwd=fwd, rev=rev, flen=flen, rlen=rlen, res=res, cs_nt=cs_nt, cs_p=cs_p,
                ctr_main=np.array([ctr[k] for k in ("count", "badr", "nofp", "norp", "noalgn", "lowq", "ok", "slow")], dtype=np.int64),
                ctr_rejected=np.asarray(ctr["rejected"]), seq_nt_head=nt[:50], seq_p_head=p[:50])


if __name__ == "__main__":
    out = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        d = build_case(name)
        np.savez_compressed(os.path.join(out, name + ".npz"), **d)
        print(name, {k: int(v) for k, v in zip(("count", "badr", "nofp", "norp", "noalgn", "lowq", "ok", "slow"), d["ctr_main"])})
