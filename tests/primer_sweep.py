#!/usr/bin/env python3
"""A randomized campaign on the lean kernel's primer scan (development aid, GPU only; not collected by pytest): primers of
random length and content planted at the start of the read, anywhere or nowhere, damaged at random rates and on random
shares of poor qualities -- the reads the shortcut settles, the ones left to the list fallback, the ones deferred -- against
the oracle on every record, row and counter.

    python tests/primer_sweep.py [rounds]        # ~6 s per round on one MI355X + the oracle on the host's cores
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import pandaseq_amd as pa
from pandaseq_amd import synth
from test_lean_long_gpu import _oracle, _plant, _run
from util import compare_counters, compare_results

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
codes = np.array([1, 2, 4, 8, 1, 2, 4, 8, 3, 5, 10, 12, 7, 14, 15], np.uint8)
bad = 0
for rnd in range(rounds):
    rng = np.random.default_rng(7000 + rnd)
    spec = synth.config_spec(2 if rnd % 3 else 1)
    L = spec.read_len
    n = 3000
    fwd, rev, flen, rlen = synth.generate_numpy(spec, int(rng.integers(0, 40_000_000)), n)
    fwd[:, :, 0] = np.where(np.arange(fwd.shape[1])[None, :] < L, fwd[:, :, 0], 0)
    fplen, rplen = int(rng.integers(1, 25)), int(rng.integers(1, 25))
    fp = rng.choice(codes if rnd % 2 else codes[:4], fplen)
    rp = rng.choice(codes[:4], rplen)
    anywhere = rng.choice([0.0, 0.1, 0.4, 1.0])
    wf = np.where(rng.random(n) < 0.9, rng.integers(0, max(1, L - fplen), n) * (rng.random(n) < anywhere), -1)
    wr = np.where(rng.random(n) < 0.9, rng.integers(0, max(1, L - rplen), n) * (rng.random(n) < anywhere), -1)
    subst, lowq = rng.choice([0.0, 0.03, 0.08, 0.2]), rng.choice([0.0, 0.1, 0.3, 0.7])
    _plant(rng, fwd, flen, fp, wf, subst, lowq)
    _plant(rng, rev, rlen, rp, wr, subst, lowq)
    if rnd % 4 == 3:    # ragged lengths: the scan's last alignments, reads shorter than the primer
        flen[:] = rng.integers(1, L + 1, n)
        rlen[:] = rng.integers(1, L + 1, n)
    algo = ["pear", "simple_bayesian", "rdp_mle"][rnd % 3]
    if algo == "pear":
        rlen = np.minimum(rlen, flen)
    thr = float(rng.choice([0.6, 0.6, 0.9, 0.3]))
    o = _oracle(algo, fwd, rev, flen, rlen, fprimer=fp, rprimer=rp, threshold=thr)
    for exact in (False, True):
        asm = pa.Assembler(0)
        asm.set_algorithm(algo)
        asm.set_exact_sums(exact)
        asm.set_threshold(thr)
        asm.set_forward_primer(fp)
        asm.set_reverse_primer(rp)
        b = _run(asm, fwd, rev, flen, rlen)
        try:
            compare_results(b.results_numpy(), b.seq_nt.cpu().numpy(), b.seq_p.cpu().numpy(), *o[:3], exact_sums=exact)
            compare_counters(b.counters_dict(), o[3])
        except AssertionError as e:
            bad += 1
            print(f"MISMATCH round {rnd} algo {algo} exact {exact} plen {fplen}/{rplen} subst {subst} lowq {lowq}: {str(e)[:300]}", flush=True)
    st = o[0]["status"]
    print(f"round {rnd}: {algo} plen {fplen}/{rplen} anywhere {anywhere} subst {subst} lowq {lowq} thr {thr}: ok {(st == 0).sum()} nofp {(st == 2).sum()} norp {(st == 3).sum()} deferred {asm.last_deferred()}", flush=True)
print("mismatching launches:", bad)
sys.exit(1 if bad else 0)
