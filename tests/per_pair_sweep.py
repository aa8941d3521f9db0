#!/usr/bin/env python3
"""A randomized campaign on the per-pair entry point (GPU only; tests/test_sweeps_gpu.py runs a fixed-seed slice of it under
pytest -m gpu): random read lengths, scorers and options -- primers before or after assembly or with a penalty, one of the two
fused modules, overlap bounds, a threshold -- through tests/c/test_per_pair.c three ways: panda_assembler_assemble() as shipped
(the host path), pamd_compat_assemble_many() (the kernels) and, on the first pairs, panda_assembler_assemble() forced onto the
device.  The three outputs must be the same bytes, and the oracle's records.

    python tests/per_pair_sweep.py [rounds [first round]]        # ~3 s per round
"""
import os
import pathlib
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import per_pair as pp
from oracle import pyoracle as po
from pandaseq_amd import synth
from util import adversarial_pairs


def run(rounds=10, first=0, log=print):
    """-> (mismatching rounds, rounds compared)"""
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        client = pp.build_client(tmp)
        tmp = pathlib.Path(tmp)
        for rnd in range(first, first + rounds):
            rng = np.random.default_rng(17000 + rnd)
            primers = rnd % 3 == 0
            algo = pp.ALGOS[rnd % 7] if not primers else ["pear", "simple_bayesian", "rdp_mle"][rnd % 3]
            options, kw, mods = [], {}, []
            n = int(rng.integers(50, 1200))
            if primers:
                spec = synth.config_spec(2)
                fwd, rev, flen, rlen = synth.generate_host(spec, int(rng.integers(0, 100_000)), n)
                fp, rp = synth.nt_codes(spec.fprimer), synth.nt_codes(spec.rprimer, complement=True)
                options += ["-p", pp.primer_arg(fp), "-q", pp.primer_arg(rp)]
                kw.update(fprimer=fp, rprimer=rp)
                how = int(rng.integers(0, 3))
                if how == 1:
                    options.append("-a")
                    kw["primers_after"] = True
                elif how == 2:
                    pen = float(rng.choice([0.001, 0.01]))
                    options += ["-T", repr(pen)]
                    kw["primer_penalty"] = pen
            else:
                L = int(rng.choice([60, 100, 150, 192, 250, 256, 300, 320]))
                spec = synth.SynthSpec(0x5EED3000 + rnd, L, max(12, L // 2), 2 * L - 10, qhi=int(rng.integers(30, 42)))
                if rnd % 2:
                    fwd, rev, flen, rlen = adversarial_pairs(spec, n, seed=int(rng.integers(1, 1 << 30)))
                else:
                    fwd, rev, flen, rlen = synth.generate_host(spec, int(rng.integers(0, 100_000)), n)
            if algo == "pear":
                rlen = np.minimum(rlen, flen)
            if rng.random() < 0.4:
                v = int(rng.integers(2, 14))
                options += ["-o", str(v)]
                kw["minoverlap"] = v
            if rng.random() < 0.25:
                v = int(rng.integers(60, 260))
                options += ["-O", str(v)]
                kw["maxoverlap"] = v
            if rng.random() < 0.4:
                t = float(rng.choice([0.4, 0.7, 0.9]))
                options += ["-t", repr(t)]
                kw["threshold"] = t
            kind = int(rng.integers(0, 3))
            if kind == 1:
                v = int(rng.integers(1, 12))
                mods, options = [("min_phred", v)], options + ["-m", "min_phred:%d" % v]
            elif kind == 2:
                v = int(rng.integers(2, 40))
                mods, options = [("min_overlapbits", float(v))], options + ["-m", "min_overlapbits:%d" % v]
            what = f"round {rnd}: n {n} {algo} {' '.join(o if len(o) < 24 else o[:8] + '..' for o in options)}"
            try:
                pairs = tmp / "pairs.bin"
                pp.write_pairs(pairs, fwd, rev, flen, rlen)
                env = dict(os.environ, PAMD_COMPAT_SUMS="exact")
                host = pp.run_client(client, pairs, tmp / "host.bin", algo, "pair", options, env=env)
                pp.run_client(client, pairs, tmp / "many.bin", algo, "many", options, env=env)
                assert (tmp / "host.bin").read_bytes() == (tmp / "many.bin").read_bytes(), "per-pair host path and the kernels differ"
                o = po.assemble_batch(po.make_config(algo, modules=mods, **kw), fwd, rev, flen, rlen)
                pp.compare_with_oracle(*host, *o, modules=bool(mods))
                m = min(150, n)
                small = tmp / "small.bin"
                pp.write_pairs(small, fwd[:m], rev[:m], flen[:m], rlen[:m])
                pp.run_client(client, small, tmp / "dev.bin", algo, "pair", options, env=dict(env, PAMD_COMPAT_PER_PAIR="device"))
                pp.run_client(client, small, tmp / "host_small.bin", algo, "pair", options, env=env)
                assert (tmp / "dev.bin").read_bytes() == (tmp / "host_small.bin").read_bytes(), "per-pair host path and per-pair device path differ"
                log(what + ": ok %d" % int(o[3]["ok"]))
            except AssertionError as e:
                bad += 1
                log("MISMATCH " + what + ": " + str(e)[:300])
    return bad, rounds


if __name__ == "__main__":
    bad, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("mismatching rounds:", bad)
    sys.exit(1 if bad else 0)
