#!/usr/bin/env python3
"""A randomized campaign through the drop-in API (development aid, GPU only; not collected by pytest): tests/c/test_compat.c --
the reference's own `while ((r = panda_assembler_next(a)) != NULL)` loop in plain C -- over adversarial and synthetic pairs
of random read lengths, every scorer, plain or with one of the two fused filters at a random threshold, both sum modes
(PAMD_COMPAT_SUMS); every record it returns (scalars, every base, every log-p) and the module's rejection count against the
oracle.

    python tests/compat_sweep.py [rounds [first round]]        # ~4 s per round on one MI355X + the oracle on the host's cores
"""
import os
import pathlib
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from oracle import pyoracle as po
from pandaseq_amd import synth
from test_compat_gpu import _build_client, _parse_output, _write_pairs
from util import adversarial_pairs

ALGOS = ["simple_bayesian", "pear", "rdp_mle", "uparse", "flash", "ea_util", "stitch"]

def run(rounds=10, first=0, log=print):
    """-> (mismatching runs of the C client, runs compared)"""
    bad = 0
    done = 0
    with tempfile.TemporaryDirectory() as tmp:
        client = _build_client(tmp)
        tmp = pathlib.Path(tmp)
        for rnd in range(first, first + rounds):
            rng = np.random.default_rng(13000 + rnd)
            L = int(rng.choice([100, 150, 192, 250, 256, 300, 320]))
            spec = synth.SynthSpec(0x5EED2000 + rnd, L, max(12, L // 2), 2 * L - 10, qhi=int(rng.integers(30, 42)))
            n = int(rng.integers(1, 3000))
            if rnd % 2:
                fwd, rev, flen, rlen = adversarial_pairs(spec, n, seed=int(rng.integers(1, 1 << 30)))
            else:
                fwd, rev, flen, rlen = synth.generate_host(spec, int(rng.integers(0, 1_000_000)), n)
            algo = ALGOS[rnd % 7]
            if algo == "pear":
                rlen = np.minimum(rlen, flen)
            mods, spec_str = [], None
            kind = rnd % 3
            if kind == 1:
                v = int(rng.integers(1, 12))
                mods, spec_str = [("min_phred", v)], "min_phred:%d" % v
            elif kind == 2:
                v = int(rng.integers(2, 40))
                mods, spec_str = [("min_overlapbits", float(v))], "min_overlapbits:%d" % v
            _write_pairs(tmp / "pairs.bin", fwd, rev, flen, rlen)
            o_res, o_nt, o_p, o_ctr = po.assemble_batch(po.make_config(algo, modules=mods), fwd, rev, flen, rlen)
            expected = [i for i in range(n) if flen[i] >= 2 and rlen[i] >= 2 and o_res["status"][i] == 0]
            for sums in ("exact", "fast"):
                env = dict(os.environ, PAMD_COMPAT_SUMS=sums)
                args = [client, str(tmp / "pairs.bin"), str(tmp / "out.bin"), algo] + (["0"] if spec_str is None else ["3", spec_str])
                try:
                    done += 1
                    subprocess.check_call(args, timeout=300, env=env)
                    got, counters, _ = _parse_output((tmp / "out.bin").read_bytes())
                    assert [g["index"] for g in got] == expected, "which pairs came back"
                    if mods:
                        assert counters[8] == o_ctr["rejected"][0], "the module's rejections"
                    for g in got:
                        i = g["index"]
                        m = g["seq_len"]
                        assert m == o_res["seq_len"][i] and g["overlap"] == o_res["overlap"][i] and g["mism"] == o_res["mismatches"][i] and g["examined"] == o_res["examined"][i], ("scalars", i)
                        assert np.array_equal(g["nt"], o_nt[i, :m]) and np.array_equal(g["p"], o_p[i, :m]), ("row", i)
                        if sums == "exact":
                            assert g["quality"] == o_res["quality"][i], ("quality", i)
                            if algo not in ("ea_util", "stitch"):
                                assert g["est"] == o_res["est_prob"][i], ("est", i)
                        else:
                            assert abs(g["quality"] - o_res["quality"][i]) <= 1e-9 * max(1.0, abs(o_res["quality"][i])), ("quality", i)
                except (AssertionError, subprocess.CalledProcessError) as e:
                    bad += 1
                    log(f"MISMATCH round {rnd} L {L} n {n} {algo} {spec_str} sums={sums}: {str(e)[:300]}")
            log(f"round {rnd}: L {L} n {n} {algo} {spec_str}: {len(expected)} pairs back")
    return bad, done


if __name__ == "__main__":
    bad, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("mismatching runs:", bad)
    sys.exit(1 if bad else 0)
