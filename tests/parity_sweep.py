#!/usr/bin/env python3
"""A wide randomized parity campaign (development aid, GPU only; not collected by pytest): many seeds x configurations x
scorers x both summation modes, the device path against the oracle on every record, row and counter.

    python tests/parity_sweep.py [rounds [wide|plain [first round [pairs]]]]        # ~20 s per round on one MI355X + the oracle on the host's cores

From 20 000 pairs per launch on (every wave of the lean kernel then goes through whole groups of its reference-order sums and
several finishes) the oracle runs on all the host's cores and the rows are compared through their checksums (sum (j+1) nt_j, sum
(j+1) bits(p_j), wrapping 64-bit: exact in any order of addition) instead of base by base in Python.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import pandaseq_amd as pa
from oracle import pyoracle as po
from pandaseq_amd import synth
from util import adversarial_pairs, compare_counters, compare_results

ALGOS = ["simple_bayesian", "pear", "rdp_mle", "uparse", "flash", "ea_util", "stitch"]

def run(rounds=5, wide=False, first=0, n=3000, log=print):
    """-> (mismatching launches, launches compared); tests/test_sweeps_gpu.py runs a fixed-seed slice under pytest -m gpu"""
    bad = 0
    done = 0
    for rnd in range(first, first + rounds):
        rng = np.random.default_rng((5000 if wide else 1000) + rnd)
        for cfg in (0, 1, 2, 4):
            spec = synth.config_spec(cfg)
            start = int(rng.integers(0, 40_000_000))
            if rnd % 2 == 0:
                f, r, fl, rl = synth.generate_host(spec, start, n)
            else:
                f, r, fl, rl = adversarial_pairs(spec, n, seed=int(rng.integers(1, 1 << 30)))
            primers = cfg == 2
            for algo in (ALGOS if not primers else ["pear", "simple_bayesian", "rdp_mle"]):
                fl2, rl2 = fl, rl
                if algo == "pear":
                    rl2 = np.minimum(rl, fl)    # (past the forward read pear reads beyond the array in the reference: kept defined)
                kw = {}
                if primers:
                    kw = dict(fprimer=synth.nt_codes(spec.fprimer), rprimer=synth.nt_codes(spec.rprimer, complement=True))
                if rnd % 3 == 2:
                    kw["minoverlap"] = int(rng.integers(2, 12))
                mods = ()
                if wide:    # the knobs the first version left alone: threshold, trims, a maximum overlap, the two fused modules
                    kw["threshold"] = float(rng.choice([0.6, 0.7, 0.9, 0.4]))
                    if not primers and rng.random() < 0.5:
                        kw["ftrim"], kw["rtrim"] = int(rng.integers(0, 40)), int(rng.integers(0, 40))
                    if rng.random() < 0.25:
                        kw["maxoverlap"] = int(rng.integers(60, 260))
                    if rng.random() < 0.5:
                        mods = tuple(m for m in (("min_phred", int(rng.integers(1, 12))), ("min_overlapbits", int(rng.integers(5, 40)))) if rng.random() < 0.7)
                big = n >= 20000
                ocfg = po.make_config(algo, modules=mods, **kw)
                if big:
                    o = po.assemble_batch(ocfg, f, r, fl2, rl2, want_seq=False, threads=os.cpu_count() or 1)
                    o_sums = po.row_checksums(ocfg, f, r, fl2, rl2, threads=os.cpu_count() or 1)
                else:
                    o = po.assemble_batch(ocfg, f, r, fl2, rl2)
                for exact in (True, False):
                    asm = pa.Assembler(0)
                    asm.set_algorithm(algo)
                    asm.set_exact_sums(exact)
                    if primers:
                        asm.set_forward_primer(kw["fprimer"])
                        asm.set_reverse_primer(kw["rprimer"])
                    if "minoverlap" in kw:
                        asm.set_minimum_overlap(kw["minoverlap"])
                    if "threshold" in kw:
                        asm.set_threshold(kw["threshold"])
                    if "ftrim" in kw:
                        asm.set_forward_trim(kw["ftrim"])
                        asm.set_reverse_trim(kw["rtrim"])
                    if "maxoverlap" in kw:
                        asm.set_maximum_overlap(kw["maxoverlap"])
                    for name, arg in mods:
                        asm.add_module("%s:%d" % (name, arg))
                    b = pa.upload_batch(f, r, fl2, rl2, device="cuda:0", seq_stride=0)
                    asm.assemble_device(b)
                    torch.cuda.synchronize()
                    done += 1
                    try:
                        if big:
                            asm.check_overflow()
                            compare_results(b.results_numpy(), None, None, o[0], None, None, exact_sums=exact, check_seq=False)
                            d_nt, d_p = pa.row_checksums(b.results, b.seq_nt, b.seq_p)
                            assert np.array_equal(d_nt.cpu().numpy().view(np.uint64), o_sums[1]), "row checksums (nt)"
                            assert np.array_equal(d_p.cpu().numpy().view(np.uint64), o_sums[2]), "row checksums (log-p)"
                        else:
                            compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=exact)
                        compare_counters(b.counters_dict(), o[3])
                    except AssertionError as e:
                        bad += 1
                        log(f"MISMATCH round {rnd} config {cfg} {algo} exact={exact} start={start} {kw} {mods}: {str(e)[:300]}")
                        if os.environ.get("SWEEP_TRACE"):
                            import traceback
                            traceback.print_exc()
        log(f"round {rnd}: {done} launches compared, {bad} mismatches")
    return bad, done


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 5, len(sys.argv) > 2 and sys.argv[2] == "wide", int(sys.argv[3]) if len(sys.argv) > 3 else 0,
                      **({"n": int(sys.argv[4])} if len(sys.argv) > 4 else {}))[0] else 0)
