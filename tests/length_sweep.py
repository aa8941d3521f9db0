#!/usr/bin/env python3
"""A randomized campaign over READ LENGTHS (development aid, GPU only; not collected by pytest): reads of 20 to 400 bases --
across the lean kernel's three geometries (at most 192, 256, 320 bases) and beyond them (the general kernel) --, fragments from
shorter than a read to barely overlapping, a share of the pairs cut to ragged lengths, every scorer, both sum modes, against the
oracle on every record, row and counter.

    python tests/length_sweep.py [rounds [first round]]        # ~10 s per round on one MI355X + the oracle on the host's cores
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
from util import compare_counters, compare_results

ALGOS = ["simple_bayesian", "pear", "rdp_mle", "uparse", "flash", "ea_util", "stitch"]
EDGES = [20, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300, 319, 320, 321, 384, 400]

def run(rounds=10, first=0, n=1500, log=print):
    """-> (mismatching launches, launches compared)"""
    bad = 0
    done = 0
    for rnd in range(first, first + rounds):
        rng = np.random.default_rng(11000 + rnd)
        L = int(rng.choice(EDGES)) if rnd % 2 else int(rng.integers(20, 401))
        tmin = int(rng.integers(max(12, L // 2), L + 1))                    # fragments shorter than a read: the overlap runs past its ends
        tmax = int(rng.integers(max(tmin, L + 1), min(2 * L - 2, 1024) + 1))
        spec = synth.SynthSpec(0x5EED1000 + rnd, L, tmin, tmax, qhi=int(rng.integers(30, 42)))
        f, r, fl, rl = synth.generate_host(spec, int(rng.integers(0, 1_000_000)), n)
        if rnd % 3 == 0:    # a third of the rounds: a quarter of the pairs cut to ragged lengths
            cut = rng.random(n) < 0.25
            fl = np.where(cut, rng.integers(1, L + 1, n), fl).astype(fl.dtype)
            rl = np.where(cut, rng.integers(1, L + 1, n), rl).astype(rl.dtype)
        for algo in ALGOS:
            rl2 = np.minimum(rl, fl) if algo == "pear" else rl     # (past the forward read pear reads beyond the array in the reference)
            o = po.assemble_batch(po.make_config(algo), f, r, fl, rl2)
            for exact in (False, True):
                asm = pa.Assembler(0)
                asm.set_algorithm(algo)
                asm.set_exact_sums(exact)
                b = pa.upload_batch(f, r, fl, rl2, device="cuda:0", seq_stride=0)
                asm.assemble_device(b)
                torch.cuda.synchronize()
                done += 1
                try:
                    asm.check_overflow()
                    compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=exact)
                    compare_counters(b.counters_dict(), o[3])
                except AssertionError as e:
                    bad += 1
                    log(f"MISMATCH round {rnd} L {L} fragments {tmin}-{tmax} {algo} exact={exact}: {str(e)[:300]}")
                    if os.environ.get("SWEEP_TRACE"):
                        import traceback
                        traceback.print_exc()
        st = o[0]["status"]
        log(f"round {rnd}: L {L} fragments {tmin}-{tmax}: ok {(st == 0).sum()} of {n}, deferred {asm.last_deferred()}")
    return bad, done


if __name__ == "__main__":
    bad, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("mismatching launches:", bad)
    sys.exit(1 if bad else 0)
