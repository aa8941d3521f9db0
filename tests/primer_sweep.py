#!/usr/bin/env python3
"""A randomized campaign on the lean kernel's primer scan (development aid, GPU only; not collected by pytest): primers of
random length and content planted at the start of the read, anywhere or nowhere, damaged at random rates and on random
shares of poor qualities -- the reads the shortcut settles, the ones left to the list fallback, the ones deferred -- against
the oracle on every record, row and counter.

    python tests/primer_sweep.py [rounds [first round [after|penalty|plain [copies]]]]        # ~6 s per round on one MI355X + the oracle on the host's cores

copies > 1: the round's 3 000 pairs that many times over in one launch (every wave of the lean kernel then goes through whole groups
of its reference-order sums with these reads in every slot), the oracle on all the host's cores, rows through their checksums.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import pandaseq_amd as pa
from pandaseq_amd import synth
from test_lean_long_gpu import _oracle, _run, sweep_case
from util import compare_counters, compare_results


def run(rounds=10, first=0, mode="plain", log=print, copies=1):
    """mode: plain | after (`-a`: the primers are looked for in the merged sequence) | penalty (a primer penalty, -T: the careful
    primer kernel, compared after exp()) -> (mismatching launches, launches compared)"""
    bad = 0
    done = 0
    after = mode == "after"
    for rnd in range(first, first + rounds):
        c = sweep_case(rnd)
        if copies > 1:
            for k in ("fwd", "rev", "flen", "rlen"):
                c[k] = np.ascontiguousarray(np.concatenate([c[k]] * copies))
        pen = 0.0 if mode != "penalty" else float(np.random.default_rng(rnd).choice([0.001, 0.01, 0.05]))
        okw = dict(fprimer=c["fp"], rprimer=c["rp"], threshold=c["thr"], primers_after=after, primer_penalty=pen)
        if copies > 1:
            from oracle import pyoracle as po
            ocfg = po.make_config(c["algo"], **okw)
            o = po.assemble_batch(ocfg, c["fwd"], c["rev"], c["flen"], c["rlen"], want_seq=False, threads=os.cpu_count() or 1)
            o_sums = po.row_checksums(ocfg, c["fwd"], c["rev"], c["flen"], c["rlen"], threads=os.cpu_count() or 1)
        else:
            o = _oracle(c["algo"], c["fwd"], c["rev"], c["flen"], c["rlen"], **okw)
        for exact in (False, True):
            asm = pa.Assembler(0)
            asm.set_algorithm(c["algo"])
            asm.set_exact_sums(exact)
            asm.set_threshold(c["thr"])
            asm.set_forward_primer(c["fp"])
            asm.set_reverse_primer(c["rp"])
            if after:
                asm.set_primers_after(True)
            if pen != 0.0:
                asm.set_primer_penalty(pen)
            b = _run(asm, c["fwd"], c["rev"], c["flen"], c["rlen"])
            done += 1
            try:
                if copies > 1:
                    compare_results(b.results_numpy(), None, None, o[0], None, None, exact_sums=exact, check_seq=False)
                    d_nt, d_p = pa.row_checksums(b.results, b.seq_nt, b.seq_p)
                    assert np.array_equal(d_nt.cpu().numpy().view(np.uint64), o_sums[1]), "row checksums (nt)"
                    assert np.array_equal(d_p.cpu().numpy().view(np.uint64), o_sums[2]), "row checksums (log-p)"
                else:
                    compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=exact)
                compare_counters(b.counters_dict(), o[3])
            except AssertionError as e:
                bad += 1
                log(f"MISMATCH round {rnd} exact {exact} {c['what']}: {str(e)[:300]}")
        st = o[0]["status"]
        log(f"round {rnd}: {c['what']}: ok {(st == 0).sum()} nofp {(st == 2).sum()} norp {(st == 3).sum()} deferred {asm.last_deferred()}")
    return bad, done


if __name__ == "__main__":
    bad, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 0, sys.argv[3] if len(sys.argv) > 3 else "plain",
                 copies=int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    print("mismatching launches:", bad)
    sys.exit(1 if bad else 0)
