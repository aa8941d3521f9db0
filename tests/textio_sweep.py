#!/usr/bin/env python3
"""A randomized campaign on the data-format rows (development aid, GPU only; not collected by pytest): random mixed-format
ragged FASTQ texts through the device parser against the I/O oracle -- rows, lengths, parsed headers, stop conditions --, then
assembly and FASTA / FASTQ formatting against the oracle chain's text, byte for byte.

    python tests/textio_sweep.py [rounds [first round]]        # ~3 s per round on one MI355X + the oracle on one host core
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import pandaseq_amd as pa
from oracle import pyoracle as po
from pandaseq_amd import textio
from tests import util
from tests.test_textio_gpu import _assemble, check_parse


def run(rounds=10, first=0, log=print):
    """-> (mismatching rounds, rounds compared)"""
    parser, formatter = textio.FastqParser(0), textio.Formatter(0)
    bad = 0
    for rnd in range(first, first + rounds):
        rng = np.random.default_rng(9000 + rnd)
        n = int(rng.integers(1, 2500))
        ft, rt = util.random_fastq(n, 100 + rnd)
        stride = 456
        algo = ["simple_bayesian", "stitch", "rdp_mle", "uparse"][rnd % 4]
        try:
            out, st = check_parse(textio, ft, rt, parser=parser)
            ng = int(st.n_good)
            ftd = textio.text_to_device(ft)
            rows, st2 = parser.parse(ftd, textio.text_to_device(rt), ft.count(b"\n") // 4 + 2, stride)
            asm, b = _assemble(pa, rows, int(st2.n_good), algo)
            pf, pr, pfl, prl, oids, ost = po.fastq_read(ft, rt, stride=stride)
            ores, ont, op, _ = po.assemble_batch(po.make_config(algo), pf, pr, pfl, prl)
            for which, name in ((textio.OUT_FASTA, "fasta"), (textio.OUT_FASTQ, "fastq")):
                o, nbytes, fixups = formatter.format(b, rows["ids"], ftd, which)
                text = bytearray(o[:nbytes].cpu().numpy().tobytes())
                textio.apply_fixups(text, fixups, b.results_numpy()["quality"])
                want = []
                for i in range(len(ores)):
                    if ores["status"][i] == 0:
                        m = int(ores["seq_len"][i])
                        want.append(po.output_fasta(oids[i], float(ores["quality"][i]), ont[i, :m]) if name == "fasta"
                                    else po.output_fastq(oids[i], float(ores["quality"][i]), ont[i, :m], op[i, :m]))
                assert bytes(text) == b"".join(want), name
            log(f"round {rnd}: {n} records, {ng} good, {algo}: ok")
        except AssertionError as e:
            bad += 1
            log(f"MISMATCH round {rnd} ({n} records, {algo}): {str(e)[:300]}")
            if os.environ.get("SWEEP_TRACE"):
                import traceback
                traceback.print_exc()
    return bad, rounds


if __name__ == "__main__":
    bad, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("mismatching rounds:", bad)
    sys.exit(1 if bad else 0)
