#!/usr/bin/env python3
"""Regenerate tests/golden/io/mixed.npz: a small paired FASTQ text (every header format, ragged reads,
IUPAC letters, odd qualities, DOS line ends, dropped records) with what the I/O ORACLE makes of it -- rows,
lengths, header fields -- and the FASTA / FASTQ text of its simple_bayesian assembly.

Like the other fixtures these pin the oracle (oracle/panda_oracle_io.c, PARITY UNPINNED against a live
reference apart from the header vectors of check_parser.c) against regressions and give the GPU tests a
committed set of expected bytes.  Run from the repo root:  python tests/golden/io/make_io_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po  # noqa: E402
import util  # noqa: E402

N, SEED, STRIDE = 300, 77, 456


def build():
    ft, rt = util.random_fastq(N, SEED)
    pf, pr, pfl, prl, ids, st = po.fastq_read(ft, rt, stride=STRIDE)
    names = [po.seqid_str(ids[i]) for i in range(st.n_pairs)]
    res, nt, p, ctr = po.assemble_batch(po.make_config("simple_bayesian"), pf, pr, pfl, prl)
    fasta = b"".join(po.output_fasta(ids[i], float(res["quality"][i]), nt[i, :int(res["seq_len"][i])])
                     for i in range(st.n_pairs) if res["status"][i] == 0)
    fastq = b"".join(po.output_fastq(ids[i], float(res["quality"][i]), nt[i, :int(res["seq_len"][i])], p[i, :int(res["seq_len"][i])])
                     for i in range(st.n_pairs) if res["status"][i] == 0)
    return dict(fwd_text=np.frombuffer(ft, dtype=np.uint8), rev_text=np.frombuffer(rt, dtype=np.uint8),
                fwd=pf, rev=pr, flen=pfl, rlen=prl, names=np.array(names),
                status=np.array([st.n_pairs, st.n_records, st.n_dropped, st.stop_code, st.seen_under_64], dtype=np.int64),
                fasta=np.frombuffer(fasta, dtype=np.uint8), fastq=np.frombuffer(fastq, dtype=np.uint8),
                ok=(res["status"] == 0))


if __name__ == "__main__":
    d = build()
    np.savez_compressed(os.path.join(HERE, "mixed.npz"), **d)
    print("pairs", d["status"].tolist(), "ok", int(d["ok"].sum()), "fasta bytes", d["fasta"].size, "fastq bytes", d["fastq"].size)
