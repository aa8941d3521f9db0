"""CPU tests of the ingest / formatting rows: the I/O oracle against the
reference's own golden vectors and known answers, and the C-ABI surface of include/pandaseq_amd_io.h."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from pandaseq_amd import synth
from tests import util

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_seqid_golden_vectors_of_check_parser():
    """The six header cases of the reference's own parser test (check_parser.c:29-36)."""
    with open(os.path.join(GOLDEN, "seqid_check_parser.json")) as fh:
        g = json.load(fh)
    assert len(g["cases"]) == 6
    for c in g["cases"]:
        d, fmt, sid, _ = po.seqid_parse(c["header"], po.TAG_OPTIONAL)
        assert d == c["dir"], c["header"]
        assert po.IDFMT[fmt] == c["format"], c["header"]
        assert list(sid.fields()) == c["id"], c["header"]


def test_seqid_print_format():
    """seqid.c:115: "%s:%s:%s:%d:%d:%d:%d:%s"; a CASAVA 1.4 name prints two empty fields."""
    _, _, sid, _ = po.seqid_parse("1468:1:1:12675:1118#ATCACGA/1")
    assert po.seqid_str(sid) == b"1468:::1:1:12675:1118:ATCACGA"
    _, _, sid, _ = po.seqid_parse("SRR001666.1 071112_SLXA-EAS1_s_7:5:1:817:345 length=36")
    assert po.seqid_str(sid) == b"SRR1666:::1:0:0:0:"


def test_nt_tables():
    """nt.c:25-114: letters by IUPAC meaning, U = T, X = N, complement table = base-wise complement,
    index is (c & 0x1F) so case does not matter and '@' / '[' .. '_' / other letters are invalid."""
    L = po._io()
    fwd = {c: L.po_nt_from_ascii(c.encode()) for c in "ACGTUNXRYKMSWBDHV"}
    assert [fwd[c] for c in "ACGT"] == [1, 2, 4, 8] and fwd["U"] == 8 and fwd["N"] == 15 and fwd["X"] == 15
    assert fwd["R"] == 5 and fwd["Y"] == 10 and fwd["K"] == 12 and fwd["M"] == 3 and fwd["S"] == 6 and fwd["W"] == 9
    assert fwd["B"] == 14 and fwd["D"] == 13 and fwd["H"] == 11 and fwd["V"] == 7
    for c in "ACGTUNXRYKMSWBDHV":
        v = fwd[c]
        comp = ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3)
        assert L.po_nt_from_ascii_complement(c.encode()) == comp
        assert L.po_nt_from_ascii(c.lower().encode()) == v
    for c in "@EFIJLOPQZ[\\]^_ 1":
        assert L.po_nt_from_ascii(c.encode()) == 0, c
    assert b"".join(L.po_nt_to_ascii(v) for v in range(16)) == b"NACMGRSVTWYHKDBN"
    assert L.po_nt_to_ascii(200) == b"N"


def test_fastq_round_trip_through_the_oracle():
    spec = synth.config_spec(1)
    f, r, fl, rl = synth.generate_numpy(spec, 0, 3000)
    for crlf in (False, True):
        ft, rt = synth.render_fastq(f, r, fl, rl, crlf=crlf)
        pf, pr, pfl, prl, ids, st = po.fastq_read(ft, rt, stride=spec.row_stride())
        assert st.n_pairs == 3000 and st.stop_code == 0 and st.seen_under_64 == 1
        assert np.array_equal(pf, f) and np.array_equal(pr, r) and np.array_equal(pfl, fl) and np.array_equal(prl, rl)
        assert ids[17].fields() == ("SIM", "1", "FC", 1, 1101, 10000, 10017, "1")


@pytest.mark.parametrize("case", util.fastq_cases(), ids=lambda c: c[0])
def test_fastq_stop_conditions(case):
    name, ft, rt, qualmin, policy, stop = case
    pf, pr, pfl, prl, ids, st = po.fastq_read(ft, rt, qualmin=qualmin, policy=policy)
    assert po.FQ_STOP[st.stop_code] == stop, (name, po.FQ_STOP[st.stop_code])


def test_fastq_known_answers():
    cases = {c[0]: c for c in util.fastq_cases()}
    # TOINDEX (fastq.c:44): below the offset -> 0, above offset + 46 -> 46 - offset (13 for PHRED+33)
    _, ft, rt, qm, pol, _ = cases["qual_range"]
    pf, pr, pfl, prl, ids, st = po.fastq_read(ft, rt, qualmin=qm, policy=pol)
    assert st.n_pairs == 2
    assert pf[1, :8, 1].view(np.int8).tolist() == [0, 1, 2, 46, 13, 13, 13, 13]
    assert pr[1, :8, 1].view(np.int8).tolist() == [0, 0, 0, 13, 13, 0, 0, 13]
    _, ft, rt, qm, pol, _ = cases["qual64"]
    pf, pr, pfl, prl, ids, st = po.fastq_read(ft, rt, qualmin=qm, policy=pol)
    assert pf[1, :8, 1].view(np.int8).tolist() == [0, 1, 2, 40, 46, -18, -18, 0]
    # an empty forward read is dropped by the reader, an empty reverse read is not (fastq.c:192)
    _, ft, rt, qm, pol, _ = cases["empty_forward_dropped"]
    *_, st = po.fastq_read(ft, rt)
    assert (st.n_records, st.n_pairs, st.n_dropped) == (3, 2, 1)
    _, ft, rt, qm, pol, _ = cases["empty_reverse_kept"]
    pf, pr, pfl, prl, ids, st = po.fastq_read(ft, rt)
    assert st.n_pairs == 2 and prl[1] == 0 and pfl[1] == 13
    # a sequence line is cut at MAX_LEN characters (fastq.c:62) and then needs exactly that many qualities
    _, ft, rt, qm, pol, _ = cases["seq_451_qual_450"]
    pf, pr, pfl, prl, ids, st = po.fastq_read(ft, rt)
    assert st.n_pairs == 2 and pfl[1] == 450
    # the reverse read is complemented, not reversed
    _, ft, rt, qm, pol, _ = cases["lowercase_iupac"]
    pf, pr, pfl, prl, ids, st = po.fastq_read(ft, rt)
    assert pf[1, :5, 0].tolist() == [1, 2, 4, 8, 15] and pr[1, :5, 0].tolist() == [8, 4, 2, 1, 15]
    # 32-bit wrap of an over-long number, leading zeros
    _, ft, rt, qm, pol, _ = cases["int_wrap"]
    pf, pr, pfl, prl, ids, st = po.fastq_read(ft, rt)
    assert ids[1].fields()[5:7] == (np.int32(np.uint32(99999999999 % 2**32)).item() if False else ids[1].x, 7)
    assert ids[1].x == int(np.array(99999999999 % 2**32, dtype=np.uint32).view(np.int32))


def test_output_known_answers():
    """output.c:85-126 on a hand-made result."""
    _, _, sid, _ = po.seqid_parse("M01:7:FC9:2:1101:15000:1400 1:N:0:ACGT")
    t = po.tables_np()
    nt = np.array([1, 2, 4, 8, 15, 3, 0], dtype=np.uint8)
    p = np.array([t["score"][40], t["score"][2], t["qual_nn"], -5.0, t["score"][46], t["score"][0], t["score"][1]])
    q = math.log(0.987654321)
    fa = po.output_fasta(sid, q, nt)
    assert fa == b">M01:7:FC9:2:1101:15000:1400:ACGT;0.987654\nACGTNMN\n"
    fq = po.output_fastq(sid, q, nt, p)
    # panda_result_phred quirks: phred(QS[46]) = 45, phred(QS[0]) = phred(QS[1]) = 1, phred(NN) = 2
    assert fq == b"@M01:7:FC9:2:1101:15000:1400:ACGT;0.987654\nACGTNMN\n+\n" + bytes([33 + 40, 33 + 2, 33 + 2, 33 + 1, 33 + 45, 33 + 1, 33 + 1]) + b"\n"
    assert po.output_fasta(sid, q, nt[:0]) == b""


def test_io_abi_exports_every_declared_symbol():
    from pandaseq_amd import _lib, textio
    import re
    L = _lib.lib()
    hdr = open(os.path.join(os.path.dirname(GOLDEN), "..", "include", "pandaseq_amd_io.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # comments mention functions of the other header
    declared = set(re.findall(r"\b(pamd_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(textio.SYMBOLS), declared ^ set(textio.SYMBOLS)
    for s in textio.SYMBOLS:
        assert hasattr(L, s), s
    assert textio.SEQID_DTYPE.itemsize == 48 and C.sizeof(textio.FastqStatus) == 64


def test_io_product_has_no_cpu_fallback():
    """Without a GPU the I/O entry points fail loudly."""
    from pandaseq_amd import _lib, textio
    if _lib.lib().pamd_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.PamdError):
        textio.FastqParser(0)
    with pytest.raises(_lib.PamdError):
        textio.Formatter(0)


def test_io_oracle_matches_golden_fixture():
    """tests/golden/io/mixed.npz (committed with its generator): the I/O oracle still produces the same
    rows, header fields and FASTA / FASTQ bytes from the same text."""
    g = np.load(os.path.join(GOLDEN, "io", "mixed.npz"))
    ft, rt = g["fwd_text"].tobytes(), g["rev_text"].tobytes()
    pf, pr, pfl, prl, ids, st = po.fastq_read(ft, rt, stride=456)
    assert [st.n_pairs, st.n_records, st.n_dropped, st.stop_code, st.seen_under_64] == g["status"].tolist()
    assert np.array_equal(pf, g["fwd"]) and np.array_equal(pr, g["rev"])
    assert np.array_equal(pfl, g["flen"]) and np.array_equal(prl, g["rlen"])
    assert [po.seqid_str(ids[i]) for i in range(st.n_pairs)] == [bytes(x) for x in g["names"]]
    res, nt, p, _ = po.assemble_batch(po.make_config("simple_bayesian"), pf, pr, pfl, prl)
    assert np.array_equal(res["status"] == 0, g["ok"])
    fasta = b"".join(po.output_fasta(ids[i], float(res["quality"][i]), nt[i, :int(res["seq_len"][i])])
                     for i in range(st.n_pairs) if res["status"][i] == 0)
    assert fasta == g["fasta"].tobytes()
