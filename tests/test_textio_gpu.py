"""GPU parity tests of the ingest / formatting rows: the HIP kernels behind
include/pandaseq_amd_io.h against the I/O oracle, on the same texts, byte for byte."""
import numpy as np
import pytest

from oracle import pyoracle as po
from pandaseq_amd import synth
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def io(built):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import pandaseq_amd as pa
    from pandaseq_amd import textio
    return pa, textio, textio.FastqParser(0), textio.Formatter(0)


def parse_both(textio, ft, rt, qualmin=33, policy=2, stride=456, parser=None, max_pairs=None):
    own = parser is None
    if own:
        parser = textio.FastqParser(0, qualmin, policy)
    n_max = ft.count(b"\n") // 4 + 2 if max_pairs is None else max_pairs
    out, st = parser.parse(textio.text_to_device(ft + b"\0" * 0) if ft else textio.text_to_device(b"\n")[:0],
                           textio.text_to_device(rt) if rt else textio.text_to_device(b"\n")[:0], n_max, stride)
    ora = po.fastq_read(ft, rt, qualmin=qualmin, policy=policy, stride=stride, max_pairs=n_max)
    if own:
        parser.close()
    return out, st, ora


def check_parse(textio, ft, rt, qualmin=33, policy=2, stride=456, parser=None):
    out, st, (pf, pr, pfl, prl, oids, ost) = parse_both(textio, ft, rt, qualmin, policy, stride, parser)
    ng = int(st.n_good)
    assert ng == ost.n_records, (st.as_dict(), ost.n_records)
    ostop = po.FQ_STOP[ost.stop_code]
    dstop = textio.FQ_STOP[st.stop_code]
    if ostop == "EOF" and ng == st.n_records:
        assert dstop == "END"  # a record that lacks lines is the chunk's tail: carried over, not an error yet
    elif ostop == "END" and ng < st.n_records:
        assert dstop == "LONG_HEADER"  # the reference ends the stream silently on such a line
    else:
        assert dstop == ostop, (dstop, ostop)
        assert int(st.stop_in_reverse) == int(ost.stop_in_reverse)
    assert int(st.n_dropped) == ost.n_dropped
    ids = out["ids"].cpu().numpy().view(textio.SEQID_DTYPE).reshape(-1)[:ng]
    keep = np.nonzero(ids["flag"] == 0)[0]
    assert np.all((ids["flag"] == 0) | (ids["flag"] == 1))
    assert keep.size == ost.n_pairs
    dfl = out["flen"].cpu().numpy().view(np.uint16)[:ng]
    drl = out["rlen"].cpu().numpy().view(np.uint16)[:ng]
    assert np.array_equal(dfl[keep], pfl) and np.array_equal(drl[keep], prl)
    assert np.all(dfl[ids["flag"] == 1] == 0)
    assert np.array_equal(out["fwd"].cpu().numpy()[:ng][keep], pf), "forward rows"
    assert np.array_equal(out["rev"].cpu().numpy()[:ng][keep], pr), "reverse rows"
    for j, k in enumerate(keep):
        assert textio.seqid_fields(ids[k], ft) == oids[j].fields(), (k, textio.seqid_fields(ids[k], ft), oids[j].fields())
    assert int(st.seen_under_64) == int(ost.seen_under_64)
    if keep.size:
        assert int(st.max_flen) == int(pfl.max()) and int(st.max_rlen) == int(prl.max())
    return out, st


@pytest.mark.parametrize("crlf", [False, True])
def test_parse_synthetic_config1(io, crlf):
    pa, textio, parser, _ = io
    spec = synth.config_spec(1)
    f, r, fl, rl = synth.generate_host(spec, 0, 20000)
    ft, rt = synth.render_fastq(f, r, fl, rl, crlf=crlf)
    out, st = check_parse(textio, ft, rt, stride=spec.row_stride(), parser=parser)
    assert st.n_good == 20000 and st.fwd_consumed == len(ft) and st.rev_consumed == len(rt)
    # round trip: text -> rows gives back the rows the text was rendered from
    assert np.array_equal(out["fwd"].cpu().numpy()[:20000], f) and np.array_equal(out["rev"].cpu().numpy()[:20000], r)


@pytest.mark.parametrize("case", util.fastq_cases(), ids=lambda c: c[0])
def test_parse_edge_cases(io, case):
    pa, textio, _, _ = io
    name, ft, rt, qualmin, policy, stop = case
    check_parse(textio, ft, rt, qualmin=qualmin, policy=policy)


def test_parse_mixed_formats_ragged(io):
    pa, textio, parser, _ = io
    ft, rt = util.random_fastq(6000, 5)
    out, st = check_parse(textio, ft, rt, parser=parser)
    assert st.n_good == 6000 and st.n_dropped > 0


def test_parse_in_chunks(io):
    """Cutting the texts anywhere and resuming at fwd_consumed / rev_consumed gives the same rows."""
    pa, textio, parser, _ = io
    ft, rt = util.random_fastq(3000, 9)
    whole, wst = check_parse(textio, ft, rt, parser=parser)
    rows_f, rows_r, lens = [], [], []
    fpos = rpos = 0
    cuts = [(len(ft) // 3 + 7, len(rt) // 2 + 3), (2 * len(ft) // 3 + 1, len(rt) // 2 + 4000), (len(ft), len(rt))]
    for fcut, rcut in cuts:
        fchunk, rchunk = ft[fpos:fcut], rt[rpos:rcut]
        out, st = parser.parse(textio.text_to_device(fchunk), textio.text_to_device(rchunk), 4000, 456)
        assert st.stop_code == 0
        ng = int(st.n_good)
        rows_f.append(out["fwd"].cpu().numpy()[:ng])
        rows_r.append(out["rev"].cpu().numpy()[:ng])
        lens.append(out["flen"].cpu().numpy()[:ng])
        fpos += int(st.fwd_consumed)
        rpos += int(st.rev_consumed)
    assert fpos == len(ft) and rpos == len(rt)
    n = int(wst.n_good)
    assert np.array_equal(np.concatenate(rows_f), whole["fwd"].cpu().numpy()[:n])
    assert np.array_equal(np.concatenate(rows_r), whole["rev"].cpu().numpy()[:n])
    assert np.array_equal(np.concatenate(lens), whole["flen"].cpu().numpy()[:n])


def test_parse_row_overflow_and_validation(io):
    pa, textio, parser, _ = io
    spec = synth.config_spec(1)
    f, r, fl, rl = synth.generate_host(spec, 0, 64)
    ft, rt = synth.render_fastq(f, r, fl, rl)
    out, st = parser.parse(textio.text_to_device(ft), textio.text_to_device(rt), 100, stride=128)
    assert textio.FQ_STOP[st.stop_code] == "ROW_OVERFLOW" and st.n_good == 0
    with pytest.raises(pa.PamdError):
        parser.parse(textio.text_to_device(ft), textio.text_to_device(rt), 100, stride=130)


def _assemble(pa, rows, n, algo="simple_bayesian", modules=()):
    import torch
    asm = pa.Assembler(0)
    asm.set_algorithm(algo)
    for m in modules:
        asm.add_module(m)
    outs = pa.alloc_outputs(n, n * 912 + 16384 * 4096, "cuda:0")  # every wave reserves a 16384-base arena chunk
    b = pa.DeviceBatch(n=n, fwd=rows["fwd"][:n], rev=rows["rev"][:n], flen=rows["flen"][:n], rlen=rows["rlen"][:n], **outs)
    asm.assemble_device(b)
    torch.cuda.synchronize()
    return asm, b


@pytest.mark.parametrize("fmt", ["fasta", "fastq"])
@pytest.mark.parametrize("source", ["config1", "mixed", "long_headers"])
def test_format_matches_oracle(io, fmt, source):
    """FASTQ text -> rows -> assembly -> FASTA / FASTQ text, all on the device, against the oracle chain
    (reader -> assembler -> output.c formatting) on the host.  long_headers: header lines too long for the writer's LDS
    slots (they go straight to the output), mixed with ordinary ones."""
    pa, textio, parser, formatter = io
    if source in ("config1", "long_headers"):
        spec = synth.config_spec(1)
        f, r, fl, rl = util.adversarial_pairs(spec, 4000, ragged=False, weird_quals=False)
        ft, rt = synth.render_fastq(f, r, fl, rl, tag="ACGT")
        if source == "long_headers":    # every record whose y coordinate ends in 7: a 60-character instrument name
            long_name = b"@" + b"SIMULATEDINSTRUMENT0" * 3 + b":1:FC:1:"
            import re
            ft = re.sub(rb"@SIM:1:FC:1:(\d{4}:\d{5}:\d{4}7 )", lambda m: long_name + m.group(1), ft)
            rt = re.sub(rb"@SIM:1:FC:1:(\d{4}:\d{5}:\d{4}7 )", lambda m: long_name + m.group(1), rt)
            assert ft.count(b"SIMULATEDINSTRUMENT0") > 300
        stride = spec.row_stride()
    else:
        ft, rt = util.random_fastq(3000, 21)
        stride = 456
    ftd = textio.text_to_device(ft)
    rows, st = parser.parse(ftd, textio.text_to_device(rt), ft.count(b"\n") // 4 + 1, stride)
    n = int(st.n_good)
    asm, b = _assemble(pa, rows, n)
    which = textio.OUT_FASTA if fmt == "fasta" else textio.OUT_FASTQ
    out, nbytes, fixups = formatter.format(b, rows["ids"], ftd, which)
    text = bytearray(out[:nbytes].cpu().numpy().tobytes())
    res = b.results_numpy()
    textio.apply_fixups(text, fixups, res["quality"])
    # oracle chain
    pf, pr, pfl, prl, oids, ost = po.fastq_read(ft, rt, stride=stride)
    cfg = po.make_config("simple_bayesian")
    ores, ont, op, _ = po.assemble_batch(cfg, pf, pr, pfl, prl)
    want = []
    for i in range(len(ores)):
        if ores["status"][i] == 0:
            m = int(ores["seq_len"][i])
            if fmt == "fasta":
                want.append(po.output_fasta(oids[i], float(ores["quality"][i]), ont[i, :m]))
            else:
                want.append(po.output_fastq(oids[i], float(ores["quality"][i]), ont[i, :m], op[i, :m]))
    want = b"".join(want)
    assert len(want) > 0
    assert bytes(text) == want


def test_format_with_rejections_and_fastq_phred_range(io):
    """rdp_mle + both check modules: rejected pairs print nothing; phred bytes span the table."""
    pa, textio, parser, formatter = io
    spec = synth.config_spec(3)
    f, r, fl, rl = synth.generate_host(spec, 0, 3000)
    ft, rt = synth.render_fastq(f, r, fl, rl)
    ftd = textio.text_to_device(ft)
    rows, st = parser.parse(ftd, textio.text_to_device(rt), 3001, spec.row_stride())
    asm, b = _assemble(pa, rows, 3000, "rdp_mle", ("min_phred:2", "min_overlapbits:15"))
    out, nbytes, fixups = formatter.format(b, rows["ids"], ftd, textio.OUT_FASTQ)
    text = bytearray(out[:nbytes].cpu().numpy().tobytes())
    res = b.results_numpy()
    textio.apply_fixups(text, fixups, res["quality"])
    pf, pr, pfl, prl, oids, ost = po.fastq_read(ft, rt, stride=spec.row_stride())
    cfg = po.make_config("rdp_mle", modules=(("min_phred", 2), ("min_overlapbits", 15)))
    ores, ont, op, _ = po.assemble_batch(cfg, pf, pr, pfl, prl)
    want = b"".join(po.output_fastq(oids[i], float(ores["quality"][i]), ont[i, :int(ores["seq_len"][i])], op[i, :int(ores["seq_len"][i])])
                    for i in range(3000) if ores["status"][i] == 0)
    assert (ores["status"] == 6).sum() > 0
    assert bytes(text) == want


def test_fastq_phred_of_every_kind_of_value(io):
    """The FASTQ writer's quality characters over log-p values chosen to sit on, next to and between the 47 table values
    (panda_result_phred's walk, nt.c:126-150, is replaced on the device by arithmetic plus three table reads): every table
    value, its two neighbouring doubles, the midpoints, 0, tiny and large negatives, and a dense random cover."""
    import torch
    pa, textio, parser, formatter = io
    spec = synth.config_spec(1)
    n = 600
    f, r, fl, rl = synth.generate_host(spec, 100, n)
    ft, rt = synth.render_fastq(f, r, fl, rl)
    ftd = textio.text_to_device(ft)
    rows, st = parser.parse(ftd, textio.text_to_device(rt), n + 1, spec.row_stride())
    asm, b = _assemble(pa, rows, n, "simple_bayesian", ())
    score = po.tables_np()["score"]
    rng = np.random.default_rng(3)
    special = np.concatenate([score, np.nextafter(score, 0.0), np.nextafter(score, -10.0), 0.5 * (score[:-1] + score[1:]),
                              [0.0, -0.0, -1e-300, -1e-12, -2.4e-5, -2.6e-5, -1.9999999, -2.0000001, -5.0, -50.0, -745.0],
                              -np.exp(rng.uniform(np.log(1e-6), np.log(3.0), 4000)), rng.uniform(-2.2, 0.0, 4000)])
    m = b.seq_p.numel()
    vals = np.resize(rng.permutation(special), m)
    b.seq_p.copy_(torch.from_numpy(vals).to(b.seq_p.device))
    out, nbytes, fixups = formatter.format(b, rows["ids"], ftd, textio.OUT_FASTQ)
    text = bytearray(out[:nbytes].cpu().numpy().tobytes())
    res = b.results_numpy()
    textio.apply_fixups(text, fixups, res["quality"])
    pf, pr, pfl, prl, oids, ost = po.fastq_read(ft, rt, stride=spec.row_stride())
    nt = b.seq_nt.cpu().numpy()
    want = []
    for i in range(n):
        if res["status"][i] == 0:
            o, k = int(res["seq_offset"][i]), int(res["seq_len"][i])
            want.append(po.output_fastq(oids[i], float(res["quality"][i]), nt[o:o + k], vals[o:o + k]))
    assert len(want) > n // 2 and bytes(text) == b"".join(want)


def test_io_golden_fixture(io):
    """The committed fixture tests/golden/io/mixed.npz: text -> rows -> assembly -> FASTA / FASTQ on the device
    against the stored bytes."""
    import os
    pa, textio, parser, formatter = io
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io", "mixed.npz"))
    ft, rt = g["fwd_text"].tobytes(), g["rev_text"].tobytes()
    ftd = textio.text_to_device(ft)
    rows, st = parser.parse(ftd, textio.text_to_device(rt), 400, 456)
    n_pairs, n_records, n_dropped, stop, under64 = g["status"].tolist()
    assert (int(st.n_good), int(st.n_dropped), int(st.stop_code), int(st.seen_under_64)) == (n_records, n_dropped, stop, under64)
    ids = rows["ids"].cpu().numpy().view(textio.SEQID_DTYPE).reshape(-1)[:n_records]
    keep = np.nonzero(ids["flag"] == 0)[0]
    assert np.array_equal(rows["fwd"].cpu().numpy()[:n_records][keep], g["fwd"])
    assert np.array_equal(rows["rev"].cpu().numpy()[:n_records][keep], g["rev"])
    assert np.array_equal(rows["flen"].cpu().numpy().view(np.uint16)[:n_records][keep], g["flen"])
    for j, k in enumerate(keep):
        f = textio.seqid_fields(ids[k], ft)
        assert ("%s:%s:%s:%d:%d:%d:%d:%s" % f).encode("latin1") == bytes(g["names"][j])
    asm, b = _assemble(pa, rows, n_records)
    for which, key in ((textio.OUT_FASTA, "fasta"), (textio.OUT_FASTQ, "fastq")):
        out, nbytes, fixups = formatter.format(b, rows["ids"], ftd, which)
        text = bytearray(out[:nbytes].cpu().numpy().tobytes())
        textio.apply_fixups(text, fixups, b.results_numpy()["quality"])
        assert bytes(text) == g[key].tobytes(), key


def test_parse_max_pairs_and_resume(io):
    """max_pairs smaller than the records present: exactly that many are parsed and fwd/rev_consumed point
    at the next record, so a second call continues where the first stopped."""
    pa, textio, parser, _ = io
    ft, rt = util.random_fastq(500, 3)
    whole, wst = check_parse(textio, ft, rt, parser=parser)
    out1, st1 = parser.parse(textio.text_to_device(ft), textio.text_to_device(rt), 123, 456)
    assert (int(st1.n_records), int(st1.n_good), int(st1.stop_code)) == (123, 123, 0)
    out2, st2 = parser.parse(textio.text_to_device(ft[int(st1.fwd_consumed):]), textio.text_to_device(rt[int(st1.rev_consumed):]), 1000, 456)
    assert int(st2.n_good) == 500 - 123
    both = np.concatenate([out1["fwd"].cpu().numpy()[:123], out2["fwd"].cpu().numpy()[:377]])
    assert np.array_equal(both, whole["fwd"].cpu().numpy()[:500])
    assert int(st1.n_dropped) + int(st2.n_dropped) == int(wst.n_dropped)


def test_format_capacity_and_empty(io):
    pa, textio, parser, formatter = io
    spec = synth.config_spec(1)
    f, r, fl, rl = synth.generate_host(spec, 0, 512)
    ft, rt = synth.render_fastq(f, r, fl, rl)
    ftd = textio.text_to_device(ft)
    rows, st = parser.parse(ftd, textio.text_to_device(rt), 512, spec.row_stride())
    asm, b = _assemble(pa, rows, 512)
    out, nbytes, _ = formatter.format(b, rows["ids"], ftd, textio.OUT_FASTA)
    assert nbytes > 0
    # a buffer one byte too small is refused, with the size that is needed in the message
    with pytest.raises(pa.PamdError, match=str(nbytes)):
        formatter.format(b, rows["ids"], ftd, textio.OUT_FASTA, capacity=nbytes - 1)
    out2, nbytes2, _ = formatter.format(b, rows["ids"], ftd, textio.OUT_FASTA, capacity=nbytes)
    assert nbytes2 == nbytes and bytes(out2[:nbytes].cpu().numpy().tobytes()) == bytes(out[:nbytes].cpu().numpy().tobytes())
    # nothing accepted -> no text: an impossible quality threshold turns every pair into LOWQ
    asm2 = pa.Assembler(0)
    asm2.set_threshold(0.999999)
    outs = pa.alloc_outputs(512, 512 * 912 + 16384 * 4096, "cuda:0")
    b2 = pa.DeviceBatch(n=512, fwd=rows["fwd"], rev=rows["rev"], flen=rows["flen"], rlen=rows["rlen"], **outs)
    asm2.assemble_device(b2)
    import torch
    torch.cuda.synchronize()
    assert (b2.results_numpy()["status"] != 0).all()
    _, nb, _ = formatter.format(b2, rows["ids"], ftd, textio.OUT_FASTQ)
    assert nb == 0
