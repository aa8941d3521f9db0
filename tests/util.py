"""Shared helpers for the parity tests: adversarial inputs and the oracle <-> device comparison."""
import numpy as np

from pandaseq_amd import synth

P_TOL = 1e-6  # north_star: per-base log-probabilities within 1e-6 (we expect exact equality)


def adversarial_pairs(spec, n, seed=1, ragged=True, weird_quals=True, iupac=True):
    """Synthetic pairs plus the edge cases the hot path has: Ns, IUPAC ambiguity codes, Q2 tails
    (B-cliff), ragged and too-short reads, qualities outside 0..46, identical-quality mismatches."""
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, n)
    rng = np.random.default_rng(seed)
    L = spec.read_len
    if iupac:
        # sprinkle ambiguity codes (3..14 non one-hot) and extra Ns
        for arr in (fwd, rev):
            m = rng.random((n, L)) < 0.004
            arr[:, :L, 0][m] = rng.integers(3, 15, size=int(m.sum())).astype(np.uint8)
            m = rng.random((n, L)) < 0.003
            arr[:, :L, 0][m] = 15
            m = rng.random((n, L)) < 0.0005
            arr[:, :L, 0][m] = 0    # not a nucleotide at all: matches nothing, hashes as A (misc.h:41)
    if weird_quals:
        for arr in (fwd, rev):
            m = rng.random((n, L)) < 0.003
            vals = rng.choice(np.array([-5, -1, 0, 1, 2, 46, 47, 50, 60, 100, 127, -128], dtype=np.int8), size=int(m.sum()))
            arr[:, :L, 1][m] = vals.view(np.uint8)
        # a few reads ending in a long run of quality 2
        for arr in (fwd, rev):
            for i in rng.choice(n, size=max(1, n // 50), replace=False):
                k = int(rng.integers(1, L))
                arr[i, L - k:L, 1] = 2
        # whole reads of quality 2 (unmasked length 0)
        if n >= 8:
            fwd[3, :L, 1] = 2
            rev[5, :L, 1] = 2
    if ragged:
        m = rng.random(n) < 0.25
        flen = flen.copy()
        rlen = rlen.copy()
        flen[m] = rng.integers(0, L + 1, size=int(m.sum()))
        m = rng.random(n) < 0.25
        rlen[m] = rng.integers(0, L + 1, size=int(m.sum()))
        if n >= 16:
            flen[7], rlen[7] = 1, L      # BADR
            flen[8], rlen[8] = L, 0      # BADR
            flen[9], rlen[9] = 2, 2
            flen[10], rlen[10] = 9, 9
            flen[11], rlen[11] = 17, L
            flen[12], rlen[12] = L, 17
    return fwd, rev, flen, rlen


SUM_TOL = 1e-9  # fast-sums mode: |quality - ref| and |est_overlap_prob - ref| (north_star allows 1e-6)


def compare_results(dev_res, dev_nt, dev_p, ora_res, ora_nt, ora_p, seq_stride=None, check_seq=True, exact_sums=True, est_rel_tol=0.0):
    """Bit-exact on every integer/index field; the two f64 scalars bit-exact in the default (exact-sums)
    mode and within SUM_TOL in fast-sums mode; per-base p within P_TOL (and required to be exactly equal)."""
    n = len(ora_res)
    assert len(dev_res) == n
    st_d = dev_res["status"].astype(np.int64)
    st_o = ora_res["status"].astype(np.int64)
    bad = np.nonzero(st_d != st_o)[0]
    assert bad.size == 0, f"status differs at pairs {bad[:10]}: dev {st_d[bad[:10]]} oracle {st_o[bad[:10]]}"
    assert np.array_equal(dev_res["reject_module"].astype(np.int64), ora_res["reject_module"].astype(np.int64))
    assert np.array_equal(dev_res["slow"].astype(np.int64), ora_res["slow"].astype(np.int64)), "slow flag"
    # fields that exist once align() succeeded
    merged = (st_o == 0) | (st_o == 5) | (st_o == 6)
    for name_d, name_o in (("overlap", "overlap"), ("mismatches", "mismatches"), ("degenerates", "degenerates"),
                           ("seq_len", "seq_len")):
        a = dev_res[name_d].astype(np.int64)[merged]
        b = ora_res[name_o].astype(np.int64)[merged]
        w = np.nonzero(a != b)[0]
        assert w.size == 0, f"{name_d} differs at {np.nonzero(merged)[0][w[:10]]}: {a[w[:10]]} vs {b[w[:10]]}"
    scanned = merged | (st_o == 4)
    # examined is defined whenever the scan ran; NOALGN from the bounds test leaves it 0 on both sides
    assert np.array_equal(dev_res["examined"].astype(np.int64)[scanned], ora_res["examined"].astype(np.int64)[scanned]), "examined"
    offs = st_o >= 4
    assert np.array_equal(dev_res["fwd_offset"].astype(np.int64)[offs | (st_o == 0)], ora_res["fwd_offset"].astype(np.int64)[offs | (st_o == 0)])
    assert np.array_equal(dev_res["rev_offset"].astype(np.int64)[offs | (st_o == 0)], ora_res["rev_offset"].astype(np.int64)[offs | (st_o == 0)])
    # f64 scalars
    if exact_sums:
        q_d = dev_res["quality"][merged].view(np.uint64)
        q_o = ora_res["quality"][merged].view(np.uint64)
        w = np.nonzero(q_d != q_o)[0]
        assert w.size == 0, f"quality bits differ at {np.nonzero(merged)[0][w[:10]]}"
        if est_rel_tol:
            # ea_util / stitch: a device log() of a ratio of small integers against the host's libm
            a, b = dev_res["est_overlap_prob"][merged], ora_res["est_prob"][merged]
            fin = np.isfinite(b)
            assert np.array_equal(a[~fin], b[~fin])
            assert np.all(np.abs(a[fin] - b[fin]) <= est_rel_tol * np.maximum(1.0, np.abs(b[fin])))
        else:
            e_d = dev_res["est_overlap_prob"][merged].view(np.uint64)
            e_o = ora_res["est_prob"][merged].view(np.uint64)
            w = np.nonzero(e_d != e_o)[0]
            assert w.size == 0, f"estimated_overlap_probability bits differ at {np.nonzero(merged)[0][w[:10]]}"
    else:
        dq = np.abs(dev_res["quality"][merged] - ora_res["quality"][merged])
        de = np.abs(dev_res["est_overlap_prob"][merged] - ora_res["est_prob"][merged])
        assert dq.size == 0 or dq.max() <= SUM_TOL, f"quality differs by {dq.max()}"
        assert de.size == 0 or de.max() <= SUM_TOL, f"estimated_overlap_probability differs by {de.max()}"
    if not check_seq:
        return
    ok = np.nonzero(st_o == 0)[0]
    max_dp = 0.0
    # every merged base and log-p of every accepted pair, a few thousand rows at a time (the rows gathered into one matrix)
    for lo in range(0, ok.size, 4096):
        rows = ok[lo:lo + 4096]
        m = ora_res["seq_len"][rows].astype(np.int64)
        assert not (dev_res["seq_offset"][rows] == np.uint64(0xFFFFFFFFFFFFFFFF)).any()
        off = dev_res["seq_offset"][rows].astype(np.int64)
        if seq_stride:
            assert np.array_equal(off, rows.astype(np.int64) * seq_stride)
        width = int(m.max()) if m.size else 0
        j = np.arange(width, dtype=np.int64)
        inside = j[None, :] < m[:, None]
        assert (off + m <= dev_nt.size).all(), "a row runs past the end of the arena"
        at = np.where(inside, off[:, None] + j[None, :], 0)
        wrong = (dev_nt[at] != ora_nt[rows][:, :width]) & inside
        assert not wrong.any(), f"merged nt differs at pair {rows[np.nonzero(wrong.any(axis=1))[0][0]]}"
        d = np.where(inside, np.abs(dev_p[at] - ora_p[rows][:, :width]), 0.0)
        if d.size:
            assert not np.isnan(d).any(), f"per-base log-p is NaN on one side only at pair {rows[np.nonzero(np.isnan(d).any(axis=1))[0][0]]}"
            max_dp = max(max_dp, float(d.max()))
    assert max_dp <= P_TOL, f"per-base log-p differs by {max_dp}"
    assert max_dp == 0.0, f"per-base log-p not bit-exact (max diff {max_dp}, within tolerance)"


def compare_counters(dev, ora):
    for k in ("count", "badr", "nofp", "norp", "noalgn", "lowq", "ok", "slow"):
        assert dev[k] == int(ora[k]), f"counter {k}: dev {dev[k]} oracle {int(ora[k])}"
    assert list(dev["rejected"]) == [int(x) for x in ora["rejected"]]
    assert np.array_equal(np.asarray(dev["overlapcount"]), np.asarray(ora["overlapcount"]))
    assert dev["longest_overlap"] == int(ora["longest_overlap"])


# ---- FASTQ ingest: hand-made edge cases ---------------------------------------
def _rec(header, seq, qual=None, plus="+", eol="\n"):
    if qual is None:
        qual = "I" * len(seq)
    return (header + eol + seq + eol + plus + eol + qual + eol).encode("latin1")


def fastq_cases():
    """(name, forward text, reverse text, qualmin, policy, expected stop) -- every stop condition of
    fastq.c:45-198 / seqid.c:142-285 / linebuf.c:57-89 and the shapes that must NOT stop the reader.
    policy: 0 PRESENT, 1 ABSENT, 2 OPTIONAL."""
    ok17 = ("@M01:7:FC9:2:1101:15000:1400 1:N:0:ACGT", "@M01:7:FC9:2:1101:15000:1400 2:N:0:ACGT")
    s, q = "ACGTNACGTRYKM", "IIII#IIII5555"
    good = (_rec(ok17[0], s, q), _rec(ok17[1], s, q))
    cases = []

    def add(name, f, r, stop, qualmin=33, policy=2, lead=True):
        # a good record first, so that the stop lands on record 1 and record 0 must survive
        cases.append((name, (good[0] if lead else b"") + f, (good[1] if lead else b"") + r, qualmin, policy, stop))

    add("plain", b"", b"", "END")
    add("casava14", _rec("@HWI-1:3:44:1000:2000#ACGTAC/1", s, q), _rec("@HWI-1:3:44:1000:2000#ACGTAC/2", s, q), "END")
    add("casava14_notag", _rec("@HWI-1:3:44:1000:2000/1", s, q), _rec("@HWI-1:3:44:1000:2000/2", s, q), "END")
    add("converted", _rec("@MS3:18:FCX:1:1101:14774:1712#GATA/1", s, q), _rec("@MS3:18:FCX:1:1101:14774:1712#GATA/2", s, q), "END")
    add("sra", _rec("@SRR001666.1 071112_SLXA-EAS1_s_7:5:1:817:345 length=13", s, q), _rec("@SRR001666.1 071112_SLXA-EAS1_s_7:5:1:817:345 length=13", s, q), "END")
    add("ebi", _rec("@ERR000017.25 a/1", s, q), _rec("@ERR000017.25 a/2", s, q), "END")
    add("sra_mismatch", _rec("@SRR001666.1 x", s, q), _rec("@SRR001667.1 x", s, q), "NOTPAIRED")
    add("crlf", _rec(ok17[0], s, q, eol="\r\n"), _rec(ok17[1], s, q, eol="\r\n"), "END")
    add("lowercase_iupac", _rec(ok17[0], "acgtnubdhvxwsk", "I" * 14), _rec(ok17[1], "ACGTNUBDHVXWSK", "I" * 14), "END")
    add("qual_range", _rec(ok17[0], "ACGTACGT", "!\"#Ojk~\x7f"[:8]), _rec(ok17[1], "ACGTACGT", " \x1f!PQ\x80\xff~"), "END")
    add("qual64", _rec(ok17[0], "ACGTACGT", "@ABhno~!"), _rec(ok17[1], "ACGTACGT", "hhhhhhhh"), "END", qualmin=64)
    add("empty_forward_dropped", _rec(ok17[0], "", "") + good[0], _rec(ok17[1], s, q)[:0] + _rec(ok17[1], "", "") + good[1], "END")
    add("empty_reverse_kept", _rec(ok17[0], s, q), _rec(ok17[1], "", ""), "END")
    add("seq_451_qual_450", _rec(ok17[0], "A" * 451, "I" * 450), _rec(ok17[1], "C" * 450, "I" * 450), "END")
    add("seq_460_qual_460", _rec(ok17[0], "A" * 460, "I" * 460), _rec(ok17[1], "C" * 450, "I" * 450), "NOQUAL")
    add("len_450", _rec(ok17[0], "ACGTN" * 90), _rec(ok17[1], "TTGCA" * 90), "END")
    add("bad_fwd_id", _rec("@garbage", s, q), _rec(ok17[1], s, q), "BADID_FWD")
    add("bad_rev_id", _rec(ok17[0], s, q), _rec("@M01:7:FC9:2:11x1:15000:1400 2:N:0:ACGT", s, q), "BADID_REV")
    add("empty_header", _rec("", s, q), _rec(ok17[1], s, q), "BADID_FWD")
    add("at_only", _rec("@", s, q), _rec(ok17[1], s, q), "BADID_FWD")
    add("mate_zero", _rec("@M01:7:FC9:2:1101:15000:1400 0:N:0:ACGT", s, q), _rec(ok17[1], s, q), "BADID_FWD")
    add("not_paired_y", _rec(ok17[0], s, q), _rec("@M01:7:FC9:2:1101:15000:1401 2:N:0:ACGT", s, q), "NOTPAIRED")
    add("not_paired_tag", _rec(ok17[0], s, q), _rec("@M01:7:FC9:2:1101:15000:1400 2:N:0:ACGA", s, q), "NOTPAIRED")
    add("not_paired_instr", _rec(ok17[0], s, q), _rec("@M02:7:FC9:2:1101:15000:1400 2:N:0:ACGT", s, q), "NOTPAIRED")
    add("same_direction", _rec(ok17[0], s, q), _rec(ok17[0], s, q), "NOTPAIRED")
    add("policy_present_fail", _rec("@M01:7:FC9:2:1101:15000:1400 1:N:0:", s, q), _rec("@M01:7:FC9:2:1101:15000:1400 2:N:0:", s, q), "BADID_FWD", policy=0, lead=False)
    add("policy_absent_fail", _rec(ok17[0], s, q), _rec(ok17[1], s, q), "BADID_FWD", policy=1, lead=False)
    add("policy_absent_ok", _rec("@M01:7:FC9:2:1101:15000:1400 1:N:0:", s, q), _rec("@M01:7:FC9:2:1101:15000:1400 2:N:0:", s, q), "END", policy=1, lead=False)
    add("tag_50", _rec("@M01:7:FC9:2:1101:15000:1400 1:N:0:" + "A" * 50, s, q), _rec("@M01:7:FC9:2:1101:15000:1400 2:N:0:" + "A" * 50, s, q), "END")
    add("tag_51", _rec("@M01:7:FC9:2:1101:15000:1400 1:N:0:" + "A" * 51, s, q), _rec("@M01:7:FC9:2:1101:15000:1400 2:N:0:" + "A" * 51, s, q), "BADID_FWD")
    add("instr_99", _rec("@" + "I" * 99 + ":7:FC9:2:1101:15000:1400 1:N:0:A", s, q), _rec("@" + "I" * 99 + ":7:FC9:2:1101:15000:1400 2:N:0:A", s, q), "END")
    add("instr_100", _rec("@" + "I" * 100 + ":7:FC9:2:1101:15000:1400 1:N:0:A", s, q), _rec("@" + "I" * 100 + ":7:FC9:2:1101:15000:1400 2:N:0:A", s, q), "BADID_FWD")
    add("int_wrap", _rec("@M01:7:FC9:2:1101:99999999999:007 1:N:0:A", s, q), _rec("@M01:7:FC9:2:1101:99999999999:7 2:N:0:A", s, q), "END")
    add("slash_in_17", _rec("@M01:7:FC9:2:1101:15000:1400 1:N:0:A/C", s, q), _rec(ok17[1], s, q), "BADID_FWD")
    add("bad_nt_fwd", _rec(ok17[0], "ACG1ACG", "IIIIIII"), _rec(ok17[1], s, q), "BADNT")
    add("bad_nt_rev", _rec(ok17[0], s, q), _rec(ok17[1], "ACGT ACG", "IIIIIIII"), "BADNT")
    add("bad_nt_beyond_450_ignored", _rec(ok17[0], "A" * 450 + "1", "I" * 450), _rec(ok17[1], s, q), "END")
    add("no_plus_seq", _rec(ok17[0], s, q, plus="ACGT"), _rec(ok17[1], s, q), "READLEN")
    add("no_plus_junk", _rec(ok17[0], s, q, plus="1junk"), _rec(ok17[1], s, q), "BADSEQ")
    add("empty_plus", _rec(ok17[0], s, q, plus=""), _rec(ok17[1], s, q), "BADSEQ")
    add("plus_with_text", _rec(ok17[0], s, q, plus="+M01 again"), _rec(ok17[1], s, q), "END")
    add("short_qual", _rec(ok17[0], s, q[:-1]), _rec(ok17[1], s, q), "NOQUAL")
    add("long_qual_rev", _rec(ok17[0], s, q), _rec(ok17[1], s, q + "I"), "NOQUAL")
    add("nul_in_seq", _rec(ok17[0], "ACGT\0ACG", "IIII"), _rec(ok17[1], "ACGT", "IIII"), "END")
    add("nul_in_qual", _rec(ok17[0], "ACGT", "IIII\0II"), _rec(ok17[1], "ACGT", "II\0II"), "NOQUAL")
    add("nul_in_header", _rec("@M01:7:FC9:2:1101:15000:1400 1:N:0:AC\0GT", s, q), _rec("@M01:7:FC9:2:1101:15000:1400 2:N:0:AC", s, q), "END")
    add("long_header_fwd", _rec("@" + "x" * 4500, s, q), _rec(ok17[1], s, q), "END")
    add("header_4499", _rec(ok17[0] + "y" * (4499 - len(ok17[0])), s, q), _rec(ok17[1], s, q), "BADID_FWD")
    add("long_seq_line", _rec(ok17[0], "A" * 4500, "I" * 450), _rec(ok17[1], s, q), "EOF")
    add("long_qual_line_rev", _rec(ok17[0], s, q), _rec(ok17[1], s, "I" * 4600), "EOF")
    add("qual_4400", _rec(ok17[0], "A" * 450, "I" * 4400), _rec(ok17[1], s, q), "NOQUAL")
    add("partial_tail", _rec(ok17[0], s, q)[:-5], _rec(ok17[1], s, q), "EOF")  # whole-file view; a chunked caller carries the tail over
    add("rev_shorter", good[0] + good[0], b"", "END")
    add("unterminated_last_line", _rec(ok17[0], s, q)[:-1], _rec(ok17[1], s, q)[:-1], "EOF")
    return cases


def random_fastq(n, seed):
    """Ragged reads, every header format, IUPAC letters, odd qualities; one text pair."""
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(b"ACGTACGTACGTACGTNRYKMSWBDHVUXacgtn", dtype=np.uint8)
    fo, ro = [], []
    for i in range(n):
        kind = i % 5
        x, y = int(rng.integers(0, 99999)), int(rng.integers(0, 99999))
        tag = "".join(rng.choice(list("ACGT"), size=int(rng.integers(0, 9))))
        if kind == 0:
            hf, hr = f"@M0{i % 7}:{i % 90}:FC-{i % 13}:{1 + i % 8}:{1101 + i % 50}:{x}:{y} 1:N:0:{tag}", None
            hr = hf.replace(" 1:N", " 2:Y")
        elif kind == 1:
            hf = f"@HWUSI-EAS100R:{1 + i % 8}:{i % 120}:{x}:{y}#{tag}/1"
            hr = hf[:-1] + "2"
        elif kind == 2:
            hf = f"@MS{i % 3}:18:000-A1{i % 9}:{1 + i % 8}:{1101 + i % 50}:{x}:{y}#{tag}/1"
            hr = hf[:-1] + "2"
        elif kind == 3:
            hf = hr = f"@SRR{i % 1000:06d}.{i} some comment {x}"
        else:
            hf = hr = f"@ERR{i % 1000}.{i} {y}/1"
        lf, lr = (0 if i % 97 == 5 else int(rng.integers(1, 451))), int(rng.integers(0, 451))
        for h, ln, dst in ((hf, lf, fo), (hr, lr, ro)):
            seq = letters[rng.integers(0, len(letters), size=ln)].tobytes()
            qual = rng.integers(33, 127, size=ln).astype(np.uint8).tobytes()
            eol = b"\r\n" if i % 11 == 0 else b"\n"
            dst.append(h.encode() + eol + seq + eol + b"+" + (h[1:].encode() if i % 4 == 0 else b"") + eol + qual + eol)
    return b"".join(fo), b"".join(ro)
