"""Shared helpers for the parity tests: adversarial inputs and the oracle <-> device comparison."""
import numpy as np

from pandaseq_amd import synth

P_TOL = 1e-6  # north_star: per-base log-probabilities within 1e-6 (we expect exact equality)


def adversarial_pairs(spec, n, seed=1, ragged=True, weird_quals=True, iupac=True):
    """Synthetic pairs plus the edge cases the hot path has: Ns, IUPAC ambiguity codes, Q2 tails
    (B-cliff), ragged and too-short reads, qualities outside 0..46, identical-quality mismatches."""
    fwd, rev, flen, rlen = synth.generate_numpy(spec, 0, n)
    rng = np.random.default_rng(seed)
    L = spec.read_len
    if iupac:
        # sprinkle ambiguity codes (3..14 non one-hot) and extra Ns
        for arr in (fwd, rev):
            m = rng.random((n, L)) < 0.004
            arr[:, :L, 0][m] = rng.integers(3, 15, size=int(m.sum())).astype(np.uint8)
            m = rng.random((n, L)) < 0.003
            arr[:, :L, 0][m] = 15
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


def compare_results(dev_res, dev_nt, dev_p, ora_res, ora_nt, ora_p, seq_stride=None, check_seq=True, exact_sums=True):
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
    for i in ok:
        m = int(ora_res["seq_len"][i])
        off = int(dev_res["seq_offset"][i])
        assert off != 0xFFFFFFFFFFFFFFFF
        if seq_stride:
            assert off == i * seq_stride
        assert np.array_equal(dev_nt[off:off + m], ora_nt[i, :m]), f"merged nt differs at pair {i}"
        d = np.abs(dev_p[off:off + m] - ora_p[i, :m])
        if d.size:
            max_dp = max(max_dp, float(d.max()))
    assert max_dp <= P_TOL, f"per-base log-p differs by {max_dp}"
    assert max_dp == 0.0, f"per-base log-p not bit-exact (max diff {max_dp}, within tolerance)"


def compare_counters(dev, ora):
    for k in ("count", "badr", "nofp", "norp", "noalgn", "lowq", "ok", "slow"):
        assert dev[k] == int(ora[k]), f"counter {k}: dev {dev[k]} oracle {int(ora[k])}"
    assert list(dev["rejected"]) == [int(x) for x in ora["rejected"]]
    assert np.array_equal(np.asarray(dev["overlapcount"]), np.asarray(ora["overlapcount"]))
    assert dev["longest_overlap"] == int(ora["longest_overlap"])
