"""The lean kernel (pamd_fast.hip) against the general one and the oracle: a launch may be served by either, results must
not depend on which.  Covers what only the two-kernel launch can get wrong: the deferred list, pairs outside the lean
domain, the in-wave many-candidate scan, the arena shared by two launches, NOSPACE."""
import numpy as np
import pytest

from util import adversarial_pairs, compare_counters, compare_results

pytestmark = pytest.mark.gpu

COUNTING = ["simple_bayesian", "uparse", "flash", "ea_util", "stitch"]


def _run(asm, fwd, rev, flen, rlen, seq_stride=0, capacity=None):
    import torch

    import pandaseq_amd as pa
    b = pa.upload_batch(fwd, rev, flen, rlen, device="cuda:0", seq_stride=seq_stride, capacity=capacity)
    asm.assemble_device(b)
    torch.cuda.synchronize()
    return b


def _oracle(algo, fwd, rev, flen, rlen, **kw):
    from oracle import pyoracle as po
    return po.assemble_batch(po.make_config(algo, **kw), fwd, rev, flen, rlen)


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("algo", COUNTING)
def test_lean_kernel_matches_oracle_and_general(built, algo, exact):
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    est_tol = 0.0   # (ea_util and stitch too: their log() comes from a table the host filled)
    for cfg, n, weird in ((1, 6000, False), (4, 4000, False), (1, 4000, True)):
        spec = synth.config_spec(cfg)
        if weird:
            fwd, rev, flen, rlen = adversarial_pairs(spec, n, seed=21)
        else:
            fwd, rev, flen, rlen = synth.generate_host(spec, 100, n)
        o_res, o_nt, o_p, o_ctr = _oracle(algo, fwd, rev, flen, rlen)
        deferred = None
        for fast_path in (True, False):
            asm = pa.Assembler(0)
            asm.set_algorithm(algo)
            asm.set_exact_sums(exact)
            asm.set_fast_path(fast_path)
            b = _run(asm, fwd, rev, flen, rlen)
            asm.check_overflow()
            compare_results(b.results_numpy(), *b.rows_numpy(), o_res, o_nt, o_p, exact_sums=exact, est_rel_tol=est_tol)
            compare_counters(b.counters_dict(), o_ctr)
            if fast_path:
                deferred = asm.last_deferred()
            else:
                assert asm.last_deferred() == -1
        # the lean kernel ran, kept the ordinary pairs for itself and left the unusual ones (if any) to the general kernel
        assert deferred is not None and 0 <= deferred < n
        if weird:
            assert deferred > n // 20, "IUPAC codes and out-of-range qualities must go to the general kernel"
        else:
            assert deferred < n // 20


def test_lean_kernel_many_candidates_and_repeats(built):
    """Pairs that scan every overlap (no k-mer in common) and tandem repeats (k-mers that occur dozens of times) stay in
    the lean kernel: a candidate per lane, and k-mer tables that may lose what the reference loses too."""
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    n = 3000
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, n)
    rng = np.random.default_rng(5)
    L = spec.read_len
    for i in range(0, n, 3):   # tandem repeats of every unit length, and a few poly-N stretches inside them
        unit = rng.choice(np.array([1, 2, 4, 8], np.uint8), size=int(rng.integers(1, 14)))
        rep = np.tile(unit, L // len(unit) + 2)
        off = int(rng.integers(0, len(unit)))
        fwd[i, :L, 0] = rep[:L]
        rev[i, :L, 0] = rep[off:off + L]
        if i % 9 == 0:
            fwd[i, 40:52, 0] = 15
    for i in range(1, n, 3):   # unrelated reads
        rev[i, :L, 0] = rng.choice(np.array([1, 2, 4, 8], np.uint8), size=L)
    flen[5], rlen[5] = 9, 200
    flen[8], rlen[8] = 200, 12
    for algo in ("simple_bayesian", "stitch"):
        o = _oracle(algo, fwd, rev, flen, rlen)
        asm = pa.Assembler(0)
        asm.set_algorithm(algo)
        asm.set_exact_sums(False)
        b = _run(asm, fwd, rev, flen, rlen)
        compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=False, est_rel_tol=0.0)
        compare_counters(b.counters_dict(), o[3])
        assert o[3]["slow"] > n // 4
        assert asm.last_deferred() < n // 50, "these pairs are the lean kernel's own"


def test_arena_overflow_through_the_device_path(built):
    """An arena that is too small: the launch reports it (pamd_assemble_status), pairs that did not fit say NOSPACE,
    are counted as seen and as nothing else, and every pair that did fit is right."""
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    n = 5000
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, n)
    o_res, o_nt, o_p, o_ctr = _oracle("simple_bayesian", fwd, rev, flen, rlen)
    for fast_path in (True, False):
        for exact in (True, False):
            asm = pa.Assembler(0)
            asm.set_exact_sums(exact)
            asm.set_fast_path(fast_path)
            b = _run(asm, fwd, rev, flen, rlen, capacity=600_000)
            with pytest.raises(pa.PamdError, match="arena too small"):
                asm.check_overflow()
            asm.check_overflow()   # the flag was reset
            res = b.results_numpy()
            nospace = res["status"] == 7
            assert 0 < int(nospace.sum()) < n
            assert np.isin(o_res["status"][nospace], (0, 5, 6)).all(), "only pairs that were merged can run out of room"
            same = ~nospace
            assert np.array_equal(res["status"][same], o_res["status"][same].astype(np.uint8))
            c = b.counters_dict()
            assert c["count"] == n and c["ok"] == int((res["status"] == 0).sum()) == int(c["overlapcount"].sum())
            nt, p = b.rows_numpy()
            for i in np.nonzero(res["status"] == 0)[0][::7]:
                off, m = int(res["seq_offset"][i]), int(res["seq_len"][i])
                assert off + m <= 600_000
                assert np.array_equal(nt[off:off + m], o_nt[i, :m]) and np.array_equal(p[off:off + m], o_p[i, :m])


@pytest.mark.parametrize("exact", [True, False])
def test_streaming_a_shard_in_chunks(built, exact):
    """pamd_assemble_device_range (the streaming form of pool.c:71-108): a resident shard assembled in three launches whose
    rows go to two arenas used in turn gives the counters and records of the single launch; rows of a chunk are intact
    until its arena is used again."""
    import torch

    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    n, chunk = 30_000, 11_000
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, n)
    asm = pa.Assembler(0)
    asm.set_exact_sums(exact)
    whole = pa.upload_batch(fwd, rev, flen, rlen, device="cuda:0", seq_stride=0)
    asm.assemble_device(whole)
    torch.cuda.synchronize()
    asm.check_overflow()
    b = pa.upload_batch(fwd, rev, flen, rlen, device="cuda:0", seq_stride=0, capacity=16)
    arenas = [pa.alloc_outputs(0, chunk * 440 + 2 * 16384 * 4096, "cuda:0") for _ in range(2)]
    res_w = whole.results_numpy()
    nt_w, p_w = whole.seq_nt.cpu().numpy(), whole.seq_p.cpu().numpy()
    for k in range(3):
        ar = arenas[k % 2]
        ar["cursor"].zero_()
        lo, cnt = k * chunk, min(chunk, n - k * chunk)
        asm.assemble_device(b, first=lo, count=cnt, arena=ar)
        torch.cuda.synchronize()
        asm.check_overflow()
        res = b.results_numpy()[lo:lo + cnt]
        nt, p = ar["seq_nt"].cpu().numpy(), ar["seq_p"].cpu().numpy()
        for name in res.dtype.names:
            if name not in ("seq_offset", "reserved", "quality"):
                assert np.array_equal(res[name], res_w[name][lo:lo + cnt]), name
        if exact:
            assert np.array_equal(res["quality"].view(np.uint64), res_w["quality"][lo:lo + cnt].view(np.uint64))
        else:
            assert np.abs(res["quality"] - res_w["quality"][lo:lo + cnt]).max() <= 1e-9
        for i in np.nonzero(res["status"] == 0)[0][::11]:
            m, o1, o2 = int(res["seq_len"][i]), int(res["seq_offset"][i]), int(res_w["seq_offset"][lo + i])
            assert np.array_equal(nt[o1:o1 + m], nt_w[o2:o2 + m]) and np.array_equal(p[o1:o1 + m], p_w[o2:o2 + m])
    assert np.array_equal(b.counters.cpu().numpy(), whole.counters.cpu().numpy())
    with pytest.raises(pa.PamdError, match="range outside"):
        asm.assemble_device(b, first=n - 5, count=6, arena=arenas[0])
