"""The lean kernel's second geometry (pamd_fast.hip, LONG / PRIM: reads of up to 320 bases, primers and trims found and
applied inside it) against the oracle: what only that geometry can get wrong -- seeding five bases per lane on windows of
the code streams, the primer scan's filter / bound / lockstep fallback, clipped merges, offsets carried to the general
kernel by deferred pairs.  Every test also checks that the lean kernel is what ran."""
import numpy as np
import pytest

from util import adversarial_pairs, compare_counters, compare_results

pytestmark = pytest.mark.gpu

ALL = ["simple_bayesian", "pear", "rdp_mle", "uparse", "flash", "ea_util", "stitch"]


def _run(asm, fwd, rev, flen, rlen, seq_stride=0):
    import torch

    import pandaseq_amd as pa
    b = pa.upload_batch(fwd, rev, flen, rlen, device="cuda:0", seq_stride=seq_stride)
    asm.assemble_device(b)
    torch.cuda.synchronize()
    asm.check_overflow()
    return b


def _oracle(algo, fwd, rev, flen, rlen, **kw):
    from oracle import pyoracle as po
    return po.assemble_batch(po.make_config(algo, **kw), fwd, rev, flen, rlen)


@pytest.fixture(params=["fast", "exact"])
def sums(request):
    """Both summation modes (pamd_set_exact_sums): any order with guards, and the reference's order."""
    return request.param


def _check(asm, b, o, n, max_deferred, exact=False):
    compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=exact)
    compare_counters(b.counters_dict(), o[3])
    d = asm.last_deferred()
    assert 0 <= d <= max_deferred, f"the lean kernel must have served this launch (deferred {d} of {n})"


@pytest.mark.parametrize("algo", ALL)
def test_long_geometry_all_scorers(built, algo, sums):
    """2x300 reads without primers: ordinary pairs, then ragged lengths up to 304, Ns, quality-2 tails, ambiguity codes and
    out-of-range qualities (deferred), tandem repeats and unrelated reads (many candidates: deferred in this geometry)."""
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.config_spec(2)
    for weird, n, limit in ((False, 4000, 400), (True, 3000, 3000)):
        if weird:
            fwd, rev, flen, rlen = adversarial_pairs(spec, n, seed=41)
            if algo == "pear":
                rlen = np.minimum(rlen, flen)   # algo_pear.c:52 reads the forward read at the reverse index
        else:
            fwd, rev, flen, rlen = synth.generate_host(spec, 200, n)
        asm = pa.Assembler(0)
        asm.set_algorithm(algo)
        asm.set_exact_sums(sums == "exact")
        b = _run(asm, fwd, rev, flen, rlen)
        _check(asm, b, _oracle(algo, fwd, rev, flen, rlen), n, limit, exact=sums == "exact")


def _plant(rng, rows, lens, primer, where, subst, lowq):
    members = [[b for b in (1, 2, 4, 8) if b & c] for c in primer]
    for i in range(len(rows)):
        if where[i] < 0 or where[i] + len(primer) > lens[i]:
            continue
        for x, m in enumerate(members):
            p = where[i] + x
            rows[i, p, 0] = rng.choice(m)
            if rng.random() < subst:
                rows[i, p, 0] = rng.choice([1, 2, 4, 8])
            if rng.random() < lowq:
                rows[i, p, 1] = rng.choice([0, 1, 2, 5, 9, 14])


@pytest.mark.parametrize("algo", ["simple_bayesian", "pear"])
@pytest.mark.parametrize("plen", [1, 3, 8, 9, 16, 17, 23, 24])
def test_primer_scan_inside_the_lean_kernel(built, plen, algo, sums):
    """offset.c:47-112 as the lean kernel runs it: primers of every length it serves, with ambiguity codes and N among
    their bases, planted at the start of the read, anywhere in it or nowhere, damaged and on poor qualities (loose bounds,
    many alignments left: the lockstep fallback), a read shorter than its primer and one exactly as long."""
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    rng = np.random.default_rng(300 + plen)
    spec = synth.config_spec(2 if plen % 2 else 1)     # 2x300 and 2x250 reads
    L = spec.read_len
    fwd, rev, flen, rlen = synth.generate_host(spec, 52000, 2500)
    fwd[:, :, 0] = np.where(np.arange(fwd.shape[1])[None, :] < L, fwd[:, :, 0], 0)
    n = len(flen)
    codes = np.array([1, 2, 4, 8, 1, 2, 4, 8, 3, 5, 10, 12, 7, 14, 15], np.uint8)   # mostly plain bases, some ambiguity codes, N
    fp = rng.choice(codes, plen)
    rp = rng.choice(np.array([1, 2, 4, 8], np.uint8), plen)
    room = max(1, L - plen)
    wf = np.where(rng.random(n) < 0.9, rng.integers(0, room, n) * (rng.random(n) < 0.4), -1)   # most at the start, some anywhere, a tenth nowhere
    wr = np.where(rng.random(n) < 0.9, rng.integers(0, room, n) * (rng.random(n) < 0.4), -1)
    _plant(rng, fwd, flen, fp, wf, 0.05, 0.2)
    _plant(rng, rev, rlen, rp, wr, 0.05, 0.2)
    flen[5] = max(1, plen - 1)
    rlen[6] = plen
    flen[7] = plen + 1
    asm = pa.Assembler(0)
    asm.set_algorithm(algo)
    asm.set_exact_sums(sums == "exact")
    asm.set_forward_primer(fp)
    asm.set_reverse_primer(rp)
    b = _run(asm, fwd, rev, flen, rlen)
    o = _oracle(algo, fwd, rev, flen, rlen, fprimer=fp, rprimer=rp)
    _check(asm, b, o, n, n // 3, exact=sums == "exact")
    st = o[0]["status"]
    assert (st == 0).sum() > n // 4 and ((st == 2) | (st == 3)).sum() > 0 or plen < 8


def sweep_case(rnd):
    """One round of tests/primer_sweep.py (a randomized campaign; rounds that found something are pinned below): primers of
    random length and content planted at the start of the read, anywhere or nowhere, damaged at random rates and on random
    shares of poor qualities; every fourth round with ragged read lengths."""
    from pandaseq_amd import synth
    codes = np.array([1, 2, 4, 8, 1, 2, 4, 8, 3, 5, 10, 12, 7, 14, 15], np.uint8)
    rng = np.random.default_rng(7000 + rnd)
    spec = synth.config_spec(2 if rnd % 3 else 1)
    L = spec.read_len
    n = 3000
    fwd, rev, flen, rlen = synth.generate_host(spec, int(rng.integers(0, 40_000_000)), n)
    fwd[:, :, 0] = np.where(np.arange(fwd.shape[1])[None, :] < L, fwd[:, :, 0], 0)
    fplen, rplen = int(rng.integers(1, 25)), int(rng.integers(1, 25))
    fp = rng.choice(codes if rnd % 2 else codes[:4], fplen)
    rp = rng.choice(codes[:4], rplen)
    anywhere = rng.choice([0.0, 0.1, 0.4, 1.0])
    wf = np.where(rng.random(n) < 0.9, rng.integers(0, max(1, L - fplen), n) * (rng.random(n) < anywhere), -1)
    wr = np.where(rng.random(n) < 0.9, rng.integers(0, max(1, L - rplen), n) * (rng.random(n) < anywhere), -1)
    subst, lowq = rng.choice([0.0, 0.03, 0.08, 0.2]), rng.choice([0.0, 0.1, 0.3, 0.7])
    _plant(rng, fwd, flen, fp, wf, subst, lowq)
    _plant(rng, rev, rlen, rp, wr, subst, lowq)
    if rnd % 4 == 3:    # ragged lengths: the scan's last alignments, reads shorter than the primer
        flen[:] = rng.integers(1, L + 1, n)
        rlen[:] = rng.integers(1, L + 1, n)
    algo = ["pear", "simple_bayesian", "rdp_mle"][rnd % 3]
    if algo == "pear":
        rlen = np.minimum(rlen, flen)
    thr = float(rng.choice([0.6, 0.6, 0.9, 0.3]))
    return dict(algo=algo, fwd=fwd, rev=rev, flen=flen, rlen=rlen, fp=fp, rp=rp, thr=thr,
                what=f"{algo} plen {fplen}/{rplen} anywhere {anywhere} subst {subst} lowq {lowq} thr {thr}")


@pytest.mark.parametrize("rnd", [91, 3, 27])
def test_primer_sweep_rounds_that_found_something(built, rnd, sums):
    """Round 91 (a two-base primer under threshold 0.9 in ragged reads): the first good alignment does not reach the threshold,
    exactly one OTHER alignment survives the filter -- and the shortcut took "one left" to mean the one it had summed: NORP
    where the reference finds the primer.  Rounds 3 and 27: the same shapes of input on the other two scorers."""
    import pandaseq_amd as pa
    c = sweep_case(rnd)
    asm = pa.Assembler(0)
    asm.set_algorithm(c["algo"])
    asm.set_exact_sums(sums == "exact")
    asm.set_threshold(c["thr"])
    asm.set_forward_primer(c["fp"])
    asm.set_reverse_primer(c["rp"])
    b = _run(asm, c["fwd"], c["rev"], c["flen"], c["rlen"])
    o = _oracle(c["algo"], c["fwd"], c["rev"], c["flen"], c["rlen"], fprimer=c["fp"], rprimer=c["rp"], threshold=c["thr"])
    compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=sums == "exact")
    compare_counters(b.counters_dict(), o[3])


def test_primers_the_lean_kernel_does_not_serve(built):
    """A primer of 25 bases, or a penalty: the launch goes to the primer kernel and the general kernel whole, results unchanged."""
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.config_spec(2)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 1200)
    fp = np.concatenate([synth.nt_codes(spec.fprimer), np.array([1, 2, 4, 8, 1, 2, 4, 8], np.uint8)])
    rp = synth.nt_codes(spec.rprimer, complement=True)
    fwd[:, 17:25, 0] = fp[17:]
    for penalty, primer in ((0.0, fp), (0.015, fp[:17])):
        asm = pa.Assembler(0)
        asm.set_algorithm("pear")
        asm.set_exact_sums(False)
        asm.set_forward_primer(primer)
        asm.set_reverse_primer(rp)
        asm.set_primer_penalty(penalty)
        b = _run(asm, fwd, rev, flen, rlen)
        o = _oracle("pear", fwd, rev, flen, rlen, fprimer=primer, rprimer=rp, primer_penalty=penalty)
        compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=False)
        compare_counters(b.counters_dict(), o[3])
        assert asm.last_deferred() == -1


@pytest.mark.parametrize("ftrim,rtrim", [(5, 0), (0, 9), (20, 31), (140, 150), (249, 3)])
def test_trims_inside_the_lean_kernel(built, ftrim, rtrim, sums):
    """Trims without primers (assembler.c:270,282): offsets enter the merge; a trim as long as the rest of the read leaves
    NOALGN (assembler.c:73-82), an overlap that runs into a trimmed stretch is clipped (VEEZ / WEDGEZ)."""
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = synth.generate_host(spec, 9000, 3000)
    rng = np.random.default_rng(ftrim * 7 + rtrim)
    flen[::7] = rng.integers(2, 251, len(flen[::7]))
    rlen[::5] = rng.integers(2, 251, len(rlen[::5]))
    asm = pa.Assembler(0)
    asm.set_exact_sums(sums == "exact")
    asm.set_forward_trim(ftrim)
    asm.set_reverse_trim(rtrim)
    b = _run(asm, fwd, rev, flen, rlen)
    o = _oracle("simple_bayesian", fwd, rev, flen, rlen, ftrim=ftrim, rtrim=rtrim)
    _check(asm, b, o, len(flen), len(flen) // 3, exact=sums == "exact")


def test_one_primer_only_and_offsets_of_deferred_pairs(built):
    """Only `-p`, only `-q`; and pairs the lean kernel defers AFTER it found their primers (tandem repeats: many candidates)
    take the offsets along to the general kernel, pairs deferred BEFORE (an ambiguity code) get them from the primer kernel."""
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.config_spec(2)
    n = 3000
    fwd, rev, flen, rlen = synth.generate_host(spec, 77000, n)
    rng = np.random.default_rng(9)
    L = spec.read_len
    for i in range(0, n, 6):    # tandem repeats behind intact primers
        unit = rng.choice(np.array([1, 2, 4, 8], np.uint8), size=int(rng.integers(2, 9)))
        rep = np.tile(unit, L // len(unit) + 2)
        fwd[i, 17:L, 0] = rep[:L - 17]
        rev[i, 17:L, 0] = rep[3:L - 14]
    for i in range(1, n, 6):    # an ambiguity code somewhere
        fwd[i, int(rng.integers(20, L)), 0] = 5
    fp, rp = synth.nt_codes(spec.fprimer), synth.nt_codes(spec.rprimer, complement=True)
    for kw in (dict(fprimer=fp), dict(rprimer=rp), dict(fprimer=fp, rprimer=rp)):
        asm = pa.Assembler(0)
        asm.set_algorithm("pear")
        asm.set_exact_sums(False)
        if "fprimer" in kw:
            asm.set_forward_primer(fp)
        if "rprimer" in kw:
            asm.set_reverse_primer(rp)
        b = _run(asm, fwd, rev, flen, rlen)
        o = _oracle("pear", fwd, rev, flen, rlen, **kw)
        _check(asm, b, o, n, n // 2)
        assert asm.last_deferred() > n // 8, "repeats and ambiguity codes are the general kernel's"
