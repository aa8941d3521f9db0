"""GPU parity of `-a`: primers looked for in the MERGED sequence and cut off it
(assembler.c:300-333, panda_compute_offset_result offset.c:114-133, panda_log1mexp offset.c:35-38)."""
import numpy as np
import pytest

from util import adversarial_pairs, compare_counters, compare_results

pytestmark = pytest.mark.gpu


def _run(asm, fwd, rev, flen, rlen, seq_stride=0):
    import torch

    import pandaseq_amd as pa
    b = pa.upload_batch(fwd, rev, flen, rlen, device="cuda:0", seq_stride=seq_stride)
    asm.assemble_device(b)
    torch.cuda.synchronize()
    return b


def _pair(algo, fwd, rev, flen, rlen, exact=True, seq_stride=0, est_rel_tol=0.0, **kw):
    """Device and oracle with the same settings, compared."""
    import pandaseq_amd as pa
    from oracle import pyoracle as po
    asm = pa.Assembler(0)
    asm.set_algorithm(algo)
    asm.set_exact_sums(exact)
    asm.set_primers_after(True)
    if kw.get("fprimer") is not None:
        asm.set_forward_primer(kw["fprimer"])
    if kw.get("rprimer") is not None:
        asm.set_reverse_primer(kw["rprimer"])
    if kw.get("ftrim"):
        asm.set_forward_trim(kw["ftrim"])
    if kw.get("rtrim"):
        asm.set_reverse_trim(kw["rtrim"])
    if kw.get("primer_penalty"):
        asm.set_primer_penalty(kw["primer_penalty"])
    for kind, val in kw.get("modules", ()):
        asm.add_module(f"{kind}:{val}")
    b = _run(asm, fwd, rev, flen, rlen, seq_stride)
    o_res, o_nt, o_p, o_ctr = po.assemble_batch(po.make_config(algo, primers_after=True, **kw), fwd, rev, flen, rlen)
    compare_results(b.results_numpy(), *b.rows_numpy(), o_res, o_nt, o_p,
                    seq_stride=seq_stride or None, exact_sums=exact, est_rel_tol=est_rel_tol)
    compare_counters(b.counters_dict(), o_ctr)
    return o_res, o_ctr


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("algo", ["simple_bayesian", "pear", "rdp_mle", "uparse"])
def test_config2_primers_after(built, algo, exact):
    from pandaseq_amd import synth
    spec = synth.config_spec(2)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 4000)
    o_res, o_ctr = _pair(algo, fwd, rev, flen, rlen, exact=exact,
                         fprimer=synth.nt_codes(spec.fprimer), rprimer=synth.nt_codes(spec.rprimer, complement=True))
    ok = o_res["status"] == 0
    assert ok.sum() > 500
    # the primers really are cut off the merged sequence: 17 + 17 bases (1-3 more where the template had a prefix)
    assert np.median(o_res["fwd_offset"][ok]) == 17 and np.median(o_res["rev_offset"][ok]) == 17


@pytest.mark.parametrize("penalty", [0.0, 0.01])
def test_primers_after_adversarial_with_modules(built, penalty):
    """Ns / IUPAC codes / Q2 tails / ragged reads, min_phred over the KEPT bases, fixed-stride rows, both modes."""
    from pandaseq_amd import synth
    spec = synth.config_spec(2)
    fwd, rev, flen, rlen = adversarial_pairs(spec, 4000, seed=31)
    rlen = np.minimum(rlen, flen)
    for exact in (True, False):
        o_res, o_ctr = _pair("rdp_mle", fwd, rev, flen, rlen, exact=exact, seq_stride=608, primer_penalty=penalty,
                             fprimer=synth.nt_codes(spec.fprimer), rprimer=synth.nt_codes(spec.rprimer, complement=True),
                             modules=[("min_phred", 3), ("min_overlapbits", 15)])
        assert o_ctr["nofp"] + o_ctr["norp"] > 0 and o_ctr["ok"] > 0


@pytest.mark.parametrize("exact", [True, False])
def test_primers_after_one_primer_and_trims(built, exact):
    """Only one primer set (the other side trimmed by a fixed count), and no primer at all (both trims
    applied to the merged sequence, assembler.c:311,322); trims that eat the whole sequence count as NOFP.
    (exact: the general kernel's `-a`; any-order sums: the lean kernel's, DESIGN.md 3.1, docs/NOTEBOOK.md 3.1e)"""
    from pandaseq_amd import synth
    spec = synth.config_spec(2)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 3000)
    _pair("simple_bayesian", fwd, rev, flen, rlen, exact=exact, fprimer=synth.nt_codes(spec.fprimer), rtrim=9)
    _pair("simple_bayesian", fwd, rev, flen, rlen, exact=exact, rprimer=synth.nt_codes(spec.rprimer, complement=True), ftrim=5)
    _pair("pear", fwd, rev, flen, rlen, exact=exact, ftrim=20, rtrim=31)
    _pair("rdp_mle", fwd, rev, flen, rlen, exact=exact, fprimer=synth.nt_codes(spec.fprimer), rtrim=9, modules=[("min_phred", 3)])
    o_res, o_ctr = _pair("simple_bayesian", fwd, rev, flen, rlen, exact=exact, ftrim=300, rtrim=300)
    assert o_ctr["nofp"] > 2500 and o_ctr["ok"] == 0


@pytest.mark.parametrize("exact", [True, False])
def test_primers_after_short_reads_and_counting_scorers(built, exact):
    from pandaseq_amd import synth
    spec = synth.config_spec(0)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 2500)
    fp = synth.nt_codes("ACGTTGCA")
    for algo, tol in (("flash", 0.0), ("ea_util", 0.0), ("stitch", 0.0), ("uparse", 0.0)):
        _pair(algo, fwd, rev, flen, rlen, exact=exact, fprimer=fp, rprimer=fp, est_rel_tol=tol)


def test_primers_after_ambiguity_codes_in_the_primer_and_long_primers(built):
    """A primer with IUPAC codes and an N (a primer N is skipped, offset.c:79; an ambiguity code agrees with any of its bases),
    a 24-base primer (the lean scan's limit: its first sixteen bases filter, the rest are checked with the sum) and a 30-base one
    (the general kernel's), planted at the start of the merged sequence; any-order sums."""
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 2000)
    rng = np.random.default_rng(5)
    for plen in (12, 24, 30):
        f2 = fwd.copy()
        primer = rng.choice(np.array([1, 2, 4, 8], dtype=np.uint8), size=plen)
        f2[:, :plen, 0] = primer      # the forward read (and so the merged sequence) starts with the primer
        f2[::7, 3, 0] = 15            # ... sometimes with an N in it (agrees with everything)
        f2[::11, 5, 0] = 1 if primer[5] != 1 else 2     # ... or a wrong base: the alignment's value is NaN, the primer is not found
        f2[::13, 6, 0] = primer[6] | (1 if primer[6] != 1 else 2)     # ... or an ambiguity code that includes the right base (the general kernel's pair)
        loose = primer.copy()
        loose[2] = 15                 # the primer as configured: an N
        loose[7] = primer[7] | (1 if primer[7] != 1 else 2)     # and an ambiguity code that includes the planted base
        o_res, o_ctr = _pair("simple_bayesian", f2, rev, flen, rlen, exact=False, fprimer=loose, rtrim=4)
        assert o_ctr["ok"] > 500 and o_ctr["nofp"] > 100
        assert np.median(o_res["fwd_offset"][o_res["status"] == 0]) == plen
