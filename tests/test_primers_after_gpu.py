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
    compare_results(b.results_numpy(), b.seq_nt.cpu().numpy(), b.seq_p.cpu().numpy(), o_res, o_nt, o_p,
                    seq_stride=seq_stride or None, exact_sums=exact, est_rel_tol=est_rel_tol)
    compare_counters(b.counters_dict(), o_ctr)
    return o_res, o_ctr


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("algo", ["simple_bayesian", "pear", "rdp_mle", "uparse"])
def test_config2_primers_after(built, algo, exact):
    from pandaseq_amd import synth
    spec = synth.config_spec(2)
    fwd, rev, flen, rlen = synth.generate_numpy(spec, 0, 4000)
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


def test_primers_after_one_primer_and_trims(built):
    """Only one primer set (the other side trimmed by a fixed count), and no primer at all (both trims
    applied to the merged sequence, assembler.c:311,322); trims that eat the whole sequence count as NOFP."""
    from pandaseq_amd import synth
    spec = synth.config_spec(2)
    fwd, rev, flen, rlen = synth.generate_numpy(spec, 0, 3000)
    _pair("simple_bayesian", fwd, rev, flen, rlen, fprimer=synth.nt_codes(spec.fprimer), rtrim=9)
    _pair("simple_bayesian", fwd, rev, flen, rlen, rprimer=synth.nt_codes(spec.rprimer, complement=True), ftrim=5)
    _pair("pear", fwd, rev, flen, rlen, ftrim=20, rtrim=31)
    o_res, o_ctr = _pair("simple_bayesian", fwd, rev, flen, rlen, ftrim=300, rtrim=300)
    assert o_ctr["nofp"] > 2500 and o_ctr["ok"] == 0


def test_primers_after_short_reads_and_counting_scorers(built):
    from pandaseq_amd import synth
    spec = synth.config_spec(0)
    fwd, rev, flen, rlen = synth.generate_numpy(spec, 0, 2500)
    fp = synth.nt_codes("ACGTTGCA")
    for algo, tol in (("flash", 0.0), ("ea_util", 0.0), ("stitch", 0.0)):
        _pair(algo, fwd, rev, flen, rlen, fprimer=fp, rprimer=fp, est_rel_tol=tol)
