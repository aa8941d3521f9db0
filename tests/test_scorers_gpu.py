"""GPU parity of the four remaining scorers (uparse, flash, ea_util, stitch) against
the oracle's restatement of algo_uparse.c / algo_flash.c / algo_ea_util.c / algo_stitch.c."""
import numpy as np
import pytest

from util import adversarial_pairs, compare_counters, compare_results

pytestmark = pytest.mark.gpu

SCORERS = ["uparse", "flash", "ea_util", "stitch"]
# est_overlap_prob of ea_util / stitch is log() of a ratio of small integers: the device looks it up in a table the HOST's
# libm filled (PamdKernelParams::count_table), so it is bit-exact like everything else.
EST_TOL = {"uparse": 0.0, "flash": 0.0, "ea_util": 0.0, "stitch": 0.0}


def _run(asm, fwd, rev, flen, rlen, seq_stride):
    import torch

    import pandaseq_amd as pa
    b = pa.upload_batch(fwd, rev, flen, rlen, device="cuda:0", seq_stride=seq_stride)
    asm.assemble_device(b)
    torch.cuda.synchronize()
    return b


def _oracle(algo, fwd, rev, flen, rlen, **kw):
    from oracle import pyoracle as po
    return po.assemble_batch(po.make_config(algo, **kw), fwd, rev, flen, rlen)


@pytest.fixture(params=["exact", "fast"])
def sums(request):
    return request.param


@pytest.mark.parametrize("algo", SCORERS)
@pytest.mark.parametrize("cfg", [0, 1, 4])
def test_synthetic_configs(built, algo, cfg, sums):
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.config_spec(cfg)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 5000)
    asm = pa.Assembler(0)
    asm.set_algorithm(algo)
    asm.set_exact_sums(sums == "exact")
    b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
    o_res, o_nt, o_p, o_ctr = _oracle(algo, fwd, rev, flen, rlen)
    compare_results(b.results_numpy(), *b.rows_numpy(), o_res, o_nt, o_p,
                    exact_sums=sums == "exact", est_rel_tol=EST_TOL[algo])
    compare_counters(b.counters_dict(), o_ctr)
    if algo in ("uparse", "ea_util"):
        assert o_ctr["ok"] > 1000  # these two behave like assemblers on clean data; flash / stitch carry the reference's quirks


@pytest.mark.parametrize("algo", SCORERS)
def test_adversarial_strided(built, algo, sums):
    """Ns, IUPAC codes (the base-by-base fallback of the packed scan), ragged and too-short reads, odd qualities."""
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = adversarial_pairs(spec, 5000, seed=23)
    asm = pa.Assembler(0)
    asm.set_algorithm(algo)
    asm.set_exact_sums(sums == "exact")
    b = _run(asm, fwd, rev, flen, rlen, seq_stride=512)
    o_res, o_nt, o_p, o_ctr = _oracle(algo, fwd, rev, flen, rlen)
    compare_results(b.results_numpy(), *b.rows_numpy(), o_res, o_nt, o_p, seq_stride=512,
                    exact_sums=sums == "exact", est_rel_tol=EST_TOL[algo])
    compare_counters(b.counters_dict(), o_ctr)


def test_uparse_error_estimation_and_modules(built):
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 4000)
    asm = pa.Assembler(0)
    asm.set_algorithm("uparse", 0.25)
    asm.add_module("min_phred:3")
    b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
    o = _oracle("uparse", fwd, rev, flen, rlen, q=0.25, modules=[("min_phred", 3)])
    compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3])
    compare_counters(b.counters_dict(), o[3])


def test_long_reads_and_primers(built):
    """450-base reads (two DPP rows per candidate in the packed scan) and primer offsets with every scorer."""
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.SynthSpec(seed=91, read_len=450, tmin=450, tmax=899, qhi=40)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 1200)
    spec2 = synth.config_spec(2)
    f2, r2, fl2, rl2 = synth.generate_host(spec2, 0, 1500)
    for algo in SCORERS:
        asm = pa.Assembler(0)
        asm.set_algorithm(algo)
        b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
        o = _oracle(algo, fwd, rev, flen, rlen)
        compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], est_rel_tol=EST_TOL[algo])
        compare_counters(b.counters_dict(), o[3])
        asm.set_forward_primer(synth.nt_codes(spec2.fprimer))
        asm.set_reverse_primer(synth.nt_codes(spec2.rprimer, complement=True))
        b = _run(asm, f2, r2, fl2, rl2, seq_stride=0)
        o = _oracle(algo, f2, r2, fl2, rl2, fprimer=synth.nt_codes(spec2.fprimer), rprimer=synth.nt_codes(spec2.rprimer, complement=True))
        compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], est_rel_tol=EST_TOL[algo])
        compare_counters(b.counters_dict(), o[3])
