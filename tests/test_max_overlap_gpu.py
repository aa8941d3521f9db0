"""A configured maximum overlap (`-O`, assembler.c:59,81-82) inside the lean kernel (round 5): within the shorter read's length it is
the kernel's own arithmetic with a smaller candidate range; a maximum beyond that, where an overlap may run past a read's end, sends
the pair to the general kernel."""
import numpy as np
import pytest

from tests.util import adversarial_pairs, compare_counters, compare_results

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("algo,cfg,maxov", [("simple_bayesian", 1, 120), ("simple_bayesian", 1, 249), ("pear", 1, 60), ("rdp_mle", 4, 100), ("uparse", 4, 150),
                                            ("simple_bayesian", 1, 300), ("pear", 2, 200), ("simple_bayesian", 0, 40)])
@pytest.mark.parametrize("sums", ["fast", "exact"])
def test_maximum_overlap_in_the_lean_kernel(built, algo, cfg, maxov, sums):
    import torch
    import pandaseq_amd as pa
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    spec = synth.config_spec(cfg)
    n = 6000
    if maxov in (249, 40):     # ragged reads: the maximum lies beyond the shorter read of some pairs (deferred), within it for others
        fwd, rev, flen, rlen = adversarial_pairs(spec, n, seed=maxov, iupac=False, weird_quals=False)
    else:
        fwd, rev, flen, rlen = synth.generate_host(spec, 7_000_000 + maxov, n)
    if algo == "pear":
        rlen = np.minimum(rlen, flen)
    kw = dict(maxoverlap=maxov)
    asm = pa.Assembler(0)
    asm.set_algorithm(algo)
    asm.set_exact_sums(sums == "exact")
    asm.set_maximum_overlap(maxov)
    if cfg == 2:
        fp, rp = synth.nt_codes(spec.fprimer), synth.nt_codes(spec.rprimer, complement=True)
        asm.set_forward_primer(fp)
        asm.set_reverse_primer(rp)
        kw.update(fprimer=fp, rprimer=rp)
    b = pa.upload_batch(fwd, rev, flen, rlen, device="cuda:0", seq_stride=0)
    asm.assemble_device(b)
    torch.cuda.synchronize()
    asm.check_overflow()
    o = po.assemble_batch(po.make_config(algo, **kw), fwd, rev, flen, rlen)
    compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=sums == "exact")
    compare_counters(b.counters_dict(), o[3])
    assert asm.last_lean_variant() != "general", "the launch went to the general kernel whole"
    deferred = asm.last_deferred()
    if maxov == 300:    # beyond every read of config 1: every pair that reaches the scan is the general kernel's
        assert deferred > n // 2
    elif maxov in (249, 40):
        assert 0 < deferred < n
    else:
        assert deferred < n // 10, deferred
