"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same inputs."""
import numpy as np
import pytest

from util import adversarial_pairs, compare_counters, compare_results

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("algo", ["simple_bayesian", "pear", "rdp_mle"])
@pytest.mark.parametrize("cfg", [0, 1, 4])
def test_synthetic_configs(built, algo, cfg):
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.config_spec(cfg)
    n = 6000
    fwd, rev, flen, rlen = synth.generate_numpy(spec, 0, n)
    asm = pa.Assembler(0)
    asm.set_algorithm(algo)
    b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
    o_res, o_nt, o_p, o_ctr = _oracle(algo, fwd, rev, flen, rlen)
    compare_results(b.results_numpy(), b.seq_nt.cpu().numpy(), b.seq_p.cpu().numpy(), o_res, o_nt, o_p)
    compare_counters(b.counters_dict(), o_ctr)


@pytest.mark.parametrize("algo", ["simple_bayesian", "pear", "rdp_mle"])
def test_adversarial_strided(built, algo):
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = adversarial_pairs(spec, 5000, seed=11)
    if algo == "pear":
        # algo_pear.c:52 indexes the forward read with the reverse index: undefined when rlen > flen
        rlen = np.minimum(rlen, flen)
    asm = pa.Assembler(0)
    asm.set_algorithm(algo)
    b = _run(asm, fwd, rev, flen, rlen, seq_stride=512)
    o_res, o_nt, o_p, o_ctr = _oracle(algo, fwd, rev, flen, rlen)
    compare_results(b.results_numpy(), b.seq_nt.cpu().numpy(), b.seq_p.cpu().numpy(), o_res, o_nt, o_p, seq_stride=512)
    compare_counters(b.counters_dict(), o_ctr)


def test_synth_device_matches_numpy(built):
    import torch

    from pandaseq_amd import synth
    for cfg in (0, 1, 2, 4):
        spec = synth.config_spec(cfg)
        f, r, fl, rl = synth.generate_numpy(spec, 1000, 3000)
        df, dr, dfl, drl = synth.generate_device(spec, 1000, 3000)
        torch.cuda.synchronize()
        L = spec.read_len
        assert np.array_equal(df.cpu().numpy()[:, :L], f[:, :L]), f"forward reads differ (config {cfg})"
        assert np.array_equal(dr.cpu().numpy()[:, :L], r[:, :L]), f"reverse reads differ (config {cfg})"
        assert np.array_equal(dfl.cpu().numpy().view(np.uint16), fl)
        assert np.array_equal(drl.cpu().numpy().view(np.uint16), rl)
