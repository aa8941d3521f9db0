"""The per-pair entry point against the batch paths, on the GPU: the same pairs through panda_assembler_assemble() as shipped (the
host path, pamd_compat.cpp host_assemble()), through panda_assembler_assemble() forced onto the device (a batch of one per call)
and through pamd_compat_assemble_many() (the kernels, 65 536 pairs per launch) give byte-identical output -- every record, merged
base, log-p and counter -- and that output is the oracle's."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import per_pair as pp
from util import adversarial_pairs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def client(built, tmp_path_factory):
    return pp.build_client(tmp_path_factory.mktemp("per_pair"))


def _three_ways(client, tmp_path, algo, fwd, rev, flen, rlen, options=(), oracle_kw=None, modules=(), device_pairs=400):
    from oracle import pyoracle as po
    if algo == "pear":
        rlen = np.minimum(rlen, flen)
    pairs = tmp_path / "pairs.bin"
    pp.write_pairs(pairs, fwd, rev, flen, rlen)
    env = dict(os.environ, PAMD_COMPAT_SUMS="exact")    # (the kernels in the reference's summation order: bit-identical scalars)
    host = pp.run_client(client, pairs, tmp_path / "host.bin", algo, "pair", options, env=env)
    many = pp.run_client(client, pairs, tmp_path / "many.bin", algo, "many", options, env=env)
    assert (tmp_path / "host.bin").read_bytes() == (tmp_path / "many.bin").read_bytes(), "per-pair host path and the kernels differ"
    o = po.assemble_batch(po.make_config(algo, modules=modules, **(oracle_kw or {})), fwd, rev, flen, rlen)
    pp.compare_with_oracle(*host, *o, modules=bool(modules))
    pp.compare_with_oracle(*many, *o, modules=bool(modules))
    # a batch of one per call on the device: the first pairs only (40 us per call)
    m = min(device_pairs, len(flen))
    small = tmp_path / "small.bin"
    pp.write_pairs(small, fwd[:m], rev[:m], flen[:m], rlen[:m])
    pp.run_client(client, small, tmp_path / "dev.bin", algo, "pair", options, env=dict(env, PAMD_COMPAT_PER_PAIR="device"))
    pp.run_client(client, small, tmp_path / "host_small.bin", algo, "pair", options, env=env)
    assert (tmp_path / "dev.bin").read_bytes() == (tmp_path / "host_small.bin").read_bytes(), "per-pair host path and per-pair device path differ"


@pytest.mark.parametrize("algo", pp.ALGOS)
def test_per_pair_equals_batches_on_adversarial_pairs(client, tmp_path, algo):
    from pandaseq_amd import synth
    fwd, rev, flen, rlen = adversarial_pairs(synth.config_spec(1), 1500, seed=51)
    _three_ways(client, tmp_path, algo, fwd, rev, flen, rlen)


@pytest.mark.parametrize("algo,cfg", [("simple_bayesian", 0), ("simple_bayesian", 4), ("pear", 4), ("rdp_mle", 1), ("uparse", 1)])
def test_per_pair_equals_batches_on_synthetic_configs(client, tmp_path, algo, cfg):
    from pandaseq_amd import synth
    fwd, rev, flen, rlen = synth.generate_host(synth.config_spec(cfg), 77_000, 1500)
    _three_ways(client, tmp_path, algo, fwd, rev, flen, rlen)


@pytest.mark.parametrize("algo,how", [("pear", "before"), ("simple_bayesian", "before"), ("pear", "after"), ("rdp_mle", "penalty")])
def test_per_pair_equals_batches_with_primers(client, tmp_path, algo, how):
    from pandaseq_amd import synth
    spec = synth.config_spec(2)
    fwd, rev, flen, rlen = synth.generate_host(spec, 3000, 800)
    fp, rp = synth.nt_codes(spec.fprimer), synth.nt_codes(spec.rprimer, complement=True)
    options = ["-p", pp.primer_arg(fp), "-q", pp.primer_arg(rp)]
    kw = dict(fprimer=fp, rprimer=rp)
    if how == "after":
        options.append("-a")
        kw["primers_after"] = True
    if how == "penalty":
        options += ["-T", "0.01"]
        kw["primer_penalty"] = 0.01
    _three_ways(client, tmp_path, algo, fwd, rev, flen, rlen, options, kw, device_pairs=200)


@pytest.mark.parametrize("algo,module,value", [("rdp_mle", "min_phred", 4), ("rdp_mle", "min_overlapbits", 15), ("pear", "min_overlapbits", 30)])
def test_per_pair_equals_batches_with_a_fused_module(client, tmp_path, algo, module, value):
    from pandaseq_amd import synth
    fwd, rev, flen, rlen = synth.generate_host(synth.config_spec(1), 9000, 1200)
    _three_ways(client, tmp_path, algo, fwd, rev, flen, rlen, ["-m", "%s:%d" % (module, value)], modules=[(module, value)])


def test_array_form_across_more_batches_than_slots(client, tmp_path):
    """pamd_compat_assemble_many() runs on the four pipelined slots of panda_assembler_next(): 300 000 pairs are more than the
    four batches of 65 536 that can be in flight, so every slot is refilled while others are being consumed; the output is the
    per-pair host path's, byte for byte (short reads keep the two files at 150 MB)."""
    import filecmp
    from pandaseq_amd import synth
    spec = synth.SynthSpec(0x5EED7700, 40, 46, 70, qhi=36)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 300_000)
    pairs = tmp_path / "pairs.bin"
    pp.write_pairs(pairs, fwd, rev, flen, rlen)
    env = dict(os.environ, PAMD_COMPAT_SUMS="exact")
    records = pp.run_client(client, pairs, tmp_path / "many.bin", "simple_bayesian", "many", env=env)
    pp.run_client(client, pairs, tmp_path / "host.bin", "simple_bayesian", "pair", env=env)
    assert filecmp.cmp(tmp_path / "many.bin", tmp_path / "host.bin", shallow=False)
    assert len(records[0]) == 300_000 and sum(r is not None for r in records[0]) > 250_000


@pytest.mark.parametrize("n", [0, 1, 5, 4097])
def test_array_form_on_few_pairs(client, tmp_path, n):
    """... and arrays shorter than one piece of a batch (one slot filled, three idle), or empty"""
    from pandaseq_amd import synth
    fwd, rev, flen, rlen = synth.generate_host(synth.config_spec(4), 500, max(n, 1))
    fwd, rev, flen, rlen = fwd[:n], rev[:n], flen[:n], rlen[:n]
    pairs = tmp_path / "pairs.bin"
    pp.write_pairs(pairs, fwd, rev, flen, rlen)
    env = dict(os.environ, PAMD_COMPAT_SUMS="exact")
    pp.run_client(client, pairs, tmp_path / "many.bin", "simple_bayesian", "many", env=env)
    pp.run_client(client, pairs, tmp_path / "host.bin", "simple_bayesian", "pair", env=env)
    assert (tmp_path / "many.bin").read_bytes() == (tmp_path / "host.bin").read_bytes()


@pytest.mark.parametrize("k", [1, 3])
def test_other_num_kmers_on_an_ordinary_assembler(client, tmp_path, k):
    """panda_assembler_new_kmer(k != 2) on an assembler that has its GPU: the per-pair entry point and the array form both give the
    oracle's output for that k (assembler.c:94 vs :99) -- pairs of such an assembler are assembled on the host path, the kernels'
    seeding being the default's."""
    from oracle import pyoracle as po
    from test_per_pair_cpu import _repeats_and_strays
    fwd, rev, flen, rlen = _repeats_and_strays(600, seed=80 + k)
    pairs = tmp_path / "pairs.bin"
    pp.write_pairs(pairs, fwd, rev, flen, rlen)
    o = po.assemble_batch(po.make_config("simple_bayesian", num_kmers=k), fwd, rev, flen, rlen)
    for path in ("pair", "many"):
        got = pp.run_client(client, pairs, tmp_path / (path + ".bin"), "simple_bayesian", path, ["-k", str(k)])
        pp.compare_with_oracle(*got, *o)
