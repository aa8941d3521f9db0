"""The per-pair entry point's host path (pamd_compat.cpp host_assemble() scoring through pamd_host_scorers.cpp), without a GPU:
panda_assembler_assemble() on a device-less assembler (the library's test hook) against the oracle -- every record, merged base,
log-p and counter, bit for bit (the host path sums in the reference's order, assembler.c:171,226,236)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import per_pair as pp
from util import adversarial_pairs


@pytest.fixture(scope="module")
def client(built, tmp_path_factory):
    return pp.build_client(tmp_path_factory.mktemp("per_pair"))


def _check(client, tmp_path, algo, fwd, rev, flen, rlen, options=(), oracle_kw=None, modules=()):
    from oracle import pyoracle as po
    if algo == "pear":
        rlen = np.minimum(rlen, flen)   # (past the forward read pear reads beyond the array in the reference: kept defined)
    pairs, out = tmp_path / "pairs.bin", tmp_path / "out.bin"
    pp.write_pairs(pairs, fwd, rev, flen, rlen)
    records, counters = pp.run_client(client, pairs, out, algo, "host", options)
    o = po.assemble_batch(po.make_config(algo, modules=modules, **(oracle_kw or {})), fwd, rev, flen, rlen)
    pp.compare_with_oracle(records, counters, *o, modules=bool(modules))
    return o


@pytest.mark.parametrize("algo", pp.ALGOS)
@pytest.mark.parametrize("cfg", [0, 1, 4])
def test_host_path_on_synthetic_configs(client, tmp_path, algo, cfg):
    from pandaseq_amd import synth
    spec = synth.config_spec(cfg)
    fwd, rev, flen, rlen = synth.generate_numpy(spec, 1000, 400)
    o = _check(client, tmp_path, algo, fwd, rev, flen, rlen)
    if algo != "flash":     # (flash's integer quotient accepts next to nothing)
        assert int(o[3]["ok"]) > 100


@pytest.mark.parametrize("algo", pp.ALGOS)
def test_host_path_on_adversarial_pairs(client, tmp_path, algo):
    """Ns, IUPAC codes, nt 0, qualities outside 0..46, quality-2 tails, ragged and too-short reads."""
    from pandaseq_amd import synth
    fwd, rev, flen, rlen = adversarial_pairs(synth.config_spec(1), 700, seed=41)
    _check(client, tmp_path, algo, fwd, rev, flen, rlen)


@pytest.mark.parametrize("algo", ["pear", "simple_bayesian", "rdp_mle"])
@pytest.mark.parametrize("how", ["before", "after", "penalty"])
def test_host_path_with_primers(client, tmp_path, algo, how):
    """config 2 (2x300 with embedded primers): -p / -q before assembly, `-a` after it, and with a primer penalty."""
    from pandaseq_amd import synth
    spec = synth.config_spec(2)
    fwd, rev, flen, rlen = synth.generate_numpy(spec, 0, 300)
    fp, rp = synth.nt_codes(spec.fprimer), synth.nt_codes(spec.rprimer, complement=True)
    options = ["-p", pp.primer_arg(fp), "-q", pp.primer_arg(rp)]
    kw = dict(fprimer=fp, rprimer=rp)
    if how == "after":
        options.append("-a")
        kw["primers_after"] = True
    if how == "penalty":
        options += ["-T", "0.01"]
        kw["primer_penalty"] = 0.01
    o = _check(client, tmp_path, algo, fwd, rev, flen, rlen, options, kw)
    assert int(o[3]["ok"]) > 100


@pytest.mark.parametrize("algo,module,value", [("rdp_mle", "min_phred", 4), ("rdp_mle", "min_overlapbits", 15), ("pear", "min_overlapbits", 30), ("simple_bayesian", "min_phred", 10)])
def test_host_path_with_the_two_modules(client, tmp_path, algo, module, value):
    from pandaseq_amd import synth
    fwd, rev, flen, rlen = synth.generate_numpy(synth.config_spec(1), 5000, 500)
    o = _check(client, tmp_path, algo, fwd, rev, flen, rlen, ["-m", "%s:%d" % (module, value)], modules=[(module, value)])
    assert int(o[3]["rejected"][0]) > 0


@pytest.mark.parametrize("algo", ["simple_bayesian", "pear"])
def test_host_path_with_overlap_bounds_and_threshold(client, tmp_path, algo):
    from pandaseq_amd import synth
    fwd, rev, flen, rlen = adversarial_pairs(synth.config_spec(1), 500, seed=43)
    _check(client, tmp_path, algo, fwd, rev, flen, rlen, ["-o", "9", "-O", "120", "-t", "0.9"], dict(minoverlap=9, maxoverlap=120, threshold=0.9))


def _repeats_and_strays(n, seed):
    """config-1 pairs, a third of them tandem repeats (k-mers that occur dozens of times: what num_kmers decides), a third unrelated reads"""
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = synth.generate_numpy(spec, 0, n)
    rng = np.random.default_rng(seed)
    L = spec.read_len
    for i in range(0, n, 3):
        unit = rng.choice(np.array([1, 2, 4, 8], np.uint8), size=int(rng.integers(1, 14)))
        rep = np.tile(unit, L // len(unit) + 2)
        off = int(rng.integers(0, len(unit)))
        fwd[i, :L, 0] = rep[:L]
        rev[i, :L, 0] = rep[off:off + L]
        if i % 9 == 0:
            fwd[i, 40:52, 0] = 15
    for i in range(1, n, 3):
        rev[i, :L, 0] = rng.choice(np.array([1, 2, 4, 8], np.uint8), size=L)
    return fwd, rev, flen, rlen


@pytest.mark.parametrize("k", [0, 1, 3, 5, 10])
def test_host_path_with_other_num_kmers(client, tmp_path, k):
    """panda_assembler_new_kmer(k != 2) (assembler_support.c:36-100): the reference looks for a free slot at kmerseen[(kmer << 1) + j]
    and writes to kmerseen[kmer * k + j] (assembler.c:94 vs :99); the host path reproduces that as written -- against the oracle's
    flat table, on pairs whose k-mers repeat (where the number of kept positions decides which overlaps are examined)."""
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    fwd, rev, flen, rlen = _repeats_and_strays(450, seed=60 + k)
    o = _check(client, tmp_path, "simple_bayesian", fwd, rev, flen, rlen, ["-k", str(k)], dict(num_kmers=k))
    o2 = po.assemble_batch(po.make_config("simple_bayesian"), fwd, rev, flen, rlen)
    # the setting is not a no-op: some pair examines another number of overlaps than with the default two positions per k-mer
    assert (o[0]["examined"] != o2[0]["examined"]).any()
    fwd, rev, flen, rlen = adversarial_pairs(synth.config_spec(1), 300, seed=70 + k)
    _check(client, tmp_path, "pear", fwd, rev, flen, rlen, ["-k", str(k)], dict(num_kmers=k))


def test_every_class_object_is_complete(built):
    """pandaseq-common.h:594-605 is public: each of the seven class objects carries both functions, and
    panda_algorithm_quality_compare (algo.c:27-32) answers through them -- checked against the oracle's tables."""
    import ctypes as C
    from oracle import pyoracle as po
    from pandaseq_amd import _lib
    L = _lib.lib()

    class Class(C.Structure):
        _fields_ = [("data_size", C.c_size_t), ("name", C.c_char_p), ("create", C.c_void_p), ("data_destroy", C.c_void_p),
                    ("overlap_probability", C.c_void_p), ("match_probability", C.c_void_p), ("prob_unpaired", C.c_double)]
    for name in ("simple_bayes", "pear", "rdp_mle", "uparse", "flash", "ea_util", "stitch"):
        cls = Class.in_dll(L, "panda_algorithm_%s_class" % name)
        assert cls.overlap_probability and cls.match_probability, name
        assert cls.prob_unpaired == -1.38629
    T = po.tables()
    L.panda_algorithm_simple_bayes_new.restype = C.c_void_p
    L.panda_algorithm_quality_compare.restype = C.c_double
    L.panda_algorithm_quality_compare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    a = L.panda_algorithm_simple_bayes_new()
    qa, qb = (C.c_uint8 * 2)(1, 30), (C.c_uint8 * 2)(2, 12)
    assert L.panda_algorithm_quality_compare(a, qa, qb) == T.mismatch_sb[30][12]
    qb[0] = 1
    assert L.panda_algorithm_quality_compare(a, qa, qb) == T.match_sb[30][12]


def test_reference_constructor_still_needs_a_gpu(built):
    """The per-pair host path is an entry point of an assembler that has its GPU: panda_assembler_new() itself refuses to build
    one without (no CPU fallback for batches and streams); only the test hook used above builds a device-less assembler."""
    import ctypes as C
    import torch
    from pandaseq_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = _lib.lib()
    L.panda_assembler_new.restype = C.c_void_p
    L.panda_assembler_new.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.panda_assembler_new(None, None, None, None) is None
    L.pamd_last_error.restype = C.c_char_p
    assert b"no CPU fallback" in L.pamd_last_error() or b"GPU" in L.pamd_last_error() or b"device" in L.pamd_last_error()
