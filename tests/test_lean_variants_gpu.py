"""Which kernel served a launch (pamd_last_lean_variant): every BASELINE configuration runs in the lean kernel's instantiation
DESIGN.md section 3 says it does -- config 4's short reads in the three-bases-per-lane one --, and what the lean kernel does not
serve goes to the general kernel whole.  (Results are compared with the oracle either way elsewhere; this pins the routing, so
that a bench line can be read as a measurement of the kernel it names.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _run(cfg, n=20000, **kw):
    import torch
    import bench
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    exact = kw.pop("exact", False)
    tweak = kw.pop("tweak", None)
    asm, spec = bench.setup_assembler(pa, synth, cfg, 0, exact_sums=exact, **kw)
    if tweak:
        tweak(asm)
    f, r, fl, rl = synth.generate_device(spec, 0, n)
    outs = pa.alloc_outputs(n, n * 640 + (1 << 25), "cuda:0")    # (every wave reserves chunks of its own: slack for 4096 of them)
    b = pa.DeviceBatch(n=n, fwd=f, rev=r, flen=fl, rlen=rl, **outs)
    asm.assemble_device(b)
    torch.cuda.synchronize()
    asm.check_overflow()
    return asm, b


@pytest.mark.parametrize("cfg,want", [(0, "short+narrow"), (1, "short"), (2, "primers"), (3, "short"), (4, "short+narrow")])
def test_baseline_configs_run_in_their_lean_instantiation(built, cfg, want):
    for exact in (False, True):
        asm, b = _run(cfg, exact=exact)
        assert asm.last_lean_variant() == want, (cfg, exact, asm.last_lean_variant())
        assert 0 <= asm.last_deferred() < 2000 and b.counters_dict()["count"] == 20000


def test_primers_after_routing(built):
    asm, _ = _run(2, primers_after=True)
    assert asm.last_lean_variant() == "after"
    asm, _ = _run(2, primers_after=True, exact=True)      # (round 5: the ordered sums come from the scratch rows of table entries, whole merge)
    assert asm.last_lean_variant() == "after" and asm.last_deferred() >= 0
    asm, _ = _run(2, primers_after=True, tweak=lambda a: a.set_primer_penalty(0.01))
    assert asm.last_lean_variant() == "general"


def test_what_the_lean_kernel_does_not_serve(built):
    asm, _ = _run(1, tweak=lambda a: a.set_maximum_overlap(200))    # (round 5: a maximum the reads can reach is served, tests/test_max_overlap_gpu.py)
    assert asm.last_lean_variant() != "general"
    asm, _ = _run(1, tweak=lambda a: (a.set_minimum_overlap(50), a.set_maximum_overlap(20)))     # a maximum below the minimum is not
    assert asm.last_lean_variant() == "general"
    asm, _ = _run(2, tweak=lambda a: a.set_primer_penalty(0.01))
    assert asm.last_lean_variant() == "general"
    asm, _ = _run(1, tweak=lambda a: a.set_fast_path(False))
    assert asm.last_lean_variant() == "general"
    asm, _ = _run(1, tweak=lambda a: (a.set_forward_trim(7), a.set_reverse_trim(3)))
    assert asm.last_lean_variant() == "primers"     # trims are served by the primers' instantiation


ALGOS = ["simple_bayesian", "pear", "rdp_mle", "uparse", "flash", "ea_util", "stitch"]
# geometry -> (config whose data selects it, expected instantiation name, what to set)
GEOMETRIES = {
    "short": (1, "short", None),
    "narrow": (4, "short+narrow", None),
    "long": (2, "long", None),                                     # 2x300 reads without primers or trims
    "prim": (2, "primers", "primers"),
    "post": (2, "after", "after"),
}


@pytest.mark.parametrize("geometry", list(GEOMETRIES))
@pytest.mark.parametrize("algo", ALGOS)
def test_every_instantiation_of_the_lean_kernel(built, algo, geometry):
    """pamd_fast.hip is compiled with its own flags (-Os, uniform regions left unstructurized: Makefile FAST_FLAGS) into 140
    instantiations -- seven scorers x {any-order, reference-order sums} x {with, without min_phred} x {short, narrow, long, primers},
    and `-a` x seven x two x two.  Each of them, on data that selects it: the instantiation the launch reports, and records, rows and
    counters of the two-kernel launch equal to the general kernel's alone and to the oracle's."""
    import torch
    import pandaseq_amd as pa
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import compare_counters, compare_results
    cfg, want, extra = GEOMETRIES[geometry]
    spec = synth.config_spec(cfg)
    n = 3000
    fwd, rev, flen, rlen = synth.generate_host(spec, 4242, n)
    if algo == "pear":
        rlen = np.minimum(rlen, flen)
    fp = rp = None
    if cfg == 2:
        fp, rp = synth.nt_codes(spec.fprimer), synth.nt_codes(spec.rprimer, complement=True)
    for exact in (False, True):
        for phred in (False, True):
            okw = {}
            if extra:
                okw = dict(fprimer=fp, rprimer=rp, primers_after=extra == "after")
            if phred:
                okw["modules"] = [("min_phred", 2)]
            o_res, o_nt, o_p, o_ctr = po.assemble_batch(po.make_config(algo, **okw), fwd, rev, flen, rlen)
            for fast in (True, False):
                asm = pa.Assembler(0)
                asm.set_algorithm(algo)
                asm.set_exact_sums(exact)
                asm.set_fast_path(fast)
                if extra:
                    asm.set_primers_after(extra == "after")
                    asm.set_forward_primer(fp)
                    asm.set_reverse_primer(rp)
                if phred:
                    asm.add_module("min_phred:2")
                b = pa.upload_batch(fwd, rev, flen, rlen, device="cuda:0", seq_stride=0, capacity=n * 640 + (1 << 25))
                asm.assemble_device(b)
                torch.cuda.synchronize()
                asm.check_overflow()
                assert asm.last_lean_variant() == (want if fast else "general"), (algo, geometry, exact, phred, asm.last_lean_variant())
                compare_results(b.results_numpy(), *b.rows_numpy(), o_res, o_nt, o_p, exact_sums=exact)
                compare_counters(b.counters_dict(), o_ctr)


@pytest.mark.parametrize("algo", ["simple_bayesian", "uparse", "flash", "ea_util", "stitch"])
def test_two_pairs_per_wave(built, algo):
    """The lean kernel's geometry for short reads, two pairs per wavefront (pamd_fast_dual.h; off by default: it is parity-green
    and measured slower): on 2 x 150 pairs of configs 0 and 4, on adversarial pairs (Ns, IUPAC codes, nt 0,
    qualities outside 0..46, quality-2 tails, ragged and too-short reads: the pairs a pass does not settle are assembled right
    behind it by fast_pair()), on tandem repeats and unrelated reads, on an odd number of pairs and on one pair -- every record,
    merged base, log-p and counter equal to the oracle's, and to a pair per wave."""
    import torch
    import pandaseq_amd as pa
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import adversarial_pairs, compare_counters, compare_results
    spec4 = synth.config_spec(4)
    cases = []
    for cfg, n in ((0, 3000), (4, 2999), (4, 1)):
        cases.append(synth.generate_host(synth.config_spec(cfg), 777, n))
    cases.append(adversarial_pairs(spec4, 2500, seed=91))
    f, r, fl, rl = synth.generate_host(spec4, 5000, 1800)
    rng = np.random.default_rng(17)
    L = spec4.read_len
    for i in range(0, 1800, 3):     # tandem repeats of every unit length (k-mers that occur dozens of times in both pairs of a pass)
        unit = rng.choice(np.array([1, 2, 4, 8], np.uint8), size=int(rng.integers(1, 14)))
        rep = np.tile(unit, L // len(unit) + 2)
        off = int(rng.integers(0, len(unit)))
        f[i, :L, 0] = rep[:L]
        r[i, :L, 0] = rep[off:off + L]
    for i in range(1, 1800, 3):     # unrelated reads: no vote at all, every overlap a candidate
        r[i, :L, 0] = rng.choice(np.array([1, 2, 4, 8], np.uint8), size=L)
    cases.append((f, r, fl, rl))
    for fwd, rev, flen, rlen in cases:
        n = len(flen)
        o_res, o_nt, o_p, o_ctr = po.assemble_batch(po.make_config(algo), fwd, rev, flen, rlen)
        for dual in (True, False):
            asm = pa.Assembler(0)
            asm.set_algorithm(algo)
            asm.set_exact_sums(False)
            asm.set_two_pairs_per_wave(dual)
            b = pa.upload_batch(fwd, rev, flen, rlen, device="cuda:0", seq_stride=0, capacity=n * 640 + (1 << 25))
            asm.assemble_device(b)
            torch.cuda.synchronize()
            asm.check_overflow()
            assert asm.last_lean_variant() == ("short+narrow+dual" if dual else "short+narrow"), asm.last_lean_variant()
            compare_results(b.results_numpy(), *b.rows_numpy(), o_res, o_nt, o_p, exact_sums=False)
            compare_counters(b.counters_dict(), o_ctr)
    if algo == "simple_bayesian":
        # at a size where every wave goes through hundreds of passes -- the epochs of the shared k-mer tables used up and started
        # over, pairs redone by fast_pair() between passes, finishes of 64 parked pairs: an odd 700 001 pairs against the oracle
        # on all host cores, every record field and the rows through their checksums
        n = 700_001
        fwd, rev, flen, rlen = synth.generate_host(spec4, 12_000_000, n)
        ocfg = po.make_config(algo)
        o_res, o_cs_nt, o_cs_p = po.row_checksums(ocfg, fwd, rev, flen, rlen, threads=os.cpu_count() or 1)
        o_ctr = po.assemble_batch(ocfg, fwd, rev, flen, rlen, want_seq=False, threads=os.cpu_count() or 1)[3]
        asm = pa.Assembler(0)
        asm.set_algorithm(algo)
        asm.set_exact_sums(False)
        asm.set_two_pairs_per_wave(True)
        b = pa.upload_batch(fwd, rev, flen, rlen, device="cuda:0", seq_stride=0, capacity=n * 330 + (1 << 27))
        asm.assemble_device(b)
        torch.cuda.synchronize()
        asm.check_overflow()
        assert asm.last_lean_variant() == "short+narrow+dual"
        compare_results(b.results_numpy(), None, None, o_res, None, None, exact_sums=False, check_seq=False)
        d_nt, d_p = pa.row_checksums(b.results, b.seq_nt, b.seq_p)
        assert np.array_equal(d_nt.cpu().numpy().view(np.uint64), o_cs_nt) and np.array_equal(d_p.cpu().numpy().view(np.uint64), o_cs_p), "row checksums"
        compare_counters(b.counters_dict(), o_ctr)
    # what the geometry does not serve stays with a pair per wave: reference-order sums, a min_phred module, a maximum overlap, pear
    fwd, rev, flen, rlen = cases[1]
    for tweak in (lambda a: a.set_exact_sums(True), lambda a: a.add_module("min_phred:2"), lambda a: a.set_maximum_overlap(120), lambda a: a.set_algorithm("pear")):
        asm = pa.Assembler(0)
        asm.set_algorithm(algo)
        asm.set_exact_sums(False)
        asm.set_two_pairs_per_wave(True)
        tweak(asm)
        b = pa.upload_batch(fwd, rev, flen, rlen, device="cuda:0", seq_stride=0, capacity=len(flen) * 640 + (1 << 25))
        asm.assemble_device(b)
        torch.cuda.synchronize()
        assert "dual" not in asm.last_lean_variant() and asm.last_lean_variant() != "general"


@pytest.mark.parametrize("cfg", [1, 4])
def test_quality_sums_through_the_merge_tree(built, cfg):
    """Any-order sums in the short geometry (pamd_fast.hip FK_LAZYQ): a pair's quality sum comes out of a merge tree over the 64
    pairs a wave parks between two finishes, in lane bitrev6(slot), and a wave's last group is closed with pairs that hand in
    nothing.  Launches in which every wave's pairs fill its groups to every level of the tree -- 1, 2, 3, 5, 17, 31 .. 33 and
    63 .. 66 pairs per wave, a quarter of the waves one more -- against the oracle: every record (statuses and counts exactly,
    the two summed doubles within the any-order tolerance), the LOWQ verdicts among them, and all counters."""
    import torch
    import pandaseq_amd as pa
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import compare_counters, compare_results
    spec = synth.config_spec(cfg)
    waves = 4096      # (256 workgroups of 16 waves; small launches hand pairs out strided, wave w takes w, w + 4096, ..)
    ocfg = po.make_config("simple_bayesian")
    for k in (1, 2, 3, 5, 17, 31, 32, 33, 63, 64, 65, 66):
        n = waves * k + waves // 4 + 7
        fwd, rev, flen, rlen = synth.generate_host(spec, 31_000_000 + 1000 * k, n)
        o_res, _, _, o_ctr = po.assemble_batch(ocfg, fwd, rev, flen, rlen, want_seq=False, threads=os.cpu_count() or 1)
        asm = pa.Assembler(0)
        asm.set_algorithm("simple_bayesian")
        asm.set_exact_sums(False)
        b = pa.upload_batch(fwd, rev, flen, rlen, device="cuda:0", seq_stride=0, capacity=n * (520 if cfg == 1 else 330) + (1 << 26))
        asm.assemble_device(b)
        torch.cuda.synchronize()
        asm.check_overflow()
        assert asm.last_lean_variant() == ("short" if cfg == 1 else "short+narrow")
        info = asm.last_launch_info()
        assert info["workgroups"] * info["waves_per_workgroup"] == waves, info
        compare_results(b.results_numpy(), None, None, o_res, None, None, exact_sums=False, check_seq=False)
        compare_counters(b.counters_dict(), o_ctr)
