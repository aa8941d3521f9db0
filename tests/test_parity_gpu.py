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


@pytest.fixture(params=["exact", "fast"])
def sums(request):
    """Both summation modes of the library (include/pandaseq_amd.h, pamd_set_exact_sums)."""
    return request.param


@pytest.mark.parametrize("algo", ["simple_bayesian", "pear", "rdp_mle"])
@pytest.mark.parametrize("cfg", [0, 1, 4])
def test_synthetic_configs(built, algo, cfg, sums):
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.config_spec(cfg)
    n = 6000
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, n)
    asm = pa.Assembler(0)
    asm.set_algorithm(algo)
    asm.set_exact_sums(sums == "exact")
    b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
    o_res, o_nt, o_p, o_ctr = _oracle(algo, fwd, rev, flen, rlen)
    compare_results(b.results_numpy(), *b.rows_numpy(), o_res, o_nt, o_p, exact_sums=sums == "exact")
    compare_counters(b.counters_dict(), o_ctr)


@pytest.mark.parametrize("algo,cfg,after", [("simple_bayesian", 1, False), ("pear", 1, False), ("rdp_mle", 1, False), ("simple_bayesian", 4, False),
                                            ("pear", 2, False), ("pear", 2, True), ("rdp_mle", 2, True)])
def test_reference_order_sums_through_full_groups(built, algo, cfg, after):
    """Sums in the reference's order at a size where every wave of the lean kernel goes through full groups of its walk (21
    pairs, 15 with a score chain), the finish after 63 / 60 pairs and a partial last group: 300 000 pairs against the oracle,
    `quality` and `est_overlap_prob` bit for bit, every record and counter; the rows through their checksums."""
    _at_scale_against_the_oracle(algo, cfg, after, True, 300_000)


@pytest.mark.parametrize("algo,cfg,after,exact", [("simple_bayesian", 1, False, False), ("simple_bayesian", 1, False, True), ("rdp_mle", 1, False, True),
                                                  ("simple_bayesian", 4, False, False), ("pear", 2, False, True), ("pear", 2, True, True),
                                                  ("uparse", 1, False, False)])
def test_pairs_in_blocks(built, monkeypatch, algo, cfg, after, exact):
    """The lean kernel's other assignment of pairs to waves (large launches: a wave takes blocks of consecutive pairs, one block per
    finish -- 64, or 63 / 60 with sums in the reference's order -- instead of every W-th pair), forced here on a launch whose last
    round of blocks is partial and whose last block is cut short: the same records, rows and counters."""
    monkeypatch.setenv("PAMD_FK_BLOCKS", "1")
    _at_scale_against_the_oracle(algo, cfg, after, exact, 300_007)
    if cfg == 1:    # fewer blocks than waves, the last one of 33 pairs
        _at_scale_against_the_oracle(algo, cfg, after, exact, 4_001)


def _at_scale_against_the_oracle(algo, cfg, after, exact, n):
    import torch
    import pandaseq_amd as pa
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    spec = synth.config_spec(cfg)
    fwd, rev, flen, rlen = synth.generate_host(spec, 31_000_000, n)
    if algo == "pear":
        rlen = np.minimum(rlen, flen)
    okw = {}
    asm = pa.Assembler(0)
    asm.set_algorithm(algo)
    asm.set_exact_sums(exact)
    if cfg == 2:
        fp, rp = synth.nt_codes(spec.fprimer), synth.nt_codes(spec.rprimer, complement=True)
        asm.set_forward_primer(fp)
        asm.set_reverse_primer(rp)
        asm.set_primers_after(after)
        okw = dict(fprimer=fp, rprimer=rp, primers_after=after)
    b = pa.upload_batch(fwd, rev, flen, rlen, device="cuda:0", seq_stride=0, capacity=n * (620 if cfg == 2 else 440) + (1 << 27))
    asm.assemble_device(b)
    torch.cuda.synchronize()
    asm.check_overflow()
    assert asm.last_lean_variant() != "general" and 0 <= asm.last_deferred() < n // 20
    ocfg = po.make_config(algo, **okw)
    o_res, o_cs_nt, o_cs_p = po.row_checksums(ocfg, fwd, rev, flen, rlen, threads=8)
    o_ctr = po.assemble_batch(ocfg, fwd, rev, flen, rlen, want_seq=False, threads=8)[3]
    res = b.results_numpy()
    assert np.array_equal(res["status"], o_res["status"])
    ok = o_res["status"] == 0
    assert int(ok.sum()) > (n // 2 if algo != "uparse" else 0)     # (uparse's log-p put most of these synthetic merges under the 0.6 threshold: LOWQ, compared below)
    for name in ("overlap", "seq_len", "mismatches", "degenerates", "examined", "fwd_offset", "rev_offset"):
        assert np.array_equal(res[name][ok], o_res[name][ok]), name
    merged = ok | (o_res["status"] == 5) | (o_res["status"] == 6)      # (LOWQ and module rejects carry both sums too, assembler.c:244-247)
    assert int(merged.sum()) > n // 2
    for name, o_name in (("quality", "quality"), ("est_overlap_prob", "est_prob")):     # the two ordered sums: the very bits
        if not exact:
            assert np.abs(res[name][merged] - o_res[o_name][merged]).max() <= 1e-9, name
            continue
        d_bits, o_bits = res[name][merged].view(np.uint64), o_res[o_name][merged].view(np.uint64)
        assert np.array_equal(d_bits, o_bits), (name, int((d_bits != o_bits).sum()), np.nonzero(merged)[0][np.nonzero(d_bits != o_bits)[0][:8]])
    assert not res["est_overlap_prob"][~merged].any() and not res["quality"][~merged].any()     # a pair that did not merge has neither
    d_nt, d_p = pa.row_checksums(b.results, b.seq_nt, b.seq_p)
    assert np.array_equal(d_nt.cpu().numpy().view(np.uint64), o_cs_nt) and np.array_equal(d_p.cpu().numpy().view(np.uint64), o_cs_p)
    compare_counters(b.counters_dict(), o_ctr)


@pytest.mark.parametrize("algo", ["simple_bayesian", "pear", "rdp_mle"])
def test_adversarial_strided(built, algo, sums):
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = adversarial_pairs(spec, 5000, seed=11)
    if algo == "pear":
        # algo_pear.c:52 indexes the forward read with the reverse index: undefined when rlen > flen
        rlen = np.minimum(rlen, flen)
    asm = pa.Assembler(0)
    asm.set_algorithm(algo)
    asm.set_exact_sums(sums == "exact")
    b = _run(asm, fwd, rev, flen, rlen, seq_stride=512)
    o_res, o_nt, o_p, o_ctr = _oracle(algo, fwd, rev, flen, rlen)
    compare_results(b.results_numpy(), *b.rows_numpy(), o_res, o_nt, o_p, seq_stride=512, exact_sums=sums == "exact")
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


def _asm_for(algo, spec=None, primers=False, modules=(), exact=True, **settings):
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    asm = pa.Assembler(0)
    asm.set_algorithm(algo)
    asm.set_exact_sums(exact)
    if primers:
        asm.set_forward_primer(synth.nt_codes(spec.fprimer))
        asm.set_reverse_primer(synth.nt_codes(spec.rprimer, complement=True))
    for m in modules:
        asm.add_module(m)
    for k, v in settings.items():
        getattr(asm, "set_" + k)(v)
    return asm


def test_config2_primers_pear(built, sums):
    """BASELINE configs[2]: 2x300 with embedded primers, -p/-q, pear."""
    from pandaseq_amd import synth
    spec = synth.config_spec(2)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 5000)
    # a few pairs whose primers are damaged or missing, and a read shorter than the primer
    rng = np.random.default_rng(3)
    for i in range(0, 200, 7):
        fwd[i, rng.integers(0, 17), 0] = 15
        rev[i + 1, :17, 0] = rng.choice(np.array([1, 2, 4, 8], np.uint8), 17)
    flen[300] = 10
    rlen[301] = 16
    asm = _asm_for("pear", spec, primers=True, exact=sums == "exact")
    b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
    o = _oracle("pear", fwd, rev, flen, rlen, fprimer=synth.nt_codes(spec.fprimer), rprimer=synth.nt_codes(spec.rprimer, complement=True))
    compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=sums == "exact")
    compare_counters(b.counters_dict(), o[3])
    assert o[3]["ok"] > 4000 and (o[0]["fwd_offset"][o[0]["status"] == 0] >= 17).all()


@pytest.mark.parametrize("penalty", [0.0, 0.01])
def test_primers_all_algorithms_with_penalty(built, penalty):
    from pandaseq_amd import synth
    spec = synth.config_spec(2)
    fwd, rev, flen, rlen = synth.generate_host(spec, 7000, 1500)
    for algo in ("simple_bayesian", "rdp_mle"):
        asm = _asm_for(algo, spec, primers=True, primer_penalty=penalty)
        b = _run(asm, fwd, rev, flen, rlen, seq_stride=608)
        o = _oracle(algo, fwd, rev, flen, rlen, fprimer=synth.nt_codes(spec.fprimer),
                    rprimer=synth.nt_codes(spec.rprimer, complement=True), primer_penalty=penalty)
        compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], seq_stride=608)
        compare_counters(b.counters_dict(), o[3])


def test_config3_rdp_with_modules(built, sums):
    """BASELINE configs[3]: rdp_mle + min_phred + min_overlapbits, modules applied in order."""
    from pandaseq_amd import synth
    spec = synth.config_spec(3)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 6000)
    for mods in ([("min_phred", 2), ("min_overlapbits", 15)], [("min_overlapbits", 40), ("min_phred", 1)], [("min_phred", 10)]):
        asm = _asm_for("rdp_mle", modules=[f"{k}:{v}" for k, v in mods], exact=sums == "exact")
        b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
        o = _oracle("rdp_mle", fwd, rev, flen, rlen, modules=mods)
        compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=sums == "exact")
        compare_counters(b.counters_dict(), o[3])
        assert sum(o[3]["rejected"]) > 0


@pytest.mark.parametrize("algo", ["simple_bayesian", "uparse"])
def test_min_phred_one_on_rows_with_both_kinds_of_low_p(built, algo, sums):
    """panda_result_phred (nt.c:126-150) answers 1 at or below qual_score[0] and 0 just above it (sic): a row that holds a log-p of
    each kind has a phred 0 although its smallest log-p has phred 1, and `min_phred:1` rejects it.  (Found by
    tests/parity_sweep.py wide, round 3: the lean kernel took the phred of the smallest log-p.)"""
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = synth.generate_host(spec, 9000, 3000)
    plain = _oracle(algo, fwd, rev, flen, rlen)
    # two disagreeing positions planted in every second pair's overlap: qualities (3, 7) give a log-p between qual_score[0] and
    # [1] under simple_bayesian and (5, 4) under uparse, qualities (4, 30) / (0, 1) one at or below qual_score[0]
    band, low = ((3, 7), (4, 30)) if algo == "simple_bayesian" else ((5, 4), (0, 1))
    for i in range(0, len(flen), 2):
        ov = int(plain[0]["overlap"][i])
        if plain[0]["status"][i] != 0 or ov < 40:
            continue
        for k, (qf, qr) in ((11, band), (23, low)):
            fi, ri = int(flen[i]) - ov + k, int(rlen[i]) - 1 - k
            fwd[i, fi] = (1, qf)
            rev[i, ri] = (2, qr)
    for mods in ([("min_phred", 1)], [("min_phred", 1), ("min_overlapbits", 5)]):
        asm = _asm_for(algo, modules=[f"{k}:{v}" for k, v in mods], exact=sums == "exact")
        b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
        o = _oracle(algo, fwd, rev, flen, rlen, modules=mods)
        compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=sums == "exact")
        compare_counters(b.counters_dict(), o[3])
    s0, s1 = po.tables_np()["score"][:2]
    plain = _oracle(algo, fwd, rev, flen, rlen)
    both = 0
    for i in np.nonzero(plain[0]["status"] == 0)[0]:
        p = plain[2][i, :int(plain[0]["seq_len"][i])]
        both += bool((p <= s0).any() and ((p > s0) & (p < s1)).any())
    want = 500 if algo == "simple_bayesian" else 5     # (uparse accepts few overlaps that disagree twice)
    assert both > want, "the data must hold rows of the kind the test is about"
    assert o[3]["rejected"][0] > want and asm.last_deferred() < 1500


@pytest.mark.parametrize("settings", [
    dict(minimum_overlap=10), dict(minimum_overlap=60, maximum_overlap=100), dict(maximum_overlap=400),
    dict(threshold=0.9), dict(threshold=0.3), dict(forward_trim=5, reverse_trim=11), dict(maximum_overlap=120, forward_trim=30),
])
def test_assembler_settings(built, settings, sums):
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = adversarial_pairs(spec, 3000, seed=5)
    okw = {}
    for k, v in settings.items():
        okw[{"minimum_overlap": "minoverlap", "maximum_overlap": "maxoverlap", "forward_trim": "ftrim", "reverse_trim": "rtrim"}.get(k, k)] = v
    for algo in ("simple_bayesian", "rdp_mle"):
        asm = _asm_for(algo, exact=sums == "exact", **settings)
        b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
        o = _oracle(algo, fwd, rev, flen, rlen, **okw)
        compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=sums == "exact")
        compare_counters(b.counters_dict(), o[3])


def test_threshold_on_a_pairs_own_quality(built, sums):
    """The LOWQ verdict when the threshold IS a pair's quality (assembler.c:334: quality < threshold).  In fast-sums
    mode this is the case the kernel re-sums in the reference's order, reading the log-p row it has just written to
    HBM; the threshold goes through exp() and the setter's log(), so it lands within an ulp or two of the quality,
    on either side of it."""
    import math

    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = adversarial_pairs(spec, 3000, seed=11)
    o0 = _oracle("simple_bayesian", fwd, rev, flen, rlen)
    ok = np.nonzero(o0[0]["status"] == 0)[0]
    flips = 0
    for k in ok[[3, len(ok) // 3, len(ok) // 2, len(ok) - 5]]:
        for nudge in (0.0, 1e-15, -1e-15):
            t = math.exp(float(o0[0]["quality"][k])) * (1.0 + nudge)
            asm = _asm_for("simple_bayesian", exact=sums == "exact", threshold=t)
            b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
            o = _oracle("simple_bayesian", fwd, rev, flen, rlen, threshold=t)
            compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=sums == "exact")
            compare_counters(b.counters_dict(), o[3])
            flips += int(o[0]["status"][k] != 0)
    assert 0 < flips < 12, "the chosen pairs must land on both sides of their threshold for the test to mean anything"


def test_algorithm_parameters(built):
    import math

    from pandaseq_amd import synth
    import pandaseq_amd as pa
    spec = synth.config_spec(0)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 3000)
    asm = pa.Assembler(0)
    asm.set_algorithm("simple_bayesian", 0.2)
    b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
    o = _oracle("simple_bayesian", fwd, rev, flen, rlen, q=0.2)
    compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3])
    asm.set_algorithm("pear", 0.1)
    b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
    o = _oracle("pear", fwd, rev, flen, rlen, pear_random_base_log=math.log(0.1))
    compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3])


def test_max_length_reads(built, sums):
    """PANDA_MAX_LEN = 450 reads: the widest rows, the longest merges (899) and the smallest launch geometry."""
    from pandaseq_amd import synth
    spec = synth.SynthSpec(seed=77, read_len=450, tmin=450, tmax=899, qhi=40)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 1500)
    for algo in ("simple_bayesian", "pear", "rdp_mle"):
        asm = _asm_for(algo, exact=sums == "exact")
        b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
        o = _oracle(algo, fwd, rev, flen, rlen)
        compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=sums == "exact")
        compare_counters(b.counters_dict(), o[3])
        assert o[0]["seq_len"].max() > 850


def test_host_buffer_entry_point(built):
    """pamd_assemble_host: same results through host pointers (upload, kernel, download)."""
    from pandaseq_amd import synth
    import pandaseq_amd as pa
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = synth.generate_host(spec, 500, 2000)
    asm = pa.Assembler(0)
    for stride in (512, 0):
        res, nt, p, ctr = asm.assemble_host(fwd, rev, flen, rlen, seq_stride=stride)
        o = _oracle("simple_bayesian", fwd, rev, flen, rlen)
        compare_results(res, nt, p, *o[:3], seq_stride=stride or None)
        compare_counters(ctr, o[3])


def test_empty_and_tiny_batches(built):
    from pandaseq_amd import synth
    import pandaseq_amd as pa
    spec = synth.config_spec(0)
    asm = pa.Assembler(0)
    for n in (1, 2, 17):
        fwd, rev, flen, rlen = synth.generate_host(spec, 0, n)
        b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
        o = _oracle("simple_bayesian", fwd, rev, flen, rlen)
        compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3])
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 4)
    res, nt, p, ctr = asm.assemble_host(fwd[:0], rev[:0], flen[:0], rlen[:0])
    assert len(res) == 0 and ctr["count"] == 0


def test_golden_fixtures(built, sums):
    """The committed fixtures (tests/golden): device output against stored expectations, in both summation modes."""
    import glob
    import os
    import sys

    from pandaseq_amd import synth
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden
    for path in sorted(glob.glob(os.path.join(here, "golden", "*.npz"))):
        name = os.path.basename(path)[:-4]
        g = np.load(path)
        cfg_index, n, kw = make_golden.CASES[name]
        spec = synth.config_spec(cfg_index)
        mods = [f"{k}:{v}" for k, v in kw.get("modules", [])]
        asm = _asm_for(kw["algo"], spec, primers=kw.get("primers", False), modules=mods, exact=sums == "exact")
        b = _run(asm, g["fwd"], g["rev"], g["flen"], g["rlen"], seq_stride=0)
        res = b.results_numpy()
        gr = g["res"]
        assert np.array_equal(res["status"], gr["status"].astype(np.uint8)), name
        ok = gr["status"] == 0
        assert np.array_equal(res["overlap"][ok], gr["overlap"][ok].astype(np.uint16)), name
        if sums == "exact":
            assert np.array_equal(res["quality"][ok].view(np.uint64), gr["quality"][ok].view(np.uint64)), name
            assert np.array_equal(res["est_overlap_prob"][ok].view(np.uint64), gr["est_prob"][ok].view(np.uint64)), name
        else:
            assert np.abs(res["quality"][ok] - gr["quality"][ok]).max() <= 1e-9, name
            assert np.abs(res["est_overlap_prob"][ok] - gr["est_prob"][ok]).max() <= 1e-9, name
        nt = b.seq_nt.cpu().numpy()
        p = b.seq_p.cpu().numpy()
        for i in np.nonzero(ok)[0]:
            m = int(gr["seq_len"][i])
            off = int(res["seq_offset"][i])
            w = np.arange(1, m + 1)
            assert int((nt[off:off + m].astype(np.uint64) * w.astype(np.uint64)).sum()) == int(g["cs_nt"][i]), (name, i)
            assert float((p[off:off + m] * w).sum()) == float(g["cs_p"][i]), (name, i)


def test_full_size_properties(built):
    """At a size the oracle would need minutes for: size-independent properties on 2 M pairs of config 1 --
    counters add up, every accepted row lies inside the arena without overlapping its neighbour, merged
    length = flen + rlen - overlap, per-base log-p <= 0, and a re-run is bit-identical up to row placement."""
    import torch

    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    n = 2_000_000
    f, r, fl, rl = synth.generate_device(spec, 0, n)
    outs = pa.alloc_outputs(n, n * 420, "cuda:0")
    b = pa.DeviceBatch(n=n, fwd=f, rev=r, flen=fl, rlen=rl, **outs)
    asm = pa.Assembler(0)
    asm.assemble_device(b)
    torch.cuda.synchronize()
    res = b.results_numpy().copy()
    c = b.counters_dict()
    assert c["count"] == n == c["ok"] + c["noalgn"] + c["lowq"] + c["badr"]
    assert int(c["overlapcount"].sum()) == c["ok"] == int((res["status"] == 0).sum())
    ok = res["status"] == 0
    assert (res["seq_len"][ok] == 500 - res["overlap"][ok]).all()
    offs = res["seq_offset"][ok].astype(np.int64)
    order = np.argsort(offs)
    ends = offs[order] + res["seq_len"][ok][order]
    assert (offs % 16 == 0).all() and (ends[:-1] <= offs[order][1:]).all() and ends[-1] <= int(b.cursor.item())
    # every written log-p is <= 0 (the arena between rows is never written: only rows are looked at)
    import torch as _t
    sample = np.nonzero(ok)[0][:: max(1, int(ok.sum()) // 20000)]
    so = _t.from_numpy(res["seq_offset"][sample].astype(np.int64)).cuda()
    sl = _t.from_numpy(res["seq_len"][sample].astype(np.int64)).cuda()
    col = _t.arange(512, device="cuda:0")
    vals = b.seq_p[(so[:, None] + col[None, :]).clamp(max=b.seq_p.numel() - 1)]
    assert float(vals[col[None, :] < sl[:, None]].max().item()) <= 0.0
    cs1 = _row_checksums(b, res, ok)
    b.cursor.zero_()
    b.counters.zero_()
    asm.assemble_device(b)
    torch.cuda.synchronize()
    res2 = b.results_numpy()
    for k in ("status", "overlap", "seq_len", "examined", "mismatches"):
        assert np.array_equal(res[k], res2[k])
    assert np.array_equal(res["quality"].view(np.uint64), res2["quality"].view(np.uint64))
    assert np.array_equal(cs1, _row_checksums(b, res2, ok))


@pytest.mark.parametrize("cfg", [2, 3, 4])
def test_full_size_properties_other_configs(built, cfg, sums):
    """The same size-independent properties on 1 M pairs of BASELINE configs 2, 3 and 4: counters add up (every pair has
    exactly one outcome), accepted rows lie inside the arena without touching, merged length = reads - overlap - clipped
    primers, and the two-kernel launch agrees with the general kernel alone on every record."""
    import torch

    import bench
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    n = 1_000_000
    spec = synth.config_spec(cfg)
    f, r, fl, rl = synth.generate_device(spec, 0, n)
    recs = []
    for fast_path in (True, False):
        asm, _ = bench.setup_assembler(pa, synth, cfg, 0, exact_sums=sums == "exact")
        asm.set_fast_path(fast_path)
        outs = pa.alloc_outputs(n, n * 600 + (1 << 28), "cuda:0")
        b = pa.DeviceBatch(n=n, fwd=f, rev=r, flen=fl, rlen=rl, **outs)
        asm.assemble_device(b)
        torch.cuda.synchronize()
        asm.check_overflow()
        res = b.results_numpy().copy()
        c = b.counters_dict()
        assert c["count"] == n == c["ok"] + c["noalgn"] + c["lowq"] + c["badr"] + c["nofp"] + c["norp"] + sum(c["rejected"])
        assert int(c["overlapcount"].sum()) == c["ok"] == int((res["status"] == 0).sum())
        ok = res["status"] == 0
        L = spec.read_len
        assert (res["seq_len"][ok].astype(np.int64) == 2 * L - res["overlap"][ok].astype(np.int64) - res["fwd_offset"][ok] - res["rev_offset"][ok]).all()
        rows = ok & (res["seq_len"] > 0)   # (primers "found" at the very end of both reads leave an empty merge: no row)
        offs = res["seq_offset"][rows].astype(np.int64)
        order = np.argsort(offs)
        ends = offs[order] + res["seq_len"][rows][order]
        assert (offs % 16 == 0).all() and (ends[:-1] <= offs[order][1:]).all() and ends[-1] <= int(b.cursor.item())
        recs.append((res, c))
        del b, outs
        torch.cuda.empty_cache()
    (ra, ca), (rb, cb) = recs
    for k in ("status", "overlap", "seq_len", "examined", "mismatches", "degenerates", "fwd_offset", "rev_offset", "slow", "reject_module"):
        assert np.array_equal(ra[k], rb[k]), k
    assert np.array_equal(ca["overlapcount"], cb["overlapcount"])
    if sums == "exact":
        assert np.array_equal(ra["quality"].view(np.uint64), rb["quality"].view(np.uint64))
    else:
        assert np.abs(ra["quality"] - rb["quality"]).max() <= 1e-9


def _row_checksums(b, res, ok, sample=4000):
    idx = np.nonzero(ok)[0][:: max(1, int(ok.sum()) // sample)]
    nt = b.seq_nt
    p = b.seq_p
    out = np.zeros((len(idx), 2))
    for k, i in enumerate(idx):
        off, m = int(res["seq_offset"][i]), int(res["seq_len"][i])
        out[k, 0] = float(nt[off:off + m].double().sum().item())
        out[k, 1] = float(p[off:off + m].sum().item())
    return out


def test_unequal_strides_and_api_validation(built):
    """Forward and reverse rows may have different strides; bad strides / alignment / a missing cursor are
    refused with an error (never silently computed)."""
    import ctypes as C

    import torch

    import pandaseq_amd as pa
    from pandaseq_amd import _lib, synth
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 1500)
    # reverse reads truncated to 180 bases in a narrower row
    rev2 = np.ascontiguousarray(rev[:, :184])
    rlen2 = np.minimum(rlen, 180).astype(np.uint16)
    asm = pa.Assembler(0)
    b = _run(asm, fwd, rev2, flen, rlen2, seq_stride=0)
    o = _oracle("simple_bayesian", fwd, rev2, flen, rlen2)
    compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3])
    compare_counters(b.counters_dict(), o[3])
    # validation: stride not a multiple of 8
    bad = pa.upload_batch(np.ascontiguousarray(fwd[:, :252]), rev, np.minimum(flen, 250), rlen, device="cuda:0", seq_stride=0)
    with pytest.raises(pa.PamdError, match="multiples of 8"):
        asm.assemble_device(bad)
    # validation: seq_stride not a multiple of 16
    b.seq_stride = 500
    with pytest.raises(pa.PamdError, match="multiple of 16"):
        asm.assemble_device(b)
    torch.cuda.synchronize()


def test_arena_overflow_is_reported(built):
    """A compact arena that is too small must surface as an error from the host entry point, and the
    pairs that did not fit must not claim a row."""
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 4000)
    asm = pa.Assembler(0)
    with pytest.raises(pa.PamdError, match="arena too small"):
        asm.assemble_host(fwd, rev, flen, rlen, seq_stride=0, capacity=200_000)


def test_primer_scan_when_exp_collapses_two_alignments(built):
    """offset.c:72-74 compares exp(sum / (index + 1)) and keeps the EARLIER index on a tie.  Two alignments whose values
    differ by less than exp() can resolve are such a tie for the reference although the values themselves differ: the
    forward primer is planted twice, with qualities searched so that the second copy's value exceeds the first's by a few
    1e-18 (same exponential) or falls short of it, or exceeds it by 1e-14 (different exponentials).  The device, which
    compares before exp, must give the reference's answer in all three cases."""
    from pandaseq_amd import synth
    from oracle import pyoracle as po
    spec = synth.config_spec(2)
    primer = synth.nt_codes(spec.fprimer)
    P = len(primer)
    score = po.tables_np()["score"]
    rng = np.random.default_rng(2024)
    n1 = 1_500_000
    q1 = rng.integers(25, 42, size=(n1, P))
    v1 = np.cumsum(score[q1], axis=1)[:, -1] / 18.0          # the first copy starts at 0: index + 1 = P + 1 = 18
    order = np.argsort(v1)
    v1s = v1[order]
    cases = []   # (quality vector 1, quality vector 2, kind)
    want = {"tie": 6, "below": 3, "clear": 3}
    for _ in range(40):
        q2 = rng.integers(20, 39, size=(400_000, P))
        v2 = np.cumsum(score[q2], axis=1)[:, -1] / 36.0      # the second copy starts at 18: index + 1 = 36
        j = np.searchsorted(v1s, v2, side="left")             # v1s[j-1] < v2 <= v1s[j]
        lo = np.clip(j - 1, 0, n1 - 1)
        d_up = v2 - v1s[lo]                                    # > 0: the later alignment is (slightly) better before exp
        hi = np.clip(j, 0, n1 - 1)
        d_dn = v1s[hi] - v2
        for k in np.nonzero((d_up > 0) & (d_up <= 3e-18))[0]:
            if want["tie"] > 0:
                cases.append((q1[order[lo[k]]], q2[k], "tie"))
                want["tie"] -= 1
        for k in np.nonzero((d_dn > 0) & (d_dn <= 3e-18))[0]:
            if want["below"] > 0:
                cases.append((q1[order[hi[k]]], q2[k], "below"))
                want["below"] -= 1
        for k in np.nonzero((d_up > 5e-15) & (d_up <= 2e-14))[0]:
            if want["clear"] > 0:
                cases.append((q1[order[lo[k]]], q2[k], "clear"))
                want["clear"] -= 1
        if not any(want.values()):
            break
    assert sum(1 for c in cases if c[2] == "tie") >= 2, "the search must produce ties for the test to mean anything"
    n = len(cases)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, n)
    other = {1: 2, 2: 4, 4: 8, 8: 1}
    for i, (qa, qb, _kind) in enumerate(cases):
        fwd[i, 0:P, 0] = primer
        fwd[i, 0:P, 1] = qa
        fwd[i, P, 0] = other[int(primer[0])]      # no third alignment
        fwd[i, P, 1] = 40
        fwd[i, 18:18 + P, 0] = primer
        fwd[i, 18:18 + P, 1] = qb
    ocfg = dict(fprimer=primer, rprimer=synth.nt_codes(spec.rprimer, complement=True))
    o = _oracle("pear", fwd, rev, flen, rlen, **ocfg)
    offs = o[0]["fwd_offset"]
    kinds = np.array([c[2] for c in cases])
    assert (o[0]["status"] != 2).all()
    assert (offs[kinds == "tie"] == P).all(), "the reference keeps the first copy when exp() cannot tell the two apart"
    assert (offs[kinds == "below"] == P).all() and (offs[kinds == "clear"] == 18 + P).all()
    for exact in (True, False):
        asm = _asm_for("pear", spec, primers=True, exact=exact)
        b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
        compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=exact)
        compare_counters(b.counters_dict(), o[3])


def _plant(rng, rows, lens, primer, where, subst, lowq):
    """Writes one member of every primer base's code set into each read at its offset, with a few wrong bases and a few
    low qualities (rows: u8[n, stride, 2] = nt, qual)."""
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


@pytest.mark.parametrize("plen,penalty", [(1, 0.0), (3, 0.0), (8, 0.0), (17, 0.0), (17, 0.02), (23, 0.0), (40, 0.0), (64, 0.0), (65, 0.0), (100, 0.0), (100, 0.01)])
def test_primer_kernel_on_planted_primers(built, plen, penalty):
    """pamd_primer.hip against offset.c:47-112 as the oracle restates it: primers of every length class (the eight-base
    filter, the whole-wave sums, the walk in step, primers longer than a wave), with ambiguity codes and N among their
    first bases, planted anywhere in the read or nowhere, damaged and on poor qualities; with and without a penalty."""
    from pandaseq_amd import synth
    import pandaseq_amd as pa
    rng = np.random.default_rng(100 + plen)
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = synth.generate_host(spec, 31000, 3000)
    n = len(flen)
    codes = np.array([1, 2, 4, 8, 1, 2, 4, 8, 3, 5, 10, 12, 7, 14, 15], np.uint8)   # mostly plain bases, some ambiguity codes, N
    fp = rng.choice(codes, plen)
    rp = rng.choice(codes, plen)
    if plen >= 8:
        fp[rng.integers(0, 8)] = 15     # an N among the first eight bases of one primer, none in the other's
        rp[:8] = rng.choice(np.array([1, 2, 4, 8], np.uint8), 8)
    room = max(1, 250 - plen)
    wf = np.where(rng.random(n) < 0.9, rng.integers(0, room, n) * (rng.random(n) < 0.5), -1)   # half at the start, the rest anywhere, a tenth nowhere
    wr = np.where(rng.random(n) < 0.9, rng.integers(0, room, n) * (rng.random(n) < 0.5), -1)
    _plant(rng, fwd, flen, fp, wf, 0.04, 0.15)
    _plant(rng, rev, rlen, rp, wr, 0.04, 0.15)
    flen[5] = max(1, plen - 1)          # a read shorter than its primer
    rlen[6] = plen                      # and one exactly as long
    asm = pa.Assembler(0)
    asm.set_algorithm("simple_bayesian")
    asm.set_forward_primer(fp)
    asm.set_reverse_primer(rp)
    asm.set_primer_penalty(penalty)
    b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
    o = _oracle("simple_bayesian", fwd, rev, flen, rlen, fprimer=fp, rprimer=rp, primer_penalty=penalty)
    compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3])
    compare_counters(b.counters_dict(), o[3])
    # (the reference finds "a primer" in nearly every read: offset.c:72 divides by index + 1, so late weak hits pass)
    ok = o[0]["status"] == 0
    if penalty == 0.0 and plen >= 8:
        assert len(np.unique(o[0]["fwd_offset"][ok])) > 8 and len(np.unique(o[0]["rev_offset"][ok])) > 8, "the case must spread its primers over the reads"


@pytest.mark.parametrize("algo", ["pear", "rdp_mle"])
def test_pairs_that_scan_every_overlap(built, algo, sums):
    """pear / rdp_mle pairs without a seeded overlap: every overlap is a candidate (assembler.c:111-117).  The kernel bounds
    most of them from above by counts and walks the rest; the winner, its score, `examined` and `slow` must be the
    reference's -- on unrelated reads of every quality mix, with Ns, ambiguity codes, repeats and ragged lengths."""
    import pandaseq_amd as pa
    rng = np.random.default_rng(77)
    n, stride = 900, 256
    fwd = np.zeros((n, stride, 2), np.uint8)
    rev = np.zeros((n, stride, 2), np.uint8)
    flen = rng.integers(20, 251, n).astype(np.uint16)
    rlen = rng.integers(20, 251, n).astype(np.uint16)
    for i in range(n):
        kind = i % 6
        for row, ln in ((fwd, flen[i]), (rev, rlen[i])):
            row[i, :ln, 0] = rng.choice(np.array([1, 2, 4, 8], np.uint8), ln)
            if kind == 0:       # good qualities throughout
                row[i, :ln, 1] = rng.integers(30, 42, ln)
            elif kind == 1:     # poor qualities throughout
                row[i, :ln, 1] = rng.integers(0, 15, ln)
            else:               # anything
                row[i, :ln, 1] = rng.integers(0, 47, ln)
            if kind == 3:       # Ns and ambiguity codes
                at = rng.integers(0, ln, max(1, ln // 12))
                row[i, at, 0] = rng.choice(np.array([15, 3, 5, 6, 9, 10, 12, 7, 14], np.uint8), len(at))
        if kind == 4:           # a two-base repeat in both reads: many good overlaps, ties
            fwd[i, :flen[i], 0] = np.tile(np.array([1, 4], np.uint8), 126)[:flen[i]]
            rev[i, :rlen[i], 0] = np.tile(np.array([4, 1], np.uint8), 126)[:rlen[i]]
        if kind == 5:           # the same quality everywhere: equal scores wherever the counts are equal
            fwd[i, :flen[i], 1] = 30
            rev[i, :rlen[i], 1] = 30
    asm = _asm_for(algo, exact=sums == "exact")
    b = _run(asm, fwd, rev, flen, rlen, seq_stride=0)
    o = _oracle(algo, fwd, rev, flen, rlen)
    compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=sums == "exact")
    compare_counters(b.counters_dict(), o[3])
    assert (o[0]["slow"] != 0).sum() > n // 3, "the case must hold pairs that scan every overlap"


@pytest.mark.parametrize("cfg,algo", [(1, "simple_bayesian"), (1, "rdp_mle"), (2, "pear")])
def test_batch_sizes_around_the_work_queues_grain(built, cfg, algo, sums):
    """The kernels hand out work sixteen pairs at a time (one at a time over the lean kernel's list) and finish parked pairs
    64 at a time: batches just below, at and above those grains, against the oracle."""
    from pandaseq_amd import synth
    spec = synth.config_spec(cfg)
    fwd, rev, flen, rlen = synth.generate_host(spec, 123456, 1100)
    kw = dict(fprimer=synth.nt_codes(spec.fprimer), rprimer=synth.nt_codes(spec.rprimer, complement=True)) if cfg == 2 else {}
    for n in (1, 2, 15, 16, 17, 31, 63, 64, 65, 127, 129, 255, 1024, 1025, 1100):
        asm = _asm_for(algo, spec, primers=cfg == 2, exact=sums == "exact")
        b = _run(asm, fwd[:n], rev[:n], flen[:n], rlen[:n], seq_stride=0)
        o = _oracle(algo, fwd[:n], rev[:n], flen[:n], rlen[:n], **kw)
        compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=sums == "exact")
        compare_counters(b.counters_dict(), o[3])
