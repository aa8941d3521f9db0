"""CPU suite: the oracle against what pins it, the product's host-side logic, and the C ABI surface.
Nothing here needs a GPU."""
import ctypes as C
import glob
import math
import os
import sys
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def po(built):
    from oracle import pyoracle
    return pyoracle


# ---------------------------------------------------------------- tables: pinned on the reference's own generator

TABLE_NAMES = ("score", "score_err", "match_sb", "mismatch_sb", "match_pear", "mismatch_pear", "mismatch_rdp", "mismatch_rdp_asm",
               "match_uparse", "mismatch_uparse")


def test_oracle_tables_equal_reference_mktable(po):
    """oracle/_ref/libreftable.so is table.c as written by the reference's mktable.c + tablebuilder.c."""
    ref = po.ref_tables_np()
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference; built by __graft_entry__.build() in the dev container)")
    mine = po.tables_np()
    assert mine["qual_nn"] == ref["qual_nn"] == -1.38629
    for k in TABLE_NAMES:
        assert np.array_equal(mine[k].view(np.uint64), ref[k].view(np.uint64)), k


def test_product_tables_equal_reference_mktable(po):
    """pamd_tables_get() (host code of the shipped library) against the same reference output."""
    from pandaseq_amd import _lib
    ref = po.ref_tables_np()
    prod = _lib.tables()
    ora = po.tables_np()
    for k in TABLE_NAMES:
        assert np.array_equal(prod[k].view(np.uint64), ora[k].view(np.uint64)), k
        if ref is not None:
            assert np.array_equal(prod[k].view(np.uint64), ref[k].view(np.uint64)), k
    assert prod["qual_nn"] == -1.38629


def test_table_known_values(po):
    """spot values quoted in SURVEY.md 8a/7 from the reference's table.c"""
    t = po.tables_np()
    assert t["score"][0] == -2.0                       # mktable.c:69-71
    assert t["mismatch_rdp_asm"][0][0] == -708.396     # DBL_MIN branch, mktable.c:103
    assert t["match_sb"][0][46] == -11.6905
    for k in TABLE_NAMES[2:-1]:
        assert np.array_equal(t[k], t[k].T), f"{k} is expected to be symmetric"
    assert not np.array_equal(t["mismatch_uparse"], t["mismatch_uparse"].T)  # the one asymmetric table
    assert np.array_equal(t["mismatch_rdp"], t["mismatch_sb"])


# ---------------------------------------------------------------- known answers recorded from a real reference run

def _perfect_pair(L, tlen, q=40, seed=1):
    rng = np.random.default_rng(seed)
    T = rng.integers(0, 4, tlen)
    nts = np.array([1, 2, 4, 8], dtype=np.uint8)
    f = np.zeros((1, L, 2), np.uint8)
    r = np.zeros((1, L, 2), np.uint8)
    f[0, :, 0] = nts[T[:L]]
    f[0, :, 1] = q
    r[0, :, 0] = nts[T[::-1][:L]]
    r[0, :, 1] = q
    return f, r


def test_known_answer_2x150_overlap40(po):
    """SURVEY.md 8c 'Observed': a synthetic 2x150 pair with a 40-base overlap gives overlap 40,
    1 overlap examined, estimated_overlap_probability = -396.13854265506922 (simple_bayesian)."""
    f, r = _perfect_pair(150, 260)
    res, nt, p, ctr = po.assemble_batch(po.make_config(), f, r, [150], [150])
    assert res["status"][0] == 0 and res["overlap"][0] == 40 and res["examined"][0] == 1
    assert res["est_prob"][0] == -396.13854265506922
    assert res["seq_len"][0] == 260 and res["mismatches"][0] == 0


def test_phred_quirks(po):
    """SURVEY.md Appendix A quirk list: phred(QS[46]) = 45, phred(QS[0]) = phred(QS[1]) = 1, phred(NN) = 2."""
    t = po.tables()
    L = po.lib()
    qs = po.tables_np()["score"]
    assert L.po_result_phred(C.byref(t), qs[46]) == 45
    assert L.po_result_phred(C.byref(t), qs[0]) == 1
    assert L.po_result_phred(C.byref(t), qs[1]) == 1
    assert L.po_result_phred(C.byref(t), -1.38629) == 2


def test_phred_is_one_below_qs0_and_monotone_above(po):
    """The device derives min_phred's verdict from the smallest log-p of a row.  That needs
    panda_result_phred (nt.c:126-150) to be 1 for p <= qual_score[0] and non-decreasing for larger p.
    The function is piecewise constant with breakpoints at the 47 table values: check each value and
    its two neighbouring doubles, plus a dense grid."""
    t = po.tables()
    L = po.lib()
    qs = po.tables_np()["score"]
    assert L.po_result_phred(C.byref(t), float(qs[0])) == 1 and L.po_result_phred(C.byref(t), -700.0) == 1
    pts = set()
    for v in qs:
        for w in (np.nextafter(v, -np.inf), v, np.nextafter(v, np.inf)):
            if w > qs[0]:
                pts.add(float(w))
    pts |= {float(v) for v in np.linspace(qs[0], 0.0, 20001)[1:]}
    ph = np.array([L.po_result_phred(C.byref(t), v) for v in sorted(pts)])
    assert (np.diff(ph) >= 0).all() and ph.min() == 0 and ph.max() == 45


def test_exported_result_phred_matches_oracle(po, built):
    """libpandaseq_amd's host panda_result_phred / panda_quality_log_probability (what a reference-compiled
    plugin binary links against, plugin_min_phred.c:16) against the oracle's restatement of nt.c:120-150
    at every breakpoint, its neighbours, and a dense grid."""
    from pandaseq_amd import _lib

    class Result(C.Structure):
        _fields_ = [("nt", C.c_char), ("p", C.c_double)]

    class Qual(C.Structure):
        _fields_ = [("nt", C.c_char), ("qual", C.c_byte)]

    assert C.sizeof(Result) == 16 and C.sizeof(Qual) == 2
    P = _lib.lib()
    P.panda_result_phred.restype = C.c_byte
    P.panda_result_phred.argtypes = [C.POINTER(Result)]
    P.panda_quality_log_probability.restype = C.c_double
    P.panda_quality_log_probability.argtypes = [C.POINTER(Qual)]
    t = po.tables()
    L = po.lib()
    qs = po.tables_np()["score"]
    pts = {-1e300, -700.0, -2.0, 0.0, 1.0, -1.38629}
    for v in qs:
        pts |= {float(np.nextafter(v, -np.inf)), float(v), float(np.nextafter(v, np.inf))}
    pts |= {float(v) for v in np.linspace(-3.0, 0.0, 5001)}
    for v in sorted(pts):
        assert P.panda_result_phred(C.byref(Result(b"\x01", v))) == L.po_result_phred(C.byref(t), v), v
    for q in range(-128, 128):
        assert P.panda_quality_log_probability(C.byref(Qual(b"\x01", q))) == qs[min(max(q, 0), 46)]


def test_kmer_iterator_rules(po):
    """misc.h:41-45: first forward k-mer at position 8; a 9-wide N exclusion window; A and ambiguity codes hash as 0."""
    L = po.lib()
    n = 40
    s = np.zeros((n, 2), np.uint8)
    s[:, 0] = 8  # all T
    km = np.zeros(n, np.uint16)
    ps = np.zeros(n, np.uint16)
    c = L.po_kmers_forward(s.ctypes.data, n, km.ctypes.data, ps.ctypes.data)
    assert c == n - 8 and ps[0] == 8 and km[0] == 0xFFFF
    s[20, 0] = 15  # an N at 20 blocks positions 20..28
    c = L.po_kmers_forward(s.ctypes.data, n, km.ctypes.data, ps.ctypes.data)
    assert list(ps[:c]) == list(range(8, 20)) + list(range(29, 40))
    c = L.po_kmers_reverse(s.ctypes.data, n, km.ctypes.data, ps.ctypes.data)
    assert list(ps[:c]) == list(range(31, 20, -1)) + list(range(11, -1, -1))
    s[:, 0] = 1  # all A -> code 0
    s[5, 0] = 3  # ambiguity code M -> code 0 too, and not an N
    c = L.po_kmers_forward(s.ctypes.data, n, km.ctypes.data, ps.ctypes.data)
    assert c == n - 8 and not km[:c].any()


def test_strict_argmax_and_defaults(po):
    """Strict '>' in ascending overlap keeps the smallest overlap among equal scores; defaults are
    minoverlap 2 / maxoverlap min(len) / threshold log 0.6 (assembler_support.c:76-97)."""
    cfg = po.make_config()
    assert cfg.minoverlap == 2 and cfg.maxoverlap == 0 and cfg.threshold == math.log(0.6)
    assert cfg.sb_pmatch == math.log(0.25 * (1 - 2 * 0.36 + 0.36 * 0.36))
    # homopolymer pair: every overlap scores identically per base -> all candidates examined (SLOW)
    L = 60
    f = np.zeros((1, L, 2), np.uint8)
    r = np.zeros((1, L, 2), np.uint8)
    f[0, :, 0] = 8
    r[0, :, 0] = 8
    f[0, :, 1] = 30
    r[0, :, 1] = 30
    res, *_ = po.assemble_batch(cfg, f, r, [L], [L])
    assert res["slow"][0] == 0 and res["status"][0] == 0
    # with 'T' homopolymers every k-mer is the same: only the first two forward positions (8, 9) seed
    assert res["examined"][0] == len({L + L - rp - fp - 1 for fp in (8, 9) for rp in range(0, L - 8) if 2 <= L + L - rp - fp - 1 <= L})


def test_counter_order_and_bad_reads(po):
    f, r = _perfect_pair(150, 260)
    cfg = po.make_config()
    res, _, _, ctr = po.assemble_batch(cfg, np.repeat(f, 3, 0), np.repeat(r, 3, 0), [150, 1, 150], [150, 150, 0])
    assert list(res["status"]) == [0, 1, 1] and ctr["count"] == 3 and ctr["badr"] == 2 and ctr["ok"] == 1
    assert ctr["overlapcount"][40] == 1 and ctr["longest_overlap"] == 40


# ---------------------------------------------------------------- golden vectors (oracle regression + inputs for the GPU suite)

@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))))
def test_oracle_matches_golden(po, path):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden
    name = os.path.basename(path)[:-4]
    g = np.load(path)
    d = make_golden.build_case(name)
    assert np.array_equal(d["fwd"], g["fwd"]) and np.array_equal(d["rev"], g["rev"]), "generator changed"
    assert np.array_equal(d["res"], g["res"])
    assert np.array_equal(d["cs_nt"], g["cs_nt"])
    assert np.array_equal(d["cs_p"].view(np.uint64), g["cs_p"].view(np.uint64))
    assert np.array_equal(d["ctr_main"], g["ctr_main"])


def test_oracle_mt_equals_single_thread(po):
    from pandaseq_amd import synth
    f, r, fl, rl = synth.generate_numpy(synth.config_spec(1), 0, 3000)
    cfg = po.make_config("rdp_mle", modules=[("min_phred", 2), ("min_overlapbits", 15)])
    a = po.assemble_batch(cfg, f, r, fl, rl, threads=1)
    b = po.assemble_batch(cfg, f, r, fl, rl, threads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint64), b[2].view(np.uint64))
    for k in ("count", "ok", "noalgn", "lowq", "slow", "longest_overlap"):
        assert a[3][k] == b[3][k]
    assert np.array_equal(a[3]["overlapcount"], b[3]["overlapcount"]) and np.array_equal(a[3]["rejected"], b[3]["rejected"])


# ---------------------------------------------------------------- the C ABI surface (no compute without a GPU)

def test_library_exports_every_declared_symbol(built):
    from pandaseq_amd import _lib
    L = _lib.lib()
    declared = set()
    for hdr in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(hdr).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b((?:pamd|panda)_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations found"
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    assert set(_lib.SYMBOLS) <= declared


def test_record_layout_matches_header(built):
    from pandaseq_amd import _lib
    assert _lib.RESULT_DTYPE.itemsize == 48
    assert _lib.RESULT_DTYPE.fields["seq_len"][1] == 24 and _lib.RESULT_DTYPE.fields["status"][1] == 38
    assert _lib.NUM_COUNTERS == 916


def test_compute_fails_loudly_without_gpu(built):
    """No CPU fallback: creating an assembler on a box without a GPU must raise."""
    import torch

    import pandaseq_amd as pa
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pa.PamdError, match="no CPU fallback"):
        pa.Assembler(0)


def test_missing_library_fails_loudly(built, monkeypatch):
    from pandaseq_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("PAMD_LIB", "/nonexistent/libpandaseq_amd.so")
    with pytest.raises(_lib.LibraryMissing):
        _lib.lib()


def test_product_never_imports_oracle():
    """The oracle is a checker only: nothing under pandaseq_amd/, include/ or tools/ may reference it (only
    tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs do)."""
    offenders = []
    for base in ("pandaseq_amd", "include", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                    text = open(os.path.join(dirpath, fn), errors="ignore").read()
                    if re.search(r"\bimport\s+oracle|from\s+oracle|liboracle|panda_oracle|pyoracle", text):
                        offenders.append(os.path.join(dirpath, fn))
    assert not offenders, offenders


# ---------------------------------------------------------------- host logic

def test_synth_is_deterministic_and_chunk_invariant():
    from pandaseq_amd import synth
    spec = synth.config_spec(2)
    a = synth.generate_numpy(spec, 100, 300)
    b = synth.generate_numpy(spec, 0, 400)
    for x, y in zip(a, b):
        assert np.array_equal(x, y[100:400])
    L = spec.read_len
    fp = synth.nt_codes(spec.fprimer)
    hits = sum(np.array_equal(a[0][i, :17, 0], fp) for i in range(300))
    assert hits > 100  # most un-prefixed pairs start with an intact forward primer


def test_stat_block_format():
    from pandaseq_amd import _lib
    from pandaseq_amd.stats import format_stat_block, shard_range
    c = np.zeros(_lib.NUM_COUNTERS, np.int64)
    c[_lib.CTR["count"]] = 10
    c[_lib.CTR["ok"]] = 7
    c[_lib.CTR["noalgn"]] = 2
    c[_lib.CTR_REJECTED0] = 1
    c[_lib.CTR_OVERLAP0 + 3] = 7
    txt = format_stat_block(c, module_names=["min_phred"], name="g0")
    lines = txt.split("\n")
    assert lines[1] == "g0\tSTAT\tNOALGN\t2" and "g0\tSTAT\tmin_phred\t1" in lines and lines[-2] == "g0\tSTAT\tOK\t7"
    assert lines[-1] == "g0\tSTAT\tOVERLAPS\t0 0 0 7"   # proxy.c:182-187: bins 0..longest_overlap
    assert [shard_range(10, r, 3) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]


# ---------------------------------------------------------------- pinned on the reference's own header-only macros

def test_kmer_macros_of_the_reference_pin_the_oracle(po):
    """FOREACH_KMER / FOREACH_KMER_REVERSE (misc.h:36-45), compiled from the reference's own text (oracle/ref_macros.c),
    against the oracle's restatement: same (k-mer, position) emissions in the same order on adversarial reads (Ns, IUPAC
    codes, nt 0, ragged lengths down to 0) -- SURVEY.md 8a row a3."""
    import ctypes as C

    import pins
    from pandaseq_amd import synth
    from util import adversarial_pairs
    if pins.lib() is None:
        pytest.skip("oracle/_ref not built (needs /root/reference; built by __graft_entry__.build() in the dev container)")
    assert pins.lib().ref_kmer_len() == 8
    L = po.lib()
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = adversarial_pairs(spec, 400, seed=3)
    emitted = 0
    for reads, lens in ((fwd, flen), (rev, rlen)):
        for i in range(len(reads)):
            n = int(lens[i])
            row = np.ascontiguousarray(reads[i])
            for reverse in (False, True):
                rk, rp = pins.kmers(row, n, reverse)
                k = np.zeros(512, dtype=np.uint16)
                p = np.zeros(512, dtype=np.uint16)
                m = (L.po_kmers_reverse if reverse else L.po_kmers_forward)(row.ctypes.data, n, k.ctypes.data, p.ctypes.data)
                assert m == len(rk) and np.array_equal(k[:m], rk) and np.array_equal(p[:m], rp), (i, reverse)
                emitted += m
    assert emitted > 100000


def test_domain_macros_of_the_reference_pin_the_oracle(po):
    """PHREDCLAMP (prob.h:23) over every char, PANDA_NT_IS_N / PANDA_NT_IS_DEGN (pandaseq-nt.h:51-55) over every byte."""
    import pins
    R = pins.lib()
    if R is None:
        pytest.skip("oracle/_ref not built")
    L = po.lib()
    for q in range(-128, 128):
        assert R.ref_phredclamp(q) == L.po_phredclamp(q) == min(46, max(0, q))
    for nt in range(16):   # panda_nt is a 4-bit IUPAC set (pandaseq-nt.h:35-50); the macro's bit trick means nothing beyond it
        assert R.ref_nt_is_degn(nt) == L.po_nt_is_degenerate(nt) == (1 if bin(nt).count("1") != 1 else 0), nt
    for nt in range(256):
        assert R.ref_nt_is_n(nt) == (1 if nt == 15 else 0)


def test_candidate_sets_from_reference_kmers_match_the_oracle(po):
    """overlaps_examined rebuilt from the reference macros' k-mers (tests/pins.py) equals the oracle's on every pair
    whose scan ran: the seeding of assembler.c:84-118 on top of pinned k-mers."""
    import pins
    from pandaseq_amd import synth
    from util import adversarial_pairs
    if pins.lib() is None:
        pytest.skip("oracle/_ref not built")
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = adversarial_pairs(spec, 1500, seed=9)
    res = po.assemble_batch(po.make_config("simple_bayesian"), fwd, rev, flen, rlen, want_seq=False)[0]
    checked = 0
    for i in range(len(res)):
        ex = pins.examined(fwd[i], rev[i], flen[i], rlen[i])
        if ex is None:
            assert res["status"][i] in (1, 4)
            continue
        assert int(res["examined"][i]) == ex, i
        checked += 1
    assert checked > 1000


def test_cpu_baseline_harness_assembles_what_the_plain_entry_point_does(built):
    """bench.py's cpu_baseline leg (po_bench_mt: pinned pthreads over private, self-touched copies of their slices) returns the
    records and counters of po_assemble_batch, whatever the thread count."""
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = synth.generate_numpy(spec, 0, 3000)
    cfg = po.make_config("rdp_mle")
    want, _, _, wctr = po.assemble_batch(cfg, fwd, rev, flen, rlen, want_seq=False)
    for threads in (1, 3, 8):
        secs, got, ctr = po.bench_mt(cfg, fwd, rev, flen, rlen, threads)
        assert secs > 0
        for name in ("status", "overlap", "seq_len", "examined", "mismatches"):
            assert np.array_equal(got[name], want[name]), (threads, name)
        assert np.array_equal(got["quality"].view(np.uint64), want["quality"].view(np.uint64))
        for k in ("count", "ok", "noalgn", "lowq", "badr", "slow"):
            assert int(ctr[k]) == int(wctr[k]), (threads, k)


def test_the_bench_kernels_do_not_spill(built, tmp_path):
    """The four lean-kernel instantiations BASELINE.json's configurations run (and the cold function two of them call) are
    compiled once more with the resource remarks on: none may use scratch memory, and the many-candidate scan of the counting
    scorers must stay within 80 vector registers -- what it clobbers the hot path cannot keep across the call.  (Round 4: a
    change inside that cold function took it from 56 to 128 registers and config 1 lost 7 % to spills nobody had written.)"""
    import subprocess
    src = os.path.join(ROOT, "pandaseq_amd", "csrc", "pamd_fast.hip")
    flags = subprocess.check_output(["make", "-s", "-C", os.path.dirname(src), "fast-flags"], text=True).split()
    assert "-ffp-contract=off" in flags
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950"] + flags + ["-DPAMD_MARK", "-DFK_NO_DUAL", "-x", "hip", src, "-S",     # (PAMD_MARK: phase names as ISA comments)
                          "--cuda-device-only", "-o", str(tmp_path / "fast.s"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    usage = {}
    name = None
    for line in out.stderr.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split()[0]
            usage[name] = {}
        elif name and "remark:" in line and ":" in line.split("remark:")[1]:
            k, v = line.split("remark:")[1].split("[-Rpass")[0].rsplit(":", 1)
            usage[name][k.strip()] = v.strip()
    listing = (tmp_path / "fast.s").read_text().splitlines()
    # <ALGO, EXACT, PHRED, LONG, PRIM, NARROW, POST>: config 1, config 2, config 3, config 4
    for inst in ("ILi0ELb0ELb0ELb0ELb0ELb0ELb0E", "ILi1ELb0ELb0ELb1ELb1ELb0ELb0E", "ILi2ELb0ELb1ELb0ELb0ELb0ELb0E", "ILi0ELb0ELb0ELb0ELb0ELb1ELb0E"):
        hits = [n for n in usage if "pamd_fast_kernel" + inst in n]
        assert len(hits) == 1, (inst, hits)
        u = usage[hits[0]]
        assert int(u["VGPRs"]) <= 128, (inst, u)
        if inst != "ILi1ELb0ELb0ELb1ELb1ELb0ELb0E":
            assert u["VGPRs Spill"] == "0" and u["ScratchSize [bytes/lane]"] == "0", (inst, u)
        else:
            # config 2's kernel keeps one loop-invariant pair of registers in scratch -- written once per wave before the pair
            # loop, read where a pair is handed to the general kernel -- and saves registers around its one cold call; what
            # it may not do is touch scratch in the phases every pair goes through
            assert int(u["VGPRs Spill"]) <= 2, (inst, u)
        start = next(i for i, t in enumerate(listing) if t.startswith(hits[0] + ":"))
        phase, per_phase = "prologue", {}
        for t in listing[start:]:
            t = t.strip()
            if t.startswith("s_endpgm"):
                break
            if t.startswith("; MARK "):
                phase = t.split()[2]
            elif t.startswith("scratch_"):
                per_phase[phase] = per_phase.get(phase, 0) + 1
        assert set(per_phase) <= {"prologue", "loop"}, (inst, per_phase)   # ("loop": the loop's end and, laid out behind it, the cold blocks)
    # ... and, over EVERY instantiation of the file (tools/isa_hazards.py): nothing touches the destination of a bare LDS read
    # (lds_read_u16_bare: a load the compiler's waitcnt insertion does not see) before its wait on any path of the kernel's
    # control-flow graph, and nothing the compiler emits uses M0, which the parking code writes behind its back
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_hazards
    assert isa_hazards.count_bare_reads(listing) > 1000
    assert isa_hazards.scan_bare_reads(listing) == []
    assert isa_hazards.scan_m0(listing) == []
    cold = re.findall(r"\.set \.L(\S*scan_manyILi0E\S*)\.num_vgpr, (\d+)", (tmp_path / "fast.s").read_text())
    assert len(cold) == 1 and int(cold[0][1]) <= 80, cold     # (56 when this was written, 65 since the uniform regions are no longer structurized)


def test_the_build_refuses_a_hazardous_listing(built, tmp_path):
    """tools/isa_hazards.py is a step of the BUILD (pandaseq_amd/csrc/Makefile: the lean kernel's object is moved into place only
    after the listing of the same compilation has passed the scan), not only of this suite: `make isa-check` passes on the listing
    the library was built from and fails on a copy with a planted hazard -- a register move that reads the destination of a bare
    LDS read before its wait, and a compiler-side use of M0."""
    import glob
    import subprocess
    csrc = os.path.join(ROOT, "pandaseq_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    assert "$(ISA_CHECK) $(FAST_LISTING)" in mk and "mv $(OBJDIR)/fast/pamd_fast.o $@" in mk
    listings = glob.glob(os.path.join(ROOT, "pandaseq_amd", "libpandaseq_amd.so.obj", "*", "fast", "pamd_fast-hip-amdgcn-amd-amdhsa-gfx950.s"))
    if not listings:
        pytest.skip("the library was not built in this tree (no listing kept)")
    listing = max(listings, key=os.path.getmtime)
    assert subprocess.run(["make", "-s", "-C", csrc, "isa-check", "ISA_LISTING=" + listing], capture_output=True).returncode == 0
    lines = open(listing).read().splitlines()
    at = next(i for i, t in enumerate(lines) if t.strip().startswith("ds_read_u16 v") and lines[i - 1].strip().startswith(";;#ASMSTART"))
    reg = lines[at].strip().split()[1].rstrip(",")
    end = next(i for i in range(at, len(lines)) if lines[i].strip().startswith(";;#ASMEND"))
    planted = lines[:end + 1] + ["\tv_mov_b32_e32 v0, " + reg] + lines[end + 1:]
    (tmp_path / "planted.s").write_text("\n".join(planted))
    out = subprocess.run(["make", "-s", "-C", csrc, "isa-check", "ISA_LISTING=" + str(tmp_path / "planted.s")], capture_output=True, text=True)
    assert out.returncode != 0 and "1 uses of a pending register" in out.stdout, out.stdout[-500:]
    planted = lines[:end + 1] + ["\ts_movrels_b32 s0, s1", "\ts_mov_b32 m0, s2"] + lines[end + 1:]
    (tmp_path / "planted_m0.s").write_text("\n".join(planted))
    out = subprocess.run(["make", "-s", "-C", csrc, "isa-check", "ISA_LISTING=" + str(tmp_path / "planted_m0.s")], capture_output=True, text=True)
    assert out.returncode != 0 and "1 compiler-emitted uses of m0" in out.stdout, out.stdout[-500:]


def test_row_checksums_two_ways(built):
    """bench.py's at-scale row parity rests on two implementations of the same per-row checksums -- the oracle's
    (po_row_checksums_mt, C, on the host) and the product's (pandaseq_amd.row_checksums, torch, where the rows lie) --: here both
    run on the same rows on the CPU, rows in a compact arena in shuffled order, and must agree bit for bit with each other and
    with the definition."""
    import torch
    import pandaseq_amd as pa
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    from pandaseq_amd._lib import RESULT_DTYPE
    spec = synth.config_spec(1)
    n = 1500
    f, r, fl, rl = synth.generate_numpy(spec, 77, n)
    cfg = po.make_config("simple_bayesian")
    res, nt, p, _ = po.assemble_batch(cfg, f, r, fl, rl)
    res2, o_nt, o_p = po.row_checksums(cfg, f, r, fl, rl, threads=3)
    assert (res == res2).all()
    # the rows into an arena, in an order that is not the pairs'
    order = np.random.default_rng(5).permutation(n)
    rec = np.zeros(n, dtype=RESULT_DTYPE)
    for k in rec.dtype.names:
        if k in res.dtype.names and k != "seq_offset":
            rec[k] = res[k]
    arena_nt = np.zeros(n * 512 + 64, dtype=np.uint8)
    arena_p = np.zeros(n * 512 + 64, dtype=np.float64)
    cur = 16
    for i in order:
        m = int(res["seq_len"][i]) if res["status"][i] == 0 else 0
        rec["seq_offset"][i] = cur if m else 0xFFFFFFFFFFFFFFFF
        arena_nt[cur:cur + m] = nt[i, :m]
        arena_p[cur:cur + m] = p[i, :m]
        cur += (m + 15) // 16 * 16
    t_res = torch.from_numpy(rec.view(np.uint8).reshape(n, 48).copy())
    a, q = pa.row_checksums(t_res, torch.from_numpy(arena_nt), torch.from_numpy(arena_p), first=100, count=1300, block=256)
    assert np.array_equal(a.numpy().view(np.uint64), o_nt[100:1400]) and np.array_equal(q.numpy().view(np.uint64), o_p[100:1400])
    i = int(np.nonzero(res["status"] == 0)[0][3])
    m = int(res["seq_len"][i])
    w = np.arange(1, m + 1, dtype=np.uint64)
    assert int(o_nt[i]) == int((nt[i, :m].astype(np.uint64) * w).sum()) and int(o_p[i]) == int((p[i, :m].view(np.uint64) * w).sum(dtype=np.uint64))
    assert int(o_nt[res["status"] != 0].sum()) == 0


def test_the_isa_hazard_scan_sees_what_it_is_for():
    """tools/isa_hazards.py on listings made for it: a register read, written or carried across a branch between a bare
    `ds_read_u16` and the wait is reported; the same code with the wait in place is not; M0 outside inline asm is."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_hazards

    def kernel(body):
        return ["_ZN12_GLOBAL__N_116pamd_fast_kernelILi0ELb0ELb0ELb0ELb0ELb0ELb0EEEv16PamdKernelParams:"] + body + ["\ts_endpgm"]
    read = ["\t;;#ASMSTART", "\tds_read_u16 v7, v3 offset:128", "\t;;#ASMEND"]
    wait = ["\ts_waitcnt lgkmcnt(0)"]
    clean = kernel(read + ["\ts_add_i32 s4, s4, s5", "\tv_mov_b32_e32 v9, v8"] + wait + ["\tv_lshrrev_b32_e32 v1, 8, v7"])
    assert isa_hazards.count_bare_reads(clean) == 1 and isa_hazards.scan_bare_reads(clean) == []
    copied = kernel(read + ["\tv_mov_b32_e32 v9, v7"] + wait)
    assert [h[3] for h in isa_hazards.scan_bare_reads(copied)] == [7]
    overwritten = kernel(read + ["\tv_add_u32_e32 v7, v1, v2"] + wait)
    assert [h[3] for h in isa_hazards.scan_bare_reads(overwritten)] == [7]
    wide = kernel(read + ["\tv_mad_u64_u32 v[12:13], s[4:5], s13, v4, v[6:7]"] + wait)     # (the upper half of a 64-bit operand counts)
    assert [h[3] for h in isa_hazards.scan_bare_reads(wide)] == [7]
    across = kernel(read + ["\ts_cbranch_scc1 .LBB0_2", "\ts_waitcnt lgkmcnt(0)", ".LBB0_2:", "\tv_mov_b32_e32 v1, v7"])      # one path skips the wait
    assert 7 in [h[3] for h in isa_hazards.scan_bare_reads(across)]    # (and -1: the kernel ends with the read still pending on that path)
    both = kernel(read + ["\ts_cbranch_scc1 .LBB0_2", "\ts_waitcnt lgkmcnt(0)", "\ts_branch .LBB0_3", ".LBB0_2:", "\ts_waitcnt lgkmcnt(0)", ".LBB0_3:", "\tv_mov_b32_e32 v1, v7"])
    assert isa_hazards.scan_bare_reads(both) == []
    assert isa_hazards.scan_m0(kernel(["\ts_mov_b32 m0, s3"])) != [] and isa_hazards.scan_m0(kernel(["\t;;#ASMSTART", "\ts_mov_b32 m0, s3", "\t;;#ASMEND"])) == []
