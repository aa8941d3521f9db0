"""panda_module_load() on plugin files and the algorithm registry: host logic of the compat layer, no GPU.
Reference behaviour: module.c:227-298 (loading), :360-391 (getters), algo.c:96-133 (registry)."""
import ctypes
import os

import numpy as np
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pandaseq_amd", "libpandaseq_amd.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        pytest.skip("libpandaseq_amd.so not built")
    L = ctypes.CDLL(LIB, mode=ctypes.RTLD_GLOBAL)
    L.panda_module_load.restype = ctypes.c_void_p
    L.panda_module_load.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    L.panda_module_unref.argtypes = [ctypes.c_void_p]
    for f in ("name", "args", "description", "usage", "version"):
        fn = getattr(L, "panda_module_get_" + f)
        fn.restype = ctypes.c_char_p
        fn.argtypes = [ctypes.c_void_p]
    L.panda_module_get_api.restype = ctypes.c_int
    L.panda_module_get_api.argtypes = [ctypes.c_void_p]
    return L


@pytest.fixture(scope="module")
def plugins(tmp_path_factory):
    d = tmp_path_factory.mktemp("plugins")
    for name in ("short_overlap", "wrong_api"):
        subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "c", "plugin_%s.c" % name), "-o", str(d / (name + ".so"))])
    return d


def test_load_plugin_file(lib, plugins):
    path = str(plugins / "short_overlap.so").encode()
    m = lib.panda_module_load(None, path + b":25")
    assert m
    assert lib.panda_module_get_api(m) == 3
    assert lib.panda_module_get_name(m) == path          # module.c:244-247: the text before the separator
    assert lib.panda_module_get_args(m) == b"25"
    assert lib.panda_module_get_description(m).startswith(b"Reject sequences")
    assert lib.panda_module_get_usage(m) == b"short_overlap:bases"
    assert lib.panda_module_get_version(m) == b"0.1"
    plug = ctypes.CDLL(str(plugins / "short_overlap.so"))
    assert plug.short_overlap_live_instances() == 1
    lib.panda_module_unref(m)
    assert plug.short_overlap_live_instances() == 0      # the destroy callback ran (module.c:347)


def test_load_plugin_by_name_and_directory(lib, plugins, monkeypatch):
    # lt_dlopenext() appends the extension; the plugin directory stands for PKGLIBDIR
    m = lib.panda_module_load(None, str(plugins / "short_overlap").encode() + b":3")
    assert m
    lib.panda_module_unref(m)
    monkeypatch.setenv("PANDASEQ_PLUGIN_DIR", str(plugins))
    m = lib.panda_module_load(None, b"short_overlap:3")
    assert m and lib.panda_module_get_name(m) == b"short_overlap"
    lib.panda_module_unref(m)


def test_load_refusals(lib, plugins, capfd):
    assert not lib.panda_module_load(None, str(plugins / "short_overlap.so").encode())           # the opener wants an argument
    assert not lib.panda_module_load(None, str(plugins / "short_overlap.so").encode() + b":x")
    assert not lib.panda_module_load(None, str(plugins / "wrong_api.so").encode() + b":1")
    assert "Invalid API" in capfd.readouterr().err
    assert not lib.panda_module_load(None, str(plugins / "no_such_plugin").encode())
    assert "Could not open module" in capfd.readouterr().err
    assert not lib.panda_module_load(None, None)


def test_fused_module_getters(lib):
    m = lib.panda_module_load(None, b"min_phred:10")
    assert m and lib.panda_module_get_name(m) == b"min_phred" and lib.panda_module_get_args(m) == b"10"
    assert lib.panda_module_get_api(m) == 3 and lib.panda_module_get_usage(m) == b"min_phred:value"
    assert lib.panda_module_get_version(m) == b"1.0"
    lib.panda_module_unref(m)
    m = lib.panda_module_load(None, b"min_overlapbits")
    assert m and lib.panda_module_get_args(m) is None and lib.panda_module_get_usage(m) == b"min_overlapbits:15"
    lib.panda_module_unref(m)
    assert not lib.panda_module_load(None, b"min_phred:ten")
    assert not lib.panda_module_load(None, b"min_phred")


def test_algorithm_registry(lib):
    class Class(ctypes.Structure):
        _fields_ = [("data_size", ctypes.c_size_t), ("name", ctypes.c_char_p), ("create", ctypes.c_void_p), ("data_destroy", ctypes.c_void_p),
                    ("overlap_probability", ctypes.c_void_p), ("match_probability", ctypes.c_void_p), ("prob_unpaired", ctypes.c_double)]

    def names():
        arr = ctypes.POINTER(ctypes.POINTER(Class)).in_dll(lib, "panda_algorithms")
        n = ctypes.c_size_t.in_dll(lib, "panda_algorithms_length").value
        return [arr[i].contents.name for i in range(n)]
    builtin = [b"ea_util", b"flash", b"pear", b"rdp_mle", b"simple_bayesian", b"stitch", b"uparse"]
    assert [x for x in names() if x in builtin] == builtin
    mine = Class(0, b"middle", None, None, None, None, 0.0)
    lib.panda_algorithm_register.argtypes = [ctypes.c_void_p]
    lib.panda_algorithm_register(ctypes.byref(mine))
    lib.panda_algorithm_register(ctypes.byref(mine))     # a second registration is a no-op (algo.c:107-111)
    got = names()
    assert got.count(b"middle") == 1 and got == sorted(got)


def test_quality_compare_of_the_builtin_classes(lib):
    """panda_algorithm_quality_compare (algo.c:27-32) for the seven built-in classes, whose merge runs on the device and whose
    class structs carry no host function here: every (match, qual a, qual b) in and around the clamp against the oracle's
    match_probability -- host tables only, no GPU."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as po

    class Qual(ctypes.Structure):
        _fields_ = [("nt", ctypes.c_ubyte), ("qual", ctypes.c_byte)]
    lib.panda_algorithm_quality_compare.restype = ctypes.c_double
    lib.panda_algorithm_quality_compare.argtypes = [ctypes.c_void_p, ctypes.POINTER(Qual), ctypes.POINTER(Qual)]
    lib.panda_algorithm_unref.argtypes = [ctypes.c_void_p]
    quals = list(range(-3, 50)) + [-128, 127]
    for algo, ctor in (("simple_bayesian", "simple_bayes"), ("pear", "pear"), ("rdp_mle", "rdp_mle"), ("uparse", "uparse"), ("flash", "flash"),
                       ("ea_util", "ea_util"), ("stitch", "stitch")):
        new = getattr(lib, "panda_algorithm_%s_new" % ctor)
        new.restype = ctypes.c_void_p
        a = new()
        cfg = po.make_config(algo)
        for match, nts in ((1, (1, 1)), (1, (3, 2)), (0, (1, 2)), (0, (4, 8))):
            for qa in quals:
                for qb in quals:
                    got = lib.panda_algorithm_quality_compare(a, ctypes.byref(Qual(nts[0], qa)), ctypes.byref(Qual(nts[1], qb)))
                    want = po.lib().po_match_probability(ctypes.byref(cfg), ctypes.byref(po.tables()), match, qa, qb)
                    assert got == want, (algo, match, qa, qb, got, want)
        lib.panda_algorithm_unref(a)


def test_exported_primer_scans(lib):
    """panda_compute_offset_qual / _result and panda_log1mexp (offset.c:35-38,103-133) as plugins call them on rows of their own:
    random reads with a primer planted, damaged or absent, both directions, with and without a penalty, against the oracle's
    scans -- host code only, no GPU."""
    import sys

    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as po

    class Result(ctypes.Structure):
        _fields_ = [("nt", ctypes.c_ubyte), ("p", ctypes.c_double)]
    lib.panda_compute_offset_qual.restype = ctypes.c_size_t
    lib.panda_compute_offset_qual.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_bool, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    lib.panda_compute_offset_result.restype = ctypes.c_size_t
    lib.panda_compute_offset_result.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_bool, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    lib.panda_log1mexp.restype = ctypes.c_double
    lib.panda_log1mexp.argtypes = [ctypes.c_double]
    L = po.lib()
    L.po_compute_offset_result.restype = ctypes.c_size_t
    L.po_compute_offset_result.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    assert ctypes.sizeof(Result) == 16
    rng = np.random.default_rng(11)
    score = po.tables_np()["score"]
    found = 0
    for case in range(300):
        n, plen = int(rng.integers(1, 120)), int(rng.integers(1, 25))
        primer = rng.choice(np.array([1, 2, 4, 8, 3, 12, 15], np.uint8), plen)
        seq = np.zeros((n, 2), np.uint8)
        seq[:, 0] = rng.choice(np.array([1, 2, 4, 8, 15], np.uint8), n, p=[.24, .24, .24, .24, .04])
        seq[:, 1] = rng.integers(0, 42, n).astype(np.uint8)
        if plen <= n and case % 4:
            at = int(rng.integers(0, n - plen + 1))
            seq[at:at + plen, 0] = np.where(rng.random(plen) < 0.08, seq[at:at + plen, 0], [int(rng.choice([b for b in (1, 2, 4, 8) if b & c])) for c in primer])
        thr, penalty = float(np.log(rng.choice([0.6, 0.9, 0.3]))), float(rng.choice([0.0, 0.0, 0.01]))
        for reverse in (False, True):
            view = np.ascontiguousarray(seq[::-1] if reverse else seq)
            want = L.po_compute_offset_qual(ctypes.byref(po.tables()), thr, penalty, view.ctypes.data, n, primer.ctypes.data, plen)
            got = lib.panda_compute_offset_qual(thr, penalty, reverse, seq.ctypes.data, n, primer.ctypes.data, plen)
            assert got == want, (case, reverse, n, plen, got, want)
            found += got != 0
            rows = (Result * n)()
            ps = score[np.minimum(seq[:, 1], 46)].copy()
            for i in range(n):
                rows[i].nt, rows[i].p = int(seq[i, 0]), float(ps[i])
            nts = np.ascontiguousarray(seq[:, 0])
            want = L.po_compute_offset_result(thr, penalty, int(reverse), nts.ctypes.data, ps.ctypes.data, n, primer.ctypes.data, plen)
            got = lib.panda_compute_offset_result(thr, penalty, reverse, ctypes.byref(rows), n, primer.ctypes.data, plen)
            assert got == want, (case, reverse, n, plen, got, want)
    assert found > 100
    import math
    for p in (1e-9, 0.1, math.log(2), 0.7, 3.0, 40.0):
        want = math.log1p(-math.exp(-p)) if p > math.log(2) else math.log(-math.expm1(-p))
        assert lib.panda_log1mexp(p) == want


def test_nt_letters(lib):
    """panda_nt_from_ascii / _from_ascii_complement / _complement / _to_ascii (nt.c:27-114,152-173): IUPAC sets by letter, either
    case, U as T, X as N, anything else empty; the complement of a SET as the reference's table has it (V and H both give V --
    sic, nt.c:35,39), while the complemented LETTER table is right for every letter."""
    for f in ("panda_nt_from_ascii", "panda_nt_from_ascii_complement"):
        getattr(lib, f).restype = ctypes.c_ubyte
        getattr(lib, f).argtypes = [ctypes.c_char]
    lib.panda_nt_complement.restype = ctypes.c_ubyte
    lib.panda_nt_complement.argtypes = [ctypes.c_ubyte]
    lib.panda_nt_to_ascii.restype = ctypes.c_char
    lib.panda_nt_to_ascii.argtypes = [ctypes.c_ubyte]
    sets = {"A": "A", "C": "C", "G": "G", "T": "T", "U": "T", "R": "AG", "Y": "CT", "S": "CG", "W": "AT", "K": "GT", "M": "AC",
            "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG", "N": "ACGT", "X": "ACGT"}
    bit = {"A": 1, "C": 2, "G": 4, "T": 8}
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for c in map(chr, range(64, 96)):
        want = sum(bit[b] for b in sets.get(c, ""))
        want_c = sum(bit[comp[b]] for b in sets.get(c, ""))
        for letter in (c, c.lower()):
            assert lib.panda_nt_from_ascii(letter.encode("latin1")) == want, letter
            assert lib.panda_nt_from_ascii_complement(letter.encode("latin1")) == want_c, letter
    table = [0, 8, 4, 12, 2, 10, 6, 7, 1, 9, 5, 7, 3, 11, 7, 15]
    assert [lib.panda_nt_complement(v) for v in range(16)] == table
    assert b"".join(lib.panda_nt_to_ascii(v) for v in range(16)) == b"NACMGRSVTWYHKDBN"
    assert lib.panda_nt_to_ascii(16) == b"N" and lib.panda_nt_to_ascii(255) == b"N"


def test_seqid_helpers(lib):
    """panda_seqid_clear / _copy / _equal / _compare / _str (seqid.c:46-127): what a client does with a result's identifier."""
    class SeqId(ctypes.Structure):
        _fields_ = [("instrument", ctypes.c_char * 100), ("run", ctypes.c_char * 100), ("flowcell", ctypes.c_char * 100), ("lane", ctypes.c_int),
                    ("tile", ctypes.c_int), ("x", ctypes.c_int), ("y", ctypes.c_int), ("tag", ctypes.c_char * 50)]
    lib.panda_seqid_str.restype = ctypes.c_char_p
    lib.panda_seqid_equal.restype = ctypes.c_bool
    a = SeqId(b"M01271", b"5", b"000000000-A1B2C", 1, 1101, 15589, 1332, b"ACGTAC")
    assert lib.panda_seqid_str(ctypes.byref(a)) == b"M01271:5:000000000-A1B2C:1:1101:15589:1332:ACGTAC"
    b = SeqId()
    lib.panda_seqid_copy(ctypes.byref(a), ctypes.byref(b))
    assert lib.panda_seqid_equal(ctypes.byref(a), ctypes.byref(b)) and lib.panda_seqid_compare(ctypes.byref(a), ctypes.byref(b)) == 0
    b.x = 15590
    assert not lib.panda_seqid_equal(ctypes.byref(a), ctypes.byref(b)) and lib.panda_seqid_compare(ctypes.byref(a), ctypes.byref(b)) < 0
    b.x, b.tag = a.x, b"ACGTAA"
    assert lib.panda_seqid_compare(ctypes.byref(a), ctypes.byref(b)) > 0
    b.instrument = b"M01272"
    assert lib.panda_seqid_compare(ctypes.byref(a), ctypes.byref(b)) < 0     # the instrument is compared first
    lib.panda_seqid_clear(ctypes.byref(b))
    assert lib.panda_seqid_str(ctypes.byref(b)) == b":::0:0:0:0:" and lib.panda_seqid_str(None) is None


def test_device_selection_is_per_thread(built):
    """Which GPU a new assembler gets: the calling thread's own choice, else PAMD_DEVICES round robin, else the process
    default -- decided without shared mutable state that two threads creating assemblers could race on (pool.c:140-160
    builds one assembler per thread).  No GPU needed: only the choice is observed."""
    import ctypes as C
    import os
    import threading

    from pandaseq_amd import _lib
    L = _lib.lib()
    L.pamd_compat_set_device(2)
    assert L.pamd_compat_device_for_new_assembler() == 2
    seen = {}

    def work(dev):
        L.pamd_compat_set_thread_device(dev)
        got = [L.pamd_compat_device_for_new_assembler() for _ in range(2000)]
        seen[dev] = set(got)

    ts = [threading.Thread(target=work, args=(d,)) for d in (3, 5, 6, 7)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert seen == {3: {3}, 5: {5}, 6: {6}, 7: {7}}
    assert L.pamd_compat_device_for_new_assembler() == 2   # the main thread never chose
    os.environ["PAMD_DEVICES"] = "4,1,6"
    try:
        got = [L.pamd_compat_device_for_new_assembler() for _ in range(6)]
        assert sorted(got[:3]) == [1, 4, 6] and got[:3] == got[3:]
        L.pamd_compat_set_thread_device(0)
        assert L.pamd_compat_device_for_new_assembler() == 0   # a thread's own choice wins
        L.pamd_compat_set_thread_device(-1)
    finally:
        del os.environ["PAMD_DEVICES"]
        L.pamd_compat_set_device(0)
    # without a GPU, concurrent constructors fail cleanly (NULL), they do not race
    if L.pamd_device_count() == 0:
        L.panda_assembler_new.restype = C.c_void_p
        L.panda_assembler_new.argtypes = [C.c_void_p] * 4
        out = []
        ts = [threading.Thread(target=lambda: out.append(L.panda_assembler_new(None, None, None, None))) for _ in range(8)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert out == [None] * 8


# ---- the data-format rows under the reference's names: seqid.c:127-285, writer.c, output.c:85-126 ----
def _lib_and_types():
    import ctypes as C
    from oracle import pyoracle as po
    from pandaseq_amd import _lib
    L = _lib.lib()
    L.panda_seqid_parse_fail.restype = C.c_int
    L.panda_seqid_parse_fail.argtypes = [C.POINTER(po.SeqId), C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_char_p)]
    L.panda_seqid_parse.restype = C.c_int
    L.panda_seqid_parse.argtypes = [C.POINTER(po.SeqId), C.c_char_p, C.c_int]
    return L


def _host_parse(L, header, policy):
    import ctypes as C
    from oracle import pyoracle as po
    raw = header if isinstance(header, bytes) else header.encode("latin1")
    buf = C.create_string_buffer(raw)
    sid, fmt = po.SeqId(), C.c_int(-1)
    end = C.c_char_p()
    d = L.panda_seqid_parse_fail(C.byref(sid), C.cast(buf, C.c_char_p), policy, C.byref(fmt), C.byref(end))
    end_off = C.cast(end, C.c_void_p).value - C.addressof(buf)
    return d, fmt.value, sid, end_off


def test_host_seqid_parse_on_the_reference_vectors(built):
    """panda_seqid_parse_fail as a host function against the six cases of the reference's own parser test (check_parser.c:29-36)."""
    import json
    from oracle import pyoracle as po
    L = _lib_and_types()
    with open(os.path.join(ROOT, "tests", "golden", "seqid_check_parser.json")) as fh:
        g = json.load(fh)
    assert len(g["cases"]) == 6
    for c in g["cases"]:
        d, fmt, sid, end = _host_parse(L, c["header"], po.TAG_OPTIONAL)
        assert d == c["dir"] and po.IDFMT[fmt] == c["format"] and list(sid.fields()) == c["id"], c["header"]
        assert end == len(c["header"]) or c["format"] != "CASAVA_1_7"      # (a header parsed to its end stops at its end)
        sid2 = po.SeqId()
        assert L.panda_seqid_parse(sid2, c["header"].encode(), po.TAG_OPTIONAL) == c["dir"] and list(sid2.fields()) == c["id"]


def test_host_seqid_parse_against_the_oracle(built):
    """... and against the oracle's restatement on headers of every format, truncated at every length, with odd characters, over-long
    fields and each tagging policy: the direction, the detected format, where parsing stopped, and -- where it parsed -- every field."""
    from oracle import pyoracle as po
    L = _lib_and_types()
    rng = np.random.default_rng(3)
    base = ["M01271:10:000000000-A3WGH:1:1101:18786:6175 1:N:0:1", "ILLUMINA-BE9C3F:29:FC:3:1:2462:1120 2:Y:18:GCTATA", "M00958:47:A3GH7:1:1101:15028:1512 3:N:0:",
            "1468:1:1:12675:1118#ATCACGA/1", "HWUSI-EAS100R:6:73:941:1973#0/2", "HWUSI:6:73:941:1973/1", "MISEQ03:18:000000000-A1REG:1:1101:14774:1712#GATAGTGCCAC/1",
            "SRR001666.1 071112_SLXA-EAS1_s_7:5:1:817:345 length=36", "ERR12.7 x/1", "SRR5", "SRRx.1 a", "ERR1.2", "", ":", "a:b:c:1:2:3:4 1:N:0:" + "ACGT" * 12 + "AC",
            "a:b:c:1:2:3:4 1:N:0:" + "ACGT" * 13, "x" * 99 + ":r:f:1:2:3:4 1:N:0:T", "i:" + "y" * 99 + ":f:1:2:3:4 1:N:0:T", "i:r:f:99999999999:2:3:4 1:N:0:T",
            "i:r:f:1:2:3:4 1:N:0", "i:r:f:1:2:3:4 1", "i:r:f:1:2:3:4 1:N:x:T", "i:1:2:3:4#" + "G" * 50 + "/1", "i:1:2:3:4#" + "G" * 51 + "/1", "i:1:2:3:4#/1", "i:1:2:3:4/"]
    cases = []
    for h in base:
        cases.append(h)
        for cut in range(0, len(h), 3):
            cases.append(h[:cut])
        for _ in range(6):
            b = bytearray(h.encode())
            if b:
                b[int(rng.integers(0, len(b)))] = int(rng.choice(list(b":/# .0a-")))
            cases.append(bytes(b).decode("latin1"))
    checked = parsed = 0
    for h in cases:
        if any(len(f) >= 100 for f in h.split(" ")[0].split("#")[0].split("/")[0].split(":")[:3]):
            # (fields of 100 characters or more: the reference writes past its 100-byte array; refused here, include/pandaseq_amd_compat.h)
            assert _host_parse(L, h, po.TAG_OPTIONAL)[0] == 0 or not h.startswith(("x" * 99, "i:" + "y" * 99))
            continue
        for policy in (po.TAG_PRESENT, po.TAG_ABSENT, po.TAG_OPTIONAL):
            d, fmt, sid, end = _host_parse(L, h, policy)
            od, ofmt, osid, oend = po.seqid_parse(h, policy)
            assert (d, fmt, end) == (od, ofmt, oend), (h, policy, (d, fmt, end), (od, ofmt, oend))
            if d != 0:
                assert sid.fields() == osid.fields(), h
                parsed += 1
            checked += 1
    assert checked > 1000 and parsed > 150


def test_writer_and_output_functions(built, tmp_path):
    """panda_output_fasta / panda_output_fastq through a PandaWriter (pandaseq.h:293-304, output.c:85-126, writer.c): the bytes of the
    oracle's formatter; a record reaches the write callback whole, at commit; an empty sequence writes nothing; slaves commit too."""
    import ctypes as C
    from oracle import pyoracle as po
    from pandaseq_amd import _lib, synth
    L = _lib.lib()

    class Result(C.Structure):
        _fields_ = [("nt", C.c_char), ("p", C.c_double)]

    class ResultSeq(C.Structure):
        _fields_ = [("quality", C.c_double), ("degenerates", C.c_size_t), ("name", po.SeqId), ("sequence", C.POINTER(Result)), ("sequence_length", C.c_size_t),
                    ("forward", C.c_void_p), ("forward_length", C.c_size_t), ("forward_offset", C.c_size_t), ("reverse", C.c_void_p), ("reverse_length", C.c_size_t),
                    ("reverse_offset", C.c_size_t), ("overlap_mismatches", C.c_size_t), ("overlaps_examined", C.c_size_t), ("overlap", C.c_size_t), ("estimated_overlap_probability", C.c_double)]
    assert C.sizeof(ResultSeq) == 480 and C.sizeof(po.SeqId) == 368
    WRITE = C.CFUNCTYPE(None, C.POINTER(C.c_char), C.c_size_t, C.c_void_p)
    chunks = []
    cb = WRITE(lambda buf, n, data: chunks.append(C.string_at(buf, n)))
    L.panda_writer_new.restype = C.c_void_p
    L.panda_writer_new.argtypes = [WRITE, C.c_void_p, C.c_void_p]
    L.panda_writer_new_null.restype = C.c_void_p
    L.panda_writer_open_file.restype = C.c_void_p
    L.panda_writer_open_file.argtypes = [C.c_char_p, C.c_bool]
    for f in ("panda_writer_unref", "panda_writer_commit", "panda_writer_flush"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = None
    L.panda_writer_set_slave.argtypes = [C.c_void_p, C.c_void_p]
    L.panda_writer_set_slave.restype = None
    L.panda_writer_get_slave.argtypes = [C.c_void_p]
    L.panda_writer_get_slave.restype = C.c_void_p
    for f in ("panda_output_fasta", "panda_output_fastq"):
        getattr(L, f).argtypes = [C.POINTER(ResultSeq), C.c_void_p]
        getattr(L, f).restype = C.c_bool
    w = L.panda_writer_new(cb, None, None)
    spec = synth.config_spec(1)
    fwd, rev, fl, rl = synth.generate_numpy(spec, 9, 60)
    res, nt, p, _ = po.assemble_batch(po.make_config("simple_bayesian"), fwd, rev, fl, rl)
    want_a, want_q, n_ok = [], [], 0
    for i in range(60):
        if res["status"][i] != 0:
            continue
        m = int(res["seq_len"][i])
        _, _, sid, _ = po.seqid_parse(f"M0{i}:7:FC:1:1101:{100 + i}:{7 * i} 1:N:0:ACGT")
        rows = (Result * m)()
        for j in range(m):
            rows[j].nt = bytes([int(nt[i, j])])
            rows[j].p = float(p[i, j])
        rs = ResultSeq(quality=float(res["quality"][i]), name=sid, sequence=rows, sequence_length=m)
        before = len(chunks)
        assert L.panda_output_fasta(C.byref(rs), w) and len(chunks) == before + 1       # one write per committed record
        assert chunks[-1] == po.output_fasta(sid, float(res["quality"][i]), nt[i, :m])
        assert L.panda_output_fastq(C.byref(rs), w)
        assert chunks[-1] == po.output_fastq(sid, float(res["quality"][i]), nt[i, :m], p[i, :m])
        n_ok += 1
    assert n_ok > 40
    empty = ResultSeq(quality=-0.1, sequence_length=0)
    before = len(chunks)
    assert L.panda_output_fasta(C.byref(empty), w) and L.panda_output_fastq(C.byref(empty), w) and len(chunks) == before
    # a file writer as slave of the callback writer: committing the master commits what was appended to the slave as well
    path = tmp_path / "slave.txt"
    fw = L.panda_writer_open_file(str(path).encode(), False)
    assert fw
    L.panda_writer_set_slave(w, fw)
    assert L.panda_writer_get_slave(w) == fw
    L.panda_writer_set_slave(fw, w)          # a cycle is refused (writer.c:288-306)
    assert L.panda_writer_get_slave(fw) is None
    L.panda_writer_append_c.argtypes = [C.c_void_p, C.c_char]
    L.panda_writer_append.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.panda_writer_append_c(fw, b"#")
    L.panda_writer_append(fw, b"%d\n", 42)
    L.panda_writer_commit(w)
    L.panda_writer_unref(fw)                 # (the master still holds it)
    L.panda_writer_unref(w)                  # closes the file through the slave's destroy
    assert path.read_bytes() == b"#42\n"
    # text appended and never committed is written when the last reference goes (writer.c:186-193), from whichever thread appended
    # it; and more writers than a thread's cache of buffers holds (64) lose nothing
    import threading
    late = []
    cb2 = WRITE(lambda buf, n, data: late.append(C.string_at(buf, n)))
    w2 = L.panda_writer_new(cb2, None, None)
    L.panda_writer_append(w2, b"main %d;", 1)
    t = threading.Thread(target=lambda: L.panda_writer_append(w2, b"other %d;", 2))
    t.start()
    t.join()
    many = [L.panda_writer_new(cb2, None, None) for _ in range(70)]
    for k, m in enumerate(many):
        L.panda_writer_append(m, b"w%d;", k)
    assert late == []
    L.panda_writer_unref(w2)
    assert sorted(late) == [b"main 1;", b"other 2;"]
    for m in many:
        L.panda_writer_unref(m)
    assert sorted(late[2:]) == sorted(b"w%d;" % k for k in range(70))


def test_bzip2_in_and_out(lib, tmp_path):
    """fileio.c:70-96 decides by an input's first two bytes -- "BZ": libbz2, anything else: zlib, which also reads plain text --
    and writer.c:147-151 writes `-j` output through libbz2.  The library binds libbz2.so.1 at run time (pamd_bz2.h): a bzip2
    writer's file inflates to what was committed, and panda_open_buffer() hands out the text of a bzip2, a gzip and a plain file
    alike (the callback's `false` after the last bytes of a bzip2 stream is the reference's, fileio.c:48-58)."""
    import bz2
    import gzip
    C = ctypes
    L = lib
    L.panda_writer_open_file.argtypes = [C.c_char_p, C.c_bool]
    L.panda_writer_open_file.restype = C.c_void_p
    L.panda_writer_append.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    for f in ("panda_writer_commit", "panda_writer_unref"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = None
    w = L.panda_writer_open_file(str(tmp_path / "out.bz2").encode(), True)
    assert w, "libbz2.so.1 is in this image"
    for i in range(5000):
        L.panda_writer_append(w, b">record %d\nACGTACGTTTGACA\n", i)
        L.panda_writer_commit(w)
    L.panda_writer_unref(w)
    want = b"".join(b">record %d\nACGTACGTTTGACA\n" % i for i in range(5000))
    assert bz2.decompress((tmp_path / "out.bz2").read_bytes()) == want

    READ = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p)
    DESTROY = C.CFUNCTYPE(None, C.c_void_p)
    L.panda_open_buffer.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.panda_open_buffer.restype = C.c_void_p
    text = b"".join(b"@r%d\nACGT\n+\nIIII\n" % i for i in range(200_000))
    for name, data in (("t.fastq.bz2", bz2.compress(text)), ("t.fastq.gz", gzip.compress(text)), ("t.fastq", text), ("noext", bz2.compress(text))):
        (tmp_path / name).write_bytes(data)
        ud, destroy = C.c_void_p(), C.c_void_p()
        fn = L.panda_open_buffer(str(tmp_path / name).encode(), None, C.byref(ud), C.byref(destroy))
        assert fn and destroy.value, name
        read = READ(fn)
        buf = C.create_string_buffer(1 << 20)
        got, n = bytearray(), C.c_size_t(0)
        while True:
            more = read(C.cast(buf, C.c_void_p), len(buf), C.byref(n), ud)
            got += buf.raw[:n.value]
            if not more or n.value == 0:
                break
        DESTROY(destroy.value)(ud)
        assert bytes(got) == text, name
    ud, destroy = C.c_void_p(), C.c_void_p()
    assert not L.panda_open_buffer(str(tmp_path / "absent").encode(), None, C.byref(ud), C.byref(destroy))
