"""panda_module_load() on plugin files and the algorithm registry: host logic of the compat layer, no GPU.
Reference behaviour: module.c:227-298 (loading), :360-391 (getters), algo.c:96-133 (registry)."""
import ctypes
import os
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
