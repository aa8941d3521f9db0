"""The reference's own per-assembler API (panda_assembler_next / _assemble), re-linked against
libpandaseq_amd.so by a plain-C client, compared with the oracle."""
import json
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_client(tmp):
    exe = os.path.join(tmp, "test_compat")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "test_compat.c"), "-o", exe,
                           "-rdynamic", "-L", os.path.join(ROOT, "pandaseq_amd"), "-lpandaseq_amd", "-ldl",
                           "-Wl,-rpath," + os.path.join(ROOT, "pandaseq_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _read_record(buf, o):
    h = struct.unpack_from("<8I", buf, o)
    o += 32
    q, e = struct.unpack_from("<2d", buf, o)
    o += 16
    m = h[2]
    nt = np.frombuffer(buf, np.uint8, m, o)
    o += m
    p = np.frombuffer(buf, np.float64, m, o)
    o += 8 * m
    return dict(index=h[0], overlap=h[1], seq_len=m, mism=h[3], degen=h[4], fo=h[5], ro=h[6], examined=h[7], quality=q, est=e, nt=nt, p=p), o


def _write_pairs(path, fwd, rev, flen, rlen):
    n, stride = fwd.shape[0], fwd.shape[1]
    blob = bytearray(struct.pack("<2I", n, stride))
    for i in range(n):
        blob += struct.pack("<2H", int(flen[i]), int(rlen[i])) + fwd[i].tobytes() + rev[i].tobytes()
    path.write_bytes(bytes(blob))


def _parse_output(buf):
    o = 0
    got = []
    while struct.unpack_from("<I", buf, o)[0] != 0xFFFFFFFF:
        rec, o = _read_record(buf, o)
        got.append(rec)
    o += 4
    counters = struct.unpack_from("<10q", buf, o)
    return got, counters, o + 80


REF_DIR = os.path.join(ROOT, "oracle", "_ref")


@pytest.mark.parametrize("algo,plugin,args", [("simple_bayesian", "min_phred", "4"), ("rdp_mle", "min_phred", "10"),
                                              ("rdp_mle", "min_overlapbits", "15"), ("pear", "min_overlapbits", "2.5"),
                                              ("simple_bayesian", "min_overlapbits", None), ("rdp_mle", "min_overlapbits", "")])
def test_reference_plugin_binaries(built, tmp_path, algo, plugin, args):
    """The reference's OWN check plugins, compiled from its plugin_*.c (oracle/Makefile `ref`), run as host
    callbacks over the results the GPU path returns -- in a client compiled against the reference's own
    pandaseq.h -- and must keep/reject exactly the pairs the fused device filter keeps/rejects (and the
    oracle says).  This pins rows a15/a16 and the panda_result_seq / panda_result layout (a17) on real
    reference code."""
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    from util import adversarial_pairs
    client = os.path.join(REF_DIR, "refclient")
    so = os.path.join(REF_DIR, plugin + ".so")
    if not (os.path.exists(client) and os.path.exists(so)):
        pytest.skip("oracle/_ref was not built (needs /root/reference at build time: make -C oracle ref)")
    spec = synth.config_spec(1)
    n = 2500
    fwd, rev, flen, rlen = adversarial_pairs(spec, n, seed=33)
    if algo == "pear":
        rlen = np.minimum(rlen, flen)
    pin = tmp_path / "pairs.bin"
    _write_pairs(pin, fwd, rev, flen, rlen)
    out_host, out_fused, out_ours = tmp_path / "host.bin", tmp_path / "fused.bin", tmp_path / "ours.bin"
    subprocess.check_call([client, str(pin), str(out_host), algo, "2", so, plugin, "(null)" if args is None else args], timeout=300)
    spec_str = plugin if args is None else plugin + ":" + args    # module.c:280: no colon -> NULL args
    subprocess.check_call([client, str(pin), str(out_fused), algo, "3", spec_str], timeout=300)
    # the same client compiled against OUR compat header must see the same bytes (header layouts agree)
    subprocess.check_call([_build_client(str(tmp_path)), str(pin), str(out_ours), algo, "3", spec_str], timeout=300)
    host, fused = out_host.read_bytes(), out_fused.read_bytes()
    assert host == fused, "reference plugin on the host and the fused device filter disagree"
    assert out_ours.read_bytes() == fused
    got, counters, _ = _parse_output(fused)
    val = 15.0 if args is None else (float(args) if args else 0.0)    # plugin_min_overlapbits.c:26-33
    o_res, o_nt, o_p, o_ctr = po.assemble_batch(po.make_config(algo, modules=[(plugin, int(val) if plugin == "min_phred" else val)]), fwd, rev, flen, rlen)
    expected = [i for i in range(n) if flen[i] >= 2 and rlen[i] >= 2 and o_res["status"][i] == 0]
    assert [g["index"] for g in got] == expected
    assert counters[8] == o_ctr["rejected"][0] and counters[8] > 0, "the filter must reject something for the test to mean anything"
    for g in got[:200]:
        i = g["index"]
        assert g["quality"] == o_res["quality"][i]
        if algo in ("ea_util", "stitch"):  # a device log() of a ratio of small integers (include/pandaseq_amd.h)
            assert abs(g["est"] - o_res["est_prob"][i]) <= 1e-14 * max(1.0, abs(o_res["est_prob"][i]))
        else:
            assert g["est"] == o_res["est_prob"][i]
        assert np.array_equal(g["nt"], o_nt[i, :g["seq_len"]]) and np.array_equal(g["p"], o_p[i, :g["seq_len"]])


@pytest.mark.parametrize("plugin,args", [("completely_miss_the_point", "0"), ("completely_miss_the_point", "2"), ("empty", None)])
def test_reference_host_plugins(built, tmp_path, plugin, args):
    """Two more of the reference's own plugin binaries (plugin_completely_miss_the_point.c, plugin_empty.c), which
    have no fused form: loaded the way module.c:227-298 loads them and run as host callbacks over the results of
    the device path, in the client compiled against the reference's own pandaseq.h.  They read
    overlap_mismatches and sequence_length through the reference's struct layout."""
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    from util import adversarial_pairs
    client = os.path.join(REF_DIR, "refclient")
    so = os.path.join(REF_DIR, plugin + ".so")
    if not (os.path.exists(client) and os.path.exists(so)):
        pytest.skip("oracle/_ref was not built (needs /root/reference at build time: make -C oracle ref)")
    spec = synth.config_spec(1)
    n = 2500
    fwd, rev, flen, rlen = adversarial_pairs(spec, n, seed=35)
    pin, out_host = tmp_path / "pairs.bin", tmp_path / "host.bin"
    _write_pairs(pin, fwd, rev, flen, rlen)
    subprocess.check_call([client, str(pin), str(out_host), "simple_bayesian", "2", so, plugin, "(null)" if args is None else args], timeout=300)
    # and the same plugin file through panda_module_load("<file>:<args>"), the call main.c makes for -C
    out_load = tmp_path / "load.bin"
    subprocess.check_call([client, str(pin), str(out_load), "simple_bayesian", "3", so if args is None else so + ":" + args], timeout=300)
    assert out_load.read_bytes() == out_host.read_bytes()
    got, counters, _ = _parse_output(out_host.read_bytes())
    o_res, o_nt, o_p, o_ctr = po.assemble_batch(po.make_config("simple_bayesian"), fwd, rev, flen, rlen)
    ok = [i for i in range(n) if flen[i] >= 2 and rlen[i] >= 2 and o_res["status"][i] == 0]
    if plugin == "completely_miss_the_point":
        expected = [i for i in ok if o_res["mismatches"][i] <= int(args)]
    else:
        expected = [i for i in ok if o_res["seq_len"][i] > 0]
    assert [g["index"] for g in got] == expected
    assert counters[8] == len(ok) - len(expected)
    if plugin == "completely_miss_the_point":
        assert counters[8] > 0, "the filter must reject something for the test to mean anything"


def test_reference_plugin_that_calls_back_into_the_library(built, tmp_path):
    """plugin_other_primer.c, the reference's own binary: its opener parses the primer with panda_nt_from_ascii and its precheck
    runs panda_compute_offset_qual(log 0.9, 0.01, ...) on every forward read -- both now functions of this library (nt.c:152-155,
    offset.c:103-112) -- and a pair in whose forward read the primer is found never reaches the device."""
    import math

    from oracle import pyoracle as po
    from pandaseq_amd import synth
    client = os.path.join(REF_DIR, "refclient")
    so = os.path.join(REF_DIR, "other_primer.so")
    if not (os.path.exists(client) and os.path.exists(so)):
        pytest.skip("oracle/_ref was not built (needs /root/reference at build time: make -C oracle ref)")
    spec = synth.config_spec(2)     # reads that begin with the forward primer, a tenth of them after 1-3 stray bases
    n = 2000
    fwd, rev, flen, rlen = synth.generate_host(spec, 4000, n)
    rng = np.random.default_rng(5)
    wipe = rng.random(n) < 0.4      # the primer destroyed in four reads of ten: those pass the filter
    fwd[wipe, :24, 0] = rng.choice(np.array([1, 2, 4, 8], np.uint8), (int(wipe.sum()), 24))
    rlen = np.minimum(rlen, flen)
    pin, out = tmp_path / "pairs.bin", tmp_path / "out.bin"
    _write_pairs(pin, fwd, rev, flen, rlen)
    subprocess.check_call([client, str(pin), str(out), "pear", "3", so + ":f:" + spec.fprimer], timeout=300)
    got, counters, _ = _parse_output(out.read_bytes())
    primer = synth.nt_codes(spec.fprimer)
    L = po.lib()
    found = np.array([L.po_compute_offset_qual(ctypes.byref(po.tables()), math.log(0.9), 0.01, np.ascontiguousarray(fwd[i]).ctypes.data, int(flen[i]),
                                                primer.ctypes.data, len(primer)) for i in range(n)])
    o_res = po.assemble_batch(po.make_config("pear"), fwd, rev, flen, rlen)[0]
    passed = [i for i in range(n) if found[i] == 0 and flen[i] >= 2 and rlen[i] >= 2]
    expected = [i for i in passed if o_res["status"][i] == 0]
    assert [g["index"] for g in got] == expected
    assert 0 < len(expected) < n and (found != 0).sum() > n // 3
    assert counters[8] == int(((found != 0) & (flen >= 2) & (rlen >= 2)).sum())     # module.c:138-154 counts a precheck's rejections


def test_loaded_plugin_file(built, tmp_path):
    """A plugin file written for this test (tests/c/plugin_short_overlap.c: a precheck and a check), found by
    panda_module_load() the way module.c:227-298 finds plugins, run on the host around the device batch."""
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    from util import adversarial_pairs
    so = str(tmp_path / "short_overlap.so")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "plugin_short_overlap.c"), "-o", so])
    spec = synth.config_spec(1)
    n = 2500
    fwd, rev, flen, rlen = adversarial_pairs(spec, n, seed=37)
    flen[5:40:7] = 2     # reach the precheck (forward_length >= 3) with reads the assembler itself accepts
    pin, out = tmp_path / "pairs.bin", tmp_path / "out.bin"
    _write_pairs(pin, fwd, rev, flen, rlen)
    limit = 60
    subprocess.check_call([_build_client(str(tmp_path)), str(pin), str(out), "simple_bayesian", "3", "%s:%d" % (so, limit)], timeout=300)
    got, counters, _ = _parse_output(out.read_bytes())
    keep = (flen >= 3) & (rlen >= 2)
    o_res, o_nt, o_p, o_ctr = po.assemble_batch(po.make_config("simple_bayesian"), fwd, rev, np.where(keep, flen, 0), rlen)
    ok = [i for i in range(n) if keep[i] and o_res["status"][i] == 0]
    expected = [i for i in ok if o_res["overlap"][i] >= limit]
    assert [g["index"] for g in got] == expected
    pre_rejected = int(((flen == 2) & (rlen >= 2)).sum())
    assert pre_rejected > 0 and len(ok) > len(expected) > 0
    assert counters[8] == pre_rejected + len(ok) - len(expected)    # module.c:138-154 and :120-136 count into the same slot
    for g in got[:100]:
        i = g["index"]
        assert g["quality"] == o_res["quality"][i] and np.array_equal(g["nt"], o_nt[i, :g["seq_len"]])


@pytest.mark.parametrize("sums", ["exact", "fast", "default"])
@pytest.mark.parametrize("algo,mode", [("simple_bayesian", 0), ("pear", 0), ("rdp_mle", 1), ("uparse", 0), ("flash", 0), ("ea_util", 0), ("stitch", 0)])
def test_reference_api_client(built, tmp_path, algo, mode, sums):
    """The summation modes of the reference API (PAMD_COMPAT_SUMS): in the reference's order every scalar is bit-identical;
    in any order (the shipped default) `quality` and the pear / rdp_mle score are within 1e-9 and everything else is
    bit-identical.  (mode 1 has a caller's `check` behind the fused module: such an assembler sums in the reference's order by
    itself -- "default": the variable unset -- unless any-order sums were asked for in so many words, PAMD_COMPAT_SUMS=fast;
    pamd_compat.cpp push_config.)"""
    if sums == "default" and mode != 1:
        pytest.skip("the default of an assembler without a host check is the fast mode, covered by `fast`")
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    from util import adversarial_pairs
    spec = synth.config_spec(1)
    n = 2500
    fwd, rev, flen, rlen = adversarial_pairs(spec, n, seed=21)
    if algo == "pear":
        rlen = np.minimum(rlen, flen)
    pin = tmp_path / "pairs.bin"
    pout = tmp_path / "out.bin"
    _write_pairs(pin, fwd, rev, flen, rlen)
    exe = _build_client(str(tmp_path))
    env = dict(os.environ, PAMD_COMPAT_SUMS=sums)
    if sums == "default":
        del env["PAMD_COMPAT_SUMS"]
    subprocess.check_call([exe, str(pin), str(pout), algo, str(mode)], timeout=300, env=env)
    buf = pout.read_bytes()
    exact = sums == "exact" or (mode == 1 and sums == "default")

    mods = [("min_overlapbits", 15)] if mode == 1 else []
    o_res, o_nt, o_p, o_ctr = po.assemble_batch(po.make_config(algo, modules=mods), fwd, rev, flen, rlen)
    # expected sequence of returned results, replaying the host-side modules of the client
    expected = []
    rej_host = rej_pre = 0
    for i in range(n):
        if flen[i] < 2 or rlen[i] < 2:
            continue
        if mode == 1 and i % 11 == 0:
            rej_pre += 1
            continue
        if o_res["status"][i] != 0:
            continue
        if mode == 1 and o_res["mismatches"][i] > 3:
            rej_host += 1
            continue
        expected.append(i)
    o = 0
    got = []
    while struct.unpack_from("<I", buf, o)[0] != 0xFFFFFFFF:
        rec, o = _read_record(buf, o)
        got.append(rec)
    o += 4
    counters = struct.unpack_from("<10q", buf, o)
    o += 80
    assert [g["index"] for g in got] == expected, "results must come back in source order"
    for g in got:
        i = g["index"]
        assert g["overlap"] == o_res["overlap"][i] and g["seq_len"] == o_res["seq_len"][i]
        assert g["mism"] == o_res["mismatches"][i] and g["degen"] == o_res["degenerates"][i] and g["examined"] == o_res["examined"][i]
        if exact:
            assert g["quality"] == o_res["quality"][i]
        else:
            assert abs(g["quality"] - o_res["quality"][i]) <= 1e-9
        if algo in ("ea_util", "stitch"):  # a device log() of a ratio of small integers (include/pandaseq_amd.h)
            assert abs(g["est"] - o_res["est_prob"][i]) <= 1e-14 * max(1.0, abs(o_res["est_prob"][i]))
        elif exact or algo not in ("pear", "rdp_mle"):
            assert g["est"] == o_res["est_prob"][i]
        else:
            assert abs(g["est"] - o_res["est_prob"][i]) <= 1e-9 * max(1.0, abs(o_res["est_prob"][i]))
        assert np.array_equal(g["nt"], o_nt[i, :g["seq_len"]]) and np.array_equal(g["p"], o_p[i, :g["seq_len"]])
    count, ok, noalgn, lowq, badr, slow, longest, noalgn_cb = counters[:8]
    assert count == n and ok == len(expected) and noalgn_cb == noalgn
    if mode == 0:
        assert (noalgn, lowq, badr, slow) == (o_ctr["noalgn"], o_ctr["lowq"], o_ctr["badr"], o_ctr["slow"])
        assert longest == o_ctr["longest_overlap"]
    # single-pair entry point
    m = struct.unpack_from("<I", buf, o)[0]
    o += 4
    for i in range(m):
        flag = struct.unpack_from("<I", buf, o)[0]
        o += 4
        should = i in expected
        assert bool(flag) == should, i
        if flag:
            rec, o = _read_record(buf, o)
            assert rec["overlap"] == o_res["overlap"][i] and rec["quality"] == o_res["quality"][i]
            assert np.array_equal(rec["nt"], o_nt[i, :rec["seq_len"]]) and np.array_equal(rec["p"], o_p[i, :rec["seq_len"]])


def test_one_assembler_per_thread(built, tmp_path):
    """pool.c:140-160's shape: several threads, each with its own assembler and its own slice of the pairs, all on one GPU
    at once; their summed counters equal one assembler's over everything."""
    from pandaseq_amd import synth
    from oracle import pyoracle as po
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, 20000)
    pin = tmp_path / "pairs.bin"
    _write_pairs(pin, fwd, rev, flen, rlen)
    exe = str(tmp_path / "test_threads")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "test_threads.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "pandaseq_amd"), "-lpandaseq_amd", "-Wl,-rpath," + os.path.join(ROOT, "pandaseq_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    o_ctr = po.assemble_batch(po.make_config("simple_bayesian"), fwd, rev, flen, rlen, want_seq=False)[3]
    want = [int(o_ctr[k]) for k in ("count", "ok", "noalgn", "lowq", "badr", "slow")]
    for threads in (1, 2, 5):
        out = subprocess.check_output([exe, str(pin), str(threads)], timeout=300)
        assert [int(x) for x in out.split()] == want, (threads, out)
    # PAMD_DEVICES: the assemblers of a process are dealt over the listed GPUs in turn (here the one GPU twice: the code path
    # of a multi-GPU embedder).  The reference never sums its workers' STAT counters (pool.c:86-104 prints each thread's block);
    # the client sums the getters of its assemblers, and that sum is one assembler's over all the pairs.
    out = subprocess.check_output([exe, str(pin), "2"], timeout=300, env=dict(os.environ, PAMD_DEVICES="0,0"))
    assert [int(x) for x in out.split()] == want, out


def test_fused_modules_after_a_host_module(built, tmp_path):
    """min_phred / min_overlapbits AFTER a host-callback module: they run on the host then (the kernel may only take a
    leading run of them), with the same results and the same module blamed."""
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    n = 3000
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, n)
    pin, pout = tmp_path / "pairs.bin", tmp_path / "out.bin"
    _write_pairs(pin, fwd, rev, flen, rlen)
    subprocess.check_call([_build_client(str(tmp_path)), str(pin), str(pout), "rdp_mle", "4"], timeout=300)
    got, counters, _ = _parse_output(pout.read_bytes())
    plain = po.assemble_batch(po.make_config("rdp_mle"), fwd, rev, flen, rlen)[0]
    mods = po.assemble_batch(po.make_config("rdp_mle", modules=[("min_phred", 4), ("min_overlapbits", 15)]), fwd, rev, flen, rlen)[0]
    expected = [i for i in range(n) if plain["status"][i] == 0 and plain["mismatches"][i] <= 3 and mods["status"][i] == 0]
    assert [g["index"] for g in got] == expected
    host_rejected = int(((plain["status"] == 0) & (plain["mismatches"] > 3)).sum())
    assert counters[8] == host_rejected and 0 < len(expected) < int((plain["status"] == 0).sum()) - host_rejected


def _build_c(tmp, name):
    exe = os.path.join(tmp, name)
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", name + ".c"), "-o", exe, "-lm",
                           "-rdynamic", "-L", os.path.join(ROOT, "pandaseq_amd"), "-lpandaseq_amd", "-ldl",
                           "-Wl,-rpath," + os.path.join(ROOT, "pandaseq_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


@pytest.mark.parametrize("primers", [False, True])
def test_algorithm_class_registered_by_the_caller(built, tmp_path, primers):
    """algo.c:106-120, assembler.c:54-57,125,209: a class the caller registers is scored through its own two functions, on
    the host path of the library.  The class here is a twin of simple_bayesian written from the published formulas, so every
    record, merged base, log-p and counter must equal what the kernels produce for the library's own simple_bayesian; a
    class with neither functions nor a device form is refused by the setter (the assembler keeps its algorithm)."""
    from pandaseq_amd import synth
    from util import adversarial_pairs
    spec = synth.config_spec(2 if primers else 1)
    if primers:
        fwd, rev, flen, rlen = synth.generate_host(spec, 0, 600)
    else:
        fwd, rev, flen, rlen = adversarial_pairs(spec, 600, seed=31)
    pairs = tmp_path / "pairs.bin"
    _write_pairs(pairs, fwd, rev, flen, rlen)
    exe = _build_c(str(tmp_path), "test_custom_algo")
    out = subprocess.run([exe, str(pairs)] + (["primers"] if primers else []), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("same 600 pairs"), out.stdout


def test_reference_api_default_sums(built, tmp_path):
    """Without PAMD_COMPAT_SUMS / pamd_compat_set_exact_sums the reference API sums in any order: every decision, merged
    base and per-base log-p as the oracle's, `quality` within 1e-9 (north_star: 1e-6)."""
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = synth.generate_host(spec, 500, 700)
    pin, pout = tmp_path / "pairs.bin", tmp_path / "out.bin"
    _write_pairs(pin, fwd, rev, flen, rlen)
    env = dict(os.environ)
    env.pop("PAMD_COMPAT_SUMS", None)
    subprocess.check_call([_build_client(str(tmp_path)), str(pin), str(pout), "simple_bayesian", "0"], timeout=300, env=env)
    got, counters, _ = _parse_output(pout.read_bytes())
    o_res, o_nt, o_p, o_ctr = po.assemble_batch(po.make_config("simple_bayesian"), fwd, rev, flen, rlen)
    assert len(got) == int(o_ctr["ok"]) > 600
    worst = 0.0
    for g in got:
        i = g["index"]
        assert o_res["status"][i] == 0 and g["overlap"] == o_res["overlap"][i] and g["seq_len"] == o_res["seq_len"][i]
        assert np.array_equal(g["nt"], o_nt[i, :g["seq_len"]]) and np.array_equal(g["p"], o_p[i, :g["seq_len"]])
        worst = max(worst, abs(g["quality"] - o_res["quality"][i]))
    assert worst <= 1e-9


def test_rows_built_on_the_callers_thread(built, tmp_path):
    """PAMD_COMPAT_ROWS=lazy (pamd_compat_set_lazy_rows): panda_assembler_next() builds each record's panda_result[] on the
    caller's thread instead of ahead of it -- the same records, bit for bit."""
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = synth.generate_host(spec, 900, 1500)
    pin = tmp_path / "pairs.bin"
    _write_pairs(pin, fwd, rev, flen, rlen)
    exe = _build_client(str(tmp_path))
    outs = []
    for rows in ("eager", "lazy"):
        pout = tmp_path / f"out_{rows}.bin"
        subprocess.check_call([exe, str(pin), str(pout), "rdp_mle", "0"], timeout=300, env=dict(os.environ, PAMD_COMPAT_ROWS=rows))
        outs.append(pout.read_bytes())
    assert len(outs[0]) > 100_000 and outs[0] == outs[1]


def test_one_assembler_per_thread_consumers(built, tmp_path):
    """pool.c:140-167's shape: several threads, an assembler each on the same GPU, every thread reading every base
    (tests/c/bench_pool.c).  Each thread's share of the pairs is assembled exactly once: the OK counts add up to what the
    oracle says for the shares."""
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    spec = synth.config_spec(1)
    distinct, total, threads = 3000, 50_000, 4
    fwd, rev, flen, rlen = synth.generate_host(spec, 0, distinct)
    pin = tmp_path / "pairs.bin"
    _write_pairs(pin, fwd, rev, flen, rlen)
    exe = _build_c(str(tmp_path), "bench_pool")
    out = subprocess.check_output([exe, str(pin), str(total), "sb", "1", str(threads)], timeout=600).decode()
    d = json.loads(out.strip().splitlines()[-1])
    ok = po.assemble_batch(po.make_config("simple_bayesian"), fwd, rev, flen, rlen, want_seq=False)[0]["status"] == 0
    expect = 0
    for k in range(threads):
        share = total * (k + 1) // threads - total * k // threads      # the first `share` pairs of the cycled source
        expect += (share // distinct) * int(ok.sum()) + int(ok[: share % distinct].sum())
    assert d["threads"] == threads and d["pairs"] == total and d["ok"] == expect == d["returned"]
