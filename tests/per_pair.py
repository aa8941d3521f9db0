"""Helpers of the per-pair entry point's tests (tests/c/test_per_pair.c): build the C client, write its input, read its
output, compare with the oracle."""
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALGOS = ["simple_bayesian", "pear", "rdp_mle", "uparse", "flash", "ea_util", "stitch"]


def build_client(tmp):
    exe = os.path.join(str(tmp), "test_per_pair")
    if not os.path.exists(exe):
        subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "c", "test_per_pair.c"), "-o", exe,
                               "-L", os.path.join(ROOT, "pandaseq_amd"), "-lpandaseq_amd", "-ldl",
                               "-Wl,-rpath," + os.path.join(ROOT, "pandaseq_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def write_pairs(path, fwd, rev, flen, rlen):
    n, stride = fwd.shape[0], fwd.shape[1]
    rec = np.zeros((n, 4 + 4 * stride), dtype=np.uint8)
    rec[:, 0:2] = np.ascontiguousarray(flen, dtype="<u2").view(np.uint8).reshape(n, 2)
    rec[:, 2:4] = np.ascontiguousarray(rlen, dtype="<u2").view(np.uint8).reshape(n, 2)
    rec[:, 4:4 + 2 * stride] = fwd.reshape(n, 2 * stride)
    rec[:, 4 + 2 * stride:] = rev.reshape(n, 2 * stride)
    with open(path, "wb") as f:
        f.write(struct.pack("<2I", n, stride))
        f.write(rec.tobytes())


def primer_arg(nt):
    return "".join("%x" % int(v) for v in nt)


def run_client(exe, pairs, out, algo, path, options=(), env=None):
    subprocess.check_call([exe, str(pairs), str(out), algo, path] + [str(o) for o in options], timeout=900, env=env)
    buf = open(out, "rb").read()
    o = 0
    records = []
    index = 0
    while o < len(buf) - 80:
        flag = struct.unpack_from("<I", buf, o)[0]
        o += 4
        if flag:
            h = struct.unpack_from("<8I", buf, o)
            o += 32
            q, e = struct.unpack_from("<2d", buf, o)
            o += 16
            m = h[2]
            nt = np.frombuffer(buf, np.uint8, m, o)
            o += m
            p = np.frombuffer(buf, np.float64, m, o)
            o += 8 * m
            assert h[0] == index
            records.append(dict(index=h[0], overlap=h[1], seq_len=m, mism=h[3], degen=h[4], fo=h[5], ro=h[6], examined=h[7], quality=q, est=e, nt=nt, p=p))
        else:
            records.append(None)
        index += 1
    counters = dict(zip(("count", "ok", "noalgn", "lowq", "badr", "slow", "nofp", "norp", "longest_overlap", "rejected0"), struct.unpack_from("<10q", buf, o)))
    return records, counters


def same_bits(a, b):
    return np.float64(a).tobytes() == np.float64(b).tobytes() or (np.isnan(a) and np.isnan(b))


def compare_with_oracle(records, counters, o_res, o_nt, o_p, o_ctr, modules=False, exact=True):
    """Every pair the oracle accepts comes back with its record; nothing else does."""
    n = len(o_res)
    assert len(records) == n
    for i, g in enumerate(records):
        ok = o_res["status"][i] == 0
        assert (g is not None) == bool(ok), (i, int(o_res["status"][i]))
        if g is None:
            continue
        m = int(o_res["seq_len"][i])
        assert g["seq_len"] == m and g["overlap"] == o_res["overlap"][i] and g["mism"] == o_res["mismatches"][i], ("scalars", i)
        assert g["degen"] == o_res["degenerates"][i] and g["examined"] == o_res["examined"][i], ("scalars", i)
        assert g["fo"] == o_res["fwd_offset"][i] and g["ro"] == o_res["rev_offset"][i], ("offsets", i, g["fo"], g["ro"], o_res["fwd_offset"][i], o_res["rev_offset"][i])
        assert np.array_equal(g["nt"], o_nt[i, :m]), ("nt", i)
        assert g["p"].tobytes() == np.ascontiguousarray(o_p[i, :m]).tobytes(), ("p", i)
        if exact:
            assert same_bits(g["quality"], o_res["quality"][i]), ("quality", i, g["quality"], o_res["quality"][i])
            assert same_bits(g["est"], o_res["est_prob"][i]), ("est", i, g["est"], o_res["est_prob"][i])
        else:
            assert abs(g["quality"] - o_res["quality"][i]) <= 1e-9, ("quality", i)
    for k in ("count", "ok", "noalgn", "lowq", "badr", "slow", "nofp", "norp", "longest_overlap"):
        assert counters[k] == int(o_ctr[k]), (k, counters[k], int(o_ctr[k]))
    if modules:
        assert counters["rejected0"] == int(o_ctr["rejected"][0]), "the first module's rejections"
