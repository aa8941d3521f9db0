#!/usr/bin/env python3
"""End-to-end rate of the drop-in API: tests/c/bench_next.c (the reference's own panda_assembler_next loop, plain C)
over an in-memory source of synthetic pairs.  Builds the client, writes the pairs, runs it, prints its JSON line.

    python tools/bench_next.py [--config 1] [--pairs 4000000] [--sums exact|fast] [--batch 65536]
"""
import argparse
import json
import os
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(tmp):
    exe = os.path.join(tmp, "bench_next")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "bench_next.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "pandaseq_amd"), "-lpandaseq_amd", "-Wl,-rpath," + os.path.join(ROOT, "pandaseq_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def write_pairs(path, fwd, rev, flen, rlen):
    import numpy as np
    n, stride = fwd.shape[0], fwd.shape[1]
    rec = np.zeros((n, 4 + stride * 4), dtype=np.uint8)
    rec[:, 0:2] = np.ascontiguousarray(flen, dtype="<u2").view(np.uint8).reshape(n, 2)
    rec[:, 2:4] = np.ascontiguousarray(rlen, dtype="<u2").view(np.uint8).reshape(n, 2)
    rec[:, 4:4 + stride * 2] = fwd.reshape(n, stride * 2)
    rec[:, 4 + stride * 2:] = rev.reshape(n, stride * 2)
    with open(path, "wb") as fh:
        fh.write(struct.pack("<2I", n, stride))
        fh.write(rec.tobytes())


def synth_pairs(spec, n):
    """n pairs of the synthetic stream as numpy rows: from the device generator when there is a GPU (the host twin takes
    200 us per pair), else the host twin (tested equal)."""
    from pandaseq_amd import synth
    import torch
    if torch.cuda.is_available():
        f, r, fl, rl = synth.generate_device(spec, 0, n)
        torch.cuda.synchronize()
        return f.cpu().numpy(), r.cpu().numpy(), fl.cpu().numpy().astype("<u2"), rl.cpu().numpy().astype("<u2")
    return synth.generate_numpy(spec, 0, n)


def run(config=1, pairs=4_000_000, sums="exact", batch=65536, distinct=8192, algo=None, consume=2, api=0, rows="eager", chunk=65536):
    from pandaseq_amd import synth
    import bench
    spec = synth.config_spec(config)
    with tempfile.TemporaryDirectory() as tmp:
        exe = build(tmp)
        fwd, rev, flen, rlen = synth_pairs(spec, min(distinct, pairs))
        pin = os.path.join(tmp, "pairs.bin")
        write_pairs(pin, fwd, rev, flen, rlen)
        del fwd, rev
        env = dict(os.environ)
        env["PAMD_COMPAT_ROWS"] = rows
        out = subprocess.check_output([exe, pin, str(pairs), algo or bench.CONFIGS[config]["algo"].replace("simple_bayesian", "sb"), "1" if sums == "exact" else "0", str(batch), str(consume), str(api), str(chunk)], timeout=1200, env=env)
    d = json.loads(out.decode().strip().splitlines()[-1])
    d.update(config=config, sums=sums, batch_pairs=batch, distinct_pairs=min(distinct, pairs), api=("panda_assembler_next", "panda_assembler_assemble", "pamd_compat_assemble_many")[api], rows=rows,
             path=("in-memory PandaNextSeq source -> panda_assembler_next (pinned staging, four batches in flight, index-coded download)",
                   "the caller's pair -> panda_assembler_assemble on the caller's thread",
                   "the caller's arrays, %d pairs a call -> pamd_compat_assemble_many (the machinery of panda_assembler_next)" % chunk)[api]
             + " -> consumer reading " + ("the records' scalars", "every merged nt", "every merged base")[consume])
    return d


def run_pool(config=1, pairs=8_000_000, sums="fast", threads=8, distinct=8192, algo=None):
    """tests/c/bench_pool.c: `threads` threads, each with an assembler of its own on the same GPU (pool.c:140-167), each
    consuming every merged base."""
    from pandaseq_amd import synth
    import bench
    spec = synth.config_spec(config)
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "bench_pool")
        subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "bench_pool.c"), "-o", exe, "-lpthread",
                               "-L", os.path.join(ROOT, "pandaseq_amd"), "-lpandaseq_amd", "-Wl,-rpath," + os.path.join(ROOT, "pandaseq_amd"), "-Wl,-rpath,/opt/rocm/lib"])
        fwd, rev, flen, rlen = synth_pairs(spec, min(distinct, pairs))
        pin = os.path.join(tmp, "pairs.bin")
        write_pairs(pin, fwd, rev, flen, rlen)
        del fwd, rev
        out = subprocess.check_output([exe, pin, str(pairs), algo or bench.CONFIGS[config]["algo"].replace("simple_bayesian", "sb"), "1" if sums == "exact" else "0", str(threads)], timeout=1200)
    d = json.loads(out.decode().strip().splitlines()[-1])
    d.update(config=config, sums=sums, distinct_pairs=min(distinct, pairs), api="panda_assembler_next, one assembler per thread", rows="eager", batch_pairs=65536)
    return d


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=4_000_000)
    ap.add_argument("--sums", default="exact")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--consume", type=int, default=2)
    ap.add_argument("--chunk", type=int, default=65536, help="pairs per pamd_compat_assemble_many call (--api 2)")
    ap.add_argument("--api", type=int, default=0, help="0 panda_assembler_next, 1 panda_assembler_assemble per pair, 2 pamd_compat_assemble_many")
    ap.add_argument("--rows", default="eager", help="eager | lazy (PAMD_COMPAT_ROWS)")
    ap.add_argument("--threads", type=int, default=0, help="> 0: bench_pool.c, that many threads with an assembler each")
    args = ap.parse_args()
    if args.threads > 0:
        print(json.dumps(run_pool(args.config, args.pairs, args.sums, args.threads)))
        sys.exit(0)
    print(json.dumps(run(args.config, args.pairs, args.sums, args.batch, consume=args.consume, api=args.api, rows=args.rows, chunk=args.chunk)))
