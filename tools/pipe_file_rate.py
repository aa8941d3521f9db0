#!/usr/bin/env python3
"""File -> file rate through the C ABI only (development aid, GPU only; never bench.py's `value`): tests/c/pamd_pipe.c -- two
FASTQ files in, one FASTA file out, parse + assemble + format on the device, plain C around it --
timed on files in a memory-backed directory, so that the figure is the pipeline's and not a disk's.

    python tools/pipe_file_rate.py [--pairs 2000000] [--config 1] [--chunk 1000000] [--dir /dev/shm]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=2_000_000)
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=1_000_000)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import bench
    import bench_next
    from pandaseq_amd import synth
    spec = synth.config_spec(args.config)
    base_n = min(args.pairs, 250_000)
    f, r, fl, rl = bench_next.synth_pairs(spec, base_n)
    ft, rt = synth.render_fastq(f, r, fl, rl)
    reps = max(1, args.pairs // base_n)
    n = base_n * reps
    algo = ["simple_bayesian", "pear", "rdp_mle", "uparse", "flash", "ea_util", "stitch"].index(bench.CONFIGS[args.config]["algo"])
    with tempfile.TemporaryDirectory(dir=args.dir) as tmp, tempfile.TemporaryDirectory() as bindir:
        exe = os.path.join(bindir, "pamd_pipe")    # (a memory-backed directory is often mounted noexec)
        subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                               os.path.join(ROOT, "tests", "c", "pamd_pipe.c"), "-o", exe,
                               "-L", os.path.join(ROOT, "pandaseq_amd"), "-lpandaseq_amd", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                               "-Wl,-rpath," + os.path.join(ROOT, "pandaseq_amd"), "-Wl,-rpath,/opt/rocm/lib"])
        pf, pr, po = (os.path.join(tmp, x) for x in ("f.fastq", "r.fastq", "out.fasta"))
        with open(pf, "wb") as fh:
            for _ in range(reps):
                fh.write(ft)
        with open(pr, "wb") as fh:
            for _ in range(reps):
                fh.write(rt)
        in_bytes = os.path.getsize(pf) + os.path.getsize(pr)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            subprocess.check_call([exe, pf, pr, po, str(algo), "fasta", str(args.chunk)], timeout=1200)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        out_bytes = os.path.getsize(po)
    d = {"what": "tests/c/pamd_pipe.c, file -> file (FASTQ x 2 -> FASTA) on %s: process start to exit, library and device start-up included" % args.dir,
         "config": args.config, "pairs": n, "pairs_per_chunk": args.chunk, "seconds": best, "pairs_per_s": n / best,
         "input_bytes": in_bytes, "output_bytes": out_bytes, "input_GBps": in_bytes / best / 1e9}
    print(json.dumps(d))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(d, fh, indent=1)


if __name__ == "__main__":
    main()
