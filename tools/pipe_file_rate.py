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
    ap.add_argument("--pairs", type=int, default=32_000_000)
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=1_000_000)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--sink", default=None, help="write the output here instead of a file in --dir (e.g. /dev/null: the pump without the output file system)")
    ap.add_argument("--gzip", action="store_true", help="gzip inputs (level 1; one member per 250 000 pairs): one inflating reader per file")
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
        if args.sink:
            po = args.sink
        if args.gzip:
            import gzip
            ft, rt = gzip.compress(ft, 1), gzip.compress(rt, 1)
            pf, pr = pf + ".gz", pr + ".gz"
        with open(pf, "wb") as fh:
            for _ in range(reps):
                fh.write(ft)
        with open(pr, "wb") as fh:
            for _ in range(reps):
                fh.write(rt)
        in_bytes = os.path.getsize(pf) + os.path.getsize(pr)
        best = None
        runs = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            text = subprocess.check_output([exe, pf, pr, po, str(algo), "fasta", str(args.chunk)], timeout=1800).decode()
            dt = time.perf_counter() - t0
            f = [ln.split("\t") for ln in text.splitlines() if ln.startswith("PIPE")][0]
            info = dict(zip(f[1::2], f[2::2]))
            assert int(info["pairs"]) == n, (info, n)
            runs.append({"process_seconds": dt, "setup_seconds": float(info["setup_s"]), "pump_seconds": float(info["pump_s"]), "chunks": int(info["chunks"])})
            if best is None or dt < best["process_seconds"]:
                best = runs[-1]
        out_bytes = os.path.getsize(po) if not args.sink else None
    d = {"what": "tests/c/pamd_pipe.c = pamd_pipe_run() -- pamd_pipe_run_multi() over the devices of PAMD_PIPE_DEVICES when that is set --, file -> file (FASTQ x 2 -> FASTA) on %s; `pairs_per_s`: process start to exit (library load, device start-up, "
                 "allocation of the pinned blocks and thread start included); `pump_pairs_per_s`: first block read to last byte written" % args.dir,
         "config": args.config, "pairs": n, "pairs_per_chunk": args.chunk, "seconds": best["process_seconds"], "pairs_per_s": n / best["process_seconds"],
         "pump_seconds": best["pump_seconds"], "pump_pairs_per_s": n / best["pump_seconds"], "setup_seconds": best["setup_seconds"],
         "input_bytes": in_bytes, "output_bytes": out_bytes, "input_GBps": in_bytes / best["process_seconds"] / 1e9, "pump_input_GBps": in_bytes / best["pump_seconds"] / 1e9,
         "pump_output_GBps": out_bytes / best["pump_seconds"] / 1e9 if out_bytes else None, "sink": args.sink, "devices": os.environ.get("PAMD_PIPE_DEVICES", "(pamd_pipe_run on device 0)"), "gzip_inputs": bool(args.gzip), "reader_threads_per_file": int(os.environ.get("PAMD_PIPE_READERS", 3)), "writer_threads": int(os.environ.get("PAMD_PIPE_WRITERS", 1)), "runs": runs}
    print(json.dumps(d))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(d, fh, indent=1)


if __name__ == "__main__":
    main()
