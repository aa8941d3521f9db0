#!/usr/bin/env python3
"""Per-phase cost of the assemble kernel by ablation (development aid, GPU only).

Build variant k abandons every pair after phase k (-DPAMD_STOP_AFTER=k); the kernel time of variant k
minus variant k-1 is what phase k costs under real occupancy.  Unlike tools/phase_timing.py this adds no
instructions to the kernel that is measured.  Build the variants first (hipcc, no GPU needed):

    python tools/phase_ablation.py --build          # here or on the GPU box
    python tools/phase_ablation.py [--config 1] [--pairs 8000000] [--sums fast]
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = ["stage", "offsets", "seed-fwd", "seed-rev", "scan"]
ap = argparse.ArgumentParser()
ap.add_argument("--build", action="store_true")
ap.add_argument("--config", type=int, default=1)
ap.add_argument("--pairs", type=int, default=8_000_000)
ap.add_argument("--sums", default="fast")
ap.add_argument("--variant", type=int, default=None, help="internal: measure one variant and print its ms")
args = ap.parse_args()
csrc = os.path.join(ROOT, "pandaseq_amd", "csrc")


def lib_path(k):
    return os.path.join(ROOT, "pandaseq_amd", f"libpandaseq_amd_stop{k}.so")


if args.build:
    procs = [subprocess.Popen(["make", "-s", "-C", csrc, "OUT=" + lib_path(k),
                               f"CXXFLAGS=-O3 -std=c++17 -fPIC -ffp-contract=off -DPAMD_STOP_AFTER={k}"]) for k in range(len(PHASES))]
    sys.exit(max(p.wait() for p in procs))

if args.variant is not None:
    if args.variant < len(PHASES):
        os.environ["PAMD_LIB"] = lib_path(args.variant)
    import torch

    import bench
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    asm, spec = bench.setup_assembler(pa, synth, args.config, 0, exact_sums=args.sums == "exact")
    n = args.pairs
    f, r, fl, rl = synth.generate_device(spec, 0, n)
    outs = pa.alloc_outputs(n, n * 420 + (1 << 27), "cuda:0")
    b = pa.DeviceBatch(n=n, fwd=f, rev=r, flen=fl, rlen=rl, **outs)
    best = 1e30
    for _ in range(4):
        b.cursor.zero_()
        b.counters.zero_()
        asm.assemble_device(b)
        torch.cuda.synchronize()
        best = min(best, asm.last_kernel_ms())
    print(best)
    sys.exit(0)

ms = []
for k in range(len(PHASES) + 1):
    out = subprocess.check_output([sys.executable, os.path.abspath(__file__), "--variant", str(k), "--config", str(args.config),
                                   "--pairs", str(args.pairs), "--sums", args.sums])
    ms.append(float(out.decode().strip().splitlines()[-1]))
names = PHASES + ["merge + finish"]
print(f"config {args.config}, {args.pairs} pairs, sums={args.sums}: full kernel {ms[-1]:.2f} ms = {args.pairs / ms[-1] / 1e3:.1f} M pairs/s")
prev = 0.0
for nm, m in zip(names, ms):
    print(f"  {nm:16s} {m - prev:8.2f} ms  {100.0 * (m - prev) / ms[-1]:5.1f} %   (cumulative {m:.2f})")
    prev = m
print("  (stage includes the loop overhead: prefetch loads, result record, counters)")
