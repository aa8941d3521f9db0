#!/usr/bin/env python3
"""Per-phase cycle breakdown of the assemble kernel (development aid, GPU only).

Builds a -DPAMD_PHASE_TIMING variant of the library next to the shipped one and runs one batch
through it.  Usage: python tools/phase_timing.py [--config 1] [--pairs 4000000]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=1)
ap.add_argument("--pairs", type=int, default=4_000_000)
ap.add_argument("--only", choices=["all", "slow", "ordinary"], default="all", help="fill the batch with copies of the pairs that scan every overlap / of the others")
args = ap.parse_args()
prof = os.path.join(ROOT, "pandaseq_amd", "libpandaseq_amd_phase.so")
csrc = os.path.join(ROOT, "pandaseq_amd", "csrc")
subprocess.check_call(["make", "-s", "-C", csrc, "OUT=" + prof, "CXXFLAGS=-O3 -std=c++17 -fPIC -ffp-contract=off -DPAMD_PHASE_TIMING"])
os.environ["PAMD_LIB"] = prof
import ctypes as C

import numpy as np
import torch

import pandaseq_amd as pa
from pandaseq_amd import _lib, synth
import bench

asm, spec = bench.setup_assembler(pa, synth, args.config, 0, exact_sums=os.environ.get("PAMD_SUMS", "exact") == "exact")
n = args.pairs
f, r, fl, rl = synth.generate_device(spec, 0, n)
outs = pa.alloc_outputs(n, n * (620 if args.config == 2 else 440) + (1 << 28), "cuda:0")
b = pa.DeviceBatch(n=n, fwd=f, rev=r, flen=fl, rlen=rl, **outs)
asm.assemble_device(b)
torch.cuda.synchronize()
if args.only != "all":
    slow = torch.from_numpy(b.results_numpy()["slow"] != 0).cuda()
    src = (slow if args.only == "slow" else ~slow).nonzero().flatten()
    idx = src[torch.arange(n, device="cuda") % src.numel()]
    f, r, fl, rl = f[idx].contiguous(), r[idx].contiguous(), fl[idx].contiguous(), rl[idx].contiguous()
    b = pa.DeviceBatch(n=n, fwd=f, rev=r, flen=fl, rlen=rl, **outs)
    b.cursor.zero_(); b.counters.zero_()
    asm.assemble_device(b)
    torch.cuda.synchronize()
buf = np.zeros(10, dtype=np.uint64)
_lib.check(_lib.lib().pamd_debug_phase_cycles(asm._h, buf.ctypes.data, 1))
b.cursor.zero_(); b.counters.zero_()
asm.assemble_device(b)
torch.cuda.synchronize()
ms = asm.last_kernel_ms()
_lib.check(_lib.lib().pamd_debug_phase_cycles(asm._h, buf.ctypes.data, 1))
names = ["stage", "offsets", "seed-fwd", "seed-rev", "scan", "merge+row stores", "-", "-", "group finish", "whole-wave"]
tot = float(buf[:9].sum())
print(f"config {args.config}, {n} pairs, kernel {ms:.2f} ms ({n / ms / 1e3:.1f} M pairs/s with timing overhead)")
for k, nm in enumerate(names):
    print(f"  {nm:14s} {buf[k] / n:10.0f} cycles/pair  {100.0 * buf[k] / tot if k < 9 else 0:5.1f}%")
