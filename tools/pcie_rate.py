#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (pamd_assemble_host: hipMalloc + H2D + kernel +
D2H + hipFree per call).  Reported in DESIGN.md only; it is never bench.py's `value`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import pandaseq_amd as pa
from pandaseq_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
spec = synth.config_spec(1)
import torch
f, r, fl, rl = synth.generate_device(spec, 0, n)
hf, hr = f.cpu().numpy(), r.cpu().numpy()
hfl, hrl = fl.cpu().numpy().view(np.uint16), rl.cpu().numpy().view(np.uint16)
asm = pa.Assembler(0)
asm.assemble_host(hf[:1000], hr[:1000], hfl[:1000], hrl[:1000], seq_stride=0, capacity=1000 * 512 + 16384 * 4096)
for rep in range(2):
    t0 = time.perf_counter()
    res, nt, p, ctr = asm.assemble_host(hf, hr, hfl, hrl, seq_stride=0, capacity=n * 420 + 16384 * 4096)
    dt = time.perf_counter() - t0
    print(f"pamd_assemble_host: {n} pairs in {dt:.3f} s = {n / dt / 1e6:.2f} M pairs/s (ok={ctr['ok']})")
