#!/usr/bin/env python3
"""Does it matter WHERE the rows go?  (development aid, GPU only)  The same 8 M config-1 pairs into the compact arena (every wave
fills 128 KB chunks of its own: 4096 write streams far apart) and into strided rows (pair i's row at i x stride: the waves, which
work on consecutive pairs at any one time, write next to each other)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pandaseq_amd as pa
from pandaseq_amd import synth
n = 8_000_000
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
spec = synth.config_spec(cfg)
f, r, fl, rl = synth.generate_device(spec, 0, n)
for stride in (0, 512, 400):
    asm = pa.Assembler(0)
    asm.set_exact_sums(False)
    outs = pa.alloc_outputs(n, n * (stride if stride else 440) + (1 << 28), "cuda:0", seq_stride=stride)
    b = pa.DeviceBatch(n=n, fwd=f, rev=r, flen=fl, rlen=rl, **outs)
    best = 1e9
    for _ in range(3):
        b.cursor.zero_(); b.counters.zero_()
        asm.assemble_device(b); torch.cuda.synchronize()
        best = min(best, asm.last_kernel_ms())
    print("seq_stride %4d: %7.3f ms  %6.1f M pairs/s" % (stride, best, n / best / 1e3))
    del b, outs
    torch.cuda.empty_cache()
