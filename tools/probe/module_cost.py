#!/usr/bin/env python3
"""What do the two fused modules cost the lean kernel?  (development aid, GPU only)  rdp_mle on config-1 data, 8 M pairs:
no module, min_phred alone, min_overlapbits alone, both (= config 3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pandaseq_amd as pa
from pandaseq_amd import synth
n = 8_000_000
spec = synth.config_spec(1)
f, r, fl, rl = synth.generate_device(spec, 0, n)
for mods in ((), ("min_phred:2",), ("min_overlapbits:15",), ("min_phred:2", "min_overlapbits:15")):
    asm = pa.Assembler(0)
    asm.set_exact_sums(False)
    asm.set_algorithm("rdp_mle")
    for m in mods:
        asm.add_module(m)
    outs = pa.alloc_outputs(n, n * 440 + (1 << 28), "cuda:0")
    b = pa.DeviceBatch(n=n, fwd=f, rev=r, flen=fl, rlen=rl, **outs)
    best = 1e9
    for _ in range(3):
        b.cursor.zero_(); b.counters.zero_()
        asm.assemble_device(b); torch.cuda.synchronize()
        best = min(best, asm.last_kernel_ms())
    print("%-40s %7.3f ms  %6.1f M pairs/s  deferred %d" % (" + ".join(mods) or "(no module)", best, n / best / 1e3, asm.last_deferred()))
    del b, outs
