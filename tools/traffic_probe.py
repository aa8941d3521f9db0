#!/usr/bin/env python3
"""HBM traffic of one assemble launch, for `roofline.traffic`.

Run under rocprofv3 twice (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2, so one pass each):

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/traffic_rd -o t -- python3 tools/traffic_probe.py
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/traffic_wr -o t -- python3 tools/traffic_probe.py

The same process also runs a 1 GiB device-to-device copy (16 B per lane, coalesced): its known byte
count calibrates the counters' unit, as MI355X_MICROARCH.md prescribes (on gfx950 FETCH_SIZE reads half
of a wide streaming read).  profiles/summarize_traffic.py turns the two databases into bytes per launch.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import pandaseq_amd as pa
from pandaseq_amd import synth
import bench

cfg = int(os.environ.get("PAMD_CONFIG", "1"))
n = int(os.environ.get("PAMD_PAIRS", "8000000"))
asm, spec = bench.setup_assembler(pa, synth, cfg, 0, exact_sums=os.environ.get("PAMD_SUMS", "exact") == "exact",
                                  primers_after=os.environ.get("PAMD_PRIMERS_AFTER", "0") == "1", min_phred=int(os.environ.get("PAMD_MIN_PHRED", "2")))
asm.set_fast_path(os.environ.get("PAMD_FAST_PATH", "1") == "1")
f, r, fl, rl = synth.generate_device(spec, 0, n)
outs = pa.alloc_outputs(n, n * (620 if cfg == 2 else 440) + (1 << 28), "cuda:0")
b = pa.DeviceBatch(n=n, fwd=f, rev=r, flen=fl, rlen=rl, **outs)
asm.assemble_device(b)          # warm (tables, code objects)
torch.cuda.synchronize()
b.cursor.zero_()
b.counters.zero_()
asm.assemble_device(b)          # the measured launch = the LAST assemble dispatch in the trace
torch.cuda.synchronize()
src = torch.empty(1 << 28, dtype=torch.int32, device="cuda:0").fill_(7)
dst = torch.empty_like(src)
torch.cuda.synchronize()
dst.copy_(src)                  # calibration: reads 1 GiB, writes 1 GiB
torch.cuda.synchronize()
print("pairs", n, "config", cfg, "arena_used_bases", int(b.cursor.item()))
