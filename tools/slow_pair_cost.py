#!/usr/bin/env python3
"""What the pairs that scan every overlap cost (development aid, GPU only): kernel time of a batch as generated
against the same batch with those pairs replaced by copies of ordinary ones.
    python tools/slow_pair_cost.py [--config 1] [--pairs 8000000] [--sums fast]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
import pandaseq_amd as pa
from pandaseq_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=1)
ap.add_argument("--pairs", type=int, default=8_000_000)
ap.add_argument("--sums", default="fast")
args = ap.parse_args()
n = args.pairs
asm, spec = bench.setup_assembler(pa, synth, args.config, 0, exact_sums=args.sums == "exact")
f, r, fl, rl = synth.generate_device(spec, 0, n)
outs = pa.alloc_outputs(n, n * 440 + (1 << 27), "cuda:0")


def run(f, r, fl, rl):
    b = pa.DeviceBatch(n=n, fwd=f, rev=r, flen=fl, rlen=rl, **outs)
    best = 1e9
    for _ in range(3):
        b.cursor.zero_()
        b.counters.zero_()
        asm.assemble_device(b)
        torch.cuda.synchronize()
        best = min(best, asm.last_kernel_ms())
    return b, best


b, t_all = run(f, r, fl, rl)
res = b.results_numpy()
slow = torch.from_numpy(res["slow"] != 0).cuda()
noal = torch.from_numpy(res["status"] != 0).cuda()
idx = torch.arange(n, device="cuda")
ok = (~slow).nonzero().flatten()
# every slow pair takes the rows of an ordinary pair
repl = ok[torch.arange(int(slow.sum()), device="cuda") % ok.numel()]
idx[slow] = repl
_, t_none = run(f[idx].contiguous(), r[idx].contiguous(), fl[idx].contiguous(), rl[idx].contiguous())
ns = int(slow.sum())
print(f"pairs {n}  slow {ns} ({100.0 * ns / n:.2f} %)  not assembled {int(noal.sum())}")
print(f"kernel as generated {t_all:.3f} ms = {n / t_all / 1e3:.1f} M pairs/s; slow pairs replaced {t_none:.3f} ms = {n / t_none / 1e3:.1f} M pairs/s")
print(f"a slow pair costs {(t_all - t_none) / ns * 1e6:.1f} ns of kernel time = {(t_all - t_none) / t_all * 100:.1f} % of the kernel; an ordinary one {t_none / n * 1e6:.2f} ns")
