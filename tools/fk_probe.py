#!/usr/bin/env python3
"""The lean kernel against the general one on the same batch (development aid, GPU only).

    python tools/fk_probe.py [--config 1] [--pairs 8000000] [--algo simple_bayesian]

Runs the batch twice -- lean + general (the default two-kernel launch) and general only -- and reports the time
of each, how many pairs the lean kernel deferred, and whether every record, merged base, log-p and counter agrees.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import pandaseq_amd as pa
from pandaseq_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=1)
ap.add_argument("--pairs", type=int, default=8_000_000)
ap.add_argument("--algo", default=None)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--sums", default="fast")
ap.add_argument("--no-compare", action="store_true")
ap.add_argument("--primers-after", action="store_true", help="`-a` (config 2 has primers)")
ap.add_argument("--min-phred", type=int, default=2)
ap.add_argument("--only", default=None, help="two-kernel or general: run just that mode (implies --no-compare)")
ap.add_argument("--read-len", type=int, default=int(os.environ.get("PAMD_PROBE_READ_LEN", "0")), help="shorter reads in the configuration's rows (the stride stays): what idle lanes cost")
args = ap.parse_args()

n = args.pairs
spec = synth.config_spec(args.config)
if args.read_len:
    import dataclasses
    cut = spec.read_len - args.read_len
    spec = dataclasses.replace(spec, read_len=args.read_len, stride=spec.row_stride(), tmin=spec.tmin - cut, tmax=spec.tmax - 2 * cut if spec.tmax - 2 * cut > spec.tmin - cut else spec.tmin - cut + 40)
f, r, fl, rl = synth.generate_device(spec, 0, n)
out = {}
if args.only:
    args.no_compare = True
for mode in ((args.only,) if args.only else ("two-kernel", "general")):
    asm, _ = bench.setup_assembler(pa, synth, args.config, 0, exact_sums=args.sums == "exact", algo=args.algo, primers_after=args.primers_after or os.environ.get("PAMD_PRIMERS_AFTER") == "1", min_phred=args.min_phred)
    asm.set_fast_path(mode == "two-kernel")
    outs = pa.alloc_outputs(n, n * (620 if args.config == 2 else 440) + (1 << 28), "cuda:0")
    b = pa.DeviceBatch(n=n, fwd=f, rev=r, flen=fl, rlen=rl, **outs)
    best = 1e30
    for _ in range(args.reps):
        b.cursor.zero_()
        b.counters.zero_()
        asm.assemble_device(b)
        torch.cuda.synchronize()
        best = min(best, asm.last_kernel_ms())
    asm.check_overflow()
    if mode == "two-kernel":
        import ctypes as C
        from pandaseq_amd import _lib
        ph = (C.c_ulonglong * 10)()
        _lib.lib().pamd_debug_phase_cycles(asm._h, ph, 1)
        why = [v // args.reps for v in ph]
        print("deferred because: domain %d, k-mer lost %d, score too close %d, many candidates %d, LOWQ guard %d, primer scan %d, clipped overlap %d; (debug builds) reads settled block by block: %d without, %d with a summed alignment" % (why[0], why[1], why[2], why[3], why[4], why[6], why[7], why[8], why[9]))
    print(f"{mode:11s} {best:9.3f} ms  {n / best / 1e3:8.1f} M pairs/s  deferred {asm.last_deferred()}  launch {asm.last_launch_info()}  arena {int(b.cursor.item())}")
    if not args.no_compare:
        res = b.results_numpy()
        order = np.argsort(res["seq_offset"], kind="stable")
        out[mode] = (res.copy(), b.counters.cpu().numpy().copy(), b)
if not args.no_compare:
    ra, ca, ba = out["two-kernel"]
    rb, cb, bb = out["general"]
    print("counters equal:", bool(np.array_equal(ca, cb)))
    same = True
    for name in ra.dtype.names:
        if name in ("seq_offset", "reserved"):
            continue
        eq = ra[name] == rb[name]
        if name in ("quality",) and args.sums != "exact":
            eq = np.abs(ra[name] - rb[name]) <= 1e-9
        if not eq.all():
            bad = np.nonzero(~eq)[0]
            print(f"field {name}: {bad.size} differ, first {bad[:8]}: {ra[name][bad[:8]]} vs {rb[name][bad[:8]]} status {ra['status'][bad[:8]]}/{rb['status'][bad[:8]]}")
            same = False
    # merged rows of a sample of accepted pairs
    ok = np.nonzero(ra["status"] == 0)[0]
    rng = np.random.default_rng(1)
    pick = rng.choice(ok, size=min(20000, ok.size), replace=False)
    nta, pa_ = ba.seq_nt, ba.seq_p
    ntb, pb_ = bb.seq_nt, bb.seq_p
    bad_rows = 0
    idx = torch.arange(512, device="cuda:0")
    for chunk in np.array_split(pick, 20):
        oa = torch.from_numpy(ra["seq_offset"][chunk].astype(np.int64)).cuda()
        ob = torch.from_numpy(rb["seq_offset"][chunk].astype(np.int64)).cuda()
        ln = torch.from_numpy(ra["seq_len"][chunk].astype(np.int64)).cuda()
        m = idx[None, :] < ln[:, None]
        ga = nta[(oa[:, None] + idx[None, :]).clamp(max=nta.numel() - 1)]
        gb = ntb[(ob[:, None] + idx[None, :]).clamp(max=ntb.numel() - 1)]
        qa = pa_[(oa[:, None] + idx[None, :]).clamp(max=nta.numel() - 1)]
        qb = pb_[(ob[:, None] + idx[None, :]).clamp(max=ntb.numel() - 1)]
        bad_rows += int((((ga != gb) | (qa != qb)) & m).any(dim=1).sum().item())
    print("records equal:", same, " sampled rows differing:", bad_rows, "of", pick.size)
