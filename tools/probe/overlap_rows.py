import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, pandaseq_amd as pa
from pandaseq_amd import synth
cfg = int(sys.argv[1]); n = 1_000_000
spec = synth.config_spec(cfg)
f, r, fl, rl = synth.generate_device(spec, 0, n)
for fast in (True, False):
  for exact in (True, False):
    asm, _ = bench.setup_assembler(pa, synth, cfg, 0, exact_sums=exact)
    asm.set_fast_path(fast)
    outs = pa.alloc_outputs(n, n * 600 + (1 << 28), "cuda:0")
    b = pa.DeviceBatch(n=n, fwd=f, rev=r, flen=fl, rlen=rl, **outs)
    asm.assemble_device(b); torch.cuda.synchronize()
    res = b.results_numpy().copy()
    ok = res["status"] == 0
    idx = np.nonzero(ok)[0]
    offs = res["seq_offset"][ok].astype(np.int64)
    order = np.argsort(offs)
    ends = offs[order] + res["seq_len"][ok][order]
    bad = np.nonzero(ends[:-1] > offs[order][1:])[0]
    print("fast", fast, "exact", exact, "overlapping rows:", bad.size, "of", ok.sum())
    for k in bad[:5]:
        i, j = idx[order[k]], idx[order[k + 1]]
        print("  pair", i, res[i], "\n  next", j, res[j])
    del b, outs; torch.cuda.empty_cache()
