"""The N > 1 path with counters the PRODUCT produced: two processes, each with its own assembler on cuda:0 and its own contiguous
shard of the pair index space (pandaseq_amd/stats.py shard_range), their int64[916] counter blocks summed by one all-reduce
(gloo here: one GPU; bench.py uses the same call over RCCL) -- the reduced block and its STAT text (format_stat_block:
pool.c:86-104's lines) equal a single assembler's over all the pairs, and the oracle's."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = 3     # rdp_mle + min_phred + min_overlapbits: the per-module rejection counters take part in the reduction


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_counters(lo, hi):
    import bench
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    asm, spec = bench.setup_assembler(pa, synth, CFG, 0, exact_sums=False)
    f, r, fl, rl = synth.generate_device(spec, lo, hi - lo)
    outs = pa.alloc_outputs(hi - lo, (hi - lo) * 520, "cuda:0")
    b = pa.DeviceBatch(n=hi - lo, fwd=f, rev=r, flen=fl, rlen=rl, **outs)
    asm.assemble_device(b)
    torch.cuda.synchronize()
    asm.check_overflow()
    return b.counters.clone()


def _worker(rank, world, port, n_total, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pandaseq_amd.stats import allreduce_counters, shard_range
    lo, hi = shard_range(n_total, rank, world)
    t = _shard_counters(lo, hi).cpu()      # (gloo reduces host tensors; on N GPUs the device tensor goes to RCCL as it is)
    allreduce_counters(t)
    if rank == 0:
        out_q.put(t.numpy().copy())
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reduce_to_a_single_run(built):
    sys.path.insert(0, ROOT)
    import bench
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    from pandaseq_amd.assembler import counters_to_dict
    from pandaseq_amd.stats import format_stat_block
    n_total, world = 200_001, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    summed = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    single = _shard_counters(0, n_total).cpu().numpy()
    assert np.array_equal(summed, single), "the reduced counter block differs from one assembler's over all pairs"
    names = ("min_phred", "min_overlapbits")
    text = format_stat_block(summed, module_names=names)
    assert text == format_stat_block(single, module_names=names)
    # ... and from the oracle's, on a prefix the host finishes in seconds
    m = 20_000
    f, r, fl, rl = synth.generate_host(synth.config_spec(CFG), 0, m)
    ctr = po.assemble_batch(bench.oracle_config(po, synth, CFG), f, r, fl, rl, want_seq=False)[3]
    d = counters_to_dict(_shard_counters(0, m).cpu().numpy())
    for k in ("count", "badr", "nofp", "norp", "noalgn", "lowq", "ok", "slow"):
        assert d[k] == int(ctr[k]), k
    assert list(d["rejected"][:2]) == [int(x) for x in ctr["rejected"][:2]]
    assert np.array_equal(d["overlapcount"], ctr["overlapcount"]) and d["longest_overlap"] == int(ctr["longest_overlap"])
    assert "STAT\tOK\t%d" % d["ok"] in format_stat_block(d, module_names=names)
