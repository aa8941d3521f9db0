"""world_size-2 gloo test of the only N>1 step on this path: shard the pairs, all-reduce the STAT counters."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from oracle import pyoracle as po
    from pandaseq_amd import _lib, synth
    from pandaseq_amd.stats import allreduce_counters, shard_range
    lo, hi = shard_range(n_total, rank, world)
    f, r, fl, rl = synth.generate_numpy(synth.config_spec(4), lo, hi - lo)
    # the shard's counters come from the oracle here (no GPU in this test); what is under test is
    # the sharding + reduction logic that bench.py uses unchanged on RCCL
    _, _, _, ctr = po.assemble_batch(po.make_config("simple_bayesian"), f, r, fl, rl, want_seq=False)
    block = np.zeros(_lib.NUM_COUNTERS, np.int64)
    for k, i in _lib.CTR.items():
        block[i] = ctr[k]
    block[_lib.CTR_REJECTED0:_lib.CTR_REJECTED0 + 8] = ctr["rejected"]
    block[_lib.CTR_OVERLAP0:] = ctr["overlapcount"]
    t = torch.from_numpy(block)
    allreduce_counters(t)
    if rank == 0:
        out_q.put(t.numpy().copy())
    dist.destroy_process_group()


def test_sharded_counters_allreduce_equals_single_run(built):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from oracle import pyoracle as po
    from pandaseq_amd import _lib, synth
    from pandaseq_amd.assembler import counters_to_dict
    n_total, world = 3001, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    summed = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    f, r, fl, rl = synth.generate_numpy(synth.config_spec(4), 0, n_total)
    _, _, _, ctr = po.assemble_batch(po.make_config("simple_bayesian"), f, r, fl, rl, want_seq=False)
    d = counters_to_dict(summed)
    for k in ("count", "badr", "nofp", "norp", "noalgn", "lowq", "ok", "slow"):
        assert d[k] == int(ctr[k]), k
    assert np.array_equal(d["overlapcount"], ctr["overlapcount"])
    assert d["longest_overlap"] == int(ctr["longest_overlap"])  # max recovered from the summed histogram
