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


def test_stat_block_in_c_equals_the_python_one(built):
    """pamd_stat_block() (pamd_pipe_multi.cpp: pool.c:86-104's lines, OVERLAPS as proxy.c:175-190 prints it) against
    pandaseq_amd.stats.format_stat_block on counter blocks as a run produces them -- primers set and not, modules that rejected
    and did not, an assembler name, an empty histogram -- and its refusal of a buffer that is too small."""
    import ctypes as C
    import numpy as np
    from pandaseq_amd import _lib, stats
    L = _lib.lib()
    L.pamd_stat_block.restype = C.c_int
    L.pamd_stat_block.argtypes = [C.POINTER(C.c_int64), C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]
    rng = np.random.default_rng(3)
    for case in range(6):
        ctr = np.zeros(_lib.NUM_COUNTERS, dtype=np.int64)
        ctr[:8] = rng.integers(0, 1_000_000_000_000, size=8)
        if case != 4:
            hi = int(rng.integers(1, 2 * _lib.MAX_LEN))
            ctr[_lib.CTR_OVERLAP0:_lib.CTR_OVERLAP0 + hi] = rng.integers(0, 50_000, size=hi)
            ctr[_lib.CTR_OVERLAP0 + hi - 1] = 7
        mods = [b"min_phred", b"min_overlapbits", b"empty"][:case % 4]
        for m in range(len(mods)):
            ctr[_lib.CTR_REJECTED0 + m] = 0 if m == 1 else int(rng.integers(1, 1000))
        names = (C.c_char_p * max(1, len(mods)))(*mods) if mods else None
        name = b"worker3" if case % 2 else None
        buf = C.create_string_buffer(1 << 16)
        n = L.pamd_stat_block(ctr.ctypes.data_as(C.POINTER(C.c_int64)), case & 1, (case >> 1) & 1, names, len(mods), name, buf, len(buf))
        want = stats.format_stat_block(ctr, has_forward_primer=bool(case & 1), has_reverse_primer=bool((case >> 1) & 1),
                                       module_names=[m.decode() for m in mods], name=name.decode() if name else None) + "\n"
        assert n == len(want) and buf.value.decode() == want, (case, buf.value.decode()[:300], want[:300])
    small = C.create_string_buffer(16)
    assert L.pamd_stat_block(ctr.ctypes.data_as(C.POINTER(C.c_int64)), 0, 0, None, 0, None, small, len(small)) == -1     # PAMD_EINVAL
