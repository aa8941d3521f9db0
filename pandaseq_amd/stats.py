"""STAT counters across GPUs.

The reference never sums its counters: every pthread prints its own STAT block
(pool.c:86-104, README.md:99-101).  With one process per GPU the natural equivalent is one
all-reduce(sum) of the int64 counter block over RCCL at the end of a run -- the ONLY collective on
this path (read pairs are independent).  ``longest_overlap`` is a running max in the
reference (assembler.c:342-344); after summing histograms it is the highest non-empty overlap bin,
so it needs no separate max-reduce.
"""
import numpy as np

from . import _lib
from .assembler import counters_to_dict


def allreduce_counters(counters, group=None):
    """In-place sum of the int64[NUM_COUNTERS] block over all ranks (RCCL on GPU tensors, gloo on CPU)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
    return counters


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of the global pair index space owned by `rank`."""
    lo = n_total * rank // world
    hi = n_total * (rank + 1) // world
    return lo, hi


def format_stat_block(counters, has_forward_primer=False, has_reverse_primer=False, module_names=(), name=None):
    """The STAT lines a reference worker prints at end of input, in its order and format
    (pool.c:86-104: NOFP/NORP only when a primer is set, then NOALGN LOWQ BADR SLOW, one line per
    module with rejects (module.c:208-216), OK, and the OVERLAPS histogram up to longest_overlap,
    proxy.c:175-190).  Lines are prefixed with the assembler name when one is set (proxy.c:165-172)."""
    d = counters if isinstance(counters, dict) else counters_to_dict(np.asarray(counters))
    pre = f"{name}\t" if name else ""
    lines = [f"{pre}STAT\tREADS\t{d['count']}"]
    if has_forward_primer:
        lines.append(f"{pre}STAT\tNOFP\t{d['nofp']}")
    if has_reverse_primer:
        lines.append(f"{pre}STAT\tNORP\t{d['norp']}")
    lines.append(f"{pre}STAT\tNOALGN\t{d['noalgn']}")
    lines.append(f"{pre}STAT\tLOWQ\t{d['lowq']}")
    lines.append(f"{pre}STAT\tBADR\t{d['badr']}")
    lines.append(f"{pre}STAT\tSLOW\t{d['slow']}")
    for i, mod in enumerate(module_names):
        if d["rejected"][i] > 0:
            lines.append(f"{pre}STAT\t{mod}\t{d['rejected'][i]}")
    lines.append(f"{pre}STAT\tOK\t{d['ok']}")
    ov = d["overlapcount"]
    lines.append(f"{pre}STAT\tOVERLAPS\t" + " ".join(str(int(x)) for x in ov[: d["longest_overlap"] + 1]))
    return "\n".join(lines)
