"""Host-side mirror of the reference's assembler interface for the hot path.

``Assembler`` carries the same knobs, with the same names and the same "silently ignore bad
values" behaviour, as the setters of pandaseq-assembler.h:199-405 / assembler_support.c:177-410,
and drives the HIP kernel through the C ABI (``include/pandaseq_amd.h``).  PyTorch is used only
for device memory and streams.  There is no CPU path here.
"""
import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from ._lib import ALGO, NUM_COUNTERS, RESULT_DTYPE, Batch, check, lib


def _torch():
    import torch
    return torch


@dataclass
class DeviceBatch:
    """A batch of read pairs resident in HBM plus the buffers its results go to."""
    n: int
    fwd: "object"       # uint8 [n, fstride, 2]   panda_qual rows
    rev: "object"       # uint8 [n, rstride, 2]
    flen: "object"      # int16 storage of uint16 [n]
    rlen: "object"
    results: "object"   # uint8 [n, 48]  pamd_pair_result records
    seq_nt: "object"    # uint8 [capacity]
    seq_p: "object"     # float64 [capacity]
    cursor: "object"    # int64 [1]   arena cursor (compact mode)
    counters: "object"  # int64 [NUM_COUNTERS]
    seq_stride: int = 0

    @property
    def capacity(self):
        return int(self.seq_nt.numel())

    def results_numpy(self):
        return self.results.cpu().numpy().view(RESULT_DTYPE).reshape(-1)

    def counters_dict(self):
        return counters_to_dict(self.counters.cpu().numpy())

    def rows_numpy(self):
        """(seq_nt, seq_p) on the host.  Of a compact arena only the part the launch's waves reserved (everything below the
        cursor: the rows lie there) -- the arena of a 2 000-pair test batch is sized for its chunks, a gigabyte, and its rows are
        a megabyte."""
        used = self.capacity
        if self.seq_stride == 0:
            used = max(0, min(used, int(self.cursor.item())))
        return self.seq_nt[:used].cpu().numpy(), self.seq_p[:used].cpu().numpy()


def counters_to_dict(c):
    c = np.asarray(c)
    d = {k: int(c[i]) for k, i in _lib.CTR.items()}
    d["rejected"] = [int(x) for x in c[_lib.CTR_REJECTED0:_lib.CTR_REJECTED0 + _lib.MAX_MODULES]]
    ov = c[_lib.CTR_OVERLAP0:_lib.CTR_OVERLAP0 + 2 * _lib.MAX_LEN]
    d["overlapcount"] = ov.astype(np.int64)
    nz = np.nonzero(ov)[0]
    # longest_overlap is a running max in the reference (assembler.c:342-344); after summing
    # histograms it is the highest non-empty bin
    d["longest_overlap"] = int(nz[-1]) if nz.size else 0
    return d


def alloc_outputs(n, capacity, device, seq_stride=0):
    torch = _torch()
    cap = ((int(capacity) + 15) // 16) * 16 + 16
    return dict(
        results=torch.empty((n, 48), dtype=torch.uint8, device=device),
        seq_nt=torch.empty(cap, dtype=torch.uint8, device=device),
        seq_p=torch.empty(cap, dtype=torch.float64, device=device),
        cursor=torch.zeros(1, dtype=torch.int64, device=device),
        counters=torch.zeros(NUM_COUNTERS, dtype=torch.int64, device=device),
        seq_stride=seq_stride,
    )


def row_checksums(results, seq_nt, seq_p, first=0, count=None, seq_stride=0, block=262144):
    """Two position-weighted checksums of every merged row of pairs [first, first + count), computed where the rows lie (in HBM):
    cs_nt[i] = sum (j+1) nt_j and cs_p[i] = sum (j+1) * (the 64 bits of p_j), both in wrapping 64-bit integer arithmetic, so
    that the result does not depend on the order of addition and can be held, bit for bit, against a checker's on another
    machine (bench.py's at-scale parity check; "a checksum of checksums" over a whole run).  0 for a pair without a row.
    `results`: uint8 [n, 48] records; rows at results.seq_offset in the compact arena, or at i * seq_stride.  Returns two int64
    tensors (the bit patterns of the uint64 sums)."""
    torch = _torch()
    n = results.shape[0]
    count = n - first if count is None else count
    dev = results.device
    cs_nt = torch.zeros(count, dtype=torch.int64, device=dev)
    cs_p = torch.zeros(count, dtype=torch.int64, device=dev)
    p_bits = seq_p.view(torch.int64)
    for lo in range(first, first + count, block):
        hi = min(first + count, lo + block)
        rec = results[lo:hi]
        m = rec[:, 24:26].contiguous().view(torch.int16).to(torch.int64).flatten() & 0xFFFF
        m = m * (rec[:, 38] == 0)      # accepted pairs only
        off = rec[:, 16:24].contiguous().view(torch.int64).flatten()
        if seq_stride:
            off = torch.arange(lo, hi, device=dev, dtype=torch.int64) * seq_stride
        off = torch.where(m > 0, off, torch.zeros_like(off))
        total = int(m.sum().item())
        if total == 0:
            continue
        seg = torch.repeat_interleave(torch.arange(hi - lo, device=dev, dtype=torch.int64), m, output_size=total)
        start = torch.cumsum(m, 0) - m
        j = torch.arange(total, device=dev, dtype=torch.int64) - start[seg]
        idx = off[seg] + j
        w = j + 1
        cs_nt[lo - first:hi - first].index_add_(0, seg, seq_nt[idx].to(torch.int64) * w)
        cs_p[lo - first:hi - first].index_add_(0, seg, p_bits[idx] * w)
    return cs_nt, cs_p


def upload_batch(fwd, rev, flen, rlen, device="cuda:0", seq_stride=None, capacity=None):
    """numpy panda_qual rows -> DeviceBatch.  seq_stride=None picks strided output rows wide enough
    for any merge; seq_stride=0 selects the compact arena (capacity bases)."""
    torch = _torch()
    fwd = np.ascontiguousarray(fwd, dtype=np.uint8)
    rev = np.ascontiguousarray(rev, dtype=np.uint8)
    n = fwd.shape[0]
    fl = np.ascontiguousarray(flen, dtype=np.uint16)
    rl = np.ascontiguousarray(rlen, dtype=np.uint16)
    if seq_stride is None:
        seq_stride = min(2 * _lib.MAX_LEN, fwd.shape[1] + rev.shape[1])
        seq_stride = ((seq_stride + 15) // 16) * 16
    if capacity is None:
        capacity = n * (seq_stride if seq_stride else ((min(900, fwd.shape[1] + rev.shape[1]) + 15) // 16) * 16) + 2 * 16384 * 4096
    outs = alloc_outputs(n, capacity if seq_stride == 0 else n * seq_stride, device, seq_stride)
    return DeviceBatch(
        n=n,
        fwd=torch.from_numpy(fwd).to(device), rev=torch.from_numpy(rev).to(device),
        flen=torch.from_numpy(fl.view(np.int16)).to(device), rlen=torch.from_numpy(rl.view(np.int16)).to(device),
        **outs)


class Assembler:
    """The configured overlap assembler on one GPU (mirror of PandaAssembler's hot-path surface)."""

    def __init__(self, device: int = 0):
        self._h = lib().pamd_assembler_create(int(device))
        if not self._h:
            raise _lib.PamdError(lib().pamd_last_error().decode())
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            lib().pamd_assembler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- configuration: names follow pandaseq-assembler.h ----
    def set_algorithm(self, name: str, param: Optional[float] = None):
        """`-A name[:param]` (args_assembler.c:26-60). param: simple_bayesian / uparse -> q; pear -> random-base
        probability (NOT its log; algo_pear.c:84-89 takes the log); the others take none."""
        if name not in ALGO:
            raise _lib.PamdError(f"algorithm {name!r} is not on the device path ({', '.join(ALGO)})")
        p = float("nan") if param is None else float(param)
        if name == "pear" and param is not None:
            p = math.log(param)
        check(lib().pamd_set_algorithm(self._h, ALGO[name], p))

    def set_threshold(self, t: float):
        check(lib().pamd_set_threshold(self._h, float(t)))

    def set_minimum_overlap(self, v: int):
        check(lib().pamd_set_minimum_overlap(self._h, int(v)))

    def set_maximum_overlap(self, v: int):
        check(lib().pamd_set_maximum_overlap(self._h, int(v)))

    def set_forward_primer(self, nt):
        a = np.ascontiguousarray(nt, dtype=np.uint8)
        check(lib().pamd_set_forward_primer(self._h, a.ctypes.data, a.size))

    def set_reverse_primer(self, nt):
        a = np.ascontiguousarray(nt, dtype=np.uint8)
        check(lib().pamd_set_reverse_primer(self._h, a.ctypes.data, a.size))

    def set_forward_trim(self, n: int):
        check(lib().pamd_set_forward_trim(self._h, int(n)))

    def set_reverse_trim(self, n: int):
        check(lib().pamd_set_reverse_trim(self._h, int(n)))

    def set_primer_penalty(self, p: float):
        check(lib().pamd_set_primer_penalty(self._h, float(p)))

    def set_primers_after(self, after: bool):
        """`-a`: strip the primers after assembly instead of before (assembler_support.c:355-359)."""
        check(lib().pamd_set_primers_after(self._h, 1 if after else 0))

    def add_module(self, spec: str):
        """`-C name:args` for the two fused check modules (module.c:244-247 splits name from args)."""
        name, _, args = spec.partition(":")
        if name == "min_phred":
            check(lib().pamd_add_module_min_phred(self._h, int(args)))
        elif name == "min_overlapbits":
            check(lib().pamd_add_module_min_overlapbits(self._h, float(args) if args else 15.0))
        else:
            raise _lib.PamdError(f"module {name!r} is not fused on the device path")

    def set_exact_sums(self, exact: bool):
        """True (default): ordered sums, every output bit-identical to the reference.  False: any-order
        sums with exactness guards on every decision (see include/pandaseq_amd.h)."""
        check(lib().pamd_set_exact_sums(self._h, 1 if exact else 0))

    def clear_modules(self):
        check(lib().pamd_clear_modules(self._h))

    def set_fast_path(self, enable: bool):
        """False: every pair goes through the general kernel (the lean kernel of pamd_fast.hip is skipped).
        Results are the same either way; tests use it to compare the two."""
        check(lib().pamd_set_fast_path(self._h, 1 if enable else 0))

    def last_deferred(self) -> int:
        """Pairs the lean kernel handed to the general one in the last launch (-1: it did not run)."""
        return int(lib().pamd_last_deferred(self._h))

    LEAN_NAMES = {0: "general", 1: "short", 2: "long", 3: "primers", 4: "after"}

    def last_lean_variant(self) -> str:
        """Which instantiation of the lean kernel served the last launch: general (none), short, short+narrow, long, primers, after."""
        v = int(lib().pamd_last_lean_variant(self._h))
        return self.LEAN_NAMES[v & 0xFF] + ("+narrow" if v & 0x100 else "") + ("+dual" if v & 0x200 else "")

    def set_two_pairs_per_wave(self, enable: bool):
        """Reads of at most 160 bases are assembled two pairs per wavefront where the configuration allows (default OFF: measured slower)."""
        check(lib().pamd_set_two_pairs_per_wave(self._h, 1 if enable else 0))

    def check_overflow(self):
        """After synchronising: raise if pairs of the last launch did not fit in the output arena
        (their records carry status NOSPACE)."""
        n = C.c_uint64(0)
        check(lib().pamd_assemble_status(self._h, C.byref(n)))

    # ---- the hot path ----
    def assemble_device(self, b: DeviceBatch, stream=None, first=None, count=None, arena=None):
        """Enqueue the assembly of a device-resident batch on the current (or given) torch stream.
        first / count: only pairs [first, first + count) (pamd_assemble_device_range); arena: a dict with seq_nt, seq_p,
        cursor (as from alloc_outputs) that receives this range's rows instead of the batch's own arena."""
        torch = _torch()
        st = stream if stream is not None else torch.cuda.current_stream(b.fwd.device)
        cb = Batch()
        cb.n_pairs = b.n
        cb.fwd, cb.fwd_stride = b.fwd.data_ptr(), b.fwd.shape[1]
        cb.rev, cb.rev_stride = b.rev.data_ptr(), b.rev.shape[1]
        cb.flen, cb.rlen = b.flen.data_ptr(), b.rlen.data_ptr()
        cb.results = b.results.data_ptr()
        cb.seq_nt, cb.seq_p = b.seq_nt.data_ptr(), b.seq_p.data_ptr()
        cb.seq_stride = b.seq_stride
        cb.seq_capacity = b.capacity - 16
        cb.seq_cursor = b.cursor.data_ptr()
        cb.counters = b.counters.data_ptr()
        if arena is not None:
            cb.seq_nt, cb.seq_p = arena["seq_nt"].data_ptr(), arena["seq_p"].data_ptr()
            cb.seq_capacity = int(arena["seq_nt"].numel()) - 16
            cb.seq_cursor = arena["cursor"].data_ptr()
        if first is not None:
            check(lib().pamd_assemble_device_range(self._h, C.byref(cb), int(first), int(count if count is not None else b.n - first), C.c_void_p(st.cuda_stream)))
        else:
            check(lib().pamd_assemble_device(self._h, C.byref(cb), C.c_void_p(st.cuda_stream)))

    def encode_rows(self, b: "DeviceBatch", n_bases=None, stream=None):
        """The batch's merged log-p as u16 indices into table_image() (2 bytes per base for the trip to the host):
        returns an int16-storage tensor of n_bases entries (default: the arena's used part)."""
        torch = _torch()
        st = stream if stream is not None else torch.cuda.current_stream(b.fwd.device)
        n = int(b.cursor.item()) if n_bases is None else int(n_bases)
        n = (n + 1) & ~1
        out = torch.empty(max(n, 2), dtype=torch.int16, device=b.seq_p.device)
        check(lib().pamd_encode_rows_device(self._h, b.seq_p.data_ptr(), n, out.data_ptr(), C.c_void_p(st.cuda_stream)))
        return out[:n]

    def table_image(self):
        """The table image the indices of encode_rows() point into (host copy, float64)."""
        buf = np.zeros(8192, dtype=np.float64)
        used = C.c_size_t(0)
        check(lib().pamd_table_image(self._h, buf.ctypes.data, buf.size, C.byref(used)))
        return buf[:used.value].copy()

    def last_kernel_ms(self) -> float:
        return float(lib().pamd_last_kernel_ms(self._h))

    def last_launch_info(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(lib().pamd_last_launch_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"workgroups": a.value, "waves_per_workgroup": b.value, "lds_bytes": c.value}

    def assemble_host(self, fwd, rev, flen, rlen, seq_stride=None, capacity=None):
        """numpy in, numpy out, through pamd_assemble_host (upload, kernel, download)."""
        fwd = np.ascontiguousarray(fwd, dtype=np.uint8)
        rev = np.ascontiguousarray(rev, dtype=np.uint8)
        fl = np.ascontiguousarray(flen, dtype=np.uint16)
        rl = np.ascontiguousarray(rlen, dtype=np.uint16)
        n = fwd.shape[0]
        if seq_stride is None:
            seq_stride = ((min(900, fwd.shape[1] + rev.shape[1]) + 15) // 16) * 16
        cap = n * seq_stride if seq_stride else (capacity or n * 912 + 16384 * 4096)
        res = np.zeros(n, dtype=RESULT_DTYPE)
        seq_nt = np.zeros(cap, dtype=np.uint8)
        seq_p = np.zeros(cap, dtype=np.float64)
        ctr = np.zeros(NUM_COUNTERS, dtype=np.int64)
        cur = np.zeros(1, dtype=np.uint64)
        cb = Batch()
        cb.n_pairs = n
        cb.fwd, cb.fwd_stride = fwd.ctypes.data, fwd.shape[1]
        cb.rev, cb.rev_stride = rev.ctypes.data, rev.shape[1]
        cb.flen, cb.rlen = fl.ctypes.data, rl.ctypes.data
        cb.results = res.ctypes.data
        cb.seq_nt, cb.seq_p = seq_nt.ctypes.data, seq_p.ctypes.data
        cb.seq_stride, cb.seq_capacity = seq_stride, cap
        cb.seq_cursor = cur.ctypes.data
        cb.counters = ctr.ctypes.data
        check(lib().pamd_assemble_host(self._h, C.byref(cb)))
        return res, seq_nt, seq_p, counters_to_dict(ctr)
