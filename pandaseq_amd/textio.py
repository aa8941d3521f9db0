"""Host-side mirror of the reference's FASTQ reader and FASTA / FASTQ writers for the device path.

``FastqParser`` stands where ``panda_create_fastq_reader`` (fastq.c:200-237) stands in the reference:
same knobs (PHRED offset ``qualmin``, tag ``policy``), same stop conditions (the PandaCode the reader
logs before it gives up), but a whole text chunk per call instead of one pair per callback.
``Formatter`` stands where ``panda_output_fasta`` / ``panda_output_fastq`` (output.c:85-126) stand.
Both drive HIP kernels through the C ABI of ``include/pandaseq_amd_io.h``; PyTorch only provides the
device buffers.  There is no CPU path here.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import check, lib

TAG_PRESENT, TAG_ABSENT, TAG_OPTIONAL = 0, 1, 2
IDFMT = ["UNKNOWN", "SRA", "CASAVA_1_4", "CASAVA_1_7", "EBI_SRA", "CASAVA_CONVERTED"]
FQ_STOP = ["END", "BADID_FWD", "BADID_REV", "NOTPAIRED", "BADNT", "READLEN", "BADSEQ", "NOQUAL", "EOF",
           "LONG_HEADER", "ROW_OVERFLOW"]
OUT_FASTA, OUT_FASTQ = 0, 1

# pamd_seqid, 48 bytes
SEQID_DTYPE = np.dtype([
    ("hdr_off", "<u8"), ("lane", "<i4"), ("tile", "<i4"), ("x", "<i4"), ("y", "<i4"), ("sra", "<i4"),
    ("instr_len", "<u2"), ("run_off", "<u2"), ("run_len", "<u2"), ("fc_off", "<u2"), ("fc_len", "<u2"),
    ("tag_off", "<u2"), ("tag_len", "<u2"), ("format", "u1"), ("dir", "u1"), ("flag", "u1"), ("reserved", "u1", (3,)),
])
assert SEQID_DTYPE.itemsize == 48

# every symbol include/pandaseq_amd_io.h declares
SYMBOLS = [
    "pamd_fastq_parser_create", "pamd_fastq_parser_destroy", "pamd_fastq_parse_device", "pamd_fastq_last_kernel_ms",
    "pamd_formatter_create", "pamd_formatter_destroy", "pamd_format_device", "pamd_format_last_kernel_ms", "pamd_pipe_run", "pamd_pipe_run_multi", "pamd_stat_block",
]


class FastqStatus(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_good", C.c_uint64), ("n_dropped", C.c_uint64),
                ("fwd_consumed", C.c_uint64), ("rev_consumed", C.c_uint64),
                ("stop_code", C.c_int32), ("stop_in_reverse", C.c_int32), ("seen_under_64", C.c_int32),
                ("max_flen", C.c_uint32), ("max_rlen", C.c_uint32), ("reserved", C.c_uint32)]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "reserved"}
        d["stop"] = FQ_STOP[self.stop_code]
        return d


_bound = False


def _io():
    global _bound
    L = lib()
    if not _bound:
        L.pamd_fastq_parser_create.restype = C.c_void_p
        L.pamd_fastq_parser_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.pamd_fastq_parser_destroy.argtypes = [C.c_void_p]
        L.pamd_fastq_parse_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                                              C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.POINTER(FastqStatus), C.c_void_p]
        L.pamd_fastq_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.pamd_formatter_create.restype = C.c_void_p
        L.pamd_formatter_create.argtypes = [C.c_int]
        L.pamd_formatter_destroy.argtypes = [C.c_void_p]
        L.pamd_format_device.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t,
                                         C.POINTER(C.c_size_t), C.c_void_p]
        L.pamd_format_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        _bound = True
    return L


def _torch():
    import torch
    return torch


def text_to_device(data: bytes, device="cuda:0"):
    """FASTQ text -> uint8 tensor in HBM (torch allocations are at least 256-byte aligned)."""
    torch = _torch()
    return torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)


class FastqParser:
    """Paired FASTQ text (in HBM) -> rows of panda_qual + parsed headers, on the GPU."""

    def __init__(self, device: int = 0, qualmin: int = 33, policy: int = TAG_OPTIONAL):
        self._h = _io().pamd_fastq_parser_create(int(device), int(qualmin), int(policy))
        if not self._h:
            raise _lib.PamdError(lib().pamd_last_error().decode())
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            lib().pamd_fastq_parser_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def parse(self, fwd_text, rev_text, max_pairs: int, stride: int = 456, out=None, stream=None):
        """fwd_text / rev_text: uint8 device tensors.  Returns (rows, status): rows is a dict with
        fwd / rev uint8 [max_pairs, stride, 2], flen / rlen int16-stored uint16 [max_pairs] and ids
        uint8 [max_pairs, 48] (pamd_seqid records); rows [0, status.n_good) are valid."""
        torch = _torch()
        dev = fwd_text.device
        st = stream if stream is not None else torch.cuda.current_stream(dev)
        if out is None:
            out = dict(
                fwd=torch.empty((max_pairs, stride, 2), dtype=torch.uint8, device=dev),
                rev=torch.empty((max_pairs, stride, 2), dtype=torch.uint8, device=dev),
                flen=torch.zeros(max_pairs, dtype=torch.int16, device=dev),
                rlen=torch.zeros(max_pairs, dtype=torch.int16, device=dev),
                ids=torch.zeros((max_pairs, 48), dtype=torch.uint8, device=dev),
            )
        status = FastqStatus()
        check(_io().pamd_fastq_parse_device(
            self._h, fwd_text.data_ptr(), fwd_text.numel(), rev_text.data_ptr(), rev_text.numel(), int(max_pairs),
            out["fwd"].data_ptr(), out["fwd"].shape[1], out["rev"].data_ptr(), out["rev"].shape[1],
            out["flen"].data_ptr(), out["rlen"].data_ptr(), out["ids"].data_ptr(), C.byref(status),
            C.c_void_p(st.cuda_stream)))
        return out, status

    def last_kernel_ms(self):
        a, b = C.c_float(), C.c_float()
        check(_io().pamd_fastq_last_kernel_ms(self._h, C.byref(a), C.byref(b)))
        return {"index_ms": a.value, "parse_ms": b.value}


def seqid_fields(rec, hdr_text: bytes):
    """One SEQID_DTYPE record + the forward text -> the fields of panda_seq_identifier
    (instrument, run, flowcell, lane, tile, x, y, tag)."""
    h = int(rec["hdr_off"])

    def span(off, ln):
        return hdr_text[h + int(off):h + int(off) + int(ln)].decode("latin1")

    instr = span(0, rec["instr_len"])
    if int(rec["sra"]) >= 0:
        instr += str(int(rec["sra"]))
    return (instr, span(rec["run_off"], rec["run_len"]), span(rec["fc_off"], rec["fc_len"]),
            int(rec["lane"]), int(rec["tile"]), int(rec["x"]), int(rec["y"]), span(rec["tag_off"], rec["tag_len"]))


class Formatter:
    """Accepted pairs of an assembled batch -> FASTA / FASTQ text, on the GPU."""

    def __init__(self, device: int = 0):
        self._h = _io().pamd_formatter_create(int(device))
        if not self._h:
            raise _lib.PamdError(lib().pamd_last_error().decode())
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            lib().pamd_formatter_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def format(self, batch, ids, hdr_text, fmt=OUT_FASTA, out=None, capacity=None, stream=None):
        """batch: an assembled DeviceBatch; ids: uint8 [n, 48] device tensor; hdr_text: the forward FASTQ
        text the ids point into (device).  Returns (out tensor, n_bytes, fixups) where fixups is a list of
        (byte offset, pair index) whose eight-character quality field the host must re-print."""
        torch = _torch()
        dev = batch.results.device
        st = stream if stream is not None else torch.cuda.current_stream(dev)
        if out is None:
            if capacity is None:
                capacity = batch.n * (2 * 912 + 480)
            out = torch.empty(int(capacity), dtype=torch.uint8, device=dev)
        nbytes, nfix = C.c_size_t(0), C.c_size_t(0)
        fix = np.zeros((1024, 2), dtype=np.uint64)
        check(_io().pamd_format_device(
            self._h, int(fmt), batch.n, batch.results.data_ptr(), batch.seq_nt.data_ptr(), batch.seq_p.data_ptr(),
            ids.data_ptr(), hdr_text.data_ptr(), out.data_ptr(), out.numel(), C.byref(nbytes),
            fix.ctypes.data, fix.shape[0], C.byref(nfix), C.c_void_p(st.cuda_stream)))
        return out, int(nbytes.value), [(int(a), int(b)) for a, b in fix[:nfix.value]]

    def last_kernel_ms(self):
        a = C.c_float()
        check(_io().pamd_format_last_kernel_ms(self._h, C.byref(a)))
        return a.value


def apply_fixups(text: bytearray, fixups, qualities):
    """Re-print the quality fields the device flagged, exactly as output.c:93 does ("%f" of exp(quality))."""
    for off, pair in fixups:
        text[off:off + 8] = ("%f" % math.exp(float(qualities[pair]))).encode()
    return text
