"""ctypes binding of libpandaseq_amd.so (the C ABI in include/pandaseq_amd.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C pandaseq_amd/csrc`` and is
the ONLY compute path of this package: if it is missing, loading fails loudly.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpandaseq_amd.so")

MAX_LEN = 450
MAX_MODULES = 8
NUM_COUNTERS = 8 + MAX_MODULES + 2 * MAX_LEN
TABLE_DOUBLES = 1 + 47 + 47 + 8 * 47 * 47

ALGO = {"simple_bayesian": 0, "pear": 1, "rdp_mle": 2, "uparse": 3, "flash": 4, "ea_util": 5, "stitch": 6}
STATUS = ["OK", "BADR", "NOFP", "NORP", "NOALGN", "LOWQ", "REJECT", "NOSPACE"]
CTR = {"count": 0, "badr": 1, "nofp": 2, "norp": 3, "noalgn": 4, "lowq": 5, "ok": 6, "slow": 7}
CTR_REJECTED0 = 8
CTR_OVERLAP0 = 8 + MAX_MODULES

# pamd_pair_result, 48 bytes
RESULT_DTYPE = np.dtype([
    ("quality", "<f8"), ("est_overlap_prob", "<f8"), ("seq_offset", "<u8"),
    ("seq_len", "<u2"), ("overlap", "<u2"), ("examined", "<u2"), ("mismatches", "<u2"),
    ("degenerates", "<u2"), ("fwd_offset", "<u2"), ("rev_offset", "<u2"),
    ("status", "u1"), ("reject_module", "i1"), ("slow", "u1"), ("reserved", "u1", (7,)),
])
assert RESULT_DTYPE.itemsize == 48


class Batch(C.Structure):
    _fields_ = [
        ("n_pairs", C.c_size_t),
        ("fwd", C.c_void_p), ("fwd_stride", C.c_size_t),
        ("rev", C.c_void_p), ("rev_stride", C.c_size_t),
        ("flen", C.c_void_p), ("rlen", C.c_void_p),
        ("results", C.c_void_p),
        ("seq_nt", C.c_void_p), ("seq_p", C.c_void_p),
        ("seq_stride", C.c_size_t), ("seq_capacity", C.c_size_t),
        ("seq_cursor", C.c_void_p), ("counters", C.c_void_p),
    ]


class SynthSpecC(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("read_len", C.c_uint32), ("tmin", C.c_uint32), ("tmax", C.c_uint32),
        ("qhi", C.c_int32), ("novaseq", C.c_int32),
        ("fprimer_len", C.c_uint32), ("rprimer_len", C.c_uint32),
        ("fprimer", C.c_uint8 * 32), ("rprimer", C.c_uint8 * 32),
    ]


class LibraryMissing(RuntimeError):
    pass


_lib = None

# every symbol include/pandaseq_amd.h declares
SYMBOLS = [
    "pamd_last_error", "pamd_device_count", "pamd_assembler_create", "pamd_assembler_destroy",
    "pamd_set_algorithm", "pamd_set_threshold", "pamd_set_minimum_overlap", "pamd_set_maximum_overlap",
    "pamd_set_forward_primer", "pamd_set_reverse_primer", "pamd_set_forward_trim", "pamd_set_reverse_trim",
    "pamd_set_primer_penalty", "pamd_set_primers_after", "pamd_add_module_min_phred", "pamd_add_module_min_overlapbits",
    "pamd_clear_modules", "pamd_set_exact_sums", "pamd_tables_get", "pamd_assemble_device", "pamd_assemble_host",
    "pamd_last_kernel_ms", "pamd_last_launch_info", "pamd_synth_generate", "pamd_debug_phase_cycles",
    "pamd_assemble_status", "pamd_set_log_threshold", "pamd_set_fast_path", "pamd_last_deferred", "pamd_last_lean_variant", "pamd_set_two_pairs_per_wave", "pamd_assemble_device_range", "pamd_encode_rows_device", "pamd_table_image",
]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PAMD_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise LibraryMissing(
            f"{path} is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C pandaseq_amd/csrc`). There is no CPU fallback for the assembly path.")
    # One HIP runtime per process: PyTorch ships its own libamdhip64 (same SONAME as /opt/rocm's).
    # Importing torch first makes the loader bind our NEEDED libamdhip64.so.7 to that copy; loading
    # ours first would pull in /opt/rocm's and torch would then bring a second runtime that sees
    # no devices ("No HIP GPUs are available").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    L.pamd_last_error.restype = C.c_char_p
    L.pamd_device_count.restype = C.c_int
    L.pamd_assembler_create.restype = C.c_void_p
    L.pamd_assembler_create.argtypes = [C.c_int]
    L.pamd_assembler_destroy.argtypes = [C.c_void_p]
    L.pamd_set_algorithm.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.pamd_set_threshold.argtypes = [C.c_void_p, C.c_double]
    L.pamd_set_log_threshold.argtypes = [C.c_void_p, C.c_double]
    L.pamd_set_minimum_overlap.argtypes = [C.c_void_p, C.c_int]
    L.pamd_set_maximum_overlap.argtypes = [C.c_void_p, C.c_int]
    L.pamd_set_forward_primer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.pamd_set_reverse_primer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.pamd_set_forward_trim.argtypes = [C.c_void_p, C.c_size_t]
    L.pamd_set_reverse_trim.argtypes = [C.c_void_p, C.c_size_t]
    L.pamd_set_primer_penalty.argtypes = [C.c_void_p, C.c_double]
    L.pamd_set_primers_after.argtypes = [C.c_void_p, C.c_int]
    L.pamd_add_module_min_phred.argtypes = [C.c_void_p, C.c_int]
    L.pamd_add_module_min_overlapbits.argtypes = [C.c_void_p, C.c_double]
    L.pamd_clear_modules.argtypes = [C.c_void_p]
    L.pamd_set_exact_sums.argtypes = [C.c_void_p, C.c_int]
    L.pamd_tables_get.argtypes = [C.c_void_p, C.c_size_t]
    L.pamd_assemble_device.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p]
    L.pamd_assemble_host.argtypes = [C.c_void_p, C.POINTER(Batch)]
    L.pamd_assemble_device_range.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_size_t, C.c_size_t, C.c_void_p]
    L.pamd_encode_rows_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.pamd_table_image.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.pamd_assemble_status.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.pamd_set_fast_path.argtypes = [C.c_void_p, C.c_int]
    L.pamd_set_two_pairs_per_wave.argtypes = [C.c_void_p, C.c_int]
    L.pamd_set_two_pairs_per_wave.restype = C.c_int
    L.pamd_last_deferred.restype = C.c_longlong
    L.pamd_last_deferred.argtypes = [C.c_void_p]
    L.pamd_last_lean_variant.restype = C.c_int
    L.pamd_last_lean_variant.argtypes = [C.c_void_p]
    L.pamd_last_kernel_ms.restype = C.c_float
    L.pamd_last_kernel_ms.argtypes = [C.c_void_p]
    L.pamd_debug_phase_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.pamd_last_launch_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.pamd_synth_generate.argtypes = [C.c_int, C.POINTER(SynthSpecC), C.c_void_p, C.c_uint64, C.c_size_t,
                                      C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = L
    return L


class PamdError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise PamdError(f"pandaseq_amd error {rc}: {lib().pamd_last_error().decode()}")


def tables():
    """The library's probability tables as a dict of numpy arrays (host-side, no GPU needed)."""
    buf = np.zeros(TABLE_DOUBLES, dtype=np.float64)
    check(lib().pamd_tables_get(buf.ctypes.data, buf.size))
    out = {"qual_nn": float(buf[0]), "score": buf[1:48].copy(), "score_err": buf[48:95].copy()}
    o = 95
    for name in ("match_sb", "mismatch_sb", "match_pear", "mismatch_pear", "mismatch_rdp", "mismatch_rdp_asm", "match_uparse", "mismatch_uparse"):
        out[name] = buf[o:o + 2209].reshape(47, 47).copy()
        o += 2209
    return out
