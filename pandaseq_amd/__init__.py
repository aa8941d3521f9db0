"""pandaseq_amd -- MI355X-native overlap assembly (the PANDAseq hot path) behind a C ABI.

Only what the path needs: csrc/ (HIP kernels + C ABI), the ctypes binding, the host-side
mirror of the reference's assembler interface, the synthetic-pair generator and the STAT
counter all-reduce.
"""
from ._lib import ALGO, STATUS, NUM_COUNTERS, RESULT_DTYPE, PamdError, LibraryMissing  # noqa: F401
from .assembler import Assembler, DeviceBatch, upload_batch, alloc_outputs, counters_to_dict, row_checksums  # noqa: F401
