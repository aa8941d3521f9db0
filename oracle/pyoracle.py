"""ctypes binding for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (pandaseq_amd) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_LEN = 450
NQ = 47
MAX_MODULES = 8

ALGO = {"simple_bayesian": 0, "pear": 1, "rdp_mle": 2, "uparse": 3, "flash": 4, "ea_util": 5, "stitch": 6}
STATUS = ["OK", "BADR", "NOFP", "NORP", "NOALGN", "LOWQ", "REJECT"]
MOD_MIN_PHRED, MOD_MIN_OVERLAPBITS = 1, 2


class Module(C.Structure):
    _fields_ = [("kind", C.c_int), ("phred", C.c_int), ("nats", C.c_double)]


class Config(C.Structure):
    _fields_ = [
        ("algo", C.c_int),
        ("sb_q", C.c_double), ("sb_pmatch", C.c_double), ("sb_pmismatch", C.c_double),
        ("pear_random_base", C.c_double),
        ("threshold", C.c_double),
        ("minoverlap", C.c_int), ("maxoverlap", C.c_int), ("num_kmers", C.c_int),
        ("fprimer", C.c_uint8 * MAX_LEN), ("fprimer_len", C.c_int),
        ("rprimer", C.c_uint8 * MAX_LEN), ("rprimer_len", C.c_int),
        ("ftrim", C.c_int), ("rtrim", C.c_int),
        ("primer_penalty", C.c_double),
        ("n_modules", C.c_int),
        ("modules", Module * MAX_MODULES),
        ("up_q", C.c_double), ("up_pmatch", C.c_double), ("up_pmismatch", C.c_double),
        ("post_primers", C.c_int),
    ]


class Tables(C.Structure):
    _fields_ = [
        ("qual_nn", C.c_double),
        ("score", C.c_double * NQ),
        ("score_err", C.c_double * NQ),
        ("match_sb", (C.c_double * NQ) * NQ),
        ("mismatch_sb", (C.c_double * NQ) * NQ),
        ("match_pear", (C.c_double * NQ) * NQ),
        ("mismatch_pear", (C.c_double * NQ) * NQ),
        ("mismatch_rdp", (C.c_double * NQ) * NQ),
        ("mismatch_rdp_asm", (C.c_double * NQ) * NQ),
        ("match_uparse", (C.c_double * NQ) * NQ),
        ("mismatch_uparse", (C.c_double * NQ) * NQ),
    ]


RESULT_DTYPE = np.dtype([
    ("status", "<i4"), ("reject_module", "<i4"), ("overlap", "<i4"), ("examined", "<i4"),
    ("mismatches", "<i4"), ("degenerates", "<i4"), ("fwd_offset", "<i4"), ("rev_offset", "<i4"),
    ("seq_len", "<i4"), ("slow", "<i4"), ("quality", "<f8"), ("est_prob", "<f8"),
])

COUNTER_DTYPE = np.dtype([
    ("count", "<i8"), ("badr", "<i8"), ("nofp", "<i8"), ("norp", "<i8"), ("noalgn", "<i8"),
    ("lowq", "<i8"), ("ok", "<i8"), ("slow", "<i8"), ("rejected", "<i8", (MAX_MODULES,)),
    ("overlapcount", "<i8", (2 * MAX_LEN,)), ("longest_overlap", "<i8"),
])

_lib = None


def build(force=False):
    """Compile liboracle.so (and, when /root/reference is present, the reference's own table
    generator into oracle/_ref/).  Building the checker is not using it."""
    so = os.path.join(HERE, "liboracle.so")
    srcs = [os.path.join(HERE, f) for f in ("panda_oracle.c", "panda_oracle.h", "panda_oracle_io.c", "panda_oracle_io.h")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"])
    if os.path.isdir("/root/reference") and not os.path.exists(os.path.join(HERE, "_ref", "libreftable.so")):
        subprocess.check_call(["make", "-C", HERE, "ref"])
    return so


def lib():
    global _lib
    if _lib is None:
        so = build()
        L = C.CDLL(so)
        L.po_build_tables.argtypes = [C.POINTER(Tables)]
        L.po_config_default.argtypes = [C.POINTER(Config)]
        L.po_config_set_sb_q.argtypes = [C.POINTER(Config), C.c_double]
        L.po_config_set_uparse_q.argtypes = [C.POINTER(Config), C.c_double]
        L.po_overlap_probability.restype = C.c_double
        L.po_overlap_probability.argtypes = [C.POINTER(Config), C.POINTER(Tables), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t]
        L.po_match_probability.restype = C.c_double
        L.po_match_probability.argtypes = [C.POINTER(Config), C.POINTER(Tables), C.c_int, C.c_int8, C.c_int8]
        L.po_compute_offset_qual.restype = C.c_size_t
        L.po_compute_offset_qual.argtypes = [C.POINTER(Tables), C.c_double, C.c_double, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.po_result_phred.restype = C.c_int
        L.po_result_phred.argtypes = [C.POINTER(Tables), C.c_double]
        L.po_kmers_forward.restype = C.c_size_t
        L.po_kmers_forward.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.po_kmers_reverse.restype = C.c_size_t
        L.po_kmers_reverse.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.po_phredclamp.argtypes = [C.c_int]
        L.po_nt_is_degenerate.argtypes = [C.c_uint]
        batch_args = [C.POINTER(Config), C.POINTER(Tables), C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.po_assemble_batch.argtypes = batch_args
        L.po_assemble_batch_mt.argtypes = batch_args + [C.c_int]
        _lib = L
    return _lib


_tables = None


def tables():
    global _tables
    if _tables is None:
        _tables = Tables()
        lib().po_build_tables(C.byref(_tables))
    return _tables


def tables_np():
    t = tables()
    out = {"qual_nn": t.qual_nn}
    for name in ("score", "score_err", "match_sb", "mismatch_sb", "match_pear", "mismatch_pear", "mismatch_rdp", "mismatch_rdp_asm",
                 "match_uparse", "mismatch_uparse"):
        out[name] = np.ctypeslib.as_array(getattr(t, name)).copy()
    return out


def make_config(algo="simple_bayesian", q=None, pear_random_base_log=None, threshold=None, minoverlap=None,
                maxoverlap=None, fprimer=None, rprimer=None, ftrim=0, rtrim=0, primer_penalty=0.0, modules=(), primers_after=False, num_kmers=None):
    """Mirror of the flag -> setter mapping of args_assembler.c:26-368 for the in-scope knobs.
    fprimer / rprimer are sequences of nt codes as the assembler stores them (the -q primer is
    already complemented, args_assembler.c:222)."""
    import math
    c = Config()
    lib().po_config_default(C.byref(c))
    c.algo = ALGO[algo]
    if q is not None:
        if algo == "uparse":
            lib().po_config_set_uparse_q(C.byref(c), q)
        else:
            lib().po_config_set_sb_q(C.byref(c), q)
    if pear_random_base_log is not None:
        c.pear_random_base = pear_random_base_log
    if threshold is not None and 0 < threshold < 1:  # assembler_support.c:391-397
        c.threshold = math.log(threshold)
    if minoverlap is not None and 1 < minoverlap < 2 * MAX_LEN:  # assembler_support.c:276-282
        c.minoverlap = minoverlap
    if maxoverlap is not None and 0 <= maxoverlap < 2 * MAX_LEN:
        c.maxoverlap = maxoverlap
    c.ftrim, c.rtrim = ftrim, rtrim
    if fprimer is not None and len(fprimer) < MAX_LEN:
        for i, v in enumerate(fprimer):
            c.fprimer[i] = int(v)
        c.fprimer_len = len(fprimer)
        c.ftrim = 0
    if rprimer is not None and len(rprimer) < MAX_LEN:
        for i, v in enumerate(rprimer):
            c.rprimer[i] = int(v)
        c.rprimer_len = len(rprimer)
        c.rtrim = 0
    if 0 <= primer_penalty < 1:
        c.primer_penalty = primer_penalty
    c.post_primers = 1 if primers_after else 0  # `-a`, args_assembler.c
    if num_kmers is not None:  # panda_assembler_new_kmer (assembler_support.c:80), args.c:384
        c.num_kmers = int(num_kmers)
    c.n_modules = len(modules)
    for i, (kind, val) in enumerate(modules):
        if kind == "min_phred":
            c.modules[i].kind = MOD_MIN_PHRED
            c.modules[i].phred = int(val)
        elif kind == "min_overlapbits":
            c.modules[i].kind = MOD_MIN_OVERLAPBITS
            c.modules[i].nats = float(val) * math.log(2.0)  # plugin_min_overlapbits.c:35, M_LN2
        else:
            raise ValueError(kind)
    return c


def assemble_batch(cfg, fwd, rev, flen, rlen, want_seq=True, threads=1):
    """fwd / rev: uint8 arrays [n, stride, 2] (nt, qual) -- the panda_qual layout; flen / rlen uint16 [n].
    Returns (results, seq_nt, seq_p, counters)."""
    fwd = np.ascontiguousarray(fwd, dtype=np.uint8)
    rev = np.ascontiguousarray(rev, dtype=np.uint8)
    flen = np.ascontiguousarray(flen, dtype=np.uint16)
    rlen = np.ascontiguousarray(rlen, dtype=np.uint16)
    n = fwd.shape[0]
    assert fwd.ndim == 3 and fwd.shape[2] == 2 and rev.shape[0] == n
    res = np.zeros(n, dtype=RESULT_DTYPE)
    ctr = np.zeros(1, dtype=COUNTER_DTYPE)
    ostride = 2 * MAX_LEN
    if want_seq:
        seq_nt = np.zeros((n, ostride), dtype=np.uint8)
        seq_p = np.zeros((n, ostride), dtype=np.float64)
        nt_p, p_p = seq_nt.ctypes.data, seq_p.ctypes.data
    else:
        seq_nt = seq_p = None
        nt_p = p_p = None
    args = [C.byref(cfg), C.byref(tables()), n, fwd.ctypes.data, fwd.shape[1], rev.ctypes.data, rev.shape[1],
            flen.ctypes.data, rlen.ctypes.data, res.ctypes.data, nt_p, p_p, ostride, ctr.ctypes.data]
    if threads > 1:
        lib().po_assemble_batch_mt(*args, threads)
    else:
        lib().po_assemble_batch(*args)
    if threads <= 1:
        # the single-thread entry point leaves longest_overlap as a running max already
        pass
    return res, seq_nt, seq_p, ctr[0]


def bench_mt(cfg, fwd, rev, flen, rlen, threads, pin=True):
    """bench.py's cpu_baseline leg (po_bench_mt): `threads` pinned pthreads, each over its own, self-touched copy of its
    slice; returns (seconds from all-ready to all-done, results, counters)."""
    fwd = np.ascontiguousarray(fwd, dtype=np.uint8)
    rev = np.ascontiguousarray(rev, dtype=np.uint8)
    flen = np.ascontiguousarray(flen, dtype=np.uint16)
    rlen = np.ascontiguousarray(rlen, dtype=np.uint16)
    n = fwd.shape[0]
    res = np.zeros(n, dtype=RESULT_DTYPE)
    ctr = np.zeros(1, dtype=COUNTER_DTYPE)
    L = lib()
    L.po_bench_mt.restype = C.c_double
    L.po_bench_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    secs = L.po_bench_mt(C.byref(cfg), C.byref(tables()), n, fwd.ctypes.data, fwd.shape[1], rev.ctypes.data, rev.shape[1],
                         flen.ctypes.data, rlen.ctypes.data, res.ctypes.data, ctr.ctypes.data, int(threads), 1 if pin else 0)
    if secs < 0:
        raise RuntimeError("po_bench_mt failed")
    return secs, res, ctr[0]


def row_checksums(cfg, fwd, rev, flen, rlen, threads=1):
    """(results, cs_nt, cs_p): per accepted pair, sum (j+1) nt_j and sum (j+1) bits(p_j) over its merged row, both in wrapping
    64-bit integer arithmetic (exact in any order of addition); 0 where a pair has no row.  po_row_checksums_mt."""
    fwd = np.ascontiguousarray(fwd, dtype=np.uint8)
    rev = np.ascontiguousarray(rev, dtype=np.uint8)
    flen = np.ascontiguousarray(flen, dtype=np.uint16)
    rlen = np.ascontiguousarray(rlen, dtype=np.uint16)
    n = fwd.shape[0]
    res = np.zeros(n, dtype=RESULT_DTYPE)
    cs_nt = np.zeros(n, dtype=np.uint64)
    cs_p = np.zeros(n, dtype=np.uint64)
    L = lib()
    L.po_row_checksums_mt.restype = None
    L.po_row_checksums_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.po_row_checksums_mt(C.byref(cfg), C.byref(tables()), n, fwd.ctypes.data, fwd.shape[1], rev.ctypes.data, rev.shape[1],
                          flen.ctypes.data, rlen.ctypes.data, res.ctypes.data, cs_nt.ctypes.data, cs_p.ctypes.data, int(threads))
    return res, cs_nt, cs_p


def ref_tables_np():
    """Tables emitted by the REFERENCE's own mktable (oracle/_ref/libreftable.so), or None when
    that build is absent (it can only be produced where /root/reference exists)."""
    so = os.path.join(HERE, "_ref", "libreftable.so")
    if not os.path.exists(so):
        return None
    L = C.CDLL(so)
    out = {"qual_nn": C.c_double.in_dll(L, "ref_qual_nn_simple_bayesian").value}
    names = {
        "score": "qual_score", "score_err": "qual_score_err",
        "match_sb": "qual_match_simple_bayesian", "mismatch_sb": "qual_mismatch_simple_bayesian",
        "match_pear": "qual_match_pear", "mismatch_pear": "qual_mismatch_pear",
        "mismatch_rdp": "qual_mismatch_rdp_mle", "mismatch_rdp_asm": "qual_mismatch_assembled_rdp_mle",
        "match_uparse": "qual_match_uparse", "mismatch_uparse": "qual_mismatch_uparse",
    }
    for k, sym in names.items():
        if k in ("score", "score_err"):
            arr = (C.c_double * NQ).in_dll(L, sym)
        else:
            arr = ((C.c_double * NQ) * NQ).in_dll(L, sym)
        out[k] = np.ctypeslib.as_array(arr).copy()
    return out


# ---- FASTQ ingest and FASTA / FASTQ formatting (panda_oracle_io.h) --------------------------------
TAG_PRESENT, TAG_ABSENT, TAG_OPTIONAL = 0, 1, 2
IDFMT = ["UNKNOWN", "SRA", "CASAVA_1_4", "CASAVA_1_7", "EBI_SRA", "CASAVA_CONVERTED"]
FQ_STOP = ["END", "BADID_FWD", "BADID_REV", "NOTPAIRED", "BADNT", "READLEN", "BADSEQ", "NOQUAL", "EOF"]


class SeqId(C.Structure):
    _fields_ = [("instrument", C.c_char * 100), ("run", C.c_char * 100), ("flowcell", C.c_char * 100),
                ("lane", C.c_int), ("tile", C.c_int), ("x", C.c_int), ("y", C.c_int), ("tag", C.c_char * 50)]

    def fields(self):
        return (self.instrument.decode("latin1"), self.run.decode("latin1"), self.flowcell.decode("latin1"),
                self.lane, self.tile, self.x, self.y, self.tag.decode("latin1"))


class FastqStatus(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("n_records", C.c_uint64), ("n_dropped", C.c_uint64),
                ("stop_code", C.c_int32), ("stop_in_reverse", C.c_int32), ("stop_record", C.c_uint64),
                ("seen_under_64", C.c_int32), ("non_empty", C.c_int32)]


_io_ready = False


def _io():
    global _io_ready
    L = lib()
    if not _io_ready:
        L.po_seqid_parse.restype = C.c_int
        L.po_seqid_parse.argtypes = [C.POINTER(SeqId), C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
        L.po_seqid_equal.restype = C.c_int
        L.po_seqid_equal.argtypes = [C.POINTER(SeqId), C.POINTER(SeqId)]
        L.po_seqid_print.restype = C.c_int
        L.po_seqid_print.argtypes = [C.POINTER(SeqId), C.c_char_p]
        L.po_fastq_read.restype = None
        L.po_fastq_read.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_size_t,
                                    C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.POINTER(FastqStatus)]
        L.po_output_fasta.restype = C.c_size_t
        L.po_output_fasta.argtypes = [C.POINTER(SeqId), C.c_double, C.c_void_p, C.c_size_t, C.c_char_p]
        L.po_output_fastq.restype = C.c_size_t
        L.po_output_fastq.argtypes = [C.POINTER(Tables), C.POINTER(SeqId), C.c_double, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p]
        L.po_nt_from_ascii.restype = C.c_uint8
        L.po_nt_from_ascii.argtypes = [C.c_char]
        L.po_nt_from_ascii_complement.restype = C.c_uint8
        L.po_nt_from_ascii_complement.argtypes = [C.c_char]
        L.po_nt_to_ascii.restype = C.c_char
        L.po_nt_to_ascii.argtypes = [C.c_uint8]
        _io_ready = True
    return L


def seqid_parse(header, policy=TAG_OPTIONAL):
    """panda_seqid_parse_fail on the text after the '@'. Returns (direction, format, SeqId, end_offset)."""
    if isinstance(header, str):
        header = header.encode("latin1")
    sid, fmt, end = SeqId(), C.c_int(0), C.c_size_t(0)
    d = _io().po_seqid_parse(C.byref(sid), header, policy, C.byref(fmt), C.byref(end))
    return d, fmt.value, sid, end.value


def seqid_str(sid):
    buf = C.create_string_buffer(512)
    n = _io().po_seqid_print(C.byref(sid), buf)
    return buf.raw[:n]


def fastq_read(fwd_text, rev_text, qualmin=33, policy=TAG_OPTIONAL, max_pairs=None, stride=456):
    """The paired reader over two FASTQ texts (bytes). Returns (fwd, rev, flen, rlen, ids, status):
    fwd / rev uint8 [n, stride, 2], ids a ctypes array of SeqId."""
    n_max = fwd_text.count(b"\n") // 4 + 1 if max_pairs is None else max_pairs
    fwd = np.zeros((n_max, stride, 2), dtype=np.uint8)
    rev = np.zeros((n_max, stride, 2), dtype=np.uint8)
    flen = np.zeros(n_max, dtype=np.uint16)
    rlen = np.zeros(n_max, dtype=np.uint16)
    ids = (SeqId * n_max)()
    st = FastqStatus()
    _io().po_fastq_read(fwd_text, len(fwd_text), rev_text, len(rev_text), qualmin, policy, n_max,
                        fwd.ctypes.data, stride, rev.ctypes.data, stride, flen.ctypes.data, rlen.ctypes.data,
                        C.cast(ids, C.c_void_p), C.byref(st))
    n = st.n_pairs
    return fwd[:n], rev[:n], flen[:n], rlen[:n], ids, st


def output_fasta(sid, quality, seq_nt):
    seq_nt = np.ascontiguousarray(seq_nt, dtype=np.uint8)
    buf = C.create_string_buffer(1024 + 2 * len(seq_nt))
    n = _io().po_output_fasta(C.byref(sid), quality, seq_nt.ctypes.data, len(seq_nt), buf)
    return buf.raw[:n]


def output_fastq(sid, quality, seq_nt, seq_p):
    seq_nt = np.ascontiguousarray(seq_nt, dtype=np.uint8)
    seq_p = np.ascontiguousarray(seq_p, dtype=np.float64)
    buf = C.create_string_buffer(1024 + 2 * len(seq_nt))
    n = _io().po_output_fastq(C.byref(tables()), C.byref(sid), quality, seq_nt.ctypes.data, seq_p.ctypes.data, len(seq_nt), buf)
    return buf.raw[:n]
