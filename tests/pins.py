"""What of the reference's own text can be executed here, as a library (oracle/_ref/librefmacros.so, built by
`make -C oracle ref` from oracle/ref_macros.c, which only #includes the reference's misc.h / prob.h / pandaseq-nt.h): the
k-mer iterator macros, PHREDCLAMP, PANDA_NT_IS_N, PANDA_NT_IS_DEGN.  Test infrastructure."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "librefmacros.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(PATH):
            return None
        L = C.CDLL(PATH)
        for f in (L.ref_kmers_forward, L.ref_kmers_reverse):
            f.restype = C.c_size_t
            f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.ref_kmer_len.restype = C.c_size_t
        _lib = L
    return _lib


def kmers(read, length, reverse):
    """(k-mer, position) pairs the reference's FOREACH_KMER / FOREACH_KMER_REVERSE emit for one read (panda_qual rows)."""
    L = lib()
    read = np.ascontiguousarray(read[:max(int(length), 1)], dtype=np.uint8)
    k = np.zeros(512, dtype=np.uint16)
    p = np.zeros(512, dtype=np.uint16)
    n = (L.ref_kmers_reverse if reverse else L.ref_kmers_forward)(read.ctypes.data, int(length), k.ctypes.data, p.ctypes.data)
    return k[:n].copy(), p[:n].copy()


def examined(fwd, rev, flen, rlen, minoverlap=2):
    """overlaps_examined of one pair as assembler.c:84-133 counts it, with the k-mers taken from the reference's macros:
    the first two forward positions of every k-mer (kmerseen[2k], [2k+1], assembler.c:94-99), one candidate bit per
    (reverse k-mer, kept forward position) inside [0, size) (assembler.c:104-110), every overlap if none
    (ALL_BITS_IF_NONE).  None when align() does not get that far (assembler.c:73-82)."""
    flen, rlen = int(flen), int(rlen)
    if flen < 2 or rlen < 2 or min(flen, rlen) < minoverlap or minoverlap >= flen or minoverlap >= rlen:
        return None
    maxov = min(flen, rlen)
    size = maxov - minoverlap + 1
    kf, pf = kmers(fwd, flen, False)
    seen = {}
    for k, p in zip(kf.tolist(), pf.tolist()):
        lst = seen.setdefault(k, [])
        if len(lst) < 2:
            lst.append(p)
    kr, pr = kmers(rev, rlen, True)
    bits = set()
    for k, p in zip(kr.tolist(), pr.tolist()):
        for fp in seen.get(k, ()):
            idx = flen + rlen - p - fp - minoverlap - 1
            if 0 <= idx < size:
                bits.add(idx)
    return len(bits) if bits else size
