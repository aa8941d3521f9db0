"""Deterministic synthetic read-pair generator, host side.

There is no network, so every benchmark and parity input is synthetic.  The generator is
counter-based: every random quantity is ``mix64(pairkey + slot * C2)`` for a fixed slot number, so
the same pairs come out of this numpy implementation (small batches, CPU tests) and of the HIP
kernel ``pamd_synth_kernel`` in csrc/pandaseq_amd.hip (the 50 M-pair bench batch, generated in
HBM).  tests/test_synth_gpu.py checks the two against each other.

Output layout is the device batch layout (DESIGN.md "Data layout"): rows of ``panda_qual``
(nt byte, qual byte; pandaseq-common.h:210-219) at a fixed stride, forward read as sequenced,
reverse read complemented but NOT reversed (what fastq.c:154 + nt.c:81-114 hand the assembler).
"""
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

C1 = np.uint64(0x9E3779B97F4A7C15)
C2 = np.uint64(0xD1B54A32D192ED03)
M1 = np.uint64(0xBF58476D1CE4E5B9)
M2 = np.uint64(0x94D049BB133111EB)

S_KIND, S_TLEN, S_UNIT, S_TAIL, S_PREFIX = 0, 1, 2, 3, 5
S_TMPL, S_TMPL2, S_BASE = 64, 64 + 1024, 4096

NT_OF = np.array([1, 2, 4, 8], dtype=np.uint8)  # A C G T (pandaseq-nt.h:39-51)
QREPL = np.array([2, 8, 15, 22, 27, 33], dtype=np.int64)

_ASCII_NT = {"A": 1, "C": 2, "G": 4, "T": 8, "N": 15}
_COMP = {1: 8, 2: 4, 4: 2, 8: 1, 15: 15}


def nt_codes(s: str, complement: bool = False) -> np.ndarray:
    """ACGTN string -> nt codes (panda_nt_from_ascii / _complement, nt.c:152-160)."""
    v = [_ASCII_NT[c] for c in s.upper()]
    if complement:
        v = [_COMP[x] for x in v]
    return np.array(v, dtype=np.uint8)


def error_thresholds() -> np.ndarray:
    """u32 thresholds: a substitution happens when a uniform u32 draw is below thr[Q] (p = 10^(-Q/10))."""
    thr = np.zeros(64, dtype=np.uint32)
    for q in range(64):
        v = int((10.0 ** (-q / 10.0)) * 4294967296.0)
        thr[q] = min(v, 0xFFFFFFFF)
    return thr


@dataclass
class SynthSpec:
    seed: int
    read_len: int
    tmin: int
    tmax: int
    qhi: int = 38
    novaseq: bool = False  # binned qualities {2, 12, 23, 37}
    fprimer: Optional[str] = None  # forward primer as given to -p
    rprimer: Optional[str] = None  # reverse primer as given to -q
    stride: int = 0  # bases per row; 0 = read_len rounded up to 8

    def row_stride(self) -> int:
        return self.stride or ((self.read_len + 7) // 8) * 8


def config_spec(index: int) -> SynthSpec:
    """The five BASELINE.json configs as generator specs."""
    seed = 0x5EED0000 + index
    if index == 0:
        return SynthSpec(seed, 150, 260, 260, qhi=40)
    if index in (1, 3):
        return SynthSpec(seed, 250, 300, 460, qhi=38)
    if index == 2:
        return SynthSpec(seed, 300, 420, 480, qhi=37, fprimer="CCTACGGGAGGCAGCAG", rprimer="ATTACCGCGGCTGCTGG")
    if index == 4:
        return SynthSpec(seed, 150, 180, 280, novaseq=True)
    raise ValueError(index)


def _mix64(x):
    x = x ^ (x >> np.uint64(30))
    x = x * M1
    x = x ^ (x >> np.uint64(27))
    x = x * M2
    x = x ^ (x >> np.uint64(31))
    return x


def _draw(pairkey, slot):
    return _mix64(pairkey + np.asarray(slot, dtype=np.uint64) * C2)


def generate_numpy(spec: SynthSpec, start: int, n: int):
    """Pairs [start, start+n) -> (fwd u8[n,stride,2], rev u8[n,stride,2], flen u16[n], rlen u16[n])."""
    L = spec.read_len
    stride = spec.row_stride()
    thr = error_thresholds().astype(np.uint64)
    with np.errstate(over="ignore"):
        pair = np.arange(start, start + n, dtype=np.uint64)
        key = _mix64(np.uint64(spec.seed) + pair * C1)[:, None]  # [n,1]
        kind_u = _draw(key, S_KIND)[:, 0] % np.uint64(100)
        kind = np.where(kind_u < 97, 0, np.where(kind_u < 98, 1, 2))
        tlen = (np.uint64(spec.tmin) + _draw(key, S_TLEN)[:, 0] % np.uint64(spec.tmax - spec.tmin + 1)).astype(np.int64)
        unit = (2 + _draw(key, S_UNIT)[:, 0] % np.uint64(11)).astype(np.int64)

        pos = np.arange(L, dtype=np.int64)[None, :]  # [1,L]
        # template coordinate of each read base
        tf = np.broadcast_to(pos, (n, L))
        tr = tlen[:, None] - 1 - pos

        def template_base(t, second):
            """base index 0..3 at template coordinate t (array [n,L])."""
            tt = np.where((kind == 1)[:, None], t % unit[:, None], t)
            slot = (S_TMPL2 if second else S_TMPL) + tt
            b = (_draw(key, slot) & np.uint64(3)).astype(np.int64)
            return b

        bf = template_base(tf, False)
        br = np.where((kind == 2)[:, None], template_base(np.broadcast_to(pos, (n, L)), True), template_base(tr, False))

        if spec.fprimer is not None:
            fp = np.array(["ACGT".index(c) for c in spec.fprimer], dtype=np.int64)
            rp = np.array(["ACGT".index(c) for c in spec.rprimer], dtype=np.int64)
            pu = _draw(key, S_PREFIX)[:, 0]
            k = np.where(pu % np.uint64(10) == 0, 1 + ((pu >> np.uint64(32)) % np.uint64(3)).astype(np.int64), 0)
            k = np.where(kind == 0, k, 0)
            prim = (kind == 0)[:, None]
            # forward primer at template [k, k+|fp|)
            off = tf - k[:, None]
            inf = prim & (off >= 0) & (off < len(fp))
            bf = np.where(inf, fp[np.clip(off, 0, len(fp) - 1)], bf)
            # template ends with revcomp(rp): T[tlen-1-j] = comp(rp[j]); the stored reverse base j is T[tlen-1-j]
            inr = prim & (pos < len(rp))
            br = np.where(inr, 3 - rp[np.clip(np.broadcast_to(pos, (n, L)), 0, len(rp) - 1)], br)

        def read_quals_and_errors(b, r):
            base_slot = S_BASE + r * 4096 + pos * 4
            u0 = _draw(key, base_slot + 0)
            u1 = _draw(key, base_slot + 1)
            u2 = _draw(key, base_slot + 2)
            if spec.novaseq:
                v = (u0 % np.uint64(100)).astype(np.int64)
                q = np.where(v < 85, 37, np.where(v < 93, 23, np.where(v < 98, 12, 2)))
            else:
                q = spec.qhi - (pos * 12) // L
                q = np.broadcast_to(q, (n, L)).copy()
                repl = (u0 % np.uint64(10)) == 0
                q = np.where(repl, QREPL[((u0 >> np.uint64(32)) % np.uint64(6)).astype(np.int64)], q)
            ut = _draw(key, S_TAIL + r)[:, 0]
            tail = np.where(ut % np.uint64(20) == 0, 1 + ((ut >> np.uint64(32)) % np.uint64(30)).astype(np.int64), 0)
            q = np.where(pos >= (L - tail)[:, None], 2, q)
            sub = (u1 & np.uint64(0xFFFFFFFF)) < thr[q]
            b2 = (b + 1 + ((u1 >> np.uint64(32)) % np.uint64(3)).astype(np.int64)) & 3
            b = np.where(sub, b2, b)
            nt = NT_OF[b]
            isn = (q == 2) & ((u2 % np.uint64(10)) < 3)
            nt = np.where(isn, np.uint8(15), nt)
            return nt.astype(np.uint8), q.astype(np.int8)

        fnt, fq = read_quals_and_errors(bf, 0)
        rnt, rq = read_quals_and_errors(br, 1)

    fwd = np.zeros((n, stride, 2), dtype=np.uint8)
    rev = np.zeros((n, stride, 2), dtype=np.uint8)
    fwd[:, :L, 0] = fnt
    fwd[:, :L, 1] = fq.view(np.uint8)
    rev[:, :L, 0] = rnt
    rev[:, :L, 1] = rq.view(np.uint8)
    flen = np.full(n, L, dtype=np.uint16)
    rlen = np.full(n, L, dtype=np.uint16)
    return fwd, rev, flen, rlen


def generate_device(spec: SynthSpec, start: int, n: int, device="cuda:0", stream=None):
    """Same pairs as generate_numpy, produced by the HIP generator straight into HBM.
    Returns torch tensors (fwd u8[n,stride,2], rev, flen i16[n], rlen i16[n])."""
    import ctypes as C

    import torch

    from . import _lib
    dev = torch.device(device)
    stride = spec.row_stride()
    fwd = torch.empty((n, stride, 2), dtype=torch.uint8, device=dev)
    rev = torch.empty((n, stride, 2), dtype=torch.uint8, device=dev)
    flen = torch.empty(n, dtype=torch.int16, device=dev)
    rlen = torch.empty(n, dtype=torch.int16, device=dev)
    cs = _lib.SynthSpecC()
    cs.seed = spec.seed
    cs.read_len, cs.tmin, cs.tmax = spec.read_len, spec.tmin, spec.tmax
    cs.qhi, cs.novaseq = spec.qhi, int(spec.novaseq)
    if spec.fprimer is not None:
        fp = ["ACGT".index(c) for c in spec.fprimer]
        rp = ["ACGT".index(c) for c in spec.rprimer]
        cs.fprimer_len, cs.rprimer_len = len(fp), len(rp)
        for i, v in enumerate(fp):
            cs.fprimer[i] = v
        for i, v in enumerate(rp):
            cs.rprimer[i] = v
    thr = error_thresholds()
    st = stream if stream is not None else torch.cuda.current_stream(dev)
    _lib.check(_lib.lib().pamd_synth_generate(dev.index or 0, C.byref(cs), thr.ctypes.data, int(start), int(n),
                                             fwd.data_ptr(), rev.data_ptr(), stride, flen.data_ptr(), rlen.data_ptr(),
                                             C.c_void_p(st.cuda_stream)))
    return fwd, rev, flen, rlen


def generate_host(spec: SynthSpec, start: int, n: int):
    """generate_numpy's arrays (the same values, padding zeroed), made by the HIP generator where a GPU is present and copied
    back: 300 000 pairs take a second instead of a minute.  tests/test_parity_gpu.py::test_synth_device_matches_numpy holds the
    two generators to each other; without a GPU (the CPU suite) this IS generate_numpy."""
    try:
        import torch
        gpu = torch.cuda.is_available()
    except Exception:
        gpu = False
    if not gpu or n < 256:
        return generate_numpy(spec, start, n)
    from ._lib import PamdError
    try:
        f, r, fl, rl = generate_device(spec, start, n)
    except PamdError:   # (a specification the device generator does not take, e.g. a read length of its own: the host generator's pairs)
        return generate_numpy(spec, start, n)
    torch.cuda.synchronize()
    L = spec.read_len
    f, r = f.cpu().numpy(), r.cpu().numpy()
    f[:, L:] = 0
    r[:, L:] = 0
    return f, r, fl.cpu().numpy().view(np.uint16).copy(), rl.cpu().numpy().view(np.uint16).copy()


# ---- FASTQ rendering of synthetic pairs (input of the ingest path) --------------
_NT_ASCII = np.frombuffer(b"NACMGRSVTWYHKDBN", dtype=np.uint8)  # ntchar, nt.c:25
_NT_COMP = np.array([((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3) for v in range(16)], dtype=np.uint8)


def fastq_ids(start: int, n: int):
    """CASAVA 1.7 names of SURVEY.md 8d: SIM:1:FC:1:<tile>:<x>:<y>, every numeric field of fixed width."""
    i = np.arange(start, start + n, dtype=np.int64)
    tile = 1101 + (i // 8_100_000_000) % 8899
    x = 10000 + (i // 90000) % 90000
    y = 10000 + i % 90000
    return tile, x, y


def render_fastq(fwd, rev, flen, rlen, start: int = 0, qualmin: int = 33, tag: str = "1", crlf: bool = False):
    """Rows of panda_qual (all reads of one length L) -> (forward FASTQ bytes, reverse FASTQ bytes).
    The reverse file shows the read as sequenced: the stored row is its complement, base by base
    (iupac_reverse, nt.c:81-114), so the text is the complement of the stored codes in the same order."""
    n = fwd.shape[0]
    L = int(flen[0]) if n else 0
    assert n == 0 or (np.all(flen == L) and np.all(rlen == L)), "render_fastq wants equal read lengths"
    tile, x, y = fastq_ids(start, n)
    out = []
    eol = b"\r\n" if crlf else b"\n"
    for mate, rows, comp in ((1, fwd, False), (2, rev, True)):
        head0 = b"@SIM:1:FC:1:"
        tail = (" %d:N:0:%s" % (mate, tag)).encode() + eol
        hw = len(head0) + 4 + 1 + 5 + 1 + 5 + len(tail)
        rec = hw + L + len(eol) + 1 + len(eol) + L + len(eol)
        buf = np.empty((n, rec), dtype=np.uint8)
        c = 0
        buf[:, c:c + len(head0)] = np.frombuffer(head0, dtype=np.uint8)
        c += len(head0)
        for val, w in ((tile, 4), (x, 5), (y, 5)):
            for d in range(w):
                buf[:, c + d] = 48 + (val // 10 ** (w - 1 - d)) % 10
            c += w
            if w != 5 or val is x:
                buf[:, c] = ord(":")
                c += 1
        buf[:, c:c + len(tail)] = np.frombuffer(tail, dtype=np.uint8)
        c += len(tail)
        nt = rows[:, :L, 0] & 15
        if comp:
            nt = _NT_COMP[nt]
        buf[:, c:c + L] = _NT_ASCII[nt]
        c += L
        sep = eol + b"+" + eol
        buf[:, c:c + len(sep)] = np.frombuffer(sep, dtype=np.uint8)
        c += len(sep)
        buf[:, c:c + L] = (rows[:, :L, 1].view(np.int8).astype(np.int16) + qualmin).astype(np.uint8)
        c += L
        buf[:, c:c + len(eol)] = np.frombuffer(eol, dtype=np.uint8)
        assert c + len(eol) == rec
        out.append(buf.tobytes())
    return out[0], out[1]
