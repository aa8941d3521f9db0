#!/usr/bin/env python3
"""Dump the lean kernel's per-wave LDS for one pair (needs a -DPAMD_FK_DEBUG build: make EXTRA=-DPAMD_FK_DEBUG) and
print why pairs were deferred.  Development aid, GPU only.   PAMD_DEBUG_PAIR=<index> python tools/fk_debug.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import pandaseq_amd as pa
from pandaseq_amd import synth, _lib

n = int(os.environ.get("PAMD_PAIRS", "65536"))
pair = int(os.environ.setdefault("PAMD_DEBUG_PAIR", "1"))
spec = synth.config_spec(1)
fwd, rev, flen, rlen = synth.generate_numpy(spec, 0, n)
asm, _ = bench.setup_assembler(pa, synth, 1, 0, exact_sums=False)
b = pa.upload_batch(fwd, rev, flen, rlen, seq_stride=0)
asm.assemble_device(b)
torch.cuda.synchronize()
L = _lib.lib()
ph = (C.c_ulonglong * 10)()
L.pamd_debug_phase_cycles(asm._h, ph, 1)
print("deferred", asm.last_deferred(), "of", n, "reasons [weird, lost, noseed, >4, lowq, deep]:", list(ph)[:6])
buf = np.zeros(3 * 8192, dtype=np.uint8)
L.pamd_debug_fetch.argtypes = [C.c_void_p, C.c_size_t]
rc = L.pamd_debug_fetch(buf.ctypes.data, buf.size)
print("fetch rc", rc)
d0, d1, d2 = buf[:8192], buf[8192:16384], buf[16384:]
print("scalars after insert [uf ur deep lost carry0 carry1]:", [hex(int(x) & 0xFFFFFFFF) for x in d0[7680:7704].view(np.int32)])
print("scalars after scan [uf ur ncand ov0 ov1 bo]:", d2[7680:7704].view(np.int32))
print("scalars after merge [A novl Bn mism degen seqlen]:", d1[7680:7704].view(np.int32))
fl, rl = int(flen[pair]), int(rlen[pair])
print("flen rlen", fl, rl)
Frow = d0[0:512].reshape(256, 2)
Rrow = d0[512:1024].reshape(256, 2)
print("F row equal:", np.array_equal(Frow[:fl], fwd[pair, :fl]), " R row equal:", np.array_equal(Rrow[:rl], rev[pair, :rl]))
if not np.array_equal(Frow[:fl], fwd[pair, :fl]):
    bad = np.nonzero((Frow[:fl] != fwd[pair, :fl]).any(axis=1))[0]
    print(" F differs at", bad[:40])
if not np.array_equal(Rrow[:rl], rev[pair, :rl]):
    bad = np.nonzero((Rrow[:rl] != rev[pair, :rl]).any(axis=1))[0]
    print(" R differs at", bad[:40])
code_of_f = {1: 0, 2: 1, 4: 2, 8: 3, 15: 0}
code_of_r = {1: 0, 2: 2, 4: 1, 8: 3, 15: 0}
def unpack2(bytes_, nb):
    out = []
    for i in range(nb):
        out.append((int(bytes_[i // 4]) >> (2 * (i % 4))) & 3)
    return np.array(out)
fc = unpack2(d0[1024:1088], fl); rc_ = unpack2(d0[1104:1168], rl)
fn = unpack2(d0[1184:1248], fl); rn = unpack2(d0[1264:1328], rl)
efc = np.array([code_of_f[int(x)] for x in fwd[pair, :fl, 0]]); erc = np.array([code_of_r[int(x)] for x in rev[pair, :rl, 0]])
print("fc ok", np.array_equal(fc, efc), "rc ok", np.array_equal(rc_, erc))
print("fn ok", np.array_equal(fn, (fwd[pair, :fl, 0] == 15) * 1), "rn ok", np.array_equal(rn, (rev[pair, :rl, 0] == 15) * 2))
bits = d0[1344:1376].view(np.uint32)
print("candidate bits:", [hex(int(x)) for x in bits])
t1 = d0[1952:1952 + 4096].view(np.uint32)
t2 = d0[1952 + 4096:1952 + 5120].view(np.uint32)
print("t1 filled", int((t1 != 0xFFFFFFFF).sum()), "t2 filled", int((t2 != 0xFFFFFFFF).sum()), "t3/t4 filled", int((d0[1952 + 5120:1952 + 5632].view(np.uint32) != 0xFFFFFFFF).sum()))
# expected forward k-mer words
def kmers_fwd(nts):
    out = []
    for p in range(8, len(nts)):
        win = nts[p - 8:p + 1]
        if (win == 15).any():
            continue
        k = 0
        for t in range(8):
            k |= code_of_f[int(nts[p - 7 + t])] << (2 * t)
        out.append((k << 16) | p)
    return sorted(out)
exp = kmers_fwd(fwd[pair, :fl, 0])
got = sorted(int(x) for x in np.concatenate([t1, t2, d0[1952 + 5120:1952 + 5632].view(np.uint32)]) if x != 0xFFFFFFFF)
print("forward k-mer words: expected", len(exp), "in tables", len(got), "equal", exp == got)
if exp != got:
    se, sg = set(exp), set(got)
    print(" missing", [hex(x) for x in sorted(se - sg)[:10]], " extra", [hex(x) for x in sorted(sg - se)[:10]])
res = b.results_numpy()[pair]
print("record:", res)
