import os, sys
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pandaseq_amd as pa
from pandaseq_amd import synth
from test_lean_long_gpu import _oracle, _plant, _run
codes = np.array([1, 2, 4, 8, 1, 2, 4, 8, 3, 5, 10, 12, 7, 14, 15], np.uint8)
rnd = 91
rng = np.random.default_rng(7000 + rnd)
spec = synth.config_spec(2 if rnd % 3 else 1)
L = spec.read_len
n = 3000
fwd, rev, flen, rlen = synth.generate_numpy(spec, int(rng.integers(0, 40_000_000)), n)
fwd[:, :, 0] = np.where(np.arange(fwd.shape[1])[None, :] < L, fwd[:, :, 0], 0)
fplen, rplen = int(rng.integers(1, 25)), int(rng.integers(1, 25))
fp = rng.choice(codes if rnd % 2 else codes[:4], fplen)
rp = rng.choice(codes[:4], rplen)
anywhere = rng.choice([0.0, 0.1, 0.4, 1.0])
wf = np.where(rng.random(n) < 0.9, rng.integers(0, max(1, L - fplen), n) * (rng.random(n) < anywhere), -1)
wr = np.where(rng.random(n) < 0.9, rng.integers(0, max(1, L - rplen), n) * (rng.random(n) < anywhere), -1)
subst, lowq = rng.choice([0.0, 0.03, 0.08, 0.2]), rng.choice([0.0, 0.1, 0.3, 0.7])
_plant(rng, fwd, flen, fp, wf, subst, lowq)
_plant(rng, rev, rlen, rp, wr, subst, lowq)
if rnd % 4 == 3:
    flen[:] = rng.integers(1, L + 1, n)
    rlen[:] = rng.integers(1, L + 1, n)
algo = ["pear", "simple_bayesian", "rdp_mle"][rnd % 3]
thr = float(rng.choice([0.6, 0.6, 0.9, 0.3]))
print("L", L, "fp", fp, "rp", rp, "thr", thr, algo)
o = _oracle(algo, fwd, rev, flen, rlen, fprimer=fp, rprimer=rp, threshold=thr)
ost = o[0]["status"]
for fast in (True, False):
    asm = pa.Assembler(0)
    asm.set_algorithm(algo); asm.set_exact_sums(False); asm.set_threshold(thr)
    asm.set_forward_primer(fp); asm.set_reverse_primer(rp)
    if not fast:
        asm.set_fast_path(False)
    b = _run(asm, fwd, rev, flen, rlen)
    st = b.results_numpy()["status"]
    bad = np.nonzero(st != ost)[0]
    print("fast", fast, "deferred", asm.last_deferred(), "bad", bad[:12], "dev", st[bad[:12]], "oracle", ost[bad[:12]])
    for i in bad[:6]:
        print("  pair", i, "flen", flen[i], "rlen", rlen[i], "fo/ro dev", b.results_numpy()["fwd_offset"][i], b.results_numpy()["rev_offset"][i], "oracle", o[0]["fwd_offset"][i] if "fwd_offset" in o[0].dtype.names else None)
        print("   rev head nt", rev[i, :8, 0], "q", rev[i, :8, 1])
