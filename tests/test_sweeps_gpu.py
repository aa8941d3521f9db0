"""Fixed-seed slices of the randomized campaigns, under the driver (pytest -m gpu).

The campaigns (tests/parity_sweep.py, primer_sweep.py, length_sweep.py, textio_sweep.py, compat_sweep.py, per_pair_sweep.py) compare the device
path with the oracle on every record, row and counter -- diff.c:26-99's notion of "same result", applied wholesale.  Round 3's
runs of them found two silent wrong answers in the lean kernel that the fixed suite had passed, and they only ever ran on the
builder's lease.  Here every driver run sees a slice of each; the seeds rotate with ROUND (bump it every round), so successive
rounds see different inputs.  A round that finds something is pinned as a test of its own (test_lean_long_gpu.py,
test_parity_gpu.py), never fixed and forgotten.

Also: the lean kernel + general kernel against the general kernel alone (tools/fk_probe.py's comparison as a test), 8 M pairs of
each BASELINE configuration: every record, every counter, every merged base and log-p of every accepted pair.

Each test id carries the campaign, the first round and the number of rounds; the launches compared are asserted (a campaign that
silently compares nothing fails) and printed (-s).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu

ROUND = 6   # the build round: every campaign starts at a round number derived from it


def _log(lines):
    def log(s):
        lines.append(s)
        print(s, flush=True)
    return log


def _verdict(name, bad, done, want_done, lines):
    assert done >= want_done, f"{name}: compared {done} launches, expected at least {want_done}"
    assert bad == 0, f"{name}: {bad} of {done} launches differ from the oracle:\n" + "\n".join(l for l in lines if l.startswith("MISMATCH"))


# (campaign, mode, first round, rounds, launches the slice must have compared, pairs per launch | copies of the round's 3 000)
# parity: per round 4 configurations x (7 + 7 + 3 + 7 scorers) x 2 sum modes = 48 launches
# The last slice of each (round 5): launches of 100 000 pairs, where every wave of the lean kernel goes through whole groups of its
# reference-order sums and several finishes (2 000 pairs are less than one per wave); the rows there through their checksums.
PARITY = [("plain", 100 * ROUND, 6, 6 * 48, 2000), ("wide", 100 * ROUND, 9, 9 * 48, 2000), ("wide", 100 * ROUND + 50, 3, 3 * 48, 100_000)]
PRIMER = [("plain", 2000 * ROUND, 45, 90, 1), ("after", 2000 * ROUND + 500, 20, 40, 1), ("penalty", 2000 * ROUND + 700, 20, 40, 1),
          ("plain", 2000 * ROUND + 800, 9, 18, 30), ("after", 2000 * ROUND + 900, 9, 18, 30)]


@pytest.mark.parametrize("mode,first,rounds,launches,n", PARITY, ids=[f"parity-{m}-first{f}-rounds{r}-launches{l}-pairs{n}" for m, f, r, l, n in PARITY])
def test_parity_campaign(built, mode, first, rounds, launches, n):
    import parity_sweep
    lines = []
    bad, done = parity_sweep.run(rounds, mode == "wide", first, n=n, log=_log(lines))
    _verdict(f"parity_sweep {mode}", bad, done, launches, lines)


@pytest.mark.parametrize("mode,first,rounds,launches,copies", PRIMER, ids=[f"primer-{m}-first{f}-rounds{r}-launches{l}-copies{c}" for m, f, r, l, c in PRIMER])
def test_primer_campaign(built, mode, first, rounds, launches, copies):
    import primer_sweep
    lines = []
    bad, done = primer_sweep.run(rounds, first, mode, log=_log(lines), copies=copies)
    _verdict(f"primer_sweep {mode}", bad, done, launches, lines)


LENGTH = (300 * ROUND, 12, 12 * 14)


@pytest.mark.parametrize("first,rounds,launches", [LENGTH], ids=["length-first%d-rounds%d-launches%d" % LENGTH])
def test_length_campaign(built, first, rounds, launches):
    import length_sweep
    lines = []
    bad, done = length_sweep.run(rounds, first, n=1200, log=_log(lines))
    _verdict("length_sweep first%d rounds%d" % (first, rounds), bad, done, launches, lines)


TEXTIO = (400 * ROUND, 60)


@pytest.mark.parametrize("first,rounds", [TEXTIO], ids=["textio-first%d-rounds%d" % TEXTIO])
def test_textio_campaign(built, first, rounds):
    import textio_sweep
    lines = []
    bad, done = textio_sweep.run(rounds, first, log=_log(lines))
    _verdict("textio_sweep first%d rounds%d" % (first, rounds), bad, done, rounds, lines)


COMPAT = (500 * ROUND, 40, 80)


@pytest.mark.parametrize("first,rounds,runs", [COMPAT], ids=["compat-first%d-rounds%d-runs%d" % COMPAT])
def test_compat_campaign(built, first, rounds, runs):
    import compat_sweep
    lines = []
    bad, done = compat_sweep.run(rounds, first, log=_log(lines))
    _verdict("compat_sweep first%d rounds%d" % (first, rounds), bad, done, runs, lines)


PERPAIR = (600 * ROUND, 25)


@pytest.mark.parametrize("first,rounds", [PERPAIR], ids=["perpair-first%d-rounds%d" % PERPAIR])
def test_per_pair_campaign(built, first, rounds):
    """panda_assembler_assemble() on the host path == pamd_compat_assemble_many() on the kernels == per pair on the device, under
    random options (primers before / after / with a penalty, a fused module, overlap bounds, a threshold), and == the oracle."""
    import per_pair_sweep
    lines = []
    bad, done = per_pair_sweep.run(rounds, first, log=_log(lines))
    _verdict("per_pair_sweep first%d rounds%d" % (first, rounds), bad, done, rounds, lines)


# ---- lean + general kernel against the general kernel alone -----------------------------------------------------------
def _launch(pa, bench, synth, cfg, mode, rows, n, sums_exact, primers_after=False):
    import torch
    f, r, fl, rl = rows
    asm, _ = bench.setup_assembler(pa, synth, cfg, 0, exact_sums=sums_exact, primers_after=primers_after)
    asm.set_fast_path(mode == "two-kernel")
    outs = pa.alloc_outputs(n, n * 620 if cfg == 2 else n * 520, "cuda:0")
    b = pa.DeviceBatch(n=n, fwd=f, rev=r, flen=fl, rlen=rl, **outs)
    asm.assemble_device(b)
    torch.cuda.synchronize()
    asm.check_overflow()
    return asm, b


@pytest.mark.parametrize("sums", ["fast", "exact"])
@pytest.mark.parametrize("cfg", [1, 2, 3, 4], ids=lambda c: f"config{c}-8Mpairs-start{ROUND}0M")
def test_lean_plus_general_equals_general_alone(built, cfg, sums):
    """The two-kernel launch defers, never approximates: on 8 M pairs of each BASELINE configuration every record field, every
    counter and every merged base and log-p of every accepted pair is what the general kernel alone produces.  Any-order sums
    (the bench's mode): `quality` and the pear / rdp_mle score are sums in any order in both, within 1e-9.  Sums in the
    reference's order (round 5): both doubles BIT for bit -- the lean kernel makes them in its group walk (63 or 60 chains at a
    time from scratch rows of table entries, every wave through some two thousand pairs: full groups, the finish after 63 / 60
    pairs, the last partial group), the general kernel in its own in-wave finish from the f64 rows."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    n = 8_000_000
    spec = synth.config_spec(cfg)
    rows = synth.generate_device(spec, ROUND * 10_000_000, n)
    exact = sums == "exact"
    asm_a, a = _launch(pa, bench, synth, cfg, "two-kernel", rows, n, exact)
    assert asm_a.last_launch_info() is not None
    deferred = asm_a.last_deferred()
    assert 0 <= deferred < n // 4, f"the lean kernel deferred {deferred} of {n} pairs"
    asm_b, b = _launch(pa, bench, synth, cfg, "general", rows, n, exact)
    assert np.array_equal(a.counters.cpu().numpy(), b.counters.cpu().numpy()), "counters"
    ra, rb = a.results_numpy(), b.results_numpy()
    for name in ra.dtype.names:
        if name in ("seq_offset", "reserved"):
            continue
        if name in ("quality", "est_overlap_prob") and not exact:
            ok = np.abs(ra[name] - rb[name]) <= 1e-9 * np.maximum(1.0, np.abs(rb[name]))
        else:
            ok = ra[name] == rb[name]
        assert ok.all(), (name, np.nonzero(~ok)[0][:8], ra[name][~ok][:8], rb[name][~ok][:8])
    accepted = np.nonzero(ra["status"] == 0)[0]
    assert accepted.size > n // 3
    idx = torch.arange(640, device="cuda:0")
    differing = 0
    for chunk in np.array_split(accepted, max(1, accepted.size // 100_000)):
        oa = torch.from_numpy(ra["seq_offset"][chunk].astype(np.int64)).cuda()
        ob = torch.from_numpy(rb["seq_offset"][chunk].astype(np.int64)).cuda()
        ln = torch.from_numpy(ra["seq_len"][chunk].astype(np.int64)).cuda()
        m = idx[None, :] < ln[:, None]
        ia = (oa[:, None] + idx[None, :]).clamp(max=a.seq_nt.numel() - 1)
        ib = (ob[:, None] + idx[None, :]).clamp(max=b.seq_nt.numel() - 1)
        diff = ((a.seq_nt[ia] != b.seq_nt[ib]) | (a.seq_p[ia] != b.seq_p[ib])) & m
        differing += int(diff.any(dim=1).sum().item())
    print(f"config {cfg}: {n} pairs, {deferred} deferred by the lean kernel, {accepted.size} rows compared", flush=True)
    assert differing == 0, f"{differing} of {accepted.size} rows differ"
    del a, b, rows
    torch.cuda.empty_cache()
