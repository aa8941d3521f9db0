#!/bin/bash
# scratch: which pairs differ between lean+general and general alone, sums in the reference's order, config 2
cd "$(dirname "$0")/.."
python3 - <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench, pandaseq_amd as pa
from pandaseq_amd import synth
import test_sweeps_gpu as t
n = 2_000_000
rows = synth.generate_device(synth.config_spec(2), 50_000_000, n)
asm_a, a = t._launch(pa, bench, synth, 2, "two-kernel", rows, n, True)
print("variant", asm_a.last_lean_variant(), "deferred", asm_a.last_deferred())
asm_b, b = t._launch(pa, bench, synth, 2, "general", rows, n, True)
ra, rb = a.results_numpy(), b.results_numpy()
bad = np.nonzero(ra["est_overlap_prob"] != rb["est_overlap_prob"])[0]
print(len(bad), "differ")
print("statuses", np.unique(ra["status"][bad], return_counts=True), np.unique(rb["status"][bad], return_counts=True))
for i in bad[:10]:
    print(i, ra[i], rb[i])
badq = np.nonzero(ra["quality"] != rb["quality"])[0]
print(len(badq), "quality differ", np.unique(ra["status"][badq], return_counts=True))
PY
