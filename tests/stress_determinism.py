#!/usr/bin/env python3
"""Repeat the same batches many times and compare every launch with the oracle (development aid, GPU only):
catches nondeterminism (races, uninitialised LDS) that a single parity run can miss."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import pandaseq_amd as pa
from oracle import pyoracle as po
from pandaseq_amd import synth
from util import adversarial_pairs, compare_counters, compare_results

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for cfg, algo in ((0, "simple_bayesian"), (1, "simple_bayesian"), (4, "simple_bayesian"), (1, "pear"), (1, "rdp_mle"), (0, "pear")):
    spec = synth.config_spec(cfg)
    f, r, fl, rl = adversarial_pairs(spec, 6000, seed=cfg + 3) if cfg == 1 else synth.generate_host(spec, 0, 6000)
    if algo == "pear":
        rl = np.minimum(rl, fl)
    o = po.assemble_batch(po.make_config(algo), f, r, fl, rl)
    for exact in (True, False):
        asm = pa.Assembler(0)
        asm.set_algorithm(algo)
        asm.set_exact_sums(exact)
        for rep in range(reps):
            b = pa.upload_batch(f, r, fl, rl, device="cuda:0", seq_stride=0)
            asm.assemble_device(b)
            torch.cuda.synchronize()
            try:
                compare_results(b.results_numpy(), *b.rows_numpy(), *o[:3], exact_sums=exact)
                compare_counters(b.counters_dict(), o[3])
            except AssertionError as e:
                bad += 1
                print(f"MISMATCH config {cfg} {algo} exact={exact} rep {rep}: {str(e)[:200]}")
        print(f"config {cfg} {algo:16s} exact={exact}: {reps} launches compared")
print("mismatching launches:", bad)
sys.exit(1 if bad else 0)
