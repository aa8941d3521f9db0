"""Which kernel served a launch (pamd_last_lean_variant): every BASELINE configuration runs in the lean kernel's instantiation
DESIGN.md section 3 says it does -- config 4's short reads in the three-bases-per-lane one --, and what the lean kernel does not
serve goes to the general kernel whole.  (Results are compared with the oracle either way elsewhere; this pins the routing, so
that a bench line can be read as a measurement of the kernel it names.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _run(cfg, n=20000, **kw):
    import torch
    import bench
    import pandaseq_amd as pa
    from pandaseq_amd import synth
    exact = kw.pop("exact", False)
    tweak = kw.pop("tweak", None)
    asm, spec = bench.setup_assembler(pa, synth, cfg, 0, exact_sums=exact, **kw)
    if tweak:
        tweak(asm)
    f, r, fl, rl = synth.generate_device(spec, 0, n)
    outs = pa.alloc_outputs(n, n * 640 + (1 << 25), "cuda:0")    # (every wave reserves chunks of its own: slack for 4096 of them)
    b = pa.DeviceBatch(n=n, fwd=f, rev=r, flen=fl, rlen=rl, **outs)
    asm.assemble_device(b)
    torch.cuda.synchronize()
    asm.check_overflow()
    return asm, b


@pytest.mark.parametrize("cfg,want", [(0, "short+narrow"), (1, "short"), (2, "primers"), (3, "short"), (4, "short+narrow")])
def test_baseline_configs_run_in_their_lean_instantiation(built, cfg, want):
    for exact in (False, True):
        asm, b = _run(cfg, exact=exact)
        assert asm.last_lean_variant() == want, (cfg, exact, asm.last_lean_variant())
        assert 0 <= asm.last_deferred() < 2000 and b.counters_dict()["count"] == 20000


def test_primers_after_routing(built):
    asm, _ = _run(2, primers_after=True)
    assert asm.last_lean_variant() == "after"
    asm, _ = _run(2, primers_after=True, exact=True)      # the reference-order finish walks rows a clipped row no longer holds whole
    assert asm.last_lean_variant() == "general" and asm.last_deferred() == -1
    asm, _ = _run(2, primers_after=True, tweak=lambda a: a.set_primer_penalty(0.01))
    assert asm.last_lean_variant() == "general"


def test_what_the_lean_kernel_does_not_serve(built):
    asm, _ = _run(1, tweak=lambda a: a.set_maximum_overlap(200))
    assert asm.last_lean_variant() == "general"
    asm, _ = _run(2, tweak=lambda a: a.set_primer_penalty(0.01))
    assert asm.last_lean_variant() == "general"
    asm, _ = _run(1, tweak=lambda a: a.set_fast_path(False))
    assert asm.last_lean_variant() == "general"
    asm, _ = _run(1, tweak=lambda a: (a.set_forward_trim(7), a.set_reverse_trim(3)))
    assert asm.last_lean_variant() == "primers"     # trims are served by the primers' instantiation
