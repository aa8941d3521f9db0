"""bench.py's contract: one JSON line with the required keys, single- and multi-rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


def _check(line, n_gpus, steps, warmup):
    d = json.loads(line)
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["unit"] == "pairs/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"] and "sums" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # value is whole-job pairs over the timed region
    assert abs(d["value"] - d["config"]["pairs_per_gpu"] * n_gpus / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-6
    assert d["stat"]["count"] == d["config"]["pairs_per_gpu"] * n_gpus
    return d


def test_bench_single_gpu(built):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pairs", "300000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0.2", "--e2e-pairs", "400000", "--other-pairs", "200000"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _check(out.stdout.strip().splitlines()[-1], 1, 2, 1)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "pairs/s" and c["cores"] >= 1 and c["matches_gpu_on_sample"] is True
    # `cores` = CPUs the process may use, `threads` = pthreads run on them; the sample check looks at rows, not only at records
    assert c["threads"] >= c["cores"] and c["cores"] <= c["effective_cores"]["used_for_efficiency"] + 0.5
    assert c["sample_check"]["records"] is True and c["sample_check"]["rows"] is True and c["sample_check"]["accepted_rows"] > 0
    # the other BASELINE.json configurations, timed by the same code, in the same record
    o = d["other_configs"]
    for k, lean in (("config2", "primers"), ("config3", "short"), ("config4", "short+narrow")):
        assert "error" not in o[k], o[k]
        assert o[k]["pairs_per_gpu"] == 200000 and o[k]["stat"]["count"] == 200000 and o[k]["pairs_per_s"] > 0 and 0 < o[k]["frac"] < 1 and o[k]["kernel_ms"] > 0
        assert lean in o[k]["lean_kernel"].lower(), o[k]["lean_kernel"]
    assert o["config3"]["rejected"][1] > 0 and o["config2"]["stat"]["nofp"] + o["config2"]["stat"]["norp"] >= 0
    e = d["end_to_end"]
    assert "error" not in e, e
    assert e["record_only"]["pairs"] == 400000 and e["record_only"]["pairs_per_s"] > 0 and e["every_base_read"]["ok"] > 0
    assert e["panda_assembler_assemble_per_pair_host"]["device"] is False


def test_bench_two_ranks_sharded(built):
    """The N>1 path (contiguous shards + counter all-reduce + max-over-ranks timing) on one GPU:
    both ranks share cuda:0 and reduce over gloo (PAMD_BENCH_SINGLE_DEVICE test hook)."""
    env = dict(os.environ, PAMD_BENCH_SINGLE_DEVICE="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29577", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--pairs", "200000", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = _check(lines[0], 2, 2, 1)
    assert "cpu_baseline" not in d
    # the two shards are pairs [0, n) and [n, 2n): their summed counters equal one 2n-pair run
    single = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pairs", "400000", "--steps", "1", "--warmup", "0", "--no-cpu"],
                            capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert single.returncode == 0, single.stderr[-2000:]
    s = json.loads(single.stdout.strip().splitlines()[-1])
    assert d["stat"] == s["stat"]
    # ... and so does the STAT block a reference worker would print from them (pool.c:86-104), overlap histogram included
    assert d["stat_block_sha1"] == s["stat_block_sha1"]


@pytest.mark.parametrize("cfg", [1, 2])
def test_bench_chunked_launches(built, cfg):
    """A shard cut into three launches (what bench.py does by itself when the output does not fit beside the input)
    reports the same counters as the single launch, and says how it ran.  (Config 2: the lean kernel finds the primers and
    hands deferred pairs' offsets on, per launch.)"""
    runs = []
    for extra in ([], ["--chunk-pairs", "110000"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(cfg), "--pairs", "300000", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-e2e"] + extra,
                             capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        runs.append(_check(out.stdout.strip().splitlines()[-1], 1, 1, 0))
    assert runs[0]["config"]["launches_per_step"] == 1 and runs[1]["config"]["launches_per_step"] == 3
    assert runs[0]["stat"] == runs[1]["stat"] and runs[0]["stat_block_sha1"] == runs[1]["stat_block_sha1"]
    assert runs[0]["roofline"]["algorithmic_bytes_per_launch"] == runs[1]["roofline"]["algorithmic_bytes_per_launch"]


def test_bench_two_ranks_over_rccl(built):
    """The N>1 path as the driver launches it -- one rank per GPU, the STAT counters all-reduced over RCCL (backend "nccl") -- whenever
    this machine has two devices; the gloo variant above covers the sharding and the reduction on one.  Skips, loudly, on a 1-GPU box."""
    import torch
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip(f"RCCL over two ranks needs two GPUs; this box has {n_dev} (the gloo test above covers the N>1 path on one device)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PAMD_BENCH_SINGLE_DEVICE", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29578",
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--pairs", "400000", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = _check(lines[0], 2, 2, 1)
    assert "RCCL" in d["config"]["sharding"]
    single = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pairs", "800000", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-e2e"],
                            capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert single.returncode == 0, single.stderr[-2000:]
    s = json.loads(single.stdout.strip().splitlines()[-1])
    assert d["stat"] == s["stat"] and d["stat_block_sha1"] == s["stat_block_sha1"]
