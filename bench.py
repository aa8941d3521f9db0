#!/usr/bin/env python3
"""bench.py -- merged read-pairs/s of the overlap-assembly hot path on MI355X.

A "step" is one pass of the hot path (one pamd_assemble_device launch) over one batch of synthetic
read pairs that is already resident in HBM.  Default workload = BASELINE.json configs[1]:
50 M synthetic 2x250 bp pairs, -A simple_bayesian, one GPU.  With --gpus N every rank owns its own
50 M-pair shard (weak scaling); the only collective is one RCCL all-reduce of the STAT counters per
step.  Rank 0 prints ONE JSON line.

    python bench.py                     # N=1, defaults
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29500 bench.py --gpus 8 --steps 5 --warmup 1
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CONFIGS = {
    0: dict(algo="simple_bayesian", pairs=10_000, name="10k synthetic 2x150 bp pairs, 40 bp overlap, simple_bayesian"),
    1: dict(algo="simple_bayesian", pairs=50_000_000, name="50M synthetic 2x250 bp pairs, simple_bayesian"),
    2: dict(algo="pear", pairs=50_000_000, name="50M MiSeq-style 2x300 bp pairs with primers (-p/-q), pear"),
    3: dict(algo="rdp_mle", pairs=50_000_000, name="2x250 bp pairs, rdp_mle + min_phred + min_overlapbits (50M per GPU)"),
    4: dict(algo="simple_bayesian", pairs=125_000_000, name="NovaSeq-style 2x150 bp pairs, simple_bayesian (125M per GPU)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=1, help="BASELINE.json configs[] index")
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU (0 = the config's size)")
    ap.add_argument("--chunk-pairs", type=int, default=0, help="pairs per launch (0 = the whole shard when its output fits in HBM beside "
                                                                "its input, else the largest chunk whose two output arenas do)")
    ap.add_argument("--cpu-seconds", type=float, default=18.0, help="wall-clock target of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--sums", choices=["fast", "exact"], default="fast",
                    help="fast: quality / pear+rdp scores summed in any order with exactness guards on every decision "
                         "(statuses, overlaps, merged bases, per-base log-p bit-identical; the two f64 scalars within 1e-9); "
                         "exact: the reference's summation order, every output bit-identical")
    ap.add_argument("--traffic-bytes", type=float, default=None, help="PMC-measured HBM bytes per launch (from profiles/)")
    ap.add_argument("--algo", default=None, help="override the config's scorer (simple_bayesian, pear, rdp_mle, uparse, flash, ea_util, stitch): "
                                                 "not a BASELINE.json configuration, for the per-scorer table of DESIGN.md")
    ap.add_argument("--primers-after", action="store_true", help="`-a`: strip primers after assembly (config 2 has primers)")
    ap.add_argument("--io", action="store_true", help="N=1 only: also time the data-format rows either side of the path (FASTQ ingest, FASTA / "
                                                      "FASTQ formatting, tools/bench_io.py) with their own CPU baseline; reported under \"io\"")
    ap.add_argument("--io-pairs", type=int, default=4_000_000)
    ap.add_argument("--no-other-configs", action="store_true", help="skip the `other_configs` leg (N=1, default configuration only: three timed steps of each other "
                                                                    "BASELINE.json configuration's single-GPU shard after the headline line is measured)")
    ap.add_argument("--other-pairs", type=int, default=0, help="pairs per GPU of the `other_configs` entries (0 = each configuration's own size; tests shrink it)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end leg (the reference's own panda_assembler_next loop "
                                                          "through the drop-in C API, tests/c/bench_next.c)")
    ap.add_argument("--e2e-pairs", type=int, default=24_000_000)
    ap.add_argument("--min-phred", type=int, default=2, help="config 3: the min_phred module's threshold.  The synthetic qualities are bimodal: nearly every "
                                                             "merged row holds a base of PHRED 2, so 2 rejects nothing and 3 rejects 96 %% of the rows -- there is no value "
                                                             "in between; the reject count is stated in config.workload either way")
    return ap.parse_args()


def setup_assembler(pa, synth, cfg_index, device_index, exact_sums=True, algo=None, primers_after=False, min_phred=2):
    asm = pa.Assembler(device_index)
    asm.set_exact_sums(exact_sums)
    spec = synth.config_spec(cfg_index)
    c = CONFIGS[cfg_index]
    asm.set_algorithm(algo or c["algo"])
    asm.set_primers_after(primers_after)
    if cfg_index == 2:
        asm.set_forward_primer(synth.nt_codes(spec.fprimer))
        asm.set_reverse_primer(synth.nt_codes(spec.rprimer, complement=True))
    if cfg_index == 3:
        asm.add_module("min_phred:%d" % min_phred)
        asm.add_module("min_overlapbits:15")
    return asm, spec


def oracle_config(po, synth, cfg_index, algo=None, primers_after=False, min_phred=2):
    spec = synth.config_spec(cfg_index)
    kw = {}
    if cfg_index == 2:
        kw = dict(fprimer=synth.nt_codes(spec.fprimer), rprimer=synth.nt_codes(spec.rprimer, complement=True))
    if cfg_index == 3:
        kw = dict(modules=[("min_phred", min_phred), ("min_overlapbits", 15)])
    return po.make_config(algo or CONFIGS[cfg_index]["algo"], primers_after=primers_after, **kw)


def run_config(env, args, cfg_index, n, steps, warmup, headline=True):
    """One configuration's shard on this rank's GPU: generate it in HBM, calibrate the arena, `warmup` untimed and `steps` timed
    passes of the hot path between barrier + synchronize fences.  Returns everything the JSON line (or an `other_configs`
    entry) is made from; the batch stays resident for the legs that follow."""
    import torch
    pa, synth, dist, dev, rank, world, local = env["pa"], env["synth"], env["dist"], env["dev"], env["rank"], env["world"], env["local"]
    from pandaseq_amd.stats import allreduce_counters
    # (the scorer / `-a` / min_phred overrides and --chunk-pairs belong to the headline configuration; an `other_configs` entry runs as BASELINE.json names it)
    asm, spec = setup_assembler(pa, synth, cfg_index, local, exact_sums=args.sums == "exact", algo=args.algo if headline else None,
                                primers_after=args.primers_after and headline, min_phred=args.min_phred if headline else 2)

    # ---- calibrate the arena on the first pairs, then build the full shard in HBM ----
    ncal = min(n, 1 << 20)
    first = rank * n  # contiguous shards of the global pair index space

    def make_batch(count, start, capacity):
        f, r, fl, rl = synth.generate_device(spec, start, count, device=dev)
        outs = pa.alloc_outputs(count, capacity, dev, seq_stride=0)
        return pa.DeviceBatch(n=count, fwd=f, rev=r, flen=fl, rlen=rl, **outs)

    cal = make_batch(ncal, first, ncal * 912)
    asm.assemble_device(cal)
    torch.cuda.synchronize()
    asm.check_overflow()
    used = int(cal.cursor.item())
    per_pair = used / ncal
    del cal
    torch.cuda.empty_cache()
    # The shard's INPUT is resident as a whole.  Its merged output is too when it fits beside it (config 1); otherwise the
    # launch is cut into chunks of pairs whose rows go to two arenas used in turn (a consumer would drain one while the
    # other fills): every pair is still assembled and every output byte written in the timed region, the arena is reused.
    # No halving of the shard: a shard that cannot be served fails.
    stride = spec.row_stride()
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    fixed = n * (2 * stride * 2 + 4 + 48) + (1 << 30)
    slack = 1 << 28

    def arena_bytes(pairs):
        return 9 * (int(pairs * per_pair * 1.03) + slack)

    budget = int(free_b * 0.94)
    if args.chunk_pairs and headline:
        chunk = min(n, args.chunk_pairs)
    elif fixed + arena_bytes(n) <= budget:
        chunk = n
    else:
        chunk = n
        while chunk > (1 << 20) and fixed + 2 * arena_bytes(chunk) > budget:
            chunk = (chunk + 1) // 2
        if fixed + 2 * arena_bytes(chunk) > budget:
            raise SystemExit(f"a shard of {n} pairs does not fit this GPU ({free_b >> 30} GiB free)")
    n_chunks = (n + chunk - 1) // chunk
    f, r, fl, rl = synth.generate_device(spec, first, n, device=dev)
    arenas = [pa.alloc_outputs(0, int(chunk * per_pair * 1.03) + slack, dev, seq_stride=0) for _ in range(1 if n_chunks == 1 else 2)]
    results = torch.empty((n, 48), dtype=torch.uint8, device=dev)
    batch = pa.DeviceBatch(n=n, fwd=f, rev=r, flen=fl, rlen=rl, results=results, seq_nt=arenas[0]["seq_nt"], seq_p=arenas[0]["seq_p"],
                           cursor=arenas[0]["cursor"], counters=arenas[0]["counters"], seq_stride=0)
    torch.cuda.synchronize()

    def one_step(ev0=None, ev1=None):
        batch.counters.zero_()
        if ev0 is not None:
            ev0.record()
        for k in range(n_chunks):
            ar = arenas[k % len(arenas)]
            ar["cursor"].zero_()
            asm.assemble_device(batch, first=k * chunk, count=min(chunk, n - k * chunk), arena=ar)
        if ev1 is not None:
            ev1.record()
        if dist is not None:
            allreduce_counters(batch.counters)  # the ONE collective: STAT counters, int64 x 916

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        one_step()
    fence()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        one_step(*evs[k])
    fence()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: nothing overflowed, every pair was seen
    ctr = batch.counters_dict()
    total_expected = n * world if dist is not None else n
    assert ctr["count"] == total_expected, (ctr["count"], total_expected)
    asm.check_overflow()   # raises if a row did not fit its arena

    kernel_ms = [a.elapsed_time(b) for a, b in evs]
    avg_kernel_s = (sum(kernel_ms) / len(kernel_ms)) / 1e3

    # algorithmic bytes per launch: 2*(Lf+Lr) + 9*M + 64 per pair, M = merged length of accepted pairs
    res = batch.results
    seq_len = res[:, 24:26].contiguous().view(torch.int16).to(torch.int64).flatten() & 0xFFFF
    status = res[:, 38].to(torch.int64)
    merged_bases = int((seq_len * (status == 0)).sum().item())
    lens = int((batch.flen.to(torch.int64) & 0xFFFF).sum().item() + (batch.rlen.to(torch.int64) & 0xFFFF).sum().item())
    algo_bytes = 2 * lens + 9 * merged_bases + 64 * n
    achieved = algo_bytes / avg_kernel_s / 1e9

    return dict(cfg_index=cfg_index, n=n, steps=steps, warmup=warmup, elapsed=elapsed, kernel_ms_avg=avg_kernel_s * 1e3, algo_bytes=algo_bytes, achieved=achieved,
                ctr=ctr, asm=asm, spec=spec, batch=batch, arenas=arenas, n_chunks=n_chunks, chunk=chunk, value=n * world * steps / elapsed,
                launch=asm.last_launch_info(), lean_kernel=asm.last_lean_variant(), deferred=asm.last_deferred())


def main():
    args = parse()
    import numpy as np
    import torch

    import pandaseq_amd as pa
    from pandaseq_amd import _lib, synth
    from pandaseq_amd.stats import allreduce_counters, format_stat_block

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with that many ranks (WORLD_SIZE={world})")
    dist = None
    # test hook (tests/test_bench_gpu.py): run a 2-rank job on a 1-GPU box -- every rank on cuda:0,
    # counters reduced over gloo instead of RCCL.  Never set by the driver.
    single_device = os.environ.get("PAMD_BENCH_SINGLE_DEVICE") == "1"
    if single_device:
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if single_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")

    env = dict(pa=pa, synth=synth, dist=dist, dev=dev, rank=rank, world=world, local=local)
    cfg_index = args.config
    n = args.pairs or CONFIGS[cfg_index]["pairs"]
    meas = run_config(env, args, cfg_index, n, args.steps, args.warmup)
    batch, asm, ctr, spec, n_chunks, chunk, elapsed = meas["batch"], meas["asm"], meas["ctr"], meas["spec"], meas["n_chunks"], meas["chunk"], meas["elapsed"]
    algo_bytes, achieved, avg_kernel_s, L = meas["algo_bytes"], meas["achieved"], meas["kernel_ms_avg"] / 1e3, spec.read_len

    # HBM bytes per launch from the PMC passes (tools/traffic_probe.py -> profiles/r03_traffic_config*.json):
    # FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes, units calibrated on a 1 GiB copy in the same
    # process (gfx950: FETCH_SIZE counts 2 KiB per unit for wide reads).  Measured per pair on the same
    # workload; scaled to this launch's pair count.
    traffic = args.traffic_bytes
    traffic_src = "--traffic-bytes" if traffic is not None else None
    # (one file per measured workload: a scorer override, `-a` or another min_phred is a different kernel and gets no other's figure)
    variant = (f"_{args.algo}" if args.algo else "") + ("_after" if args.primers_after else "") + (f"_minphred{args.min_phred}" if cfg_index == 3 and args.min_phred != 2 else "")
    tname = tpath = None
    for rnd in ("r06", "r05", "r04", "r03"):
        cand = f"{rnd}_traffic_config{cfg_index}{variant}_{args.sums}.json"
        if os.path.exists(os.path.join(ROOT, "profiles", cand)):
            tname, tpath = cand, os.path.join(ROOT, "profiles", cand)
            break
    if traffic is None and tpath is not None:
        with open(tpath) as fh:
            tj = json.load(fh)
        traffic = tj["hbm_bytes_per_pair"] * n
        traffic_src = (f"profiles/{tname}: FETCH_SIZE + WRITE_SIZE of one {tj['pairs']}-pair launch of the build of {tj.get('build', 'round 3')}, per pair, "
                       f"scaled to this launch's {n} pairs (tools/refresh_profiles.sh re-measures it)")

    value = n * world * args.steps / elapsed
    out = {
        "metric": "merged read-pairs/sec",
        "value": value,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8+f64",
        "data": "synthetic",
        "config": {
            "workload": CONFIGS[cfg_index]["name"] + (f" [scorer overridden: {args.algo}]" if args.algo else "") + (" [-a]" if args.primers_after else "")
                        + (f" [min_phred:{args.min_phred} rejected {ctr['rejected'][0]}, min_overlapbits:15 rejected {ctr['rejected'][1]} of {ctr['count']} pairs]" if cfg_index == 3 else ""),
            "config_index": cfg_index,
            "pairs_per_gpu": n,
            "launches_per_step": n_chunks,
            "pairs_per_launch": chunk,
            "read_len": L,
            "algorithm": args.algo or CONFIGS[cfg_index]["algo"],
            "primers_after": bool(args.primers_after),
            "sharding": f"{world} contiguous shard(s), one RCCL all-reduce of STAT counters per step" if world > 1 else "single GPU",
            "output": "compact arena, nt u8 + log-p f64 per merged base" + ("" if n_chunks == 1 else f"; the shard's input is resident, its output goes to two arenas used in turn, {chunk} pairs per launch"),
            "sums": ("reference order (all outputs bit-identical)" if args.sums == "exact" else
                     "any order + exactness guards (decisions, bases, per-base log-p bit-identical; quality / est_overlap_prob within 1e-9)"),
            "launch": asm.last_launch_info(),
            "lean_kernel": asm.last_lean_variant(),     # which instantiation of pamd_fast_kernel served the launches ("general": none)
        },
        "stat": {k: ctr[k] for k in ("count", "ok", "noalgn", "lowq", "badr", "nofp", "norp", "slow")},
        # the STAT lines a reference worker prints at the end of its input (pool.c:86-104), from the counters of one step summed
        # over the ranks: the whole block as one checksum, so that an N-rank run can be held against a single run of the same pairs
        "stat_block_sha1": hashlib.sha1(format_stat_block(ctr, has_forward_primer=cfg_index == 2 and not args.primers_after, has_reverse_primer=cfg_index == 2 and not args.primers_after).encode()).hexdigest(),
        "roofline": {
            "bound": "hbm",
            "kernel": ("pamd_fast_kernel + pamd_assemble_kernel over the pairs it deferred" + (" (+ pamd_primer_kernel over those deferred before their primers were found)" if cfg_index == 2 and not args.primers_after else "") + ", all inside the HIP events" if asm.last_deferred() >= 0
                       else "pamd_primer_kernel + pamd_assemble_kernel (both inside the HIP events)" if cfg_index == 2 and not args.primers_after
                       else "pamd_assemble_kernel"),
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": algo_bytes,
            "bytes_per_pair": algo_bytes / n,
            "kernel_ms_avg": avg_kernel_s * 1e3,
        },
    }
    if ctr["rejected"][0] or ctr["rejected"][1]:
        out["stat"]["rejected"] = ctr["rejected"][:2]
    # which build was measured (the GPU box has no .git: tools/refresh_profiles.sh writes the commit next to the sources)
    build_id = os.environ.get("PAMD_BUILD")
    if not build_id and os.path.exists(os.path.join(ROOT, ".build_id")):
        build_id = open(os.path.join(ROOT, ".build_id")).read().strip()
    if build_id:
        out["build"] = build_id

    # ---- CPU baseline: the oracle (a scalar C port of the reference path) on this host's cores ----
    # pool.c:140-167 gives every thread an assembler of its own over its share of the input; so does this leg (po_bench_mt:
    # pinned pthreads, each over a private copy of its slice that it allocated and first touched itself), at 1, 8, 32, 64,
    # 128 and all CPUs of the affinity mask.  `value` is the best of them, `cores` the thread count that reached it.
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import pyoracle as po
        affinity = len(os.sched_getaffinity(0))
        quota = None
        try:
            q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            quota = None if q == "max" else float(q) / float(period)
        except (OSError, ValueError):
            pass
        effective = min(affinity, quota) if quota else affinity
        ocfg = oracle_config(po, synth, cfg_index, args.algo, args.primers_after, args.min_phred)
        pool_n = int(min(n, 4_000_000))
        hf = batch.fwd[:pool_n].cpu().numpy()
        hr = batch.rev[:pool_n].cpu().numpy()
        hfl = batch.flen[:pool_n].cpu().numpy().view(np.uint16)
        hrl = batch.rlen[:pool_n].cpu().numpy().view(np.uint16)
        probe = min(20000, pool_n)
        secs, _, _ = po.bench_mt(ocfg, hf[:probe], hr[:probe], hfl[:probe], hrl[:probe], 1)
        rate1 = probe / secs
        ladder = sorted({t for t in (1, 8, 32, 64, 128, affinity) if t <= affinity})
        per_run = args.cpu_seconds / len(ladder)
        scaling, best, o_res, sample = [], None, None, 0
        for t in ladder:
            m = int(min(pool_n, max(probe, rate1 * min(t, effective) * per_run)))
            secs, res_t, ctr_t = po.bench_mt(ocfg, hf[:m], hr[:m], hfl[:m], hrl[:m], t)
            rate = m / secs
            scaling.append({"threads": t, "pairs": m, "seconds": secs, "pairs_per_s": rate, "efficiency_vs_1": rate / (rate1 * t)})
            if best is None or rate > best["pairs_per_s"]:
                best = scaling[-1]
            if m > sample:
                o_res, sample = res_t, m
        # The largest sample doubles as an at-scale parity check -- of ROWS, not only of records: for every pair of the sample the
        # device's status, overlap, merged length, mismatches and two position-weighted checksums of its merged row in the
        # 50 M-pair arena (sum (j+1) nt_j and sum (j+1) bits(p_j), wrapping 64-bit integers: exact in any order; SURVEY Appendix B's
        # checksum made bit-exact) against the oracle's (po_row_checksums_mt: a second, untimed pass that keeps the rows).
        dres = batch.results[:sample].cpu().numpy().view(_lib.RESULT_DTYPE).reshape(-1)
        ok_o = o_res["status"] == 0
        same_records = bool((dres["status"] == o_res["status"]).all() and all((dres[k][ok_o] == o_res[k][ok_o]).all() for k in ("overlap", "seq_len", "mismatches", "degenerates", "examined")))
        c0 = time.perf_counter()
        o_res2, o_cs_nt, o_cs_p = po.row_checksums(ocfg, hf[:sample], hr[:sample], hfl[:sample], hrl[:sample], threads=int(max(1, min(affinity, 2 * effective))))
        d_cs_nt, d_cs_p = pa.row_checksums(batch.results, batch.seq_nt, batch.seq_p, first=0, count=sample)
        same_rows = bool(np.array_equal(d_cs_nt.cpu().numpy().view(np.uint64), o_cs_nt) and np.array_equal(d_cs_p.cpu().numpy().view(np.uint64), o_cs_p)
                         and (o_res2["status"] == o_res["status"]).all()) if n_chunks == 1 else None   # (a shard streamed through two arenas no longer holds the first rows)
        rows_seconds = time.perf_counter() - c0
        same = same_records and same_rows is not False
        eff_at_cores = best["pairs_per_s"] / (rate1 * min(best["threads"], effective))
        out["cpu_baseline"] = {
            "value": best["pairs_per_s"],
            "unit": "pairs/s",
            "cores": int(min(best["threads"], effective)),   # CPUs the process may use (cgroup quota / affinity) ...
            "threads": best["threads"],                     # ... and the pthreads that reached the best rate on them
            "kind": "port",
            "sample": (f"first {best['pairs']} pairs of the same batch, oracle C port, {best['threads']} pinned pthreads, one assembler each over a private, "
                       "self-touched copy of its contiguous slice (pool.c:140-167), in memory (no FASTQ parsing), merged sequence discarded; "
                       "clock from all-threads-ready to all-done"),
            "single_thread_pairs_per_s": rate1,
            "threads_scaling": scaling,
            "effective_cores": {"sched_getaffinity": affinity, "cgroup_cpu_max": quota, "os_cpu_count": os.cpu_count(), "used_for_efficiency": effective},
            "parallel_efficiency_at_cores": eff_at_cores,
            "note": ("efficiency = rate / (single-thread rate x min(threads, effective cores)); SMT siblings share a core's pipelines, so "
                     "the upper half of the ladder (threads > physical cores) adds well under 1x per thread"),
            "matches_gpu_on_sample": same,
            "sample_check": {"pairs": sample, "records": same_records, "rows": same_rows, "accepted_rows": int(ok_o.sum()),
                             "what": "status of every pair; overlap, merged length, mismatches, degenerates, examined of every accepted pair; per accepted row "
                                     "sum (j+1) nt_j and sum (j+1) bits(log-p_j) in wrapping 64-bit arithmetic, device rows in the arena against the oracle's",
                             "seconds": rows_seconds},
        }
    # ---- the other BASELINE.json configurations' single-GPU shards, in the same record (N=1, default configuration only) ----
    if world == 1 and cfg_index == 1 and (not args.pairs or args.other_pairs) and not args.algo and not args.primers_after and not args.no_other_configs:
        batch = asm = None
        meas.clear()
        torch.cuda.empty_cache()
        others = {}
        for oc in (2, 3, 4):
            try:
                r = run_config(env, args, oc, args.other_pairs or CONFIGS[oc]["pairs"], 3, 1, headline=False)
                octr = r["ctr"]
                others[f"config{oc}"] = {
                    "workload": CONFIGS[oc]["name"], "pairs_per_gpu": r["n"], "steps": 3, "warmup": 1, "pairs_per_s": r["value"], "ms_per_step": r["elapsed"] / 3 * 1e3,
                    "kernel_ms": r["kernel_ms_avg"], "frac": r["achieved"] / HBM_PEAK_GBPS, "achieved_GBps": r["achieved"], "bytes_per_pair": r["algo_bytes"] / r["n"],
                    "lean_kernel": r["lean_kernel"], "launch": r["launch"], "launches_per_step": r["n_chunks"], "deferred_pairs": r["deferred"],
                    "stat": {k: octr[k] for k in ("count", "ok", "noalgn", "lowq", "badr", "nofp", "norp", "slow")}, "rejected": octr["rejected"][:2],
                }
                r.clear()
            except BaseException as exc:   # (SystemExit of a shard that does not fit included: the headline line does not depend on this leg)
                others[f"config{oc}"] = {"error": str(exc)[:300]}
            torch.cuda.empty_cache()
        others["note"] = ("each: the configuration's single-GPU shard resident in HBM, 1 warmup + 3 timed steps by the same code as the headline line "
                          "(run_config), " + ("sums in the reference's order" if args.sums == "exact" else "any-order sums") + "; frac = algorithmic bytes / HIP-event kernel time / 8 TB/s")
        out["other_configs"] = others
    # ---- end to end through the drop-in API (never `value`): the reference's loop, PCIe both ways, plain C ----
    if rank == 0 and world == 1 and not args.no_e2e and cfg_index in (1, 3, 4):
        batch = asm = None
        meas.clear()
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_next
        try:
            e2e = {}
            keys = ("pairs", "seconds", "pairs_per_s", "ok", "batch_pairs", "distinct_pairs", "api", "rows")
            for name, consume, rows in (("record_only", 0, "eager"), ("every_base_read", 2, "eager"), ("every_base_read_rows_on_callers_thread", 2, "lazy")):
                # (record_only over a million distinct pairs: a cycled source of a few thousand would sit in the caller's cache)
                d = bench_next.run(config=cfg_index, pairs=args.e2e_pairs if consume == 0 else args.e2e_pairs // 4, sums=args.sums, consume=consume, rows=rows, distinct=1 << 20 if consume == 0 else 8192)
                e2e[name] = {k: d[k] for k in keys}
            # the reference's own deployment: one assembler per thread (pool.c:140-167), every thread reading every base
            pool_threads = max(2, min(8, len(os.sched_getaffinity(0)) // 2))
            d = bench_next.run_pool(config=cfg_index, pairs=(args.e2e_pairs // 2) * pool_threads, sums=args.sums, threads=pool_threads)
            e2e["every_base_read_one_assembler_per_thread"] = {k: d[k] for k in keys + ("threads",)}
            # the per-pair entry point (what panda_diff calls, diff.c:26-99) and its array form
            d = bench_next.run(config=cfg_index, pairs=min(400_000, args.e2e_pairs), sums=args.sums, consume=2, api=1, distinct=1 << 16)
            e2e["panda_assembler_assemble_per_pair_host"] = dict({k: d[k] for k in keys}, device=False)   # (a CPU thread: the per-pair entry point runs on the host path)
            d = bench_next.run(config=cfg_index, pairs=args.e2e_pairs // 4, sums=args.sums, consume=2, api=2)
            e2e["pamd_compat_assemble_many"] = {k: d[k] for k in keys}
            e2e["unit"] = "pairs/s"
            e2e["path"] = ("tests/c/bench_next.c: while ((r = panda_assembler_next(a)) != NULL) over an in-memory PandaNextSeq source; pinned staging, four "
                           "batches in flight, log-p downloaded as u16 table indices, panda_result[] rows expanded by helper threads (rows: eager) or on the "
                           "caller's thread as each record is handed over (rows: lazy, PAMD_COMPAT_ROWS); record_only: the consumer reads each record's "
                           "scalars; every_base_read: it reads nt and p of every merged base (16 B per base on its one thread); ..._one_assembler_per_thread: "
                           "tests/c/bench_pool.c, the same loop in `threads` threads with an assembler each, as pool.c:140-167 runs it; the last two entries: the "
                           "caller holds the pairs and calls panda_assembler_assemble() per pair (assembled on its own thread through the class's two functions, "
                           "pamd_host_scorers.cpp, as assembler.c:368-383 does) / pamd_compat_assemble_many() per 65536 pairs (the device)")
            out["end_to_end"] = e2e
        except Exception as exc:  # the headline number does not depend on this leg
            out["end_to_end"] = {"error": str(exc)[:300]}
        batch = None
    # ---- optional: the data-format rows either side of the path, with the oracle's reader / formatter as CPU baseline ----
    if rank == 0 and world == 1 and args.io:
        batch = None
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_io
        io_args = argparse.Namespace(pairs=args.io_pairs, config=cfg_index, steps=3, out=None)
        io_out, ctx = bench_io.run(io_args)
        if not args.no_cpu:
            from oracle import pyoracle as po
            m = min(200_000, ctx["base_n"])
            ft, rt = ctx["ft"], ctx["rt"]
            cut_f, cut_r = ft[:len(ft) // ctx["base_n"] * m], rt[:len(rt) // ctx["base_n"] * m]
            c0 = time.perf_counter()
            pf, pr, pfl, prl, oids, ost = po.fastq_read(cut_f, cut_r, stride=ctx["stride"], max_pairs=m)
            cpu_parse = time.perf_counter() - c0
            same = bool(ost.n_pairs == m and np.array_equal(pf, ctx["rows"]["fwd"][:m].cpu().numpy()) and np.array_equal(pr, ctx["rows"]["rev"][:m].cpu().numpy()))
            res_io, okm = ctx["res"], ctx["okm"]
            k = 0
            c0 = time.perf_counter()
            for i in range(min(m, 50_000)):
                if okm[i]:
                    po.output_fasta(oids[i], float(res_io["quality"][i]), np.ones(int(res_io["seq_len"][i]), dtype=np.uint8))
                    k += 1
            cpu_fmt = time.perf_counter() - c0
            io_out["cpu_baseline"] = {
                "ingest_pairs_per_s": m / cpu_parse, "format_fasta_pairs_per_s": k / cpu_fmt if cpu_fmt > 0 else None, "cores": 1, "kind": "port",
                "sample": f"first {m} pairs of the same text through the oracle's reader (one thread, in memory); {k} results through its FASTA formatter (ctypes call per pair)",
                "matches_gpu_on_sample": same,
            }
        out["io"] = io_out
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
