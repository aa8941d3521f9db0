#!/usr/bin/env python3
"""Turn rocprofv3's rocpd sqlite output (gpurun_out/<dir>/*_results.db) into a small text summary
that can be committed under profiles/.  Usage: summarize_rocprof.py [--like PATTERN] <results.db> [<results.db> ...]
PATTERN: SQL LIKE pattern of the kernels whose dispatches and counters are listed (default %assemble%)."""
import sqlite3
import sys

argv = sys.argv[1:]
LIKE = "%assemble%"
if argv and argv[0] == "--like":
    LIKE = argv[1]
    argv = argv[2:]

for db in argv:
    con = sqlite3.connect(db)
    print(f"== {db}")
    print("-- kernel-trace stats (rocprofv3 --kernel-trace --stats): name, calls, total_us, avg_us, pct")
    try:
        for r in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc limit 12"):
            print(f"{r[0][:90]:90s} calls={r[1]:5d} total_us={r[2]:.0f} avg_us={r[3]:.0f} pct={r[4]:.2f}")
    except sqlite3.Error as e:
        print("   (no kernel stats)", e)
    print("-- per-dispatch durations of the selected kernels (ns)")
    try:
        rows = list(con.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count from kernels where name like ? order by start", (LIKE,)))
        for r in rows:
            print(f"   {r[0][:60]:60s} dur_ns={r[2]-r[1]} grid={r[3]} wg={r[4]} lds={r[5]} vgpr={r[6]} sgpr={r[7]}")
        # the stats table above averages EVERY dispatch of a kernel, bench.py's small start-up launch included: the timed
        # launches are the full-size ones
        by = {}
        for r in rows:
            by.setdefault(r[0], []).append(r[2] - r[1])
        print("-- full-size dispatches (at least half the longest) of the selected kernels: name, n, avg_us")
        for name, durs in by.items():
            full = [d for d in durs if 2 * d >= max(durs)]
            print(f"   {name[:60]:60s} n={len(full):3d} avg_us={sum(full) / len(full) / 1e3:.0f}")
    except sqlite3.Error as e:
        print("   (n/a)", e)
    print("-- PMC counters summed over the selected kernels' dispatches: kernel, counter, dispatches, sum")
    try:
        for r in con.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection where kernel_name like ? group by kernel_name, counter_name", (LIKE,)):
            print(f"   {r[0][:48]:48s} {r[1]:28s} n={r[2]:3d} sum={r[3]:.0f}")
    except sqlite3.Error as e:
        print("   (no counters)", e)
