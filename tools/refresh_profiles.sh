#!/bin/bash
# Everything profiles/ holds for one build, in one go on the GPU box (writes gpurun_out/<tag>/; copy what is to be
# kept into profiles/):   bash tools/refresh_profiles.sh r02
# bench lines of configs 1 (fast sums with the CPU baseline and the end-to-end leg; exact sums), 2, 3, 4; rocprofv3
# --kernel-trace --stats of the bench command without its `other_configs` leg (config 1 alone: the kernels of configs 2-4 have their own lines); SQ counters and the per-phase ablation of the lean kernel, SQ counters of
# config 2's two kernels; the HBM
# traffic passes of tools/traffic_probe.py (FETCH_SIZE and WRITE_SIZE in separate passes, calibrated on a 1 GiB copy);
# the other scorers, `-a`, and the data-format rows (bench.py --io, rocprofv3 of tools/bench_io.py).
TAG=${1:-run}
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
# which build this is: the GPU box has no .git, so the commit is written next to the sources before the trip
# (git rev-parse --short HEAD > .build_id)
export PAMD_BUILD=$(cat .build_id 2>/dev/null || git rev-parse --short HEAD 2>/dev/null || echo unknown)
python3 bench.py --steps 5 --warmup 1 2> "$OUT/bench_config1_fast.err" | tail -1 > "$OUT/bench_config1_fast.json"
python3 bench.py --steps 5 --warmup 1 --sums exact --no-cpu --no-e2e --no-other-configs 2> /dev/null | tail -1 > "$OUT/bench_config1_exact.json"
for c in 2 3 4; do python3 bench.py --steps 3 --warmup 1 --config $c --no-cpu --no-e2e 2> /dev/null | tail -1 > "$OUT/bench_config$c.json"; done
for c in 2 3 4; do python3 bench.py --steps 3 --warmup 1 --config $c --sums exact --no-cpu --no-e2e 2> /dev/null | tail -1 > "$OUT/bench_config${c}_exact.json"; done
python3 bench.py --steps 3 --warmup 1 --config 2 --primers-after --sums exact --no-cpu --no-e2e 2> /dev/null | tail -1 > "$OUT/bench_config2_primers_after_exact.json"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o t -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu --no-e2e --no-other-configs > "$OUT/prof.log" 2>&1)
python3 profiles/summarize_rocprof.py --like '%pamd_%' $(find "$OUT/prof" -name '*_results.db' | head -1) > "$OUT/rocprof.txt"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof2" -o t -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --config 2 --no-cpu --no-e2e --no-other-configs > "$OUT/prof2.log" 2>&1)
python3 profiles/summarize_rocprof.py --like '%pamd_%' $(find "$OUT/prof2" -name '*_results.db' | head -1) > "$OUT/rocprof_config2.txt"
traffic() {	# config sums pairs [tag]
	export PAMD_CONFIG=$1 PAMD_SUMS=$2 PAMD_PAIRS=$3
	T=$1$4_$2
	rm -rf "$OUT/traffic_rd_$T" "$OUT/traffic_wr_$T"
	(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/traffic_rd_$T" -o t -- python3 "$ROOT/tools/traffic_probe.py" > "$OUT/traffic_rd_$T.log" 2>&1)
	(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/traffic_wr_$T" -o t -- python3 "$ROOT/tools/traffic_probe.py" > "$OUT/traffic_wr_$T.log" 2>&1)
	python3 profiles/summarize_traffic.py $(find "$OUT/traffic_rd_$T" -name '*_results.db' | head -1) $(find "$OUT/traffic_wr_$T" -name '*_results.db' | head -1) $3 > "$OUT/traffic_config$T.json"
}
traffic 1 fast 50000000
traffic 1 exact 50000000
traffic 2 fast 20000000
traffic 3 fast 20000000
traffic 4 fast 40000000
# `-a` on config 2 (its own kernel instantiation: its own traffic file, bench.py looks for the _after name)
export PAMD_PRIMERS_AFTER=1
traffic 2 fast 20000000 _after
bash tools/pmc_fk.sh 8000000 2 > "$OUT/pmc_config2_after_fast.txt" 2>&1
unset PAMD_PRIMERS_AFTER
for c in 1 2 3 4; do bash tools/pmc_fk.sh 8000000 $c > "$OUT/pmc_config${c}_fast.txt" 2>&1; done
# what one SIMD issues per cycle, by instruction class and waves per SIMD (tools/probe/valu_issue.hip; build it before the trip)
[ -x tools/probe/valu_issue ] && timeout 400 tools/probe/valu_issue 2000 > "$OUT/valu_issue.txt" 2>&1
# the per-phase ablation wants six more builds of the library: make them BEFORE going to the GPU box (bash tools/fk_phases.sh build);
# built there only if they did not travel
ls "$ROOT"/pandaseq_amd/libpandaseq_amd_fkstop5.so > /dev/null 2>&1 || bash tools/fk_phases.sh build > "$OUT/phases_build.log" 2>&1
bash tools/fk_phases.sh 8000000 1 > "$OUT/phases_config1_fast.txt" 2>&1
bash tools/fk_phases.sh 8000000 2 > "$OUT/phases_config2_fast.txt" 2>&1
PAMD_PRIMERS_AFTER=1 bash tools/fk_phases.sh 8000000 2 > "$OUT/phases_config2_after_fast.txt" 2>&1
# the other scorers on config-1 data, primers after assembly, and the data-format rows (not bench lines: DESIGN.md sections 8 and 9)
for a in uparse flash ea_util stitch pear rdp_mle; do python3 bench.py --algo $a --no-cpu --no-e2e --steps 3 --warmup 1 2> /dev/null | tail -1 > "$OUT/bench_algo_$a.json"; done
python3 bench.py --config 2 --primers-after --no-cpu --no-e2e --steps 3 --warmup 1 2> /dev/null | tail -1 > "$OUT/bench_config2_primers_after.json"
# config 3 with a min_phred value that rejects (the synthetic qualities know no value between "nothing" and "96 %")
python3 bench.py --config 3 --min-phred 3 --no-cpu --no-e2e --steps 3 --warmup 1 2> /dev/null | tail -1 > "$OUT/bench_config3_minphred3.json"
# PCIe-inclusive: text -> text through pinned host memory, file -> file through the plain-C pipe, the per-pair entry point
python3 tools/pipeline_pcie.py --out "$OUT/pipeline_pcie.json" > "$OUT/pipeline_pcie.log" 2>&1
python3 tools/pipe_file_rate.py --pairs 64000000 --reps 2 --out "$OUT/pipe_file_rate.json" > "$OUT/pipe_file_rate.log" 2>&1
python3 tools/pipe_file_rate.py --pairs 64000000 --reps 2 --sink /dev/null --out "$OUT/pipe_null_rate.json" > "$OUT/pipe_null_rate.log" 2>&1
# the same run dealt to two workers on this box's one GPU (pamd_pipe_run_multi: chunk boundaries by counting lines on the host; both share one PCIe link)
PAMD_PIPE_DEVICES=0,0 python3 tools/pipe_file_rate.py --pairs 64000000 --reps 2 --sink /dev/null --out "$OUT/pipe_null_rate_multi.json" > "$OUT/pipe_null_rate_multi.log" 2>&1
python3 tools/pipe_file_rate.py --pairs 8000000 --reps 2 --gzip --sink /dev/null --out "$OUT/pipe_gzip_rate.json" > "$OUT/pipe_gzip_rate.log" 2>&1
python3 bench.py --io --no-e2e --cpu-seconds 0 --steps 3 --warmup 1 2> /dev/null | tail -1 | python3 -c "import json,sys; json.dump(json.load(sys.stdin)['io'], open(sys.argv[1], 'w'), indent=1)" "$OUT/io_bench.json"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_io" -o t -- python3 "$ROOT/tools/bench_io.py" --pairs 4000000 > "$OUT/prof_io.log" 2>&1)
python3 profiles/summarize_rocprof.py --like '%f%_kernel%' $(find "$OUT/prof_io" -name '*_results.db' | head -1) > "$OUT/io_rocprof.txt"
for f in "$OUT"/bench_*.json; do python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']/1e6,1), 'M pairs/s', round(d['roofline']['frac'],4))" $f; done
for f in "$OUT"/traffic_config*.json; do python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['hbm_bytes_per_pair'],1), 'B/pair')" $f; done
