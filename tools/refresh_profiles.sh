#!/bin/bash
# Everything profiles/ holds for one build, in one go on the GPU box (writes gpurun_out/<tag>/; copy what is to be
# kept into profiles/):   bash tools/refresh_profiles.sh r01_s5
# bench lines of configs 1 (fast sums with the CPU baseline, exact sums), 2, 3, 4; rocprofv3 --kernel-trace --stats
# of the default bench command; the HBM traffic passes of tools/traffic_probe.py for both summation modes.
TAG=${1:-run}
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py --steps 5 --warmup 1 2> "$OUT/bench_config1_fast.err" | tail -1 > "$OUT/bench_config1_fast.json"
python3 bench.py --steps 5 --warmup 1 --sums exact --no-cpu 2> /dev/null | tail -1 > "$OUT/bench_config1_exact.json"
for c in 2 3 4; do python3 bench.py --steps 3 --warmup 1 --config $c --no-cpu 2> /dev/null | tail -1 > "$OUT/bench_config$c.json"; done
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o t -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu > "$OUT/prof.log" 2>&1)
for m in fast exact; do
	export PAMD_SUMS=$m PAMD_PAIRS=50000000
	(cd /tmp && rocprofv3 --pmc FETCH_SIZE -d "$OUT/traffic_rd_$m" -o t -- python3 "$ROOT/tools/traffic_probe.py" > "$OUT/traffic_rd_$m.log" 2>&1)
	(cd /tmp && rocprofv3 --pmc WRITE_SIZE -d "$OUT/traffic_wr_$m" -o t -- python3 "$ROOT/tools/traffic_probe.py" > "$OUT/traffic_wr_$m.log" 2>&1)
	python3 profiles/summarize_traffic.py $(find "$OUT/traffic_rd_$m" -name '*_results.db' | head -1) $(find "$OUT/traffic_wr_$m" -name '*_results.db' | head -1) 50000000 > "$OUT/traffic_config1_$m.json"
done
python3 profiles/summarize_rocprof.py $(find "$OUT/prof" -name '*_results.db' | head -1) > "$OUT/rocprof.txt"
for f in "$OUT"/bench_*.json; do python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']/1e6,1), 'M pairs/s', round(d['roofline']['frac'],4))" $f; done
tail -3 "$OUT/traffic_config1_fast.json"
