export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT
export PAMD_BUILD=$(cat .build_id)
traffic() {
	export PAMD_CONFIG=$1 PAMD_SUMS=$2 PAMD_PAIRS=$3
	T=$1$4_$2
	rm -rf "$OUT/traffic_rd_$T" "$OUT/traffic_wr_$T"
	(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/traffic_rd_$T" -o t -- python3 "$ROOT/tools/traffic_probe.py" > "$OUT/traffic_rd_$T.log" 2>&1)
	(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/traffic_wr_$T" -o t -- python3 "$ROOT/tools/traffic_probe.py" > "$OUT/traffic_wr_$T.log" 2>&1)
	python3 profiles/summarize_traffic.py $(find "$OUT/traffic_rd_$T" -name '*_results.db' | head -1) $(find "$OUT/traffic_wr_$T" -name '*_results.db' | head -1) $3 > "$OUT/traffic_config$T.json"
}
traffic 2 fast 20000000
export PAMD_PRIMERS_AFTER=1
traffic 2 fast 20000000 _after
unset PAMD_PRIMERS_AFTER
grep hbm_ $OUT/traffic_config2_fast.json $OUT/traffic_config2_after_fast.json
python3 tools/pipe_file_rate.py --out "$OUT/pipe_file_rate.json" 2>&1 | tail -2
python3 tools/pipe_file_rate.py --pairs 4000000 --chunk 2000000 2>&1 | tail -1
