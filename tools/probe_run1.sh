export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r05a; mkdir -p $OUT
export PAMD_CONFIG=1 PAMD_SUMS=exact PAMD_PAIRS=50000000
T=1_exact
rm -rf "$OUT/traffic_rd_$T" "$OUT/traffic_wr_$T"
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/traffic_rd_$T" -o t -- python3 "$ROOT/tools/traffic_probe.py" > "$OUT/traffic_rd_$T.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/traffic_wr_$T" -o t -- python3 "$ROOT/tools/traffic_probe.py" > "$OUT/traffic_wr_$T.log" 2>&1)
python3 profiles/summarize_traffic.py $(find "$OUT/traffic_rd_$T" -name '*_results.db' | head -1) $(find "$OUT/traffic_wr_$T" -name '*_results.db' | head -1) 50000000 > "$OUT/traffic_config$T.json"
cat $OUT/traffic_config$T.json
rm -rf "$OUT/traffic_rd_$T" "$OUT/traffic_wr_$T"
