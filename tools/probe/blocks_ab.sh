#!/bin/bash
# Strided pairs (PAMD_FK_BLOCKS=0) against pairs in blocks (=1), the same library, one box: full-size bench lines, then the L2 <-> fabric
# traffic passes of tools/traffic_probe.py.     bash tools/probe/blocks_ab.sh "<configs>"        (development aid, GPU box only)
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/blocks_ab; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
CONFIGS=${1:-"1 4"}
run() { PAMD_FK_BLOCKS=$1 python bench.py "${@:2}" --steps 3 --warmup 1 --no-cpu --no-e2e --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['roofline']['kernel_ms_avg'],2))"; }
for rep in 1 2; do for c in $CONFIGS; do
	echo "bench config $c: strided $(run 0 --config $c) | blocks $(run 1 --config $c)"
	echo "bench config $c exact: strided $(run 0 --config $c --sums exact) | blocks $(run 1 --config $c --sums exact)"
done; done
[ -n "$NO_TRAFFIC" ] && exit 0
for which in 0 1; do
	export PAMD_FK_BLOCKS=$which
	for c in $CONFIGS; do
		export PAMD_CONFIG=$c PAMD_SUMS=fast PAMD_PAIRS=$([ $c = 4 ] && echo 40000000 || echo 50000000)
		(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/rd_$which$c" -o t -- python3 "$ROOT/tools/traffic_probe.py" > "$OUT/rd_$which$c.log" 2>&1)
		(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/wr_$which$c" -o t -- python3 "$ROOT/tools/traffic_probe.py" > "$OUT/wr_$which$c.log" 2>&1)
		python3 profiles/summarize_traffic.py $(find "$OUT/rd_$which$c" -name '*_results.db' | head -1) $(find "$OUT/wr_$which$c" -name '*_results.db' | head -1) $PAMD_PAIRS > "$OUT/traffic_$which$c.json"
		python3 -c "import json,sys; d=json.load(open(sys.argv[1])); n=d['pairs']; print('blocks', sys.argv[2], 'config', sys.argv[3], 'read B/pair', round(d['hbm_read_bytes']/n,1), 'write B/pair', round(d['hbm_write_bytes']/n,1), 'total', round(d['hbm_bytes_per_pair'],1))" "$OUT/traffic_$which$c.json" $which $c
		rm -rf "$OUT/rd_$which$c" "$OUT/wr_$which$c"
	done
done
