#!/bin/bash
# full-size bench lines of two builds on one box: bash ab_full.sh "<suffixes>"
ROOT=$(pwd)
run() { PAMD_LIB=$ROOT/pandaseq_amd/libpandaseq_amd$1.so python bench.py "${@:2}" --steps 3 --warmup 1 --no-cpu --no-e2e 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['roofline']['kernel_ms_avg'],2))"; }
for rep in 1 2; do for l in $1; do
 [ "$l" = built ] && s="" || s=$l
 echo "lib$s cfg1 $(run "$s" --config 1) | exact $(run "$s" --config 1 --sums exact) | cfg2 $(run "$s" --config 2) | cfg2-a $(run "$s" --config 2 --primers-after) | cfg3 $(run "$s" --config 3) | cfg4 $(run "$s" --config 4)"
done; done
