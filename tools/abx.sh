#!/bin/bash
# Several builds of the library on one box, correctness first (lean + general against general alone), then the rate, interleaved:
#   bash tools/abx.sh "<lib suffixes>" [pairs] [configs...]      (development aid; GPU box only)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIBS=$1; shift
PAIRS=${1:-8000000}; shift
CONFIGS=${@:-1}
for l in $LIBS; do
	for c in $CONFIGS; do
		echo "check $l config $c: $(PAMD_LIB=$ROOT/pandaseq_amd/libpandaseq_amd$l.so python3 $ROOT/tools/fk_probe.py --config $c --pairs 2000000 2>&1 | grep -E "counters equal|records equal" | tr '\n' ' ')"
	done
done
for c in $CONFIGS; do
	for rep in 1 2 3; do
		for l in $LIBS; do
			echo "config $c $l: $(PAMD_LIB=$ROOT/pandaseq_amd/libpandaseq_amd$l.so python3 $ROOT/tools/fk_probe.py --config $c --pairs $PAIRS --only two-kernel --reps 5 2>&1 | tail -1 | cut -c1-60)"
		done
	done
done
