#!/bin/bash
# Several builds of the library on one box (development aid; boxes differ by a few per cent, builds often by less):
#   bash tools/abn_probe.sh "<lib suffixes, '' = the built library>" [pairs] [configs...]
# e.g. bash tools/abn_probe.sh "_B _X1 _X2" 8000000 1 4   runs pandaseq_amd/libpandaseq_amd{,_B,_X1,_X2}.so
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIBS=$1; shift
PAIRS=${1:-8000000}; shift
CONFIGS=${@:-1}
for c in $CONFIGS; do
	for rep in 1 2; do
		echo "config $c (built): $(python3 $ROOT/tools/fk_probe.py --config $c --pairs $PAIRS --only two-kernel 2>&1 | tail -1 | cut -c1-60)"
		for l in $LIBS; do
			echo "config $c $l: $(PAMD_LIB=$ROOT/pandaseq_amd/libpandaseq_amd$l.so python3 $ROOT/tools/fk_probe.py --config $c --pairs $PAIRS --only two-kernel 2>&1 | tail -1 | cut -c1-60)"
		done
	done
done
