#!/bin/bash
# A/B on one box: the built library against pandaseq_amd/libpandaseq_amd_B.so (development aid; boxes differ by a few per cent)
#   bash tools/ab_probe.sh [pairs] [configs...]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PAIRS=${1:-8000000}; shift
CONFIGS=${@:-1 2}
for c in $CONFIGS; do
	for rep in 1 2; do
		echo "config $c A: $(python3 $ROOT/tools/fk_probe.py --config $c --pairs $PAIRS --only two-kernel 2>&1 | tail -1 | cut -c1-60)"
		echo "config $c B: $(PAMD_LIB=$ROOT/pandaseq_amd/libpandaseq_amd_B.so python3 $ROOT/tools/fk_probe.py --config $c --pairs $PAIRS --only two-kernel 2>&1 | tail -1 | cut -c1-60)"
	done
done
