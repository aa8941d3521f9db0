#!/bin/bash
# A handful of SQ counters of the lean kernel for several builds of the library on one box (development aid; GPU box only):
#   bash tools/pmc_ab.sh "<lib suffixes>" [pairs] [config]
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIBS=$1
PAIRS=${2:-8000000}
CONFIG=${3:-1}
for l in $LIBS; do
	OUT=$ROOT/gpurun_out/pmc_ab$l
	rm -rf "$OUT"; mkdir -p "$OUT"
	export PAMD_LIB=$ROOT/pandaseq_amd/libpandaseq_amd$l.so
	(cd /tmp && timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_WAIT_INST_ANY --output-format csv -d "$OUT/g1" -o p -- python3 "$ROOT/tools/fk_probe.py" --only two-kernel --pairs $PAIRS --config $CONFIG --sums ${SUMS:-fast} --reps 1 > "$OUT/g1.log" 2>&1)
	python3 - "$OUT" "$PAIRS" "$l" <<'PY'
import csv, glob, sys
out, pairs, lib = sys.argv[1], float(sys.argv[2]), sys.argv[3]
d = {}
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "pamd_fast_kernel" in row.get("Kernel_Name", ""):
            d[row["Counter_Name"]] = float(row["Counter_Value"])
print(lib, "  ".join(f"{c[3:]} {d[c] / pairs:.1f}" for c in sorted(d)))
PY
done
