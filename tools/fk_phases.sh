#!/bin/bash
# Per-phase cost of the lean kernel by ablation: builds that abandon every pair after phase k (-DFK_STOP_AFTER=k), each
# under one rocprofv3 counter pass.  Development aid.  Build here (no GPU needed):  bash tools/fk_phases.sh build
# then on the GPU box:  bash tools/fk_phases.sh [pairs]
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
	for k in 5 0 1 2 3 4; do
		make -s -C "$ROOT/pandaseq_amd/csrc" OUT=../libpandaseq_amd_fkstop$k.so EXTRA=-DFK_STOP_AFTER=$k &
	done
	wait
	exit 0
fi
PAIRS=${1:-8000000}
CONFIG=${2:-1}
OUT=$ROOT/gpurun_out/fk_phases
rm -rf "$OUT"; mkdir -p "$OUT"
for k in 5 0 1 2 3 4 full; do
	if [ $k = full ]; then unset PAMD_LIB; else export PAMD_LIB=$ROOT/pandaseq_amd/libpandaseq_amd_fkstop$k.so; fi
	(cd /tmp && timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/s$k" -o p -- python3 "$ROOT/tools/fk_probe.py" --only two-kernel --config $CONFIG --pairs $PAIRS --reps 1 > "$OUT/s$k.log" 2>&1)
	(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t$k" -o p -- python3 "$ROOT/tools/fk_probe.py" --only two-kernel --config $CONFIG --pairs $PAIRS --reps 3 > "$OUT/t$k.log" 2>&1)
done
python3 - "$OUT" "$PAIRS" <<'PY'
import csv, glob, sys
from collections import defaultdict
names = ["stage+loop", "primers", "seed-fwd", "seed-rev", "cand+scan", "merge", "rowout+sum"]
n = float(sys.argv[2])
prev = defaultdict(float)
for i, k in enumerate(["5", "0", "1", "2", "3", "4", "full"]):
    acc = {}
    for f in glob.glob(sys.argv[1] + "/s%s/**/*counter_collection.csv" % k, recursive=True):
        for row in csv.DictReader(open(f)):
            if "pamd_fast_kernel" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]] = float(row["Counter_Value"])
    for f in glob.glob(sys.argv[1] + "/t%s/**/*kernel_stats.csv" % k, recursive=True):
        for row in csv.DictReader(open(f)):
            if "pamd_fast_kernel" in row.get("Name", ""):
                acc["kernel_us"] = float(row["AverageNs"]) / 1e3 * n   # scaled so that the division below gives us
    print("%-11s" % names[i] + "  ".join("%s %+8.1f" % (c.replace("SQ_", ""), (acc.get(c, 0) - prev[c]) / n) for c in sorted(acc)) )
    prev = defaultdict(float, acc)
PY
