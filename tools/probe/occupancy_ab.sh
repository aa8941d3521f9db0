#!/bin/bash
# Waves in flight of the lean kernel, built library (A) against libpandaseq_amd_B.so (B): SQ_WAVE_CYCLES / SQ_BUSY_CYCLES and the
# instruction counters of one rocprofv3 --pmc pass each (development aid; GPU box only).   bash tools/probe/occupancy_ab.sh [pairs] [config]
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
PAIRS=${1:-8000000}
CONFIG=${2:-1}
OUT=$ROOT/gpurun_out/occupancy_ab
rm -rf "$OUT"; mkdir -p "$OUT"
for which in A B; do
	[ $which = B ] && export PAMD_LIB=$ROOT/pandaseq_amd/libpandaseq_amd_B.so
	(cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/$which" -o p -- python3 "$ROOT/tools/fk_probe.py" --only two-kernel --pairs $PAIRS --config $CONFIG --reps 1 > "$OUT/$which.log" 2>&1)
	(cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/${which}2" -o p -- python3 "$ROOT/tools/fk_probe.py" --only two-kernel --pairs $PAIRS --config $CONFIG --reps 1 > "$OUT/${which}2.log" 2>&1)
done
python3 - "$OUT" "$PAIRS" <<'PY'
import csv, glob, sys
out, pairs = sys.argv[1], float(sys.argv[2])
for which in "AB":
    d = {}
    for f in sorted(glob.glob(out + "/" + which + "*/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            if "pamd_fast_kernel" in row.get("Kernel_Name", ""):
                d[row["Counter_Name"]] = float(row["Counter_Value"])
                d["_wg"] = row.get("Workgroup_Size"); d["_grid"] = row.get("Grid_Size"); d["_lds"] = row.get("LDS_Block_Size"); d["_vgpr"] = row.get("VGPR_Count"); d["_acc"] = row.get("Accum_VGPR_Count")
    if not d:
        print(which, "no counters"); continue
    print(f"{which}: grid {d['_grid']} wg {d['_wg']} lds {d['_lds']} vgpr {d['_vgpr']} agpr {d['_acc']}  waves in flight per SQ_BUSY cycle {d['SQ_WAVE_CYCLES'] / d['SQ_BUSY_CYCLES']:.2f}")
    for c in sorted(k for k in d if not k.startswith('_')):
        print(f"   {c:24s} {d[c]:16.0f}  per pair {d[c] / pairs:9.2f}")
PY
