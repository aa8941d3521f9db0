#!/bin/bash
# The lean kernel's vector instruction mix (which of its VALU instructions are double-rate-limited): one rocprofv3 --pmc pass per group.
#   bash tools/probe/valu_mix.sh [pairs] [config]      (PAMD_LIB selects the library; development aid, GPU box only)
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
PAIRS=${1:-8000000}
CONFIG=${2:-1}
OUT=$ROOT/gpurun_out/valu_mix
rm -rf "$OUT"; mkdir -p "$OUT"
i=0
for group in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
             "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_IOPS SQ_INSTS_VSKIPPED" \
             "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
             "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE"; do
	i=$((i + 1))
	(cd /tmp && timeout 200 rocprofv3 --pmc $group --output-format csv -d "$OUT/g$i" -o p -- python3 "$ROOT/tools/fk_probe.py" --only two-kernel --pairs $PAIRS --config $CONFIG --reps 1 > "$OUT/g$i.log" 2>&1)
done
python3 - "$OUT" "$PAIRS" <<'PY'
import csv, glob, sys
out, pairs = sys.argv[1], float(sys.argv[2])
d = {}
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "pamd_fast_kernel" in row.get("Kernel_Name", ""):
            d[row["Counter_Name"]] = float(row["Counter_Value"])
for c in sorted(d):
    print(f"   {c:28s} {d[c]:16.0f}  per pair {d[c] / pairs:9.2f}")
PY
