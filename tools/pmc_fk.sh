#!/bin/bash
# (SQ counters only: a pass with TA_* counters aborted inside rocprofv3 on this pool and hung until killed.)
# Hardware counters of the two assemble kernels of a default (two-kernel) launch, one rocprofv3 pass per group, plus the
# kernel-trace split (development aid; GPU box only).   bash tools/pmc_fk.sh [pairs] [config]
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PAIRS=${1:-8000000}
CONFIG=${2:-1}
MODE=${MODE:-two-kernel}
OUT=$ROOT/gpurun_out/pmc_fk
rm -rf "$OUT"; mkdir -p "$OUT"
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o p -- python3 "$ROOT/tools/fk_probe.py" --only $MODE --pairs $PAIRS --config $CONFIG --sums ${SUMS:-fast} --reps 3 > "$OUT/trace.log" 2>&1)
i=0
for group in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
             "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
             "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
             "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"; do
	i=$((i + 1))
	(cd /tmp && timeout 150 rocprofv3 --pmc $group --output-format csv -d "$OUT/g$i" -o p -- python3 "$ROOT/tools/fk_probe.py" --only $MODE --pairs $PAIRS --config $CONFIG --sums ${SUMS:-fast} --reps 1 > "$OUT/g$i.log" 2>&1)
done
python3 - "$OUT" "$PAIRS" <<'PY'
import csv, glob, sys
from collections import defaultdict
out, pairs = sys.argv[1], float(sys.argv[2])
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "pamd_" in row.get("Name", ""):
            print("trace:", row["Name"][:70], "calls", row["Calls"], "avg_ns", row["AverageNs"], "total_ns", row["TotalDurationNs"])
acc = defaultdict(dict)
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "pamd_fast_kernel" in k or "pamd_assemble_kernel" in k:
            acc["fast" if "pamd_fast_kernel" in k else "general"][row["Counter_Name"]] = float(row["Counter_Value"])   # last launch wins
for k, d in acc.items():
    print(f"== {k} kernel, per pair of the launch ({pairs:.0f} pairs; the general kernel only sees the deferred ones)")
    for c in sorted(d):
        print(f"   {c:26s} {d[c]:16.0f}   per pair {d[c] / pairs:10.2f}")
PY
