#!/bin/bash
# SQ counters of the primer kernel and of the assembly kernel behind it (config 2), per pair (development aid; GPU box only).
#   bash tools/pmc_primer.sh [pairs]
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PAMD_PAIRS=${1:-8000000} PAMD_CONFIG=2 PAMD_SUMS=fast
OUT=$ROOT/gpurun_out/pmc_primer
rm -rf "$OUT"; mkdir -p "$OUT"
i=0
for group in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
             "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
	i=$((i + 1))
	(cd /tmp && timeout 150 rocprofv3 --pmc $group --output-format csv -d "$OUT/g$i" -o p -- python3 "$ROOT/tools/traffic_probe.py" > "$OUT/g$i.log" 2>&1)
done
python3 - "$OUT" "$PAMD_PAIRS" <<'PY'
import csv, glob, sys
from collections import defaultdict
out, pairs = sys.argv[1], float(sys.argv[2])
acc = defaultdict(dict)
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "pamd_primer_kernel" in k or "pamd_assemble_kernel" in k:
            name = "assemble" if "pamd_assemble_kernel" in k else ("primer (fast)" if "<true>" in k else "primer (careful)")
            acc[name][row["Counter_Name"]] = float(row["Counter_Value"])   # last launch wins
for k, d in acc.items():
    print(f"== {k} kernel, per pair ({pairs:.0f} pairs)")
    for c in sorted(d):
        print(f"   {c:26s} {d[c]:16.0f}   per pair {d[c] / pairs:10.2f}")
PY
