#!/bin/bash
# Hardware counters of the assemble kernel, one rocprofv3 pass per group (development aid; GPU box only).
#   bash tools/pmc_probe.sh            -> gpurun_out/pmc/<group>/..._counter_collection.csv, summary on stdout
# The workload is tools/traffic_probe.py (two launches of 8 M config-1 pairs, fast sums unless PAMD_SUMS=exact).
export TMPDIR=/tmp
export PAMD_SUMS=${PAMD_SUMS:-fast}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc
mkdir -p "$OUT"
i=0
for group in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
             "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SMEM" \
             "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
	i=$((i + 1))
	(cd /tmp && rocprofv3 --pmc $group --output-format csv -d "$OUT/g$i" -o p -- python3 "$ROOT/tools/traffic_probe.py" > "$OUT/g$i.log" 2>&1)
done
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
for f in sorted(glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True)):
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "pamd_assemble_kernel" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:28s} last launch {v[-1]:.6g}   (launches {len(v)})")
PY
