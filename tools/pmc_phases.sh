#!/bin/bash
# Instruction and wait counters of the assemble kernel per phase: the ablation builds of tools/phase_ablation.py
# (python tools/phase_ablation.py --build) each under one rocprofv3 counter pass.  Development aid, GPU box only.
export TMPDIR=/tmp
export PAMD_SUMS=${PAMD_SUMS:-fast}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_phases
mkdir -p "$OUT"
for k in 0 1 2 3 4 full; do
	if [ $k = full ]; then unset PAMD_LIB; else export PAMD_LIB=$ROOT/pandaseq_amd/libpandaseq_amd_stop$k.so; fi
	(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/s$k" -o p -- python3 "$ROOT/tools/traffic_probe.py" > "$OUT/s$k.log" 2>&1)
done
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
names = ["stage", "offsets", "seed-fwd", "seed-rev", "scan", "merge+finish"]
prev = defaultdict(float)
for i, k in enumerate(["0", "1", "2", "3", "4", "full"]):
    acc = {}
    for f in glob.glob(sys.argv[1] + "/s%s/**/*counter_collection.csv" % k, recursive=True):
        for row in csv.DictReader(open(f)):
            if "pamd_assemble_kernel" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]] = float(row["Counter_Value"])   # last launch wins
    n = 8e6
    print("%-13s" % names[i] + "  ".join("%s %+8.1f" % (c.replace("SQ_", ""), (acc.get(c, 0) - prev[c]) / n) for c in sorted(acc)))
    prev = defaultdict(float, acc)
PY
