export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/wbisect; rm -rf $OUT; mkdir -p $OUT
export PAMD_CONFIG=2 PAMD_SUMS=fast PAMD_PAIRS=8000000
for l in "" _C200f92d _C202b4fa _C185e552; do
	if [ -z "$l" ]; then unset PAMD_LIB; else export PAMD_LIB=$ROOT/pandaseq_amd/libpandaseq_amd$l.so; fi
	(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/w$l" -o t -- python3 "$ROOT/tools/traffic_probe.py" > "$OUT/w$l.log" 2>&1)
	python3 - "$OUT/w$l" "$l" <<'PY'
import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        rows.append((int(row.get("Start_Timestamp", row.get("Dispatch_Id", 0)) or 0), row["Kernel_Name"][:60], float(row["Counter_Value"])))
rows.sort()
print("lib", sys.argv[2] or "(built)")
for r in rows[-8:]:
    print("   ", r[1], r[2] * 1024 / 8e6, "B/pair")
PY
done
