#!/bin/bash
# The randomized campaigns on seeds no earlier run has seen (round 6; the builder's lease, beyond the suite's fixed slices):
#   bash tools/campaign_r6.sh [seed shift]     writes gpurun_out/campaign_r6/*.log, one summary line per campaign at the end
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/campaign_r6
mkdir -p "$OUT"
cd "$ROOT"
S=${1:-0}
run() { name=$1; shift; timeout 900 python3 "$@" > "$OUT/$name.log" 2>&1; echo "$name rc=$? $(grep -c MISMATCH "$OUT/$name.log") mismatch lines; $(tail -1 "$OUT/$name.log")"; }
run parity_small tests/parity_sweep.py 60 wide $((61000 + S)) 3000
run parity_plain tests/parity_sweep.py 30 plain $((62000 + S)) 3000
run parity_100k tests/parity_sweep.py 6 wide $((63000 + S)) 100000
run parity_600k tests/parity_sweep.py 2 wide $((64000 + S)) 600000
run primer_plain tests/primer_sweep.py 80 $((65000 + S)) plain
run primer_after tests/primer_sweep.py 40 $((66000 + S)) after
run primer_penalty tests/primer_sweep.py 30 $((67000 + S)) penalty
run primer_x30 tests/primer_sweep.py 10 $((68000 + S)) plain 30
run length tests/length_sweep.py 30 $((69000 + S))
run textio tests/textio_sweep.py 100 $((70000 + S))
run compat tests/compat_sweep.py 40 $((71000 + S))
run per_pair tests/per_pair_sweep.py 20 $((72000 + S))
