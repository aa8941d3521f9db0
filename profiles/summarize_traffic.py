#!/usr/bin/env python3
"""usage: summarize_traffic.py <fetch_results.db> <write_results.db> <pairs>
Prints calibrated HBM bytes of the last assemble dispatch (see tools/traffic_probe.py)."""
import json
import sqlite3
import sys

GIB = float(1 << 30)


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    rows = list(con.execute("select kernel_name, value, start from counters_collection where counter_name = ? order by start", (counter,)))
    # a launch is two dispatches when the lean kernel serves it (pamd_fast_kernel, then pamd_assemble_kernel over the pairs
    # it deferred): the measured launch is the LAST dispatch of each
    gen = [v for k, v, _ in rows if "pamd_assemble_kernel" in k]
    lean = [v for k, v, _ in rows if "pamd_fast_kernel" in k]
    # with `-p` / `-q` a launch starts with the primer kernel (pamd_primer.hip), which reads the input rows once more, and
    # its (normally idle) careful twin
    primer = [v for k, v, _ in rows if "pamd_primer_kernel<true>" in k or "pamd_primer_kernelILb1" in k]
    careful = [v for k, v, _ in rows if "pamd_primer_kernel<false>" in k or "pamd_primer_kernelILb0" in k]
    # the calibration copy is the 1 GiB device-to-device copy: by far the largest copy dispatch
    cp = [v for k, v, _ in rows if "copybuffer" in k.lower()]
    return gen[-1] + (lean[-1] if lean else 0.0) + (primer[-1] if primer else 0.0) + (careful[-1] if careful else 0.0), max(cp)


fetch_db, write_db, pairs = sys.argv[1], sys.argv[2], int(sys.argv[3])
a_rd, c_rd = per_kernel(fetch_db, "FETCH_SIZE")
a_wr, c_wr = per_kernel(write_db, "WRITE_SIZE")
rd_unit = GIB / c_rd   # bytes per FETCH_SIZE count, from the 1 GiB copy
wr_unit = GIB / c_wr
out = {
    "pairs": pairs,
    "raw": {"FETCH_SIZE_assemble": a_rd, "FETCH_SIZE_copy_1GiB": c_rd, "WRITE_SIZE_assemble": a_wr, "WRITE_SIZE_copy_1GiB": c_wr},
    "bytes_per_count": {"FETCH_SIZE": rd_unit, "WRITE_SIZE": wr_unit},
    "hbm_read_bytes": a_rd * rd_unit,
    "hbm_write_bytes": a_wr * wr_unit,
}
import os
if os.environ.get("PAMD_BUILD"):
    out["build"] = os.environ["PAMD_BUILD"]   # the commit the measured library was built from (tools/refresh_profiles.sh)
out["hbm_bytes_per_launch"] = out["hbm_read_bytes"] + out["hbm_write_bytes"]
out["hbm_bytes_per_pair"] = out["hbm_bytes_per_launch"] / pairs
print(json.dumps(out, indent=1))
