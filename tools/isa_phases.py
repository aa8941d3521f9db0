#!/usr/bin/env python3
"""Static instruction mix per phase of the lean kernel: counts the instructions between the `; MARK <phase>` comments of
an ISA listing built with -DPAMD_MARK (loops count once; rare blocks count too -- read it as an upper bound per pass).

    make -C pandaseq_amd/csrc asm EXTRA=-DPAMD_MARK && python tools/isa_phases.py /tmp/pamd_fast.s [kernel-substring]
"""
import re
import sys
from collections import OrderedDict

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/pamd_fast.s"
want = sys.argv[2] if len(sys.argv) > 2 else "pamd_fast_kernelILi0E"
phases = OrderedDict()
cur = None
on = False
for line in open(path):
    t = line.strip()
    if re.match(r"^[A-Za-z_][\w.$]*:", t) and want in t.split(":")[0]:
        on = True
        cur = "prologue"
        continue
    if not on:
        continue
    if t.startswith("s_endpgm"):
        break
    m = re.match(r";\s*MARK (\S+)", t)
    if m:
        cur = m.group(1)
        continue
    if not t or t.startswith(";") or t.startswith(".") or re.match(r"^[\w.$]+:", t):
        continue
    op = t.split()[0]
    d = phases.setdefault(cur, {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "wait": 0, "branch": 0})
    if op.startswith("s_waitcnt"):
        d["wait"] += 1
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        d["branch"] += 1
    elif op.startswith("v_"):
        d["valu"] += 1
    elif op.startswith("s_"):
        d["salu"] += 1
    elif op.startswith("ds_"):
        d["lds"] += 1
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        d["vmem"] += 1
tot = {}
for k, d in phases.items():
    print(f"{k:14s} " + "  ".join(f"{n} {v:4d}" for n, v in d.items()))
    for n, v in d.items():
        tot[n] = tot.get(n, 0) + v
print(f"{'total':14s} " + "  ".join(f"{n} {v:4d}" for n, v in tot.items()))
