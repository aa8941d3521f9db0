#!/usr/bin/env python3
"""Hazards the compiler cannot see in the lean kernel's ISA (pamd_fast.hip), checked on a listing of EVERY instantiation:

1. lds_read_u16_bare() issues `ds_read_u16` from inline asm -- a load the compiler's waitcnt bookkeeping does not count -- and
   lds_wait_all() waits for it later, after a stretch of scalar work that overlaps the LDS round trip.  Nothing between the two
   may touch the destination register: a register-allocator copy, a spill or a rematerialisation there would read the register
   before the data lands.  scan_bare_reads() walks the control-flow graph of every kernel (blocks split at labels and branches,
   the set of pending registers carried along every edge to a fixed point) and reports any instruction that reads or writes a
   pending register, and any call, before an `s_waitcnt lgkmcnt(0)` on any path.
2. The parking code writes M0 inside inline asm (`s_mov_b32 m0` + `v_writelane_b32`) without the compiler knowing; that is
   harmless as long as nothing the COMPILER emits for these kernels uses M0 (no s_movrel, LDS-DMA, GDS, sendmsg operand).
   scan_m0() reports every mention of m0 outside the inline-asm blocks.

    make -C pandaseq_amd/csrc asm && python tools/isa_hazards.py /tmp/pamd_fast.s
tests/test_oracle_cpu.py::test_the_bench_kernels_do_not_spill runs both on every build of the CPU suite.
"""
import re
import sys
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan_bare_reads(listing, only=None):
    """lds_read_u16_bare() issues `ds_read_u16` from inline asm: a load the compiler's waitcnt bookkeeping does not see, whose
    destination is valid only after the `s_waitcnt lgkmcnt(0)` of lds_wait_all().  Over the control-flow graph of every kernel
    of the listing (blocks split at labels and branches, the pending set carried along every edge to a fixed point): between
    such a read and a wait for lgkmcnt(0) on ANY path, no instruction may read or write the destination register, and no call
    may be made.  Returns [(kernel, line, text, register)]."""
    bad = []
    # cut the listing into kernels
    kernels, cur = [], None
    for no, raw in enumerate(listing, 1):
        t = raw.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            cur = (m.group(1), [])
            kernels.append(cur)
            continue
        if cur is not None:
            cur[1].append((no, t))
            if t.split(";")[0].strip().startswith("s_endpgm"):
                pass
    for name, lines in kernels:
        if only and only not in name:     # (every function of the listing: the kernels, and what they call out of line -- redo_pair() holds fast_pair()'s reads too)
            continue
        # instructions and blocks
        blocks, label_of, block = [], {}, None
        in_asm = False

        def new_block():
            b = {"ins": [], "succ": [], "fall": True}
            blocks.append(b)
            return b
        block = new_block()
        for no, t in lines:
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            lm = re.match(r"^([.\w$]+):", t)
            if lm and not t.startswith(";"):
                if block["ins"] or id(block) in [id(b) for b in label_of.values()]:
                    block = new_block()
                label_of[lm.group(1)] = block
                continue
            if not t or t.startswith(";") or t.startswith("."):
                continue
            code = t.split(";")[0].strip()
            if not code:
                continue
            block["ins"].append((no, code, in_asm))
            op = code.split()[0]
            if op == "s_branch":
                block["succ"].append(code.split()[1])
                block["fall"] = False
                block = new_block()
            elif op.startswith("s_cbranch"):
                block["succ"].append(code.split()[-1])
                block = new_block()
            elif op == "s_endpgm" or op == "s_setpc_b64":
                block["fall"] = False
                block = new_block()
        idx = {id(b): i for i, b in enumerate(blocks)}
        succs = []
        for i, b in enumerate(blocks):
            s = [idx[id(label_of[l])] for l in b["succ"] if l in label_of]
            if b["fall"] and i + 1 < len(blocks):
                s.append(i + 1)
            succs.append(s)

        def transfer(b, pending, report):
            pending = dict(pending)
            for no, code, in_asm in b["ins"]:
                op = code.split()[0]
                if op == "s_waitcnt" and "lgkmcnt(0)" in code:
                    pending = {}
                    continue
                if in_asm and op == "ds_read_u16":
                    d = _regs(code.split(",")[0])
                    if report:
                        for r in (_regs(code) - d) & set(pending):
                            bad.append((name, no, code, r))
                    for r in d:
                        pending[r] = no
                    continue
                if pending:
                    if op in ("s_swappc_b64", "s_setpc_b64", "s_endpgm") and report:
                        bad.append((name, no, code, -1))
                    if report:
                        for r in _regs(code) & set(pending):
                            bad.append((name, no, code, r))
            return pending
        pin = [dict() for _ in blocks]
        work = list(range(len(blocks)))
        while work:
            i = work.pop()
            out = transfer(blocks[i], pin[i], False)
            for s in succs[i]:
                merged = dict(pin[s])
                merged.update(out)
                if merged.keys() != pin[s].keys():
                    pin[s] = merged
                    work.append(s)
        for i, b in enumerate(blocks):
            transfer(b, pin[i], True)
    return bad


def scan_m0(listing):
    """[(line, text)] of every instruction outside an inline-asm block that mentions m0, in every function of the listing."""
    out, in_asm, on = [], False, False
    for no, raw in enumerate(listing, 1):
        t = raw.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            on = True   # (every function of the listing)
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif on and not in_asm and re.search(r"\bm0\b", t.split(";")[0]):
            out.append((no, t))
    return out


def count_bare_reads(listing):
    return sum(1 for i, raw in enumerate(listing) if raw.strip().startswith("ds_read_u16 v") and i > 0 and listing[i - 1].strip().startswith(";;#ASMSTART"))


if __name__ == "__main__":
    lst = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/pamd_fast.s").read().splitlines()
    bad, m0 = scan_bare_reads(lst), scan_m0(lst)
    print(f"{count_bare_reads(lst)} bare reads; {len(bad)} uses of a pending register; {len(m0)} compiler-emitted uses of m0")
    for x in bad[:30] + m0[:30]:
        print(x)
    sys.exit(1 if bad or m0 else 0)
