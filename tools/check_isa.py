#!/usr/bin/env python
"""Post-compile lints of the generated gfx950 code.

(1) Packed-fp32 hazard (every file): no `v_pk_*_f32` may carry an `op_sel:[...]` modifier with a bit set, i.e. take the HIGH
dword of a source pair for its LOW result half.  On MI355X that form returns a wrong low half in lanes 48..63 while another
kernel's wave issues MFMAs on the same SIMD (measured: csrc/probe.hip::probe_pk_kernel under tools/race_repro.py; DESIGN.md
6.3) -- a run-to-run difference that only shows when two streams share a CU.  Kernels where hipcc forms such operands are
built with XP_NO_PK_F32 (csrc/common.h).

(2) Inline-asm LDS transpose reads (csrc/common.h::lds_read_tr16_async), files named with --asm-reads:

The asm `ds_read_b64_tr_b16` is invisible to hipcc's waitcnt bookkeeping, so the source places an explicit
`s_waitcnt lgkmcnt(0)` before the first use of its result.  This script checks the GENERATED code: between every
`ds_read_b64_tr_b16 v[a:b], ...` that came from inline asm (marked by the `;;#ASMSTART` / `;;#ASMEND` pair hipcc emits) and
the next `s_waitcnt` that waits lgkmcnt(0) ON EVERY CONTROL-FLOW PATH, no instruction may mention v[a..b] as an operand.  Usage:
    check_isa.py file.s [...] [--asm-reads file.s [...]]      exit status 1 on a violation
"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


PK_OPSEL = re.compile(r"^v_pk_[a-z0-9]+_f32\b.*\bop_sel:\[([01,]+)\]")


def check_pk(path):
    """lint (1): packed fp32 instructions whose low half selects a high source dword"""
    bad = 0
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip()
        m = PK_OPSEL.match(line)
        if m and "1" in m.group(1):
            print(f"{path}:{ln}: `{line}`: packed fp32 with op_sel low-half redirection (gfx950 MFMA co-issue hazard); "
                  "mark the kernel XP_NO_PK_F32")
            bad += 1
    return bad


def check(path):
    """lint (2): from every inline-asm transpose read, follow the control flow (fall-through, s_branch, both arms of s_cbranch_*) until
    an `s_waitcnt` with lgkmcnt(0) or `s_endpgm`; no instruction on the way may mention the read's destination registers.  (hipcc
    rotates loops: the block holding the wait is often laid out BEFORE the reads and reached by a branch -- a linear scan would flag
    whatever happens to follow the loop in the file.)"""
    insts, labels = [], {}          # (line number, text, is_asm_read)
    in_asm = False
    for ln, raw in enumerate(open(path), 1):
        st = raw.lstrip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if line.endswith(":"):
            labels[line[:-1]] = len(insts)
            continue
        if line.startswith("."):
            continue
        insts.append((ln, line, in_asm and line.startswith("ds_read_b64_tr_b16")))
    bad = 0
    reported = set()
    for start, (ln0, line0, is_read) in enumerate(insts):
        if not is_read:
            continue
        dst = regs_of(line0.split(",")[0])
        stack, seen = [start + 1], set()
        while stack:
            i = stack.pop()
            while i < len(insts) and i not in seen:
                seen.add(i)
                ln, line, rd = insts[i]
                if line.startswith("s_endpgm") or (line.startswith("s_waitcnt") and "lgkmcnt(0)" in line):
                    break
                if rd:                      # another asm read: its own walk covers its registers; this one's stay pending
                    i += 1
                    continue
                used = regs_of(line) & dst
                if used and (ln, ln0) not in reported:
                    reported.add((ln, ln0))
                    print(f"{path}:{ln}: `{line}` touches v{sorted(used)} written by the asm transpose read at line {ln0} "
                          "before s_waitcnt lgkmcnt(0)")
                    bad += 1
                if line.startswith("s_branch") or line.startswith("s_cbranch"):
                    tgt = line.split()[-1]
                    if tgt in labels:
                        stack.append(labels[tgt])
                    if line.startswith("s_branch"):
                        break
                i += 1
    return bad


if __name__ == "__main__":
    args = sys.argv[1:]
    asm_reads = []
    if "--asm-reads" in args:
        i = args.index("--asm-reads")
        asm_reads, args = args[i + 1:], args[:i]
    total = sum(check_pk(p) for p in args + asm_reads) + sum(check(p) for p in asm_reads)
    print(f"check_isa: {len(args) + len(asm_reads)} file(s), {total} violation(s)")
    sys.exit(1 if total else 0)
