#!/usr/bin/env python3
"""ISA peephole pass over the device assembly hipcc emits for the kernel units (Makefile: .s -> this -> .o).

gfx950 issues a VOP2-encoded v_cndmask_b32 (`_e32`, mask implicitly in VCC) that directly follows another one at
16-23 cycles instead of 4.4, and the stall holds the whole SIMD (it does not shrink with more resident waves);
the VOP3 encoding of the very same instruction (`_e64 ..., vcc`) issues at 4.4 cycles back to back
(scripts/ubench/cnd_issue.hip, profiles/r03_ubench.md).  The compiler shrinks every select whose mask is in VCC
to VOP2 and emits the two halves of a 64-bit select back to back, so a `cond ? a : b` on doubles costs ~48 cycles
instead of ~13.  This pass re-encodes them as VOP3: same operation, same operands, same result, 4 more bytes.
A VOP3 instruction cannot carry a 32-bit literal on gfx9, so a select with a literal source stays VOP2.

usage: isa_peephole.py in.s out.s   (prints what it changed to stderr)
"""
import re
import sys

CND = re.compile(r"^(\s+)v_cndmask_b32_e32(\s+)(v\d+|v\[\d+:\d+\]),\s*([^,]+),\s*(v\d+),\s*vcc\s*(;.*)?$")
# operands a VOP3 encoding accepts in src0: a VGPR, an SGPR, or an inline constant
INLINE_F = {"0.5", "-0.5", "1.0", "-1.0", "2.0", "-2.0", "4.0", "-4.0", "0.15915494", "0"}


def vop3_src_ok(src):
    src = src.strip()
    if re.fullmatch(r"v\d+|s\d+|vcc_lo|vcc_hi|m0|exec_lo|exec_hi", src):
        return True
    if src in INLINE_F:
        return True
    if re.fullmatch(r"-?\d+", src):
        return -16 <= int(src) <= 64
    return False


def main():
    src, dst = sys.argv[1], sys.argv[2]
    changed = kept = 0
    out = []
    with open(src) as f:
        for line in f:
            m = CND.match(line.rstrip("\n"))
            if m and vop3_src_ok(m.group(4)):
                out.append("%sv_cndmask_b32_e64%s%s, %s, %s, vcc%s\n" % (m.group(1), m.group(2), m.group(3), m.group(4).strip(), m.group(5),
                                                                       " " + m.group(6) if m.group(6) else ""))
                changed += 1
            else:
                if m:
                    kept += 1
                out.append(line)
    with open(dst, "w") as f:
        f.writelines(out)
    print("isa_peephole: %s: %d v_cndmask_b32_e32 re-encoded as VOP3, %d kept (literal source)" % (src, changed, kept), file=sys.stderr)


if __name__ == "__main__":
    main()
