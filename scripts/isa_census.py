#!/usr/bin/env python3
"""Static ISA census of one kernel instance: instructions by class, attributed to the source function whose
statement each instruction was generated from (the most recent `.loc` line of a -gline-tables-only build; the
whole trace is inlined into the kernel, so the line's enclosing function in c2rt_kernels.hip names the region).

usage: isa_census.py <unit> <mangled-name-substring> [lean|exact]
The production instances hold the trace twice (c2rt_trace.inc: lean:: first, then — after the atomic that counts a
redo — exact::); `lean` / `exact` restricts the census to that half.
Builds build/isa/u<unit>g.s with the Makefile's flags + -gline-tables-only (code generation is unchanged).
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "chess2rt_amd/csrc/c2rt_kernels.hip")
TRACE = os.path.join(ROOT, "chess2rt_amd/csrc/c2rt_trace.inc")


def make_flags():
    flags = {}
    for line in open(os.path.join(ROOT, "Makefile")):
        m = re.match(r"^(FPFLAGS|KERNELFLAGS|KERNELFLAGS_u\d)\s*:=\s*(.*)$", line)
        if m:
            flags[m.group(1)] = m.group(2).strip()
    return flags


def function_spans(path=None):
    """(first line, last line, name) of every function body in a source file (brace matching from a
    line that looks like a definition at namespace scope)."""
    lines = open(path or SRC).read().split("\n")
    spans = []
    i = 0
    sig = re.compile(r"^(?:template\s*<[^>]*>\s*)?(?:DEV|__device__|__global__|static|inline|int|void)\b.*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;]*$")
    while i < len(lines):
        m = sig.match(lines[i])
        if m and not lines[i].startswith(" "):
            name = m.group(1)
            # find the opening brace
            j = i
            depth = 0
            opened = False
            while j < len(lines):
                for ch in lines[j]:
                    if ch == "{":
                        depth += 1
                        opened = True
                    elif ch == "}":
                        depth -= 1
                if opened and depth == 0:
                    break
                if not opened and lines[j].rstrip().endswith(";"):
                    break
                j += 1
            if opened:
                spans.append((i + 1, j + 1, name))
                i = j
        i += 1
    return spans


CLASSES = [
    ("fp64 add/mul/fma", re.compile(r"^v_(add|mul|fma|fmac)_f64")),
    ("fp64 transcendental seed (rcp/rsq/sqrt)", re.compile(r"^v_(rcp|rsq|sqrt)_f64")),
    ("fp64 divide/sqrt scaffolding (div_scale/fmas/fixup, ldexp, cmp_class)", re.compile(r"^v_(div_scale|div_fmas|div_fixup|ldexp|frexp_\w+)_f64|^v_cmp_class_f64")),
    ("fp64 compare", re.compile(r"^v_cmpx?_\w+_f64")),
    ("fp64 min/max/floor/cvt", re.compile(r"^v_(max|min|floor|ceil|trunc|rndne|fract)_f64|^v_cvt_\w*f64|^v_cvt_f64")),
    ("select (v_cndmask)", re.compile(r"^v_cndmask")),
    ("64-bit move", re.compile(r"^v_mov_b64|^v_accvgpr")),
    ("32-bit move", re.compile(r"^v_mov_b32")),
    ("spill-lane traffic (v_readlane/v_writelane)", re.compile(r"^v_(readlane|writelane|readfirstlane)")),
    ("fp32 arithmetic", re.compile(r"^v_\w+_f32|^v_pk_\w+_f32")),
    ("integer / logic VALU", re.compile(r"^v_")),
    ("LDS", re.compile(r"^ds_")),
    ("global / scratch memory", re.compile(r"^(global|flat|scratch|buffer)_")),
    ("scalar memory", re.compile(r"^s_(load|buffer_load|store)")),
    ("branch", re.compile(r"^s_(cbranch|branch|setpc|swappc|call)")),
    ("wait / nop", re.compile(r"^s_(waitcnt|nop|sleep)")),
    ("SALU", re.compile(r"^s_")),
]


def classify(op):
    for name, rx in CLASSES:
        if rx.match(op):
            return name
    return "other"


def main():
    unit, pat = sys.argv[1], sys.argv[2]
    half = sys.argv[3] if len(sys.argv) > 3 else None
    flags = make_flags()
    out = os.path.join(ROOT, "build", "isa", "u%sg.s" % unit)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + flags["FPFLAGS"].split() + \
          ["-fhip-fp32-correctly-rounded-divide-sqrt"] + flags["KERNELFLAGS"].split() + flags.get("KERNELFLAGS_u%s" % unit, "").split() + \
          ["-DC2RT_UNIT=%s" % unit, "-gline-tables-only", "--offload-device-only", "-S", SRC, "-o", out, "-I" + os.path.join(ROOT, "include")]
    csrc = os.path.dirname(SRC)
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc) if f.endswith(('.hip', '.inc', '.h')))
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    spans = {"c2rt_kernels.hip": function_spans(SRC), "c2rt_trace.inc": function_spans(TRACE)}

    files = {}
    for raw in open(out):
        m = re.match(r'^\s*\.file\s+(\d+)\s+(?:"[^"]*"\s+)?"([^"]*)"', raw)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(2))

    def region_of(fileno, line):
        name = files.get(fileno, "?")
        if name == "x87.h":
            return "x87.h (Sphere u,v in x87 extended precision)"
        if name == "fp64_lean.h":
            return "fp64_lean.h (lean divide / sqrt / reciprocal length)"
        if name not in spans:
            return "device libm / HIP headers"
        for a, b, fn in spans[name]:
            if a <= line <= b:
                return fn
        if name == "c2rt_trace.inc":
            return "vector helpers (D3 / F3 operators, dot, sqmag)" if line < 140 else "(file scope)"
        return "kernel entry (c2rt_kernels.hip)"

    per_region = collections.defaultdict(collections.Counter)
    total = collections.Counter()
    inside = False
    cur = ("?", 0)
    kname = None
    in_exact = False
    for raw in open(out):
        s = raw.strip()
        if not inside:
            if re.match(r"^_Z\w+:", s) and pat in s:
                inside = True
                kname = s.split(":")[0]
            continue
        if s.startswith(".Lfunc_end") or s.startswith(".section") and "rodata" in s:
            break
        m = re.match(r"^\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith(".") or s.startswith(";") or s.endswith(":"):
            continue
        op = s.split()[0]
        if op.startswith("global_atomic_add_x2"):
            in_exact = True      # render_one: atomicAdd(P.redo_counter) sits between the two copies
        if half == "lean" and in_exact:
            continue
        if half == "exact" and not in_exact:
            continue
        c = classify(op)
        per_region[region_of(*cur)][c] += 1
        total[c] += 1
    n = sum(total.values())
    print("# Static ISA census of `%s`%s\n" % (kname, " — the %s:: half" % half if half else ""))
    print("%d instructions.  By class:\n" % n)
    print("| class | instructions | share |\n|---|---|---|")
    for name, _ in CLASSES + [("other", None)]:
        if total[name]:
            print("| %s | %d | %.1f %% |" % (name, total[name], 100.0 * total[name] / n))
    print("\nBy source region (function of c2rt_trace.inc / c2rt_kernels.hip the statement belongs to), largest first; columns: all / fp64 arithmetic / "
          "fp64 seeds / divide-sqrt scaffolding / compares / selects / moves (32+64) / spill lanes / SALU+branch+wait:\n")
    print("| region | all | fp64 arith | seeds | scaffolding | fp64 cmp | selects | moves | spill lanes | scalar |\n|---|---|---|---|---|---|---|---|---|---|")
    for reg, cnt in sorted(per_region.items(), key=lambda kv: -sum(kv[1].values())):
        tot = sum(cnt.values())
        if tot < 10:
            continue
        print("| %s | %d | %d | %d | %d | %d | %d | %d | %d | %d |" % (
            reg, tot, cnt["fp64 add/mul/fma"], cnt["fp64 transcendental seed (rcp/rsq/sqrt)"],
            cnt["fp64 divide/sqrt scaffolding (div_scale/fmas/fixup, ldexp, cmp_class)"], cnt["fp64 compare"], cnt["select (v_cndmask)"],
            cnt["64-bit move"] + cnt["32-bit move"], cnt["spill-lane traffic (v_readlane/v_writelane)"],
            cnt["SALU"] + cnt["branch"] + cnt["wait / nop"] + cnt["scalar memory"]))


if __name__ == "__main__":
    main()
