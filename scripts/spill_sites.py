#!/usr/bin/env python3
"""Where a kernel instance's scratch (spill) instructions sit: scripts/spill_sites.py <unit> <mangled-name-substring>
(source lines of a -gline-tables-only build, as in scripts/isa_census.py)."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
unit, pat = sys.argv[1], sys.argv[2]
out = os.path.join(ROOT, "build", "isa", "u%sg.s" % unit)
if not os.path.exists(out):
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_census.py"), unit, pat], check=True, stdout=subprocess.DEVNULL)
src = open(out).read().split("\n")
files = {}
for l in src:
    m = re.match(r'^\s*\.file\s+(\d+)\s+(?:"[^"]*"\s+)?"([^"]*)"', l)
    if m:
        files[int(m.group(1))] = m.group(2).split("/")[-1]
inside = False
cur = None
c = collections.Counter()
for l in src:
    s = l.strip()
    if not inside:
        if re.match(r"^_Z\w+:", s) and pat in s:
            inside = True
        continue
    if s.startswith(".Lfunc_end"):
        break
    m = re.match(r"^\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        cur = (files.get(int(m.group(1))), int(m.group(2)))
        continue
    if s.startswith("scratch_"):
        c[(s.split()[0], cur)] += 1
for k, v in sorted(c.items(), key=lambda kv: (kv[0][1] or ("", 0))):
    print(v, k[0], "%s:%s" % (k[1] or ("?", 0)))
