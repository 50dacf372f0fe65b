#!/usr/bin/env python3
"""`make -s resource-usage | python3 scripts/format_resource_usage.py > profiles/rNN_resource_usage.md`:
the compiler's kernel-resource-usage remarks as one markdown table."""
import subprocess
import sys

blocks = sys.stdin.read().split("Name: ")[1:]
out = ["# Kernel resource usage as reported by the compiler (`make resource-usage`: hipcc -Rpass-analysis=kernel-resource-usage,",
       "# gfx950, the Makefile's flags).  rocprofv3's `VGPR_Count` column is in allocation units (128 registers show as 64) and its",
       "# `LDS_Block_Size` omits dynamic LDS; these are the authoritative figures.  Template arguments: render_kernel<CSG depth,",
       "# DOF mode (0 none, 1 mono depth of field, 2 stereo), several lights, ray-counting instance>; render_kernel_dof<depth,",
       "# several lights, mode, counting>; render_kernel_planes<mode, counting> (scenes of axis planes only).  The production",
       "# instances are the `..., false>` ones; the counting instances (`..., true>`) run only when opts->count_rays is set.",
       "# Dynamic LDS per wave: hit stack entries x 640 B (depth 1/2: 16, depth 3/4: 20 entries on the first pass — kCsgFirstCap,",
       "# c2rt_device.h; 16 x depth on the retry pass).  Since round 3 the production instances of depth <= 3 hold the trace TWICE (lean:: + the\n# exact:: redo, c2rt_trace.inc); the counting instances and depth 4 hold exact:: only.",
       "",
       "| kernel | VGPRs | AGPRs | scratch B/lane | waves/SIMD | SGPRs spilled (to VGPR lanes) | VGPRs spilled |", "|---|---|---|---|---|---|---|"]
for b in blocks:
    lines = b.strip().splitlines()
    name = subprocess.check_output(["c++filt", lines[0].strip()]).decode().strip()
    name = name.replace("c2rt::(anonymous namespace)::", "").replace("(c2rt::RenderParams)", "")
    d = {}
    for l in lines[1:]:
        if ":" in l:
            k, v = l.rsplit(":", 1)
            d[k.strip()] = v.strip()
    out.append("| `%s` | %s | %s | %s | %s | %s | %s |" % (name, d.get("VGPRs"), d.get("AGPRs"), d.get("ScratchSize [bytes/lane]"),
                                                         d.get("Occupancy [waves/SIMD]"), d.get("SGPRs Spill"), d.get("VGPRs Spill")))
print("\n".join(out))
