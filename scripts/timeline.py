#!/usr/bin/env python3
"""Per-frame GPU timeline of a workload from a rocprofv3 --kernel-trace CSV: durations of the tile-mask pre-pass and of
the frame kernel, and the idle gaps between consecutive dispatches (usage: timeline.py <trace_kernel_trace.csv>)."""
import csv, statistics, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "render_kernel" in r["Kernel_Name"] or "tile_masks" in r["Kernel_Name"]]
rows = rows[len(rows) // 3:]          # the tail of the run: the timed back-to-back frames
pre, frame, gap_pf, gap_fp, period = [], [], [], [], []
last_frame_end = None
last_frame_start = None
for a, b in zip(rows, rows[1:]):
    sa, ea, sb, eb = int(a["Start_Timestamp"]), int(a["End_Timestamp"]), int(b["Start_Timestamp"]), int(b["End_Timestamp"])
    if "tile_masks" in a["Kernel_Name"] and "render_kernel" in b["Kernel_Name"]:
        pre.append(ea - sa); gap_pf.append(sb - ea)
    if "render_kernel" in a["Kernel_Name"]:
        frame.append(ea - sa); gap_fp.append(sb - ea)
        if last_frame_start is not None: period.append(sa - last_frame_start)
        last_frame_start = sa
m = lambda v: statistics.median(v) / 1e3 if v else float("nan")
print("dispatches %d | pre-pass %.2f us | gap pre->frame %.2f us | frame kernel %.2f us | gap frame->next %.2f us | frame period %.2f us" % (
    len(rows), m(pre), m(gap_pf), m(frame), m(gap_fp), m(period)))
