#!/usr/bin/env python3
"""Summarises a scripts/profile.sh output directory into one JSON + the
kernel-stats CSV (per-launch averages for the render kernel), for committing
under profiles/."""
import csv, json, os, sys, statistics
from collections import defaultdict

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
workload = sys.argv[4] if len(sys.argv) > 4 else None   # also write profiles/traffic_<workload>.json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402
KERNEL = "render_kernel"
summ = {"tag": tag, "kernel": None}

with open(os.path.join(src, "trace", "trace_kernel_stats.csv")) as f:
    rows = list(csv.DictReader(f))
summ["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in rows]
with open(os.path.join(src, "trace", "trace_kernel_trace.csv")) as f:
    all_rows = list(csv.DictReader(f))
tr_all = [r for r in all_rows if KERNEL in r["Kernel_Name"]]
# the per-tile culling masks are written by a pre-pass kernel in front of every frame launch (one lane per tile)
pre = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in all_rows if "tile_masks_kernel" in r["Kernel_Name"]][-20:]
if pre:
    summ["tile_masks_prepass_avg_duration_us_timed"] = statistics.mean(pre) / 1e3
# Nested-CSG scenes launch the frame kernel twice per frame: the whole grid with the reduced hit stack,
# then a fixed 2048-workgroup grid over the tiles that overflowed it (usually none).  The frame launch
# is the one with the larger grid; the retry launches are reported beside it.
main_grid = max(int(r["Grid_Size_X"]) for r in tr_all)
timed_name = [r for r in tr_all if int(r["Grid_Size_X"]) == main_grid][-1]["Kernel_Name"]  # not the ray-counting instance
tr = [r for r in tr_all if int(r["Grid_Size_X"]) == main_grid and r["Kernel_Name"] == timed_name]
retry = [r for r in tr_all if int(r["Grid_Size_X"]) != main_grid and r["Kernel_Name"] == timed_name]
durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr]
summ["kernel"] = tr[0]["Kernel_Name"]
summ["launches"] = len(durs)
timed = durs[-20:]  # the last 20 launches are bench.py's timed steps
summ["avg_duration_us_timed"] = statistics.mean(timed) / 1e3
summ["median_duration_us_timed"] = statistics.median(timed) / 1e3
if retry:
    rd = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in retry][-20:]
    summ["retry_launches"] = len(retry)
    summ["retry_avg_duration_us_timed"] = statistics.mean(rd) / 1e3
# metadata of the TIMED launches (the first launch of a run is the ray-counting instance, another kernel)
summ["kernel"] = tr[-1]["Kernel_Name"]
for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size_X", "Grid_Size_X"):
    summ[k] = tr[-1][k]

counters = defaultdict(list)
pmc_dur = {}
for sub in sorted(os.listdir(src)):
    p = os.path.join(src, sub, "pmc_counter_collection.csv")
    if not os.path.exists(p):
        continue
    with open(p) as f:
        for r in csv.DictReader(f):
            if r["Kernel_Name"] == summ["kernel"] and int(r["Grid_Size"]) == main_grid:
                counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
                pmc_dur.setdefault(sub, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
summ["pmc_per_launch_avg"] = {k: statistics.mean(v[-20:]) for k, v in counters.items()}
summ["pmc_pass_avg_duration_us"] = {k: statistics.mean(v[-20 * (len(v) // 24 or 1):]) / 1e3 for k, v in pmc_dur.items()}
c = summ["pmc_per_launch_avg"]
d = {}
if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
    # SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count quad-cycles summed over the chip's 1024 SIMDs;
    # GRBM_GUI_ACTIVE is the busy-cycle count summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS section)
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
    d["valu_busy_fraction"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles
    d["resident_waves_per_simd"] = 4.0 * c["SQ_WAVE_CYCLES"] / simd_cycles
    d["clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (statistics.mean(pmc_dur.get("pmc_sq2", [0])[-20:]) or 1)
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
    d["lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
    d["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # rocprofv3 reports both in KiB
    d["fetch_bytes"] = c["FETCH_SIZE"] * 1024
    d["write_bytes"] = c["WRITE_SIZE"] * 1024
    d["hbm_bytes_per_launch"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
summ["derived"] = d
summ["kernel_source_hash"] = kernel_source_hash()
os.makedirs(dst, exist_ok=True)
with open(os.path.join(dst, "%s_summary.json" % tag), "w") as f:
    json.dump(summ, f, indent=1)
with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w") as f:
    f.write(open(os.path.join(src, "trace", "trace_kernel_stats.csv")).read())
if workload and "hbm_bytes_per_launch" in d and "SQ_INSTS_VALU" in c:
    # the per-launch counts bench.py may combine with a live kernel time — only for the kernels hashed here
    fetch, write = d["fetch_bytes"], d["write_bytes"]
    tr_json = {
        "workload": workload,
        "kernel_source_hash": summ["kernel_source_hash"],
        "source": "profiles/%s_summary.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, averages over the timed launches)" % tag,
        "fetch_bytes": fetch,
        "write_bytes": write,
        "hbm_bytes_per_launch_uncorrected": fetch + write,
        # MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-B read requests as 64 B -> FETCH_SIZE doubled; WRITE_SIZE as is
        "hbm_bytes_per_launch": 2 * fetch + write,
        "valu_insts_per_launch": c["SQ_INSTS_VALU"],
        "salu_insts_per_launch": c.get("SQ_INSTS_SALU"),
        "valu_busy_fraction": d.get("valu_busy_fraction"),
        "resident_waves_per_simd": d.get("resident_waves_per_simd"),
        "lane_utilisation": d.get("lane_utilisation"),
        "valu_insts_per_wave": d.get("valu_insts_per_wave"),
        "kernel_avg_us": summ["avg_duration_us_timed"],
    }
    # the fp64 arithmetic the kernel actually EXECUTES (wave-instructions per launch; x 64 lanes = lane-operations),
    # for bench.py's roofline.fp64_executed
    f64 = {k: c.get("SQ_INSTS_VALU_%s_F64" % k) for k in ("ADD", "MUL", "FMA", "TRANS")}
    if all(v is not None for v in f64.values()):
        tr_json["fp64_wave_insts_per_launch"] = {k.lower(): v for k, v in f64.items()}
    mix = {k: c.get(k) for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
                                 "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM",
                                 "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")}
    if any(v is not None for v in mix.values()):
        tr_json["other_wave_insts_per_launch"] = {k.replace("SQ_INSTS_", "").lower(): v for k, v in mix.items() if v is not None}
    with open(os.path.join(dst, "traffic_%s.json" % workload), "w") as f:
        json.dump(tr_json, f, indent=1)
print(json.dumps(summ, indent=1))
