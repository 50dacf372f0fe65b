#!/bin/bash
# Round profiles for the judged workloads: scripts/profile_all.sh <round-tag> "<workloads>"
# (run on the GPU box from the repo root; summaries land in gpurun_out/profiles_<tag>/, copy them to profiles/)
set -o pipefail
TAG=$1; WL=${2:-"lecture5_4k_aa5 csg_stress_4k_4spp zaphod_4k_dof25"}
for w in $WL; do
  bash scripts/profile.sh ${TAG}_$w --workload $w > gpurun_out/profile_${TAG}_$w.log 2>&1 || { echo "profile $w failed"; tail -5 gpurun_out/profile_${TAG}_$w.log; exit 1; }
  python3 scripts/summarize_profile.py gpurun_out/prof_${TAG}_$w gpurun_out/profiles_$TAG ${TAG}_$w $w > /dev/null || exit 1
  python3 - "$TAG" "$w" <<'PY'
import json,sys
tag,w=sys.argv[1:3]
d=json.load(open('gpurun_out/profiles_%s/%s_%s_summary.json'%(tag,tag,w)))
c=d['pmc_per_launch_avg']; k=d['derived']
print('%-20s kernel %.3f ms  VGPR %s scratch %s LDS %s | VALU busy %.3f lane util %.3f VALU/wave %.0f | fetch %.1f MB write %.1f MB' % (
  w, d['avg_duration_us_timed']/1e3, d['VGPR_Count'], d['Scratch_Size'], d['LDS_Block_Size'], k.get('valu_busy_fraction',0), k.get('lane_utilisation',0), k.get('valu_insts_per_wave',0), k.get('fetch_bytes',0)/1e6, k.get('write_bytes',0)/1e6))
PY
done
