#!/bin/bash
# quick VALU instruction census per workload: scripts/pmc_quick.sh "<workloads>"
export TMPDIR=/tmp
for w in $1; do
  rm -rf gpurun_out/pq_$w
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pq_$w -o pmc -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-others --no-boundary --no-sustained --no-pipelined --workload $w > gpurun_out/pq_$w.log 2>&1
  python3 - "$w" <<'PY'
import csv,sys,collections,statistics
w=sys.argv[1]
c=collections.defaultdict(list); dur=[]
for r in csv.DictReader(open('gpurun_out/pq_%s/pmc_counter_collection.csv'%w)):
    if 'render_kernel' in r['Kernel_Name']:
        c[r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name']=='SQ_WAVES': dur.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
m={k:statistics.mean(v[-5:]) for k,v in c.items()}
print('%-18s waves %7d  VALU/wave %7.0f  SALU/wave %6.0f  SMEM/wave %5.0f  wavecyc/wave(x4) %8.0f  lane-util %.2f  valu-act/wavecyc %.2f  dur %.3f ms' % (
  w, m['SQ_WAVES'], m['SQ_INSTS_VALU']/m['SQ_WAVES'], m['SQ_INSTS_SALU']/m['SQ_WAVES'], m['SQ_INSTS_SMEM']/m['SQ_WAVES'],
  4*m['SQ_WAVE_CYCLES']/m['SQ_WAVES'], m['SQ_THREAD_CYCLES_VALU']/(64*m['SQ_ACTIVE_INST_VALU']), m['SQ_ACTIVE_INST_VALU']/m['SQ_WAVE_CYCLES'], statistics.mean(dur[-5:])/1e6))
PY
done
