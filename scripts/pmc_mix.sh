#!/bin/bash
# VALU instruction mix of one workload's frame kernel: scripts/pmc_mix.sh <workload>
export TMPDIR=/tmp
W=$1
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-others --no-boundary --no-sustained --no-pipelined --workload $W"
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
            "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
            "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED"; do
  tag=$(echo $pass | md5sum | cut -c1-6)
  rm -rf gpurun_out/mix_$tag
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/mix_$tag -o pmc -- python3 $ARGS > gpurun_out/mix_$tag.log 2>&1 || { tail -3 gpurun_out/mix_$tag.log; continue; }
  python3 - gpurun_out/mix_$tag/pmc_counter_collection.csv <<'PY'
import csv,collections,statistics,sys
c=collections.defaultdict(list); name=None
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'render_kernel' in r['Kernel_Name']]
last=rows[-1]['Kernel_Name']
for r in rows:
    if r['Kernel_Name']==last: c[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(statistics.mean(v[-5:])/1e6,1) for k,v in c.items()}, "(millions per launch)")
PY
done
