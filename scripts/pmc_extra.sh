#!/bin/bash
# instruction-mix and cache counters for the default bench workload
export TMPDIR=/tmp
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-others --no-boundary --no-sustained --no-pipelined $@"
rm -rf gpurun_out/px1 gpurun_out/px2
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES --output-format csv -d gpurun_out/px1 -o pmc -- python3 $ARGS > gpurun_out/px1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_INST_CYCLES_SMEM --output-format csv -d gpurun_out/px2 -o pmc -- python3 $ARGS > gpurun_out/px2.log 2>&1
python3 - <<'PY'
import csv,collections,statistics
for d in ('px1','px2'):
    c=collections.defaultdict(list)
    try:
        for r in csv.DictReader(open('gpurun_out/%s/pmc_counter_collection.csv'%d)):
            if 'render_kernel' in r['Kernel_Name']: c[r['Counter_Name']].append(float(r['Counter_Value']))
    except Exception as e:
        print(d,'failed',e); continue
    print(d, {k: round(statistics.mean(v[-5:])) for k,v in c.items()})
PY
