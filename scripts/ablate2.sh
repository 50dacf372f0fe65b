#!/bin/bash
# second ablation pass: where the waves wait (same dispatch order as ablate.sh)
export TMPDIR=/tmp
rm -rf gpurun_out/abl2
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/abl2 -o pmc -- python3 scripts/ablate.py > gpurun_out/abl2.log 2>&1
python3 - <<'PY'
import csv,collections,sys
sys.path.insert(0,'scripts')
from ablate import SUBSETS
rows=collections.OrderedDict()
for r in csv.DictReader(open('gpurun_out/abl2/pmc_counter_collection.csv')):
    if 'render_kernel' in r['Kernel_Name']:
        rows.setdefault(r['Dispatch_Id'],{})[r['Counter_Name']]=float(r['Counter_Value'])
        rows[r['Dispatch_Id']]['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
for (name,_),(k,m) in zip(SUBSETS,rows.items()):
    w=m['SQ_WAVES']
    print('%-14s life %6.0f wait_mem %6.0f (%.2f) wait_inst %6.0f (%.2f) active %6.0f  SMEM %4.0f LDS %4.0f VMEM %3.0f  dur %.3f'%(name,m['SQ_WAVE_CYCLES']/w,m['SQ_WAIT_ANY']/w,m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES'],m['SQ_WAIT_INST_ANY']/w,m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES'],m['SQ_ACTIVE_INST_ANY']/w,m['SQ_INSTS_SMEM']/w,m['SQ_INSTS_LDS']/w,m['SQ_INSTS_VMEM_RD']/w,m['dur']))
PY
