#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/abl
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/abl -o pmc -- python3 scripts/ablate.py > gpurun_out/abl.log 2>&1
python3 - <<'PY'
import csv,collections,sys
sys.path.insert(0,'scripts')
from ablate import SUBSETS
rows=collections.OrderedDict()
for r in csv.DictReader(open('gpurun_out/abl/pmc_counter_collection.csv')):
    if 'render_kernel' in r['Kernel_Name']:
        rows.setdefault(r['Dispatch_Id'],{})[r['Counter_Name']]=float(r['Counter_Value'])
        rows[r['Dispatch_Id']]['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
        rows[r['Dispatch_Id']]['k']=r['Kernel_Name'][30:70]
for (name,_),(k,m) in zip(SUBSETS,rows.items()):
    print('%-14s VALU/wave %6.0f SALU/wave %5.0f lane-util %.2f  dur %.3f ms  %s'%(name,m['SQ_INSTS_VALU']/m['SQ_WAVES'],m['SQ_INSTS_SALU']/m['SQ_WAVES'],m['SQ_THREAD_CYCLES_VALU']/(64*m['SQ_ACTIVE_INST_VALU']),m['dur'],m['k']))
PY
