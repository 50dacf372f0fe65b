#!/bin/bash
# A/B of library variants on the GPU box: scripts/ab.sh "<variants>" [bench args]
VARS=$1; shift
for v in $VARS; do
  if [ "$v" = "base" ]; then export C2RT_LIB_VARIANT=; else export C2RT_LIB_VARIANT=$v; fi
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-others --no-boundary "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %-18s %8.0f Mray/s  %7.3f ms/frame  kernel %7.3f ms' % ('$v', d['config']['name'], d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done
