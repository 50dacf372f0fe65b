#!/bin/bash
# A/B of environment knobs of ONE library on the GPU box: scripts/ab_env.sh "<VAR=val|-> ..." [bench args]
# e.g. scripts/ab_env.sh "- C2RT_PALETTE=0" --workload zaphod_4k_dof25   ("-" = no variable set)
VARS=$1; shift
for v in $VARS; do
  if [ "$v" = "-" ]; then pre=""; else pre="$v"; fi
  env $pre python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-others --no-boundary --no-sustained --no-pipelined "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s %-18s %8.0f Mray/s  %7.3f ms/frame  kernel %7.3f ms' % ('$v', d['config']['name'], d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done
