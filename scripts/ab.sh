#!/bin/bash
# A/B of library variants on the GPU box: scripts/ab.sh "<variants>" [bench args]
# variants: base = chess2rt_amd/libc2rt.so; exact = the diagnostics library with C2RT_EXACT=1 (every tile through the
# compiler's IEEE divide / sqrt: the round-2 arithmetic); anything else = chess2rt_amd/libc2rt_<name>.so
# (make VARIANT=<name> EXTRA_HIPFLAGS=...)
VARS=$1; shift
for v in $VARS; do
  unset C2RT_EXACT
  if [ "$v" = "base" ]; then export C2RT_LIB_VARIANT=; elif [ "$v" = "exact" ]; then export C2RT_LIB_VARIANT=diag; export C2RT_EXACT=1;
  else export C2RT_LIB_VARIANT=$v; fi
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-others --no-boundary --no-sustained --no-pipelined "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %-18s %8.0f Mray/s  %7.3f ms/frame  kernel %7.3f ms' % ('$v', d['config']['name'], d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done
