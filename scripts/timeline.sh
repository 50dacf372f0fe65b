#!/bin/bash
# scripts/timeline.sh "<workloads>": rocprofv3 kernel trace of back-to-back frames, then scripts/timeline.py
export TMPDIR=/tmp
for w in $1; do
  rm -rf gpurun_out/tl_$w
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$w -o tl -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-others --no-boundary --no-sustained --no-pipelined --workload $w > gpurun_out/tl_$w.log 2>&1
  echo -n "$w: "; python3 scripts/timeline.py $(find gpurun_out/tl_$w -name "*kernel_trace.csv" | head -1)
done
