#!/bin/bash
# A/B of library variants over several workloads inside ONE gpurun call: scripts/ab_multi.sh "<variants>" "<workloads>" [rounds]
# (variants as in scripts/ab.sh; each (variant, workload) pair is measured `rounds` times, interleaved, so that drift
# of the box shows as spread between rounds rather than as a difference between variants)
VARS=$1; WLS=$2; ROUNDS=${3:-2}
for r in $(seq 1 $ROUNDS); do
  for w in $WLS; do
    bash scripts/ab.sh "$VARS" --workload $w
  done
done
