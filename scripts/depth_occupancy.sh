#!/bin/bash
# occupancy / hit-stack capacity sweep of the nested-CSG instances (scripts/depth_occupancy.py), on the GPU box:
# scripts/depth_occupancy.sh "<variants>" "<caps>"   ("-" = the built-in first-pass capacity)
for d in 2 3 4; do for v in $1; do for cap in $2; do
  if [ "$v" = "base" ]; then export C2RT_LIB_VARIANT=; else export C2RT_LIB_VARIANT=$v; fi
  if [ "$cap" = "-" ]; then unset C2RT_CSG_FIRST_CAP; else export C2RT_CSG_FIRST_CAP=$cap; fi
  python scripts/depth_occupancy.py $d 2>/dev/null | tail -1
done; done; done
