#!/bin/bash
# True lane utilisation: the counter ratio SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) counts an SGPR-spill
# v_readlane / v_writelane as ONE active lane.  Build the kernels with SGPR spills sent to scratch instead
# (make VARIANT=nolane EXTRA_KERNEL_FLAGS="-mllvm -amdgpu-spill-sgpr-to-vgpr=0") and compare the ratio.
for v in "" nolane; do
  export C2RT_LIB_VARIANT=$v
  echo "== library variant: ${v:-default}"
  bash scripts/pmc_quick.sh "$1"
done
