#!/bin/bash
# Long run of the device check of chess2rt_amd/csrc/fp64_lean.h (on the GPU box, from the repo root):
#   scripts/fp64_lean_sweep.sh [log2 operands per routine, default 38 = 2.7e11] > gpurun_out/fp64_lean_sweep.json
# (tests/fp64_lean_check walks the operand space in launches of 2^30; 2^38 takes about two minutes on an MI355X.)
set -e
make -s tests/fp64_lean_check
exec ./tests/fp64_lean_check ${1:-38}
