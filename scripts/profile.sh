#!/bin/bash
# rocprofv3 passes for the bench workload (run on the GPU box, from the repo root).
# usage: scripts/profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{trace,pmc_*}/ ; counters are collected in their
# own runs (never together with tracing other than --kernel-trace).
set -e -o pipefail
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-others --no-boundary --no-sustained --no-pipelined $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o pmc -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/pmc_f64 -o pmc -- python3 $ARGS > $OUT/pmc_f64.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/pmc_mix -o pmc -- python3 $ARGS > $OUT/pmc_mix.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_mix2 -o pmc -- python3 $ARGS > $OUT/pmc_mix2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -30
