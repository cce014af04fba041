#!/bin/bash
# SQ / LDS counters of the list-driven permutation kernel (separate --pmc passes).
#   tools/profile_lists.sh [tag] [bench args...]      -> gpurun_out/prof_lists_<tag>/
set -u
TAG=${1:-cfg3}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_lists_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --kernel lists --no-cpu-baseline --steps 2 --warmup 1 $*"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_sq -o sq -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM -d $OUT/pmc_lds -o lds -- $BENCH > $OUT/pmc_lds.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
