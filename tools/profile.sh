#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py and
# the HBM-traffic PMC passes (separate runs, as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE cannot share a pass; never combined with sys-trace).
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries into profiles/.
#   tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# no sustained window, no telemetry thread: under the profiler they only add thousands of kernel records
# ... and no K1 sweep / strong-scaling extras: they launch the same kernels on OTHER shapes, whose counters would be averaged in
BENCH="python $REPO/bench.py --no-cpu-baseline --sustain-seconds 0 --telemetry-ms 0 --no-k1-cold --strong-extra off $*"
# the kernel sources these counters belong to (bench.py refuses a summary whose hash differs)
python -c "import sys; sys.path.insert(0, '$REPO'); import bench; print(bench.kernel_source_sha())" > "$OUT/kernel_source_sha256.txt"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- $BENCH > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o fetch -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o write -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d "$OUT/pmc_sq" -o sq -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d "$OUT/pmc_lds" -o lds -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_lds.log" 2>&1
find "$OUT" -type f | head -50
