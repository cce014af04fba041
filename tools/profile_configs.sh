#!/bin/bash
# Runs on the GPU box: counter profiles (tools/profile.sh) of the non-headline BASELINE
# configs, summarised on the box so that the bench lines taken right after carry a
# sha-matched roofline.  Summaries and bench lines land in gpurun_out/; re-run
# tools/rocpd_summary.py on the merged gpurun_out/prof_<tag>_<cfg> to install them under profiles/.
#   tools/profile_configs.sh <tag> cfg4 cfg2 ...
set -u
TAG=${1:-r02h}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
for CFG in "$@"; do
  bash tools/profile.sh ${TAG}_$CFG --config $CFG > gpurun_out/profile_${TAG}_$CFG.log 2>&1
  python tools/rocpd_summary.py gpurun_out/prof_${TAG}_$CFG profiles/${TAG}_$CFG $CFG \
      >> gpurun_out/profile_${TAG}_$CFG.log 2>&1
  python bench.py --config $CFG 2> gpurun_out/bench_${TAG}_$CFG.err | tail -1 > gpurun_out/bench_${TAG}_$CFG.json
  if [ "$CFG" = cfg2 ]; then
    python bench.py --config cfg2 --graph 2>> gpurun_out/bench_${TAG}_$CFG.err | tail -1 \
        > gpurun_out/bench_${TAG}_cfg2_graph.json
  fi
  # keep the merged output small: the sqlite traces are what rocpd_summary.py needs
  find gpurun_out/prof_${TAG}_$CFG -type f ! -name '*.db' ! -name '*.txt' ! -name '*.log' -delete
  du -sh gpurun_out/prof_${TAG}_$CFG
done
tail -c 600 gpurun_out/bench_${TAG}_*.json
