#!/bin/bash
# A/B on one box: the round-1 library (worktree _ab/old at 1884429) against the current one,
# same bench command, interleaved.  Prints kernel ms and ms/step per variant.
cd "$(dirname "$0")/.."
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-10s step %.3f ms  k_permute_lists %.3f ms  tiles %.3f  value %.3e" % (sys.argv[1], d["ms_per_step"], d["kernel_ms"]["k_permute_lists"], d["kernel_ms"]["k_perm_generate_tiles"], d["value"]))'
cfg=${1:-cfg3}
for i in 1 2 3; do
  (cd _ab/old && python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$pick" old)
  python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$pick" new
done
