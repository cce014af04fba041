#!/bin/bash
# A/B, round-1 library (worktree _ab/old) vs current, on other tile widths.
cd "$(dirname "$0")/.."
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-4s %-34s step %.3f ms  k_permute_lists %.3f ms" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["kernel_ms"]["k_permute_lists"]))'
run() {  # label dir args...
  local lab=$1 dir=$2; shift 2
  (cd $dir && python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "$pick" $lab "$*")
}
for shape in "--isolates 5000 --genes 25000 --traits 4 --permutations 4096" \
             "--isolates 10000 --genes 20000 --traits 2 --permutations 4096" \
             "--isolates 10000 --genes 20000 --traits 50 --permutations 2048" \
             "--isolates 20000 --genes 20000 --traits 2 --permutations 4096" \
             "--isolates 40000 --genes 10000 --traits 2 --permutations 2048"; do
  run old _ab/old $shape; run new . $shape; run old _ab/old $shape; run new . $shape
done
