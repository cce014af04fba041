#!/bin/bash
# Same-box A/B of the index-list chunk size of k_permute_lists (list_geom: bytes of lists one
# (trait, tile) block walks): alternative builds _ab/lib_chunk<MB>.so made from a copy of csrc/
# with the constant replaced; 2 MB is the tree's value.
cd "$(dirname "$0")/.."
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-10s %-28s step %9.3f ms  k_permute_lists %9.3f ms" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["kernel_ms"]["k_permute_lists"]))'
for rep in 1 2; do
for v in default 1 4 8 16; do
  if [ $v = default ]; then unset SCOARY_HIP_LIB; else export SCOARY_HIP_LIB=$(pwd)/_ab/lib_chunk$v.so; fi
  python bench.py --no-cpu-baseline --config cfg5 --genes 30000 --permutations 12800 --steps 3 --warmup 1 2>/dev/null | python -c "$pick" $v "cfg5 proxy 30k x 10k x 50"
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "$pick" $v cfg3
  python bench.py --no-cpu-baseline --config cfg4 2>/dev/null | python -c "$pick" $v cfg4
done
done
