#!/bin/bash
# round 6: blocks per launch of k_permute_lists on launch-bound shards -- SCOARY_LIST_ROUNDS (rounds of blocks over the
# CUs the geometry asks for; 16 = the tree) on one box, interleaved.  A block walks >= 16 wave groups (one per
# wavefront), so fewer rounds = more groups per wavefront and fewer tile loads / block starts.
cd "$(dirname "$0")/.."
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("rounds %-3s %-28s step %.4f ms  k_permute_lists %.4f ms  value %.4e" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["kernel_ms"]["k_permute_lists"], d["value"]))'
for args in "--config cfg4 --genes 25000" "--config cfg4 --genes 50000" "--config cfg3 --genes 6250" "--config cfg2" "--config cfg4"; do
  for i in 1 2; do
    for r in 16 4 2; do
      SCOARY_LIST_ROUNDS=$r python bench.py --no-cpu-baseline --no-k1-cold --sustain-seconds 0 --strong-extra off --telemetry-ms 0 $args 2>/dev/null | python -c "$pick" $r "$args"
    done
  done
done
