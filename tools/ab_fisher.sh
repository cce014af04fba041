#!/bin/bash
# round 6: k_fisher with the canonical orientation + the N <= 170 SciPy restatement (tree) against the kernel before
# (_ab/prefisher.so = tools/build_alt.sh prefisher aec7976), same box, interleaved
cd "$(dirname "$0")/.."
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-5s %-28s step %.4f ms  k_fisher %.4f ms  k_permute_lists %.4f ms  value %.4e" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["kernel_ms"]["k_fisher"], d["kernel_ms"]["k_permute_lists"], d["value"]))'
for args in "--config cfg3" "--config cfg4" "--config cfg4 --genes 25000" "--config cfg2"; do
  for i in 1 2; do
    for v in prev new; do
      lib=""; [ $v = prev ] && lib="$PWD/_ab/prefisher.so"
      SCOARY_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-k1-cold --sustain-seconds 0 --strong-extra off --telemetry-ms 0 $args 2>/dev/null | python -c "$pick" $v "$args"
    done
  done
done
