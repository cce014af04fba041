#!/bin/bash
# Shader clock and socket power while the headline workload runs back to back
# (rocm-smi sampled every 1.5 s; the first samples fall into warm-up / list build).
#   tools/clock_sample.sh > gpurun_out/clock_power.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
(timeout 100 python bench.py --steps 3000 --warmup 5 --no-cpu-baseline > /tmp/clock_bench.log 2>&1 &)
sleep 12
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Package Power" | sed 's/^GPU\[0\]\s*: //'
  sleep 1.5
done
wait
python - <<'PY'
import json
d = json.loads(open('/tmp/clock_bench.log').read().strip().splitlines()[-1])
print("bench: %.3e %s, %.3f ms/step, k_permute_lists %.3f ms (3000 steps)" % (
    d["value"], d["unit"], d["ms_per_step"], d["kernel_ms"]["k_permute_lists"]))
PY
