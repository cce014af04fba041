#!/bin/bash
# Shader clock and socket power under a pure v_bitop3_b32 stream at 4 and at 8 waves per SIMD
# (tools/valu_banks.hip, sustained mode), next to the achieved lane-ops/s: what the op itself
# reaches on this chip under the power cap, at both occupancies.
#   tools/clock_valu.sh > gpurun_out/clock_valu.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
hipcc --offload-arch=gfx950 -O3 tools/valu_banks.hip -o /tmp/valu_banks 2>/dev/null || exit 1
for w in 4 8; do
  echo "== $w waves per SIMD"
  (/tmp/valu_banks $w 12 > /tmp/valu_$w.log 2>&1 &)
  sleep 5
  for i in 1 2 3 4; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Package Power" | sed 's/^GPU\[0\]\s*: //'
    sleep 1.5
  done
  sleep 3
  cat /tmp/valu_$w.log
done
