#!/bin/bash
# Round 6: the XCD-aware block map of k_permute_lists for the one-lane-per-gene geometry
# (SCOARY_LIST_XCD_MAP=1: every chunk of index lists walked by ONE XCD, snake order over the rows of eight),
# same box, same process image: time, sustained clock / power and FETCH_SIZE per launch.
#   tools/ab_xcd_map.sh [bench args; default: --config cfg5 --steps 3 --warmup 1]
cd "$(dirname "$0")/.."
REPO=$(pwd)
ARGS="${*:---config cfg5 --steps 3 --warmup 1}"
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; t=d.get("telemetry") or {}; print("%-10s step %.2f ms  k_permute_lists %.2f ms  value %.4e  sclk %s MHz  power %s W  useful_frac %.3f" % (sys.argv[1], d["ms_per_step"], d["kernel_ms"]["k_permute_lists"], d["value"], t.get("sclk_mhz_mean") and round(t["sclk_mhz_mean"]), t.get("socket_power_w_mean") and round(t["socket_power_w_mean"]), r["useful_frac"]))'
for i in 1 2; do
  for v in 0 1; do
    SCOARY_LIST_XCD_MAP=$v python bench.py --no-cpu-baseline --no-k1-cold --sustain-seconds 0 $ARGS 2>/dev/null | python -c "$pick" xcd_map=$v
  done
done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  for c in FETCH_SIZE "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
    tag=$(echo $c | cut -d' ' -f1)
    rm -rf /tmp/abx_${v}_$tag
    SCOARY_LIST_XCD_MAP=$v rocprofv3 --kernel-trace --pmc $c -d /tmp/abx_${v}_$tag -o p -- python $REPO/bench.py --no-cpu-baseline --no-k1-cold --sustain-seconds 0 --telemetry-ms 0 --steps 1 --warmup 1 ${ARGS/--steps 3 --warmup 1/} > /tmp/abx_${v}_$tag.log 2>&1
    python - "$v" /tmp/abx_${v}_$tag <<'PY'
import sqlite3, sys, glob
v, d = sys.argv[1:3]
for db in glob.glob(d + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    for name, ctr, n, avg in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if "k_permute_lists" in name:
            extra = "  (x2 x1024 on gfx950 = %.1f GB)" % (avg * 2048 / 1e9) if ctr == "FETCH_SIZE" else ""
            print("xcd_map=%s %-22s per launch (%d launches): %.4e%s" % (v, ctr, n, avg, extra))
PY
  done
done
