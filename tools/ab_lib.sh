#!/bin/bash
# A/B of two builds of libscoary_hip.so on one box: time (bench.py) and FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc,
# separate passes).   tools/ab_lib.sh <alt.so> [bench args]
cd "$(dirname "$0")/.."
REPO=$(pwd); ALT=$REPO/$1; shift
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-8s step %.3f ms  k_permute_lists %.3f ms  tiles %.4f ms  value %.3e" % (sys.argv[1], d["ms_per_step"], d["kernel_ms"]["k_permute_lists"], d["kernel_ms"].get("k_perm_generate_tiles", 0), d["value"]))'
for i in 1 2 3; do
  python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "$pick" default
  SCOARY_HIP_LIB=$ALT python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "$pick" alt
done
cd /tmp && export TMPDIR=/tmp
for v in default alt; do
  [ $v = alt ] && export SCOARY_HIP_LIB=$ALT
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/abpmc_$v_$c
    rocprofv3 --kernel-trace --pmc $c -d /tmp/abpmc_${v}_$c -o p -- python $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > /dev/null 2>&1
    python - "$v" "$c" /tmp/abpmc_${v}_$c <<'PY'
import sqlite3, sys, glob
v, c, d = sys.argv[1:4]
for db in glob.glob(d + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    for name, n, avg in con.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (c,)):
        if "k_permute_lists" in name:
            print("%-8s %s per launch: %.1f MiB (x2 for FETCH on gfx950 = %.1f MB)" % (v, c, avg / 1024, avg * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e6))
PY
  done
done
