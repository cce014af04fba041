#!/bin/bash
# Same-box A/B of SCOARY_LIST_CHUNK_MB on the cfg5 geometry (N = 10 000, TW = 4, one lane per gene): kernel time and
# FETCH_SIZE of k_permute_lists per launch.  A block = one 160 KB label tile x one chunk of index lists, so the tile is
# re-loaded once per chunk: 2 MB chunks re-load every tile 8x as often as 16 MB chunks.   tools/ab_chunk_cfg5.sh [genes]
cd "$(dirname "$0")/.."
REPO=$(pwd); G=${1:-30000}
ARGS="--no-cpu-baseline --sustain-seconds 0 --telemetry-ms 0 --config cfg5 --genes $G --permutations 43520 --steps 3 --warmup 1"
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("chunk %-3s MB  step %9.3f ms  k_permute_lists %9.3f ms  value %.4e" % (sys.argv[1], d["ms_per_step"], d["kernel_ms"]["k_permute_lists"], d["value"]))'
for v in 2 8 16 32 2 16; do
  SCOARY_LIST_CHUNK_MB=$v python bench.py $ARGS 2>/dev/null | python -c "$pick" $v
done
cd /tmp && export TMPDIR=/tmp
for v in 2 8 16 32; do
  rm -rf /tmp/abpmc_$v
  SCOARY_LIST_CHUNK_MB=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/abpmc_$v -o p -- python $REPO/bench.py $ARGS --steps 1 --warmup 1 > /dev/null 2>&1
  python - "$v" /tmp/abpmc_$v <<'PY'
import sqlite3, sys, glob
v, d = sys.argv[1:3]
for db in glob.glob(d + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    for name, n, avg in con.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name"):
        if "k_permute_lists" in name:
            print("chunk %-3s MB  FETCH_SIZE per launch x2 (gfx950): %.1f GB" % (v, avg * 1024 * 2 / 1e9))
PY
done
