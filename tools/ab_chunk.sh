#!/bin/bash
# Same-box A/B of the bytes of index lists one k_permute_lists block walks against its LDS tile
# (list_geom; SCOARY_LIST_CHUNK_MB overrides the tree's 2 MB): kernel time on the three big
# shapes, then FETCH_SIZE / WRITE_SIZE of the kernel at the headline config (separate --pmc passes).
# A block that walks k times as many lists re-fetches its 128 KB tile k times less often
# (VERDICT round 2, item 6).     tools/ab_chunk.sh [out file]
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=${1:-$REPO/gpurun_out/ab_chunk.txt}
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("chunk %-3s MB  %-28s step %9.3f ms  k_permute_lists %9.3f ms" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["kernel_ms"]["k_permute_lists"]))'
{
echo "# tools/ab_chunk.sh: MB of index lists per k_permute_lists block (tree: 2)"
for v in 2 8 16 2 8; do          # (2, 4, 8, 16 MB twice over: profiles/r02_ab_list_chunk_bytes.txt)
  export SCOARY_LIST_CHUNK_MB=$v
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "$pick" $v cfg3
  python bench.py --no-cpu-baseline --config cfg4 2>/dev/null | python -c "$pick" $v cfg4
done
for v in 2 8; do
  export SCOARY_LIST_CHUNK_MB=$v
  python bench.py --no-cpu-baseline --config cfg5 --genes 30000 --permutations 12800 --steps 3 --warmup 1 2>/dev/null | python -c "$pick" $v "cfg5 proxy 30k x 10k x 50"
done
cd /tmp && export TMPDIR=/tmp
for v in 2 8; do
  export SCOARY_LIST_CHUNK_MB=$v
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/abpmc_${v}_$c
    rocprofv3 --kernel-trace --pmc $c -d /tmp/abpmc_${v}_$c -o p -- python $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
    python - "$v" "$c" /tmp/abpmc_${v}_$c <<'PY'
import sqlite3, sys, glob
v, c, d = sys.argv[1:4]
for db in glob.glob(d + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    for name, n, avg in con.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (c,)):
        if "k_permute_lists" in name:
            print("chunk %-3s MB  cfg3  %s per launch: %.1f MiB-units (x2 for FETCH on gfx950: %.1f MB)" % (v, c, avg / 1024, avg * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e6))
PY
  done
done
} 2>&1 | tee $OUT
