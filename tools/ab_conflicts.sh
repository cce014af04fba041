#!/bin/bash
# tools/ab_conflicts.py on the box: kernel times of the three gene sets per shape, then the LDS
# counters of k_permute_lists per variant (separate rocprofv3 --pmc passes, one process each).
#   tools/ab_conflicts.sh [out file]
cd "$(dirname "$0")/.."
REPO=$(pwd); OUT=${1:-$REPO/gpurun_out/ab_conflicts.txt}
{
echo "# tools/ab_conflicts.sh: LDS bank conflicts of k_permute_lists -- random genes (spec S6 order), class-balanced genes"
echo "# (no hole fillers), random genes with the entry order shuffled (no class rotation); same box, same list lengths"
python tools/ab_conflicts.py --shapes cfg3 cfg4 cfg5
cd /tmp && export TMPDIR=/tmp
for shape in cfg3 cfg4 cfg5; do
for v in random balanced shuffled; do
  rm -rf /tmp/abc_$v
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU \
      -d /tmp/abc_$v -o p -- python $REPO/tools/ab_conflicts.py --shapes $shape --variant $v --steps 2 --warmup 1 > /dev/null 2>&1
  python - "$shape" "$v" /tmp/abc_$v <<'PY'
import sqlite3, sys, glob
shape, v, d = sys.argv[1:4]
vals = {}
for db in glob.glob(d + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    for name, ctr, avg in con.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
        if "k_permute_lists" in name:
            vals[ctr] = avg
if vals:
    c, a = vals.get("SQ_LDS_BANK_CONFLICT", 0), vals.get("SQ_LDS_IDX_ACTIVE", 1)
    print("%-5s %-9s SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.1f %%   SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES = %.3f   SQ_INSTS_VALU %.4g"
          % (shape, v, 100.0 * c / a, vals.get("SQ_WAIT_INST_LDS", 0) / max(vals.get("SQ_WAVE_CYCLES", 1), 1), vals.get("SQ_INSTS_VALU", 0)))
else:
    print(shape, v, "no counters")
PY
done
done
} 2>&1 | tee $OUT
