#!/bin/bash
# GPU-box sessions collecting the round-6 evidence (outputs under gpurun_out/r06/).
#   tools/round6_run.sh [part ...]     parts: tests bench driver balance rank8 e2e k1 xcd tree treepmc pairwise cfg5cli seg profiles timeline soak curve fuzzextra longsoak
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
PARTS="${*:-tests bench balance rank8 e2e}"
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
if has tests; then
(time python -m pytest tests/ -q -m gpu -x --durations=12) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
fi
if has bench; then
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_cfg3_driver_command.json 2> $O/bench_cfg3_driver_command.err
tail -c 600 $O/bench_cfg3_driver_command.err
fi
if has driver; then
# the driver's command, twice, with its wall clock (the default line now also times cfg4 for scaling_strong)
for i in 1 2; do
  SECONDS=0
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_cfg3_driver_command_$i.json 2> $O/bench_cfg3_driver_command_$i.err
  echo "driver command run $i: wall ${SECONDS} s"
done
fi
if has balance; then
python tools/shard_balance.py --config cfg3 --gene-order sorted > $O/shard_balance_cfg3_sorted.txt 2>&1
python tools/shard_balance.py --config cfg3 --gene-order config > $O/shard_balance_cfg3_config_order.txt 2>&1
python tools/shard_balance.py --exampledata tests/golden/exampledata/Gene_presence_absence.csv.gz > $O/shard_balance_exampledata.txt 2>&1
grep -h "^#\|max / mean\|partition\|=>" $O/shard_balance_*.txt
# the 8-rank shared-GPU rehearsal on frequency-sorted genes, both partitions (functional + per_rank lines)
for part in stride contiguous; do
python bench.py --gpus 8 --share-gpu --backend gloo --config cfg3 --scaling strong --gene-order sorted --partition $part \
  --verify-gather --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_cfg3_sorted_8ranks_shared_$part.json 2> $O/bench_cfg3_sorted_8ranks_shared_$part.err
done
fi
if has rank8; then
for i in 1 2; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$i \
  bench.py --exercise-exchange --verify-gather --no-cpu-baseline --config cfg4 --genes 25000 --scaling strong \
  > $O/bench_cfg4_rank_of_8_sharded_$i.json 2> $O/bench_cfg4_rank_of_8_sharded_$i.err
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 \
  bench.py --exercise-exchange --verify-gather --no-cpu-baseline --config cfg4 --genes 25000 --scaling strong --no-graph \
  > $O/bench_cfg4_rank_of_8_sharded_eager.json 2>/dev/null
python bench.py --config cfg4 --genes 25000 --no-cpu-baseline > $O/bench_cfg4_rank_of_8_single_process.json 2>/dev/null
fi
if has e2e; then
python tools/e2e_vcf.py > $O/e2e_cli_cfg4_vcf.txt 2>&1
python tools/e2e_synth.py --genes 50000 --isolates 2000 --traits 10 --permute 10000 > $O/e2e_cli_cfg3.txt 2>&1
tail -25 $O/e2e_cli_cfg4_vcf.txt; tail -20 $O/e2e_cli_cfg3.txt
fi
if has k1; then
# K1 as an HBM stream, T = 1 ... 50 (roofline_k1.cold), the round-5 kernel (_ab/r5.so: tools/build_alt.sh r5 <rev>) against
# the double-buffered one, interleaved on one box
for i in 1 2; do
  for v in r5 new; do
    lib=""; [ $v = r5 ] && lib="$PWD/_ab/r5.so"
    SCOARY_HIP_LIB=$lib python bench.py --no-cpu-baseline --sustain-seconds 0 --strong-extra off --steps 5 --warmup 2 2>/dev/null \
      | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['roofline_k1']['cold']
print('$v pass $i: copy %.0f GB/s | ' % c['measured_copy_peak_gbs'] + ' | '.join('T=%d %.1f us %.2f TB/s (%.3f of 8, %s %.2f)' % (r['traits'], r['cold_ms_median'] * 1e3, r['gbs'] / 1e3, r['hbm_frac'], r.get('bound', '-'), r.get('frac_of_bound', 0)) for r in c['runs']))
json.dump(c, open('$O/k1_stream_${v}_$i.json', 'w'))"
  done
done
fi
if has xcd; then
bash tools/ab_xcd_map.sh > $O/ab_xcd_map_cfg5.txt 2>&1
cat $O/ab_xcd_map_cfg5.txt
fi
if has tree; then
for n in 2000 5000 10000; do
  python tools/bench_tree.py --isolates $n > $O/bench_tree_$n.json 2> $O/bench_tree_$n.err
  tail -c 1200 $O/bench_tree_$n.json; echo
done
fi
if has treepmc; then
( cd /tmp
  rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof_tree/stats -o stats -- python $OLDPWD/tools/bench_tree.py --isolates 5000 > $OLDPWD/$O/prof_tree_stats.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-include-regex "k_hamming|k_tree_dp" -d $OLDPWD/$O/prof_tree/pmc_sq -o sq -- python $OLDPWD/tools/bench_tree.py --isolates 5000 --kernels-only > $OLDPWD/$O/prof_tree_pmc.log 2>&1 )
python tools/rocpd_summary.py $O/prof_tree profiles/r06_tree5000 tree5000 > $O/prof_tree_summary.log 2>&1
find $O/prof_tree -type f ! -name '*.db' -delete; find $O/prof_tree -name '*.db' -size +20M -delete
fi
if has pairwise; then
# default mode (pairwise-comparison tree stage) end to end on the cfg3-sized table
python tools/e2e_synth.py --pairwise --genes 50000 --isolates 2000 --traits 10 --permute 1000 > $O/e2e_cli_cfg3_pairwise.txt 2>&1
tail -30 $O/e2e_cli_cfg3_pairwise.txt
fi
if has cfg5cli; then
python tools/e2e_cfg5_shard.py > $O/e2e_cli_cfg5_shard.txt 2>&1
cat $O/e2e_cli_cfg5_shard.txt
fi
if has seg; then
python tools/ab_conflicts.py --shapes wide > $O/ab_seg_conflicts.txt 2>&1
cat $O/ab_seg_conflicts.txt
for p in 1024 4096; do
python bench.py --genes 20000 --isolates 50000 --traits 2 --permutations $p --no-cpu-baseline --strong-extra off > $O/bench_wide_50000_P$p.json 2>/dev/null
done
fi
if has profiles; then
bash tools/profile_configs.sh r06 cfg3 cfg4 cfg2 > $O/profile_configs.log 2>&1
bash tools/profile_configs.sh r06 cfg5 > $O/profile_cfg5.log 2>&1
bash tools/profile.sh r06_wide --genes 20000 --isolates 50000 --traits 2 --permutations 1024 > $O/profile_wide.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_r06_wide profiles/r06_wide50000 wide50000 >> $O/profile_wide.log 2>&1
find gpurun_out/prof_r06_wide gpurun_out/prof_r06* -type f ! -name '*.db' ! -name '*.txt' ! -name '*.log' -delete 2>/dev/null
mkdir -p gpurun_out/profiles_r06; cp profiles/r06_* gpurun_out/profiles_r06/ 2>/dev/null
fi
if has timeline; then
# kernel timeline of a rank of cfg4's 8-way split: single process (graph replay) against the sharded rank (graph replay +
# record packing + RCCL gather), a few steps from the middle of the timed region
( cd /tmp
  rocprofv3 --kernel-trace -d $OLDPWD/$O/tl_single -o tl -- python $OLDPWD/bench.py --config cfg4 --genes 25000 --no-cpu-baseline --no-k1-cold --sustain-seconds 0 --telemetry-ms 0 > /dev/null 2>&1
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 rocprofv3 --kernel-trace -d $OLDPWD/$O/tl_sharded -o tl -- python $OLDPWD/bench.py --exercise-exchange --config cfg4 --genes 25000 --scaling strong --no-cpu-baseline --no-k1-cold --sustain-seconds 0 --telemetry-ms 0 > /dev/null 2>&1 )
python tools/step_timeline.py $O/tl_single > $O/timeline_cfg4_rank_of_8.txt 2>&1
python tools/step_timeline.py $O/tl_sharded >> $O/timeline_cfg4_rank_of_8.txt 2>&1
rm -rf $O/tl_single $O/tl_sharded
cat $O/timeline_cfg4_rank_of_8.txt
fi
if has curve; then
python tools/strong_curve.py > $O/strong_curve.txt 2>&1
python tools/strong_curve.py --configs cfg3 --gene-order sorted > $O/strong_curve_cfg3_sorted.txt 2>&1
grep -v "^JSON" $O/strong_curve.txt $O/strong_curve_cfg3_sorted.txt
fi
if has fuzzextra; then
# one-off extra corpora of tests/golden/make_fuzz.py (other seeds, wider tables, the inputs the reference crashes on),
# generated in the build container into _ab/fuzz/ (git-ignored; they travel with the snapshot)
for c in _ab/fuzz/*.json.gz; do
  n=$(basename $c .json.gz)
  SCOARY_FUZZ_CORPUS=$PWD/$c timeout 900 python tools/fuzz_report.py $O/fuzz_$n.txt > $O/fuzz_$n.log 2>&1
  echo "$n: $(grep 'cases differ' $O/fuzz_$n.txt)"
  grep "^reference crashes" $O/fuzz_$n.txt | head -40
done
fi
if has longsoak; then
# new cases of the five randomised cross-checks (case numbers 600 ... : the generator is seeded by the case number)
for t in lists tiles listbuild seglists counts; do
  timeout ${SOAK_SECONDS:-240} python tools/stress_$t.py 1000000 ${SOAK_FIRST:-600} > $O/longsoak_$t.log 2>&1
  echo "$t rc=$? (124 = stopped by the clock) cases $(grep -c ' ok$' $O/longsoak_$t.log) ok, $(grep -c MISMATCH $O/longsoak_$t.log) mismatching" >> $O/longsoak.txt
done
cat $O/longsoak.txt
fi
if has soak; then
for t in lists tiles listbuild seglists counts; do
  timeout 200 python tools/stress_$t.py 600 > $O/stress_$t.log 2>&1
  echo "$t rc=$? $(tail -1 $O/stress_$t.log) ($(grep -c ' ok$' $O/stress_$t.log) ok)" >> $O/stress_soak.txt
done
cat $O/stress_soak.txt
fi
for f in $(ls $O/bench_*.json 2>/dev/null); do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    pr = d.get("per_rank") or []
    print("%-52s %.4e tests/s %9.4f ms/step | k3 %9.4f ms frac %s useful %s | graph %s | ranks %s | exposed %s"
          % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], r["kernel_ms"],
             r["frac"] and round(r["frac"], 3), r["useful_frac"] and round(r["useful_frac"], 3),
             d["config"].get("hip_graph"), d["rccl_ranks"], [round(x["exchange_exposed_ms"], 3) for x in pr][:8]))
    ss = d.get("scaling_strong")
    if ss:
        for k in ("cfg3", "cfg4"):
            print("    scaling_strong.%s: %.4e tests/s %.4f ms/step" % (k, ss[k]["value"], ss[k]["ms_per_step"]))
    if pr and "list_entries" in pr[0]:
        print("    per rank list entries", [x["list_entries"] for x in pr], "k3 ms", [round(x["kernel_ms"].get("k_permute_lists", 0), 4) for x in pr])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
