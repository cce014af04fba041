#!/bin/bash
# GPU-box sessions collecting the round-5 evidence (outputs under gpurun_out/r05/; counter summaries are written
# straight to profiles/r05_<cfg>_* by tools/profile_configs.sh on the box and come back under gpurun_out/).
#   tools/round5_run.sh [part ...]     parts: configs cfg5 extras soak   (default: all)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05
mkdir -p $O gpurun_out/profiles_r05
PARTS="${*:-configs cfg5 extras soak}"
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
if has configs; then
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_cfg3_driver_command.json 2> $O/bench_cfg3_driver_command.err
bash tools/profile_configs.sh r05 cfg3 cfg4 cfg2 > $O/profile_configs.log 2>&1
python bench.py --config cfg2 --no-graph > $O/bench_cfg2_eager.json 2>/dev/null
python bench.py --config cfg4 --genes 25000 --no-cpu-baseline > $O/bench_cfg4_rank_of_8.json 2>/dev/null
fi
if has cfg5; then
bash tools/profile_configs.sh r05 cfg5 > $O/profile_cfg5.log 2>&1
fi
if has extras; then
bash tools/profile.sh r05_wide --genes 20000 --isolates 50000 --traits 2 --permutations 1024 > $O/profile_wide.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_r05_wide profiles/r05_wide50000 wide50000 >> $O/profile_wide.log 2>&1
find gpurun_out/prof_r05_wide -type f ! -name '*.db' ! -name '*.txt' ! -name '*.log' -delete
python bench.py --genes 20000 --isolates 50000 --traits 2 --permutations 1024 --no-cpu-baseline > $O/bench_wide_50000_lists.json 2>/dev/null
python bench.py --gene-kind balanced --no-cpu-baseline > $O/bench_cfg3_balanced.json 2>/dev/null
python bench.py --gene-kind ushaped --no-cpu-baseline > $O/bench_cfg3_ushaped.json 2>/dev/null
for sc in weak strong; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --exercise-exchange --verify-gather --no-cpu-baseline --steps 10 --warmup 2 --scaling $sc \
    > $O/bench_exchange_${sc}_1rank.json 2> $O/bench_exchange_$sc.err
done
for c in "2000 10 10000" "5000 1 10000" "10000 50 4096" "50000 2 1024"; do python tools/gen_time.py $c; done 2>&1 | grep "^N=" > $O/generator_alone.txt
export TMPDIR=/tmp
python tools/e2e_vcf.py > $O/e2e_cli_cfg4_vcf.txt 2>&1
python tools/e2e_synth.py --genes 50000 --isolates 2000 --traits 10 --permute 10000 > $O/e2e_cli_cfg3.txt 2>&1
fi
if has soak; then
(time python -m pytest tests/ -q -m gpu --durations=8) > $O/pytest_gpu.log 2>&1
for t in lists tiles listbuild seglists counts; do
  timeout 330 python tools/stress_$t.py 1000 > $O/stress_$t.log 2>&1
  echo "$t rc=$? $(tail -1 $O/stress_$t.log) ($(grep -c ' ok$' $O/stress_$t.log) ok)" >> $O/stress_soak.txt
done
cat $O/stress_soak.txt; tail -4 $O/pytest_gpu.log
fi
cp profiles/r05_* gpurun_out/profiles_r05/ 2>/dev/null
for f in $O/bench_*.json gpurun_out/bench_r05_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-46s %.4e tests/s %9.4f ms/step | k3 %9.4f ms frac %s useful %s | gen alone %s | ranks %s"
          % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], r["kernel_ms"],
             r["frac"] and round(r["frac"], 3), r["useful_frac"] and round(r["useful_frac"], 3),
             d.get("kernel_ms_isolated", {}).get("k_perm_generate_tiles"), d["rccl_ranks"]))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
