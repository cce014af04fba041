#!/bin/bash
# GPU-box sessions collecting the round-4 evidence (outputs under gpurun_out/r04/; the summaries are
# installed under profiles/ afterwards: tools/rocpd_summary.py on the merged gpurun_out/prof_r04_<cfg>).
#   tools/round4_run.sh [part ...]     parts: configs cfg5 wide exchange soak (default: all, in that order);
# one part per gpurun call keeps a lost box cheap.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04
mkdir -p $O
PARTS="${*:-configs cfg5 wide exchange soak}"
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
if has configs; then
# the driver's own command, first thing on the fresh box
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_cfg3_driver_command.json 2> $O/bench_cfg3_driver_command.err
# counter profiles + bench lines (with the CPU baselines) of the single-GPU BASELINE configs
bash tools/profile_configs.sh r04 cfg3 cfg4 cfg2 > $O/profile_configs.log 2>&1
python bench.py --config cfg2 --no-graph > $O/bench_cfg2_eager.json 2>/dev/null
fi
if has cfg5; then
bash tools/profile_configs.sh r04 cfg5 > $O/profile_cfg5.log 2>&1
fi
if has wide; then
# the wide (segmented) shape of VERDICT r3 item 3
bash tools/profile.sh r04_wide --genes 20000 --isolates 50000 --traits 2 --permutations 1024 > $O/profile_wide.log 2>&1
find gpurun_out/prof_r04_wide -type f ! -name '*.db' ! -name '*.txt' ! -name '*.log' -delete
python bench.py --genes 20000 --isolates 50000 --traits 2 --permutations 1024 --no-cpu-baseline > $O/bench_wide_50000_lists.json 2>/dev/null
python bench.py --genes 20000 --isolates 50000 --traits 2 --permutations 1024 --no-cpu-baseline --kernel dense > $O/bench_wide_50000_dense.json 2>/dev/null
python tools/sweep_isolates.py --permutations 8192 --traits 4 > $O/sweep_isolates.txt 2>&1
# a pan-genome-shaped evidence line next to cfg3
python bench.py --gene-kind ushaped --no-cpu-baseline > $O/bench_cfg3_ushaped.json 2>/dev/null
python bench.py --config cfg4 --gene-kind ushaped --no-cpu-baseline > $O/bench_cfg4_shape_ushaped.json 2>/dev/null
fi
if has exchange; then
# one rank through RCCL, weak and strong
for sc in weak strong; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --exercise-exchange --verify-gather --no-cpu-baseline --steps 10 --warmup 2 --scaling $sc \
    > $O/bench_exchange_${sc}_1rank.json 2> $O/bench_exchange_$sc.err
done
# sustained clock / power (rocm-smi next to 3000 back-to-back steps; the line itself carries amdsmi's samples)
tools/clock_sample.sh > $O/clock_power.txt 2>&1
python bench.py --steps 3000 --warmup 5 --no-cpu-baseline > $O/bench_cfg3_sustained_3000_steps.json 2>/dev/null
# the command line end to end on a cfg3-sized table
python tools/e2e_synth.py --genes 50000 --isolates 2000 --traits 10 --permute 10000 > $O/e2e_cli_cfg3.txt 2>&1
fi
if has soak; then
# the GPU test suite and the stress soaks
(time python -m pytest tests/ -q -m gpu --durations=8) > $O/pytest_gpu.log 2>&1
for t in lists tiles listbuild seglists counts; do
  timeout 330 python tools/stress_$t.py 1000 > $O/stress_$t.log 2>&1
  echo "$t rc=$? $(tail -1 $O/stress_$t.log) ($(grep -c ' ok$' $O/stress_$t.log) ok)" >> $O/stress_soak.txt
done
cat $O/stress_soak.txt; tail -4 $O/pytest_gpu.log
fi
for f in $O/bench_*.json gpurun_out/bench_r04_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r, t = d["roofline"], d.get("telemetry") or {}
    print("%-52s %.4e tests/s %9.4f ms/step (median %9.4f) setup %6.2f | k3 %9.4f ms frac %s useful %s opc_frac %s | sclk %s W %s n %s | ranks %s"
          % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["ms_per_step_median"], d["setup_ms"], r["kernel_ms"],
             r["frac"] and round(r["frac"], 3), r["useful_frac"] and round(r["useful_frac"], 3),
             r.get("ops_per_clock_frac") and round(r["ops_per_clock_frac"], 3), t.get("sclk_mhz_mean") and round(t["sclk_mhz_mean"]),
             t.get("socket_power_w_mean") and round(t["socket_power_w_mean"]), t.get("samples"), d["rccl_ranks"]))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
