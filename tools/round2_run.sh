#!/bin/bash
# One GPU-box session collecting the round-2 evidence (outputs under gpurun_out/).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out
python bench.py > $O/bench_r02h.json 2> $O/bench_r02h.err
tools/profile.sh r02h_cfg3 > $O/profile_r02h_cfg3.log 2>&1
python bench.py --config cfg2 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_r02h_cfg2.json 2>/dev/null
python bench.py --config cfg2 --steps 200 --warmup 20 --no-cpu-baseline --graph > $O/bench_r02h_cfg2_graph.json 2>/dev/null
python bench.py --config cfg4 --no-cpu-baseline > $O/bench_r02h_cfg4.json 2>/dev/null
python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_r02h_cfg5_shard.json 2> $O/bench_r02h_cfg5.err
for sc in weak strong; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --exercise-exchange --no-cpu-baseline --steps 5 --warmup 2 --scaling $sc > $O/bench_r02h_exchange_$sc.json 2> $O/bench_r02h_exchange_$sc.err
done
python bench.py --gpus 2 > $O/bench_r02h_gpus2.out 2>&1; echo "exit code of --gpus 2 on a 1-GPU box: $?" >> $O/bench_r02h_gpus2.out
bash tools/check_dist_cli.sh > $O/check_dist_cli.txt 2>&1
tools/clock_sample.sh > $O/clock_power_r02.txt 2>&1
for f in $O/bench_r02h*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s | %.4f ms/step (median %.4f) value %.4e setup %.2f ms | k3 %.4f ms frac %s ops/test %s | rccl_ranks %s" % (
        d["config"]["workload"][:40], d["ms_per_step"], d["ms_per_step_median"], d["value"], d["setup_ms"],
        r["kernel_ms"], r["frac"], r["ops_per_test"], d["rccl_ranks"]))
    for k in ("cpu_baseline", "cpu_baseline_port"):
        if k in d: print("  ", k, d[k].get("value"), d[k].get("cores"), d[k].get("matches_gpu"))
except Exception as e:
    print("unreadable:", e)
PY
done
tail -3 $O/bench_r02h_gpus2.out; tail -2 $O/check_dist_cli.txt; cat $O/clock_power_r02.txt
