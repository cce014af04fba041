#!/bin/bash
# GPU-box sessions collecting the round-6 evidence (outputs under gpurun_out/r06/).
#   tools/round6_run.sh [part ...]     parts: tests bench balance rank8 e2e
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
if has balance; then
python tools/shard_balance.py --config cfg3 --gene-order sorted > $O/shard_balance_cfg3_sorted.txt 2>&1
python tools/shard_balance.py --config cfg3 --gene-order config > $O/shard_balance_cfg3_config_order.txt 2>&1
python tools/shard_balance.py --exampledata tests/golden/exampledata/Gene_presence_absence.csv > $O/shard_balance_exampledata.txt 2>&1
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
for f in $O/bench_*.json; do python - "$f" <<'PY'
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
