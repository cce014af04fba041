#!/bin/bash
cd $GRAFT_REPO_ROOT
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %.4e  step %.3f ms  kernel %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"; }
cp _ab/libC.so scoary_amd/csrc/libscoary_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
for v in B C; do
  cp _ab/lib$v.so scoary_amd/csrc/libscoary_hip.so
  echo "== $v cfg3"; timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | one
  echo "== $v n5k"; timeout 200 python bench.py --kernel lists --no-cpu-baseline --steps 10 --warmup 3 --genes 25000 --isolates 5000 --traits 1 --permutations 10000 2>/dev/null | one
  echo "== $v n10k"; timeout 200 python bench.py --kernel lists --no-cpu-baseline --steps 10 --warmup 3 --genes 20000 --isolates 10000 --traits 2 --permutations 10000 2>/dev/null | one
done
done
