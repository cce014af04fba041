# Dense permutation kernel template variants side by side (SCOARY_PERMUTE_VARIANT, experiments only).
mkdir -p gpurun_out
for v in reg c8x8 c8x16 c4x16 c8x4; do
  echo "== cfg4 N=5000 G=50000 variant $v"
  SCOARY_PERMUTE_VARIANT=$v timeout 300 python bench.py --config cfg4 --genes 50000 --permutations 4000 --kernel dense --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['kernel_ms'])"
done
for v in reg c8x8 c8x16; do
  echo "== cfg3 variant $v"
  SCOARY_PERMUTE_VARIANT=$v timeout 300 python bench.py --config cfg3 --kernel dense --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['kernel_ms'])"
done
