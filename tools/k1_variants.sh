mkdir -p gpurun_out/r06
for i in 1 2; do
for v in cur qs4 qs2 qs4u2 qs4u1 qs2u2; do
  lib=""; [ $v != cur ] && lib="$PWD/_ab/$v.so"
  SCOARY_HIP_LIB=$lib python bench.py --no-cpu-baseline --sustain-seconds 0 --strong-extra off --steps 3 --warmup 1 --telemetry-ms 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['roofline_k1']['cold']
print('$v pass $i: copy %.0f | ' % c['measured_copy_peak_gbs'] + ' | '.join('T=%d %.1f us (%.3f)' % (r['traits'], r['cold_ms_median'] * 1e3, r['hbm_frac']) for r in c['runs']))"
done
done
