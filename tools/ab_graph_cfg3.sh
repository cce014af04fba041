for i in 1 2 3; do
  for v in eager graph; do
    f=""; [ $v = graph ] && f="--graph"
    python bench.py --no-cpu-baseline --no-k1-cold --sustain-seconds 0 --strong-extra off --steps 40 --warmup 10 $f 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v pass $i: %.4f ms/step  %.4e tests/s  k3 %.4f ms  graph %s' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['config'].get('hip_graph')))"
  done
done
