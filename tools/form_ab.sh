# usage (GPU box): bash tools/form_ab.sh  -- same-box A/B of the kernel forms on the bench workloads
for W in fft mtm16k mtm; do
  for F in h w x; do
    echo "== workload $W  GLFER_FORM=$F"
    GLFER_FORM=$F python3 bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   %.2f M frames/s  kernel %.3f ms  hbm frac %.3f  valu frac %.3f' % (d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['frac'], d['valu']['frac']))"
  done
done
