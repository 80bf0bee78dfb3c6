# usage (GPU box): [GLFER_FORM=h|w|x|"" (the routed form)] bash tools/variant_ab.sh "<workloads>" <variant> [<variant> ...]   (variant "product" = the in-tree library; GLFER_FORM unset = w)
WL=$1; shift
for W in $WL; do
  for V in "$@"; do
    if [ $V = product ]; then unset GLFER_LIB_PATH; else export GLFER_LIB_PATH=$PWD/tools/bin/variants/$V/libglfer_hip.so; fi
    GLFER_FORM=${GLFER_FORM-w} python3 bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %-14s %8.2f M frames/s  kernel %.3f ms  hbm %.3f  valu %.3f' % ('$W', '$V', d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['frac'], d['valu']['frac']))"
  done
done
