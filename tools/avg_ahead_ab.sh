# GPU box: update_avg_* and the average-and-map waterfall with two frames requested ahead (product) against one (tools/build_variant.sh ahead1 "-DGLFER_AVG_AHEAD=1" aux_kernels)
for V in product ahead1 product ahead1; do
  if [ $V = product ]; then unset GLFER_LIB_PATH; else export GLFER_LIB_PATH=$PWD/tools/bin/variants/$V/libglfer_hip.so; fi
  echo "== $V"; python3 tools/aux_sweep.py 2>/dev/null | grep "update_avg"; python3 tools/waterfall_time.py 2>/dev/null | grep "depth 4 levbuf yes"
done
