# GPU box: forms of the average-and-map kernel (tools/build_variant.sh <name> "<flags>" aux_kernels), whole-waterfall rates
for V in "$@"; do
  if [ $V = product ]; then unset GLFER_LIB_PATH; else export GLFER_LIB_PATH=$PWD/tools/bin/variants/$V/libglfer_hip.so; fi
  echo "== $V"; python3 tools/waterfall_time.py 2>/dev/null | grep "levbuf yes" | grep -v none
done
