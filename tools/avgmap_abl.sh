# GPU box: timing ablations of the average-and-map kernel (tools/build_variant.sh amabl<bits> "-DGLFER_AVGMAP_ABL=<bits>" aux_kernels)
for V in product amabl1 amabl2 amabl3 amabl4 amabl8 amabl15; do
  if [ $V = product ]; then unset GLFER_LIB_PATH; else export GLFER_LIB_PATH=$PWD/tools/bin/variants/$V/libglfer_hip.so; fi
  echo "== $V"; python3 tools/waterfall_time.py 2>/dev/null | grep "plain      depth 4 levbuf yes\|sumavg     depth 4 levbuf yes"
done
