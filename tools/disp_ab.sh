# usage (GPU box): bash tools/disp_ab.sh <variant> ...   -- display rates per library variant (tools/aux_sweep.py lines)
for V in "$@"; do
  if [ $V = product ]; then unset GLFER_LIB_PATH; else export GLFER_LIB_PATH=$PWD/tools/bin/variants/$V/libglfer_hip.so; fi
  echo "== $V"
  python3 tools/aux_sweep.py 2>/dev/null | grep "display\|stage by stage"
done
