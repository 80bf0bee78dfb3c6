# usage (GPU box): bash tools/floor_ab.sh <variant> ...   -- compute_floor / update_avg / display rates per library variant
for V in "$@"; do
  if [ $V = product ]; then unset GLFER_LIB_PATH; else export GLFER_LIB_PATH=$PWD/tools/bin/variants/$V/libglfer_hip.so; fi
  echo "== $V"
  python3 tools/aux_sweep.py 2>/dev/null | grep -v "stage by stage"
  python3 tools/floor_sizes.py 2>/dev/null
done
