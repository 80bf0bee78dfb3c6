#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra -D flags>" [source-stems...]
# Builds tools/bin/variants/<name>/libglfer_hip.so: the product library with the listed kernel
# sources (default: spectro16w) recompiled with the extra flags.  For same-box A/B runs:
#   GLFER_LIB_PATH=tools/bin/variants/<name>/libglfer_hip.so python3 bench.py ...
# ONLY_LOGN=<n> in the environment: of the per-block-size sources only that size is recompiled (minutes saved per variant).
set -e
cd "$(dirname "$0")/../glfer_amd/csrc"
NAME=$1; FLAGS=$2; shift 2 || true
STEMS=${@:-spectro16w}
OUT=../../tools/bin/variants/$NAME
mkdir -p $OUT/obj
HIPCC=/opt/rocm/bin/hipcc
BASE="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-result -fno-slp-vectorize"
OBJS=""
for o in build/*.o; do
  b=$(basename $o .o); stem=${b%_n*}
  rebuilt=0
  for s in $STEMS; do
    logn=${b##*_n}
    if [ -n "$ONLY_LOGN" ] && [ "$logn" != "$b" ] && [ "$logn" != "$ONLY_LOGN" ]; then continue; fi   # ONLY_LOGN=12: the other block sizes keep the product's objects
    if [ "$stem" = "$s" ]; then
      if [ "$logn" != "$b" ]; then D="-DGLFER_LOGN=$logn"; else D=""; fi
      $HIPCC $BASE $D $FLAGS -c $s.hip -o $OUT/obj/$b.o &
      OBJS="$OBJS $OUT/obj/$b.o"; rebuilt=1
    fi
  done
  [ $rebuilt = 0 ] && OBJS="$OBJS $o"
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/libglfer_hip.so $OBJS -lpthread
echo "built $OUT/libglfer_hip.so  ($FLAGS)"
