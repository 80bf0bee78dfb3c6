#!/bin/bash
# Same-box A/B of the periodogram kernel's row-offset form (spectro16h.hip, GLFER16H_OPAQUE_ROW): product against
# tools/bin/variants/h_visible_row (built by tools/build_variant.sh h_visible_row "-DGLFER16H_OPAQUE_ROW=0" spectro16h), alternating.
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2 3; do
  GLFER_FORM= bash tools/variant_ab.sh "fft fft1k" product h_visible_row || exit 1
done
for V in product h_visible_row; do
  if [ $V = product ]; then unset GLFER_LIB_PATH; else export GLFER_LIB_PATH=$PWD/tools/bin/variants/$V/libglfer_hip.so; fi
  echo "== $V"; timeout -k 10 200 python3 tools/exact_mean_time.py table 2>/dev/null | grep "C1\|C2 \|glfer default: N=1024"
done
