#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (GPU ASAN is not available on this pool):
#  1. the oracle, through the whole oracle test module;
#  2. the product's host table builders (windows, DPSS eigen-solve, twiddles, palettes, the display's dB-step table).
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
cp oracle/liboracle.so "$tmp/liboracle.so"
trap 'cp "$tmp/liboracle.so" oracle/liboracle.so; rm -rf "$tmp"' EXIT
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -shared -fPIC oracle/glfer_oracle.c -o oracle/liboracle.so -lm
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_oracle_pinning.py -x -q
cat > "$tmp/ht_main.cpp" <<'CPP'
#include "host_tables.h"
#include <cstdio>
#include <vector>
int main() {
  for (int n : {256, 1024, 4096}) for (int w = 0; w < 8; w++) { std::vector<float> v(n); glfer::make_window(w, n, v.data()); }
  for (int n : {256, 4096}) { std::vector<double> t((size_t)9 * n), s(9); if (!glfer::make_dpss(n, 8, 4.5, t.data(), s.data())) return 1; }
  for (int l = 8; l <= 14; l++) { int slots = glfer::make_twiddles16(l, nullptr); std::vector<float> tw((size_t)2 * slots * ((1 << l) / 16)); glfer::make_twiddles16(l, tw.data()); }
  unsigned char tab[768];
  for (int p = -1; p < 10; p++) glfer::make_palette(p, tab);
  const double *thr = glfer::log_thresholds();       // the whole-dB steps of the display mapping: monotonic, and exact at both ends
  for (int k = -glfer::kLogThrK + 1; k <= glfer::kLogThrK; k++) if (!(thr[glfer::kLogThrK + k] > thr[glfer::kLogThrK + k - 1])) return 2;
  puts("host tables: clean");
  return 0;
}
CPP
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -Iglfer_amd/csrc "$tmp/ht_main.cpp" glfer_amd/csrc/host_tables.cpp -o "$tmp/ht_asan"
"$tmp/ht_asan"
