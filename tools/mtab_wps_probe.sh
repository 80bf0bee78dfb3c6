#!/bin/bash
# Does the hop-means launch of the next piece hide beside the periodogram's table form when that runs at TWO wavefronts per SIMD
# (GLFER_MTAB_WPS=2: registers and LDS left for a means block on every CU)?  C1 and C2, reference order, M frames/s.
for c in C1 C2; do
  echo "$c  no mean removal: $(python3 tools/one_rate.py $c 0)   in-kernel sums: $(python3 tools/one_rate.py $c 2)   reference order (default): $(python3 tools/one_rate.py $c 1)"
  for wps in 3 2; do
    for st in 1 2 3; do
      for piece in 0 256 512 1024; do
        [ $piece = 0 ] && [ $st != 1 ] && continue
        r=$(GLFER_MTAB_WPS=$wps GLFER_EXACT_STREAMS=$st GLFER_EXACT_PIECE_MB=$piece python3 tools/one_rate.py $c 1)
        echo "   table form at $wps wavefronts/SIMD, streams $st, piece $piece MB: $r"
      done
    done
  done
done
