#!/bin/bash
# Frames in flight per CU against throughput for the HP-ARMA kernel (C5): the launcher is asked for more LDS than a frame needs
# (22 KB: 7 per CU), so fewer wavefronts share a CU.  Linear in the count = latency-bound; flat = issue-bound.
cd "${GRAFT_REPO_ROOT:-.}"
for kb in 0 26 32 40 53; do
  echo "GLFER_HPARMA_LDS_KB=$kb"
  GLFER_HPARMA_LDS_KB=$kb timeout -k 10 120 python bench.py --workload hparma --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  %.3f M frames/s  %.2f ms' % (d['value']/1e6, d['ms_per_step']))" || exit 1
done
