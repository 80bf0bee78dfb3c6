#!/bin/bash
# rocprofv3 kernel durations of the piecewise mean pass (tools/piece_probe.py): off, one piece, 64 and 128 MB pieces on one stream
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/piece_probe
mkdir -p $O
cd $R
for cfg in "C1 0" "C1 1 0" "C1 1 64 1" "C1 1 128 1" "C3 0" "C3 1 0" "C3 1 128 1"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/$tag -o p -- python3 tools/piece_probe.py $cfg > $O/$tag.log 2>&1 || exit 1
  f=$(find $O/$tag -name "*kernel_stats.csv" | head -1)
  echo "== $cfg"; head -6 "$f" | cut -c1-220
done
