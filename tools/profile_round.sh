# usage (on the GPU box, from the repo root): bash tools/profile_round.sh [workload ...]
# Each rocprofv3 pass runs under `timeout -k 10 240` (a pass that aborts inside rocprofv3 must not hang the box
# until its silence limit), the program itself still directly after `--`.
# For each bench workload (default: mtm fft mtm16k): one rocprofv3 --kernel-trace --stats pass and three
# separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ_*), the program itself right after `--`.
# tools/summarize_prof.py condenses the trees into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
WL="$@"
[ -z "$WL" ] && WL="mtm fft mtm16k"
for W in $WL; do
  if [ $W = mtm ]; then D=$R/gpurun_out/prof; else D=$R/gpurun_out/prof_$W; fi
  rm -rf $D; mkdir -p $D
  cd $R
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --workload $W > $D/stats.log 2>&1
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --workload $W > $D/fetch.log 2>&1
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --workload $W > $D/write.log 2>&1
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $D/sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --workload $W > $D/sq.log 2>&1
  # keep only the small CSVs (the trees also hold agent info and per-dispatch traces)
  find $D -name '*_kernel_trace.csv' -size +256k -delete
  find $D -name '*agent_info.csv' -delete
  echo "$W passes done"
done
# the per-column stages (floor / average / display): kernel stats of tools/aux_sweep.py
if [ -z "$SKIP_AUX" ]; then
  D=$R/gpurun_out/prof_aux; rm -rf $D; mkdir -p $D; cd $R
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 tools/aux_sweep.py > $D/stats.log 2>&1
  find $D -name '*_kernel_trace.csv' -size +2M -delete
  echo "aux pass done"
fi
