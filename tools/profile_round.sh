set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for W in mtm fft; do
  if [ $W = mtm ]; then D=$R/gpurun_out/prof; WF=""; else D=$R/gpurun_out/prof_fft; WF="--workload fft"; fi
  mkdir -p $D
  cd $R
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline $WF > $D/stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $WF > $D/fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $WF > $D/write.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $D/sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $WF > $D/sq.log 2>&1
  echo "$W passes done"
done
cd $R
python3 bench.py > gpurun_out/bench_mtm.json 2> gpurun_out/bench_mtm.err
python3 bench.py --workload fft > gpurun_out/bench_fft.json 2> gpurun_out/bench_fft.err
python3 bench.py --workload hparma > gpurun_out/bench_hparma.json 2> gpurun_out/bench_hparma.err
cut -c1-200 gpurun_out/bench_mtm.json; cut -c1-200 gpurun_out/bench_fft.json
