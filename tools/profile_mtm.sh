set -e
# Each rocprofv3 pass runs under `timeout -k 10 240` (a pass that aborts inside rocprofv3 must not hang the box
# until its silence limit), the program itself still directly after `--`.
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
D=$R/gpurun_out/prof; mkdir -p $D; cd $R
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $D/stats.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $D/fetch.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $D/write.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $D/sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $D/sq.log 2>&1
python3 bench.py --no-cpu-baseline | cut -c1-200
