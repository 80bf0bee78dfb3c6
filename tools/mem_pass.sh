# usage (GPU box): [GLFER_LIB_PATH=...] bash tools/mem_pass.sh <workload> <tag>
# Vector-memory path counters (TA / TCP / UTCL1 / TCC) of one bench workload's estimator kernel, in small groups, one
# rocprofv3 --pmc pass each (the program itself right after `--`).  Output: gpurun_out/mem_<tag>.txt
# (A group with the TCP_UTCL1_* counters, TCP_PENDING_STALL_CYCLES and TCP_TCP_LATENCY aborted inside rocprofv3 (signal 6) and left
# the run hanging until the box's silence limit: round 3, 17 GPU-minutes.  Cause, as far as the kept records go: all of them ARE
# listed for gfx950 by `rocprofv3 --list-avail` (gpurun_out/counters_avail.txt:3107-3517), so "unsupported" is not it; that group
# asked for SIX counters of the TCP block in one pass where every group below asks for at most four of one block -- more than the
# block has counter registers, and this rocprofv3 aborts instead of splitting the request.  Not reproduced (the pass's own logs were
# removed by this script's clean-up, and an abort that hangs a box is not worth a second try): the rule kept here is <= 4 counters
# of one hardware block per pass, every pass under `timeout`, and the per-pass logs stay if a pass fails.)
W=$1; TAG=$2; R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
D=gpurun_out/mem_$TAG; rm -rf $D; mkdir -p $D
i=0
for G in "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TCP_TOTAL_ACCESSES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUSY_sum" \
         "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $D/g$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --workload $W > $D/g$i.log 2>&1 || echo "group $i failed: $G" >> $D/failed.txt
done
python3 - $D <<'PY' > gpurun_out/mem_$TAG.txt
import csv, glob, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/g*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if ("spectro16y" in k or "spectro16h" in k or "spectro16w" in k):
            agg[k[:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k)
    for n, v in sorted(c.items()):
        print("   %-40s %18.0f   (%d dispatches)" % (n, sum(v) / len(v), len(v)))
PY
cat gpurun_out/mem_$TAG.txt; cat $D/failed.txt 2>/dev/null
[ -f $D/failed.txt ] || rm -rf $D        # a failed pass keeps its logs: they are the evidence
