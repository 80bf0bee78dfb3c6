# usage (GPU box): [GLFER_LIB_PATH=...] bash tools/mem_pass.sh <workload> <tag>
# Vector-memory path counters (TA / TCP / UTCL1 / TCC) of one bench workload's estimator kernel, in small groups, one
# rocprofv3 --pmc pass each (the program itself right after `--`).  Output: gpurun_out/mem_<tag>.txt
# (Round 3: a group made of the TCP_UTCL1_* counters, TCP_PENDING_STALL_CYCLES and TCP_TCP_LATENCY aborted inside rocprofv3 (signal 6)
# and left the run hanging until the box's silence limit: 17 GPU-minutes.  The CAUSE IS NOT ESTABLISHED by the kept records: all of
# those counters are listed for gfx950 by `rocprofv3 --list-avail` (gpurun_out/counters_avail.txt), so "unsupported" is not it, and a
# per-block register limit is not it either -- groups 3 and 4 below ask for six and five TCC_* counters in one pass and run (an
# earlier version of this comment claimed "<= 4 counters of one block per pass"; the script's own groups contradict that).  What
# the aborting group had that no group below has is those seven counter NAMES; the pass's logs were removed by the clean-up this
# script then had, so nothing more can be said.  What is kept: none of those counters is requested here, every pass runs under
# `timeout -k`, the program comes directly after `--`, and a failed pass keeps its logs.  The aborting group has not been run again:
# an abort that hangs a box costs more than those counters are worth.)
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
