# usage (GPU box): bash tools/stall_cmd.sh <tag> <script.py> [args...]
# The counter groups of tools/stall_pass.sh over an arbitrary python program (the program itself directly after `--`, every pass
# under `timeout -k`); per kernel name the mean of every counter over its dispatches.  Output: gpurun_out/stall_<tag>/summary.txt
TAG=$1; shift; R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
D=gpurun_out/stall_$TAG; rm -rf $D; mkdir -p $D
i=0
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $D/g$i -- python3 "$@" > $D/g$i.log 2>&1 || echo "group $i failed: $G" >> $D/failed.txt
done
python3 - $D <<'PY' > $D/summary.txt
import csv, glob, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/g*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "spectro16" in k or "avg_" in k or "lmp" in k or "hop_means" in k:
            agg[k[:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k)
    for n, v in sorted(c.items()):
        print("   %-28s %18.0f   (%d dispatches)" % (n, sum(v) / len(v), len(v)))
PY
cat $D/summary.txt; cat $D/failed.txt 2>/dev/null
find $D -name '*_kernel_trace.csv' -delete
find $D -name '*_counter_collection.csv' -delete
find $D -name '*agent_info.csv' -delete
true
