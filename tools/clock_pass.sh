# usage (GPU box): [GLFER_LIB_PATH=...] bash tools/clock_pass.sh <workload> <tag>
# Each rocprofv3 pass runs under `timeout -k 10 240` (a pass that aborts inside rocprofv3 must not hang the box
# until its silence limit), the program itself still directly after `--`.
# shader clock of the workload's estimator kernel: GRBM_GUI_ACTIVE / 8 over the kernel's own duration (one rocprofv3 --pmc pass)
W=$1; TAG=$2; R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
D=gpurun_out/clock_$TAG; rm -rf $D; mkdir -p $D
timeout -k 10 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU --output-format csv -d $D/a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --workload $W > $D/a.log 2>&1
python3 - $D $TAG <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
dur = {}
for f in glob.glob(d + "/a/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "spectro16" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:60])
cnt = collections.defaultdict(dict)
for f in glob.glob(d + "/a/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Dispatch_Id"] in dur:
            cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
for k, (ns, name) in sorted(dur.items(), key=lambda kv: int(kv[0])):
    c = cnt.get(k, {})
    if "GRBM_GUI_ACTIVE" in c:
        print("%-10s %s  %.3f ms  clock %.3f GHz  valu instr %.0f" % (tag, name, ns / 1e6, c["GRBM_GUI_ACTIVE"] / 8 / ns, c.get("SQ_INSTS_VALU", 0)))
PY
rm -rf $D/a
