# usage (GPU box): bash tools/pmc_pass.sh <workload> <tag>   -- SQ counter passes of one bench workload (GLFER_FORM / GLFER_LIB_PATH from the environment)
# Each rocprofv3 pass runs under `timeout -k 10 240` (a pass that aborts inside rocprofv3 must not hang the box
# until its silence limit), the program itself still directly after `--`.
W=$1; TAG=$2; R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
D=gpurun_out/pmc_$TAG; rm -rf $D; mkdir -p $D
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $D/a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --workload $W > $D/a.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $D/b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --workload $W > $D/b.log 2>&1
python3 - $D <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
for part in ("a", "b"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/" + part + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "spectro16" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        print(k)
        for n, v in sorted(c.items()):
            print("   %-24s %16.0f" % (n, sum(v) / len(v)))
PY
