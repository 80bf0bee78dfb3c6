# usage (GPU box, repo root): bash tools/aux_pmc.sh <outdir under gpurun_out>
# Each rocprofv3 pass runs under `timeout -k 10 240` (a pass that aborts inside rocprofv3 must not hang the box
# until its silence limit), the program itself still directly after `--`.
# FETCH_SIZE / WRITE_SIZE / SQ passes (separate runs) over tools/aux_sweep.py: HBM bytes per row of the
# per-column kernels (floor / average / levels / map), per kernel the LARGEST launch (131072 rows).
R=$PWD; D=$R/gpurun_out/$1; rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp && cd $R
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 tools/aux_sweep.py > $D/fetch.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 tools/aux_sweep.py > $D/write.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $D/sq -- python3 tools/aux_sweep.py > $D/sq.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 tools/aux_sweep.py > $D/stats.log 2>&1
python3 - $D <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
rows = 131072
out = collections.defaultdict(dict)
for part in ("fetch", "write", "sq"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/" + part + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        for n, v in c.items():
            out[k][n] = max(v)
dur = collections.defaultdict(list)
for f in glob.glob(d + "/stats/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0][:60]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print("# per kernel: the largest launch of tools/aux_sweep.py (131072 rows of 2049 bins).  FETCH doubled (MI355X_MICROARCH.md), KiB -> bytes")
for k, c in sorted(out.items()):
    if "FETCH_SIZE" not in c: continue
    fb, wb = 2 * c["FETCH_SIZE"] * 1024, c.get("WRITE_SIZE", 0) * 1024
    print("%-62s read %7.0f B/row  write %7.0f B/row  longest launch %8.1f us" % (k, fb / rows, wb / rows, max(dur.get(k, [0])) / 1e3))
    print("      " + "  ".join("%s %.3g" % (n, v) for n, v in sorted(c.items()) if n.startswith("SQ_")))
PY
find $D -name '*_kernel_trace.csv' -size +2M -delete
