# usage (GPU box): bash tools/avg_pmc.sh     -- HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) of C2 + update_avg_plain depth 4 by form
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
D=gpurun_out/avg_pmc; rm -rf $D; mkdir -p $D
for F in fused fused_rows two; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D/${F}_$C -- python3 tools/avg_one.py $F > $D/${F}_$C.log 2>&1 || echo "$F $C failed" >> $D/failed.txt
  done
done
python3 - $D <<'PY' > $D/summary.txt
import csv, glob, sys, collections
d = sys.argv[1]
frames = 262144
print("# C2 (N 4096, 75 %) + update_avg_plain depth 4, 262 144 frames a call: HBM bytes per FRAME by kernel (rocprofv3 --pmc, separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md)")
for form in ("fused", "fused_rows", "two"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("%s/%s_%s/*/*_counter_collection.csv" % (d, form, c)):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if ("spectro16h" in k or "avg_fused" in k) and r["Counter_Name"] == c:
                    agg[k[:100]][c].append(float(r["Counter_Value"]))
    tot_r = tot_w = 0.0
    for k, v in agg.items():
        fr = 2 * 1024 * max(v["FETCH_SIZE"]) if v["FETCH_SIZE"] else 0.0      # the body launch (the largest dispatch of that kernel)
        wr = 1024 * max(v["WRITE_SIZE"]) if v["WRITE_SIZE"] else 0.0
        if fr + wr > frames * 100:
            print("%-12s %-100s read %8.0f B  written %8.0f B per frame" % (form, k, fr / frames, wr / frames))
            tot_r += fr; tot_w += wr
    print("%-12s total: read %.0f + written %.0f = %.0f B per frame   (by construction: hop 4096 in; averages 16392 + 32 out; PSD row 8196 out [+ in again])" % (form, tot_r / frames, tot_w / frames, (tot_r + tot_w) / frames))
PY
cat $D/summary.txt; cat $D/failed.txt 2>/dev/null
find $D -name '*_kernel_trace.csv' -delete; find $D -name '*agent_info.csv' -delete; find $D -name '*_counter_collection.csv' -delete
true
