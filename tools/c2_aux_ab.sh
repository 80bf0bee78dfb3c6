# usage (GPU box): bash tools/c2_aux_ab.sh <variant> ...  -- rate and WRITE_SIZE of C2 per store-policy variant
# Each rocprofv3 pass runs under `timeout -k 10 240` (a pass that aborts inside rocprofv3 must not hang the box
# until its silence limit), the program itself still directly after `--`.
R=$PWD; cd /tmp && export TMPDIR=/tmp && cd $R
GLFER_FORM=h bash tools/variant_ab.sh "fft" "$@"
for V in "$@"; do
  if [ $V = product ]; then unset GLFER_LIB_PATH; else export GLFER_LIB_PATH=$R/tools/bin/variants/$V/libglfer_hip.so; fi
  D=gpurun_out/c2_aux_$V; rm -rf $D; mkdir -p $D
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --workload fft > $D.log 2>&1
  python3 - $D $V <<'PY'
import csv, glob, sys
v = [float(r["Counter_Value"]) for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv") for r in csv.DictReader(open(f))
     if "spectro16h" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
w = sum(v) / len(v) * 1024
print("%-8s WRITE_SIZE %.4f GB per launch = %.4f x the row bytes" % (sys.argv[2], w / 1e9, w / (1048576 * 8196)))
PY
  rm -rf $D $D.log
done
