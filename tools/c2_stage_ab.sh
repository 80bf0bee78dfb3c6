# usage (GPU box): bash tools/c2_stage_ab.sh  -- C2 with the PSD rows staged through LDS and stored 16 bytes per lane from
# Each rocprofv3 pass runs under `timeout -k 10 240` (a pass that aborts inside rocprofv3 must not hang the box
# until its silence limit), the program itself still directly after `--`.
# 64-byte boundaries (variant h_stage: -DGLFER16H_STAGE_ROWS=1) against the product: rate, and WRITE_SIZE of both
R=$PWD; cd /tmp && export TMPDIR=/tmp && cd $R
GLFER_FORM=h bash tools/variant_ab.sh "fft" product h_stage product h_stage
for V in product h_stage; do
  if [ $V = product ]; then unset GLFER_LIB_PATH; else export GLFER_LIB_PATH=$R/tools/bin/variants/$V/libglfer_hip.so; fi
  D=gpurun_out/c2_stage_$V; rm -rf $D; mkdir -p $D
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --workload fft > $D.log 2>&1
  python3 - $D $V <<'PY'
import csv, glob, sys
v = [float(r["Counter_Value"]) for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv") for r in csv.DictReader(open(f))
     if "spectro16h" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
w = sum(v) / len(v) * 1024
print("%-8s WRITE_SIZE %.4f GB per launch = %.4f x the row bytes (1048576 rows of 8196 B)" % (sys.argv[2], w / 1e9, w / (1048576 * 8196)))
PY
  rm -rf $D
done
