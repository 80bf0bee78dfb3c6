#!/usr/bin/env python3
"""Condense a gpurun_out/prof[_<workload>]/ tree (rocprofv3 --kernel-trace --stats and --pmc
passes over bench.py) into the small summaries kept under profiles/.
usage: summarize_prof.py <round-tag> [mtm|fft|mtm16k|fft1k|hparma]"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
workload = sys.argv[2] if len(sys.argv) > 2 else "mtm"
suffix = "" if workload == "mtm" else "_" + workload
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "gpurun_out", "prof" + suffix)
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)


def one(pattern):
    # the newest file: gpurun merges a call's files into what earlier calls left under gpurun_out/
    return max(glob.glob(os.path.join(prof, pattern)), key=os.path.getmtime)


rows = list(csv.DictReader(open(one("stats/*/*_kernel_stats.csv"))))
with open(os.path.join(out, tag + "_kernel_stats" + suffix + ".csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                    r["MaxNs"], r["StdDev"]])
k = max((r for r in rows if "spectro16" in r["Name"] or "hparma_kernel" in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))
KNAME = k["Name"]


def pmc(path, name):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
         if r["Kernel_Name"] == KNAME and r["Counter_Name"] == name]
    return sum(v) / len(v)


if workload == "hparma":          # not an HBM-bound row: the kernel stats are the summary
    print(k)
    sys.exit(0)
fetch = pmc(one("fetch/*/*_counter_collection.csv"), "FETCH_SIZE")
write = pmc(one("write/*/*_counter_collection.csv"), "WRITE_SIZE")
sq = collections.defaultdict(list)
for r in csv.DictReader(open(one("sq/*/*_counter_collection.csv"))):
    if r["Kernel_Name"] == KNAME:
        sq[r["Counter_Name"]].append(float(r["Counter_Value"]))
sq = {n: sum(v) / len(v) for n, v in sq.items()}
frames, hop, bins, wname = {
    "mtm": (262144, 4096, 2049, "C3 MTM N=4096 NW=2.5 mtm_k=4 overlap 0, %d frames per launch"),
    "fft": (1048576, 1024, 2049, "C2 periodogram Hanning N=4096 overlap 75 %%, %d frames per launch"),
    "mtm16k": (65536, 16384, 8193, "C4 MTM N=16384 NW=4.5 mtm_k=8 (9 tapers) overlap 0, %d frames per launch"),
    "fft1k": (2097152, 512, 513, "C1 periodogram Hanning N=1024 overlap 50 %%, %d frames per launch"),
}[workload]
alg = frames * (4 * hop + 4 * bins)
# MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports exactly half
# the bytes of a coalesced streaming read -> doubled here; WRITE_SIZE reads the bytes exactly.
traffic = (2 * fetch + write) * 1024
summary = {
    "command": "rocprofv3 --kernel-trace {--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_*} "
               "-- python3 bench.py --steps K --warmup 1 --no-cpu-baseline%s   (separate passes)"
               % (" --workload " + workload),
    "workload": wname % frames,
    "kernel": k["Name"], "kernel_avg_ns_profiled": float(k["AverageNs"]), "kernel_calls": int(k["Calls"]),
    "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
    "hbm_traffic_bytes_per_launch_corrected": traffic,
    "hbm_traffic_bytes_per_frame_corrected": traffic / frames,
    "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_per_frame": alg / frames,
    "traffic_over_algorithmic": traffic / alg,
    "sq_counters_per_launch": sq, "frames_per_launch": frames,
}
json.dump(summary, open(os.path.join(out, tag + "_hbm_traffic" + suffix + ".json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
