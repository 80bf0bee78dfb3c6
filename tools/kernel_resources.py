#!/usr/bin/env python3
"""Resource usage of the built kernels from the -Rpass-analysis=kernel-resource-usage logs under
glfer_amd/csrc/build/ (usage: kernel_resources.py [name-filter])."""
import glob, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for path in sorted(glob.glob(os.path.join(root, "glfer_amd", "csrc", "build", "*.log"))):
    if flt not in os.path.basename(path):
        continue
    for b in open(path).read().split("Function Name: ")[1:]:
        name = b.split(" ")[0]
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
        g = lambda k: re.search(k + r": (\d+)", b).group(1)
        print("%-28s %-60s VGPR %3s spill %2s SGPRspill %2s scratch %3s occ %s LDS %6s" % (
            os.path.basename(path)[:-4], name.replace("void glfer::", "").replace("(SpectroParams)", "")[:60], g("VGPRs"), g("VGPRs Spill"),
            g("SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
