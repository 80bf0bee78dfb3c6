"""compute_floor throughput by row length (GLFER_LIB_PATH selects another build for an A/B)."""
import os, sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G
import glfer_amd.api as A
if os.environ.get("GLFER_LIB_PATH"):
    A.LIB_PATH = os.environ["GLFER_LIB_PATH"]
for bins in (129, 513, 1025, 2049, 4097, 8193):
    rows = (1 << 30) // (bins * 4)              # 1 GiB: well past the 256 MiB Infinity Cache, kernel time well past the call overhead
    psd = (torch.rand((rows, bins), device='cuda') ** 4).contiguous()
    G.compute_floor(psd); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): G.compute_floor(psd)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("compute_floor %5d bins: %8.2f M rows/s  %5.0f GB/s" % (bins, rows / dt / 1e6, rows * bins * 4 / dt / 1e9), flush=True)
