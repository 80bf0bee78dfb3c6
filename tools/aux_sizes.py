"""update_avg / display throughput by row length (GLFER_LIB_PATH selects another build for an A/B)."""
import os, sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G
import glfer_amd.api as A
if os.environ.get("GLFER_LIB_PATH"):
    A.LIB_PATH = os.environ["GLFER_LIB_PATH"]

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

for bins in (129, 513, 2049, 8193):
    rows = (256 << 20) // (bins * 4)
    psd = (torch.rand((rows, bins), device='cuda') ** 4).contiguous()
    stats = G.compute_floor(psd)
    lo, hi = max(1, bins // 80), bins - bins // 40
    for mode, name in ((G.AVG_PLAIN, "plain"), (G.AVG_SUMAVG, "sumavg")):
        for depth in (4, 64):
            dt = timeit(lambda: G.update_avg(mode, psd, depth, lo, hi, max0=1))
            print("update_avg %-6s depth %2d %5d bins: %8.2f M rows/s %5.0f GB/s" % (name, depth, bins, rows / dt / 1e6, rows * bins * 12 / dt / 1e9), flush=True)
    for scale, auto in ((G.SCALE_LOG, 1), (G.SCALE_LIN, 0)):
        dt = timeit(lambda: G.display(G.Display(scale_type=scale, autoscale=auto, overlap=0.5), psd, stats))
        print("display scale=%d auto=%d   %5d bins: %8.2f M rows/s %5.0f GB/s" % (scale, auto, bins, rows / dt / 1e6, rows * bins * 9 / dt / 1e9), flush=True)
