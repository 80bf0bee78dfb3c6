"""Throughput of the per-column kernels that follow the estimator (floor statistics, moving average,
display mapping) on BASELINE-sized batches: rows/s and achieved HBM GB/s (algorithmic bytes)."""
import sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

rows, bins = 131072, 2049
psd = (torch.rand((rows, bins), device='cuda') ** 4).contiguous()
dt = timeit(lambda: G.compute_floor(psd))
print("compute_floor   : %.2f M rows/s  %.0f GB/s (read %d B/row)" % (rows / dt / 1e6, rows * bins * 4 / dt / 1e9, bins * 4))
stats = G.compute_floor(psd)
for mode, name in ((G.AVG_PLAIN, "plain"), (G.AVG_SUMAVG, "sumavg"), (G.AVG_SUMEXTREME, "sumextreme")):
    dt = timeit(lambda: G.update_avg(mode, psd, 4, 25, 2000, max0=1))
    print("update_avg %-10s: %.2f M rows/s  %.0f GB/s (read 4 B + write 8 B per bin)" % (name, rows / dt / 1e6, rows * bins * 12 / dt / 1e9))
for scale, auto in ((G.SCALE_LOG, 1), (G.SCALE_LOG, 0), (G.SCALE_LIN, 1)):
    d = G.Display(scale_type=scale, autoscale=auto, overlap=0.5)
    dt = timeit(lambda: G.display(G.Display(scale_type=scale, autoscale=auto, overlap=0.5), psd, stats))
    print("display scale=%d autoscale=%d: %.2f M rows/s  %.0f GB/s (read 4 B, write 5 B per bin)" % (scale, auto, rows / dt / 1e6, rows * bins * 9 / dt / 1e9))

# the whole chain of main_window_draw for a batch: stage by stage over the whole batch (each stage
# sweeps 1 GiB of rows from HBM) against glfer_hip_waterfall_device (tiles that stay on the die)
def separate(avg):
    st = G.compute_floor(psd)
    if avg:
        a, _ = G.update_avg(G.AVG_PLAIN, psd, 4, 25, 2000)
        return G.display(G.Display(scale_type=G.SCALE_LOG, autoscale=1, overlap=0.5), a, st)
    return G.display(G.Display(scale_type=G.SCALE_LOG, autoscale=1, overlap=0.5), psd, st)

for avg in (0, 1):
    dt1 = timeit(lambda: separate(avg), reps=3)
    dt2 = timeit(lambda: G.waterfall(G.Display(scale_type=G.SCALE_LOG, autoscale=1, overlap=0.5), psd,
                                     avg_mode=G.AVG_PLAIN if avg else 0, depth=4, minbin=25, maxbin=2000), reps=3)
    print("floor%s + display, log autoscale: stage by stage %.2f M rows/s, tiled (glfer_hip_waterfall_device) %.2f M rows/s"
          % (" + avg(plain,4)" if avg else "", rows / dt1 / 1e6, rows / dt2 / 1e6))
