"""C2 + update_avg_plain depth D: the average taken inside the periodogram launch (glfer_hip_spectrogram_avg_device) against the two
launches, M rows/s on a device-resident stream.  GLFER_AVG_GRID_MULT (workgroups per resident slot of the fused launch) is read once
per process: run once per value.   python tools/avg_fused_time.py [frames] [depth]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import glfer_amd as G

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
L = G.api.lib()
for wl in ("fft", "fft1k"):
    sp = G.Spectrogram(bench.make_params(G, wl))
    x = bench.synth_on_device(torch, frames * sp.hop, dev, seed=0)
    bins = sp.bins
    psd = torch.empty((frames, bins), dtype=torch.float32, device=dev)
    avg = torch.empty((frames, bins), dtype=torch.float64, device=dev)
    ret = torch.empty((frames, 4), dtype=torch.float64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def two():
        sp.run(x, out=psd)
        assert L.glfer_hip_avg_device(G.AVG_PLAIN, psd.data_ptr(), frames, bins, bins, depth, 0, bins, 0, avg.data_ptr(), ret.data_ptr(), st) == 0

    def fused(rows, want_ret):
        assert L.glfer_hip_spectrogram_avg_device(sp._h, C.c_void_p(x.data_ptr()), x.numel(), 0, frames, G.AVG_PLAIN, depth, 0, bins, 0, bins,
                                                  C.c_void_p(psd.data_ptr() if rows else None), C.c_void_p(avg.data_ptr()),
                                                  C.c_void_p(ret.data_ptr() if want_ret else None), st) == 0

    def rate(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 5)
        return frames / best / 1e6

    print("%s depth %d, %d frames, GLFER_AVG_GRID_MULT=%s:  plain rows only %.1f | two launches %.1f | fused %.1f | fused, no return values %.1f | fused + rows %.1f  M rows/s"
          % (wl, depth, frames, os.environ.get("GLFER_AVG_GRID_MULT", "default"), rate(lambda: sp.run(x, out=psd)), rate(two), rate(lambda: fused(False, True)),
             rate(lambda: fused(False, False)), rate(lambda: fused(True, True))), flush=True)
    sp.close()
    del x, psd, avg, ret
    torch.cuda.empty_cache()
