"""Multitaper throughput by block size and taper count (f32, overlap 0), for A/B runs of two builds:
GLFER_LIB_PATH=<other libglfer_hip.so> python3 tools/mtm_sizes.py"""
import os, sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G
import glfer_amd.api as A
if os.environ.get("GLFER_LIB_PATH"):
    A.LIB_PATH = os.environ["GLFER_LIB_PATH"]
for n, nw, k, frames in ((2048, 2.5, 4, 262144), (4096, 2.5, 4, 262144), (8192, 2.5, 4, 65536), (8192, 4.0, 7, 65536),
                         (8192, 4.5, 8, 65536), (16384, 2.5, 4, 32768), (16384, 4.0, 7, 32768), (16384, 4.5, 8, 32768)):
    sp = G.Spectrogram(G.MtmParams(n=n, overlap=0.0, w=nw, kmax=k))
    x = torch.randn(frames * sp.hop, device='cuda')
    out = torch.empty((sp.num_frames(x.numel()), sp.bins), device='cuda')
    best = 1e9
    for rep in range(3):
        for _ in range(2): sp.run(x, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): sp.run(x, out=out)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 5)
    nf = out.shape[0]
    print("MTM n=%d tapers=%d: %.2f M frames/s, %.0f GB/s algorithmic" % (n, k + 1, nf / best / 1e6, nf * (4 * sp.hop + 4 * sp.bins) / best / 1e9), flush=True)
