"""The packed kernel (spectro16_kernel) by block size: even taper counts (its own cases) at overlap 0, for A/B runs of
two builds: GLFER_LIB_PATH=<other libglfer_hip.so> python3 tools/packed_sizes.py"""
import os, sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G
for n, nw, k, frames in ((256, 4.0, 7, 1 << 22), (512, 4.0, 7, 1 << 21), (1024, 4.0, 7, 1 << 20), (1024, 2.0, 3, 1 << 20), (2048, 4.0, 7, 1 << 19),
                         (4096, 4.0, 7, 1 << 18), (4096, 2.0, 3, 1 << 18), (2048, 2.5, 4, 1 << 19)):
    sp = G.Spectrogram(G.MtmParams(n=n, overlap=0.0, w=nw, kmax=k))
    x = torch.randn(frames * sp.hop, device='cuda')
    out = torch.empty((sp.num_frames(x.numel()), sp.bins), device='cuda')
    best = 1e9
    for rep in range(3):
        sp.run(x, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4): sp.run(x, out=out)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 4)
    nf = out.shape[0]
    print("MTM n=%d tapers=%d: %.2f M frames/s, %.0f GB/s algorithmic" % (n, k + 1, nf / best / 1e6, nf * (4 * sp.hop + 4 * sp.bins) / best / 1e9), flush=True)
