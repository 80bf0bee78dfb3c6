import sys, time
sys.path.insert(0, '.')
import torch, numpy as np
import glfer_amd as G
for (n, ovl, w, k, frames) in ((16384, 0.0, 4.5, 8, 16384), (4096, 0.75, 2.5, 4, 262144), (1024, 0.5, 4.0, 7, 524288), (4096, 0.0, 4.0, 7, 131072)):
    sp = G.Spectrogram(G.MtmParams(n=n, overlap=ovl, w=w, kmax=k))
    x = torch.randn(frames * sp.hop + (n - sp.hop), device='cuda')
    out = torch.empty((sp.num_frames(x.numel()), sp.bins), device='cuda')
    nf = out.shape[0]
    for _ in range(2): sp.run(x, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): sp.run(x, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("MTM n=%d overlap=%.2f tapers=%d: %.2f M frames/s, %.0f GB/s algorithmic" % (n, ovl, k + 1, nf / dt / 1e6, nf * (4 * sp.hop + 4 * sp.bins) / dt / 1e9))
for (n, ovl, frames) in ((1024, 0.5, 1048576), (16384, 0.5, 32768), (256, 0.0, 2097152)):
    sp = G.Spectrogram(G.FftParams(n=n, overlap=ovl, window_type=0))
    x = torch.randn(frames * sp.hop + (n - sp.hop), device='cuda')
    out = torch.empty((sp.num_frames(x.numel()), sp.bins), device='cuda')
    nf = out.shape[0]
    for _ in range(2): sp.run(x, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): sp.run(x, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("FFT n=%d overlap=%.2f: %.2f M frames/s, %.0f GB/s algorithmic" % (n, ovl, nf / dt / 1e6, nf * (4 * sp.hop + 4 * sp.bins) / dt / 1e9))
