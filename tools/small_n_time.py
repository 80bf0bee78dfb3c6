"""Block sizes below the 16-points-per-lane range (spectro_small.hip, N = 8..128) and above it (N = 32768, 65536)."""
import sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G
def t(params, frames, label):
    sp = G.Spectrogram(params)
    x = torch.randn(frames * sp.hop + sp.n, device='cuda')
    out = torch.empty((frames, sp.bins), device='cuda')
    sp.run(x, nframes=frames, out=out); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4): sp.run(x, nframes=frames, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    print("%-44s %9.1f M frames/s  %6.0f GB/s algorithmic" % (label, frames / dt / 1e6, frames * (4 * sp.hop + 4 * sp.bins) / dt / 1e9), flush=True)
for n in (8, 16, 32, 64, 128):
    t(G.FftParams(n=n, window_type=7, overlap=0.0), (1 << 28) // n, "periodogram N=%d overlap 0" % n)
    t(G.MtmParams(n=n, overlap=0.5, w=2.0, kmax=3), (1 << 27) // n, "multitaper N=%d 4 tapers overlap 0.5" % n)
for n in (32768, 65536):
    t(G.FftParams(n=n, window_type=7, overlap=0.5), (1 << 29) // n, "periodogram N=%d overlap 0.5" % n)
    t(G.MtmParams(n=n, overlap=0.0, w=4.0, kmax=7), (1 << 28) // n, "multitaper N=%d 8 tapers" % n)
