"""Cost of per-hop mean removal (sub_mean = opt.autoscale, the reference's default): C1/C2/C3 with and without it."""
import sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G
def bench(params, frames, label):
    sp = G.Spectrogram(params)
    x = torch.rand(frames * sp.hop + sp.n, device='cuda') - 0.5
    out = sp.run(x); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): sp.run(x, out=out) if 'out' in sp.run.__code__.co_varnames else sp.run(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("%-40s %8.2f M frames/s  (%.3f ms)" % (label, out.shape[0] / dt / 1e6, dt * 1e3), flush=True)
for sm in (0, 1):
    bench(G.FftParams(n=4096, window_type=0, overlap=0.75, sub_mean=sm), 1 << 20, "C2 periodogram sub_mean=%d" % sm)
    bench(G.MtmParams(n=4096, overlap=0.0, w=2.5, kmax=4, sub_mean=sm), 1 << 18, "C3 multitaper sub_mean=%d" % sm)
    bench(G.FftParams(n=1024, window_type=0, overlap=0.5, sub_mean=sm), 1 << 21, "C1 periodogram sub_mean=%d" % sm)
    bench(G.MtmParams(n=1024, overlap=0.0, w=4.0, kmax=7, sub_mean=sm), 1 << 20, "reference defaults: MTM N=1024 8 tapers sub_mean=%d" % sm)
    bench(G.FftParams(n=1024, window_type=7, overlap=0.0, sub_mean=sm), 1 << 21, "reference defaults: FFT N=1024 Kaiser sub_mean=%d" % sm)
