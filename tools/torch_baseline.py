"""The same estimators composed from library pieces on the same GPU -- torch (rocFFT behind torch.fft.rfft)
with the stream resident in HBM -- next to the fused kernels: frames/s for C3 (multitaper) and C2 (periodogram).
A context number for DESIGN.md, not a parity test (tests/ compare against the oracle)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import glfer_amd as G

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

dev = torch.device("cuda")
# ---- C3: multitaper N=4096, 5 tapers, overlap 0
n, frames = 4096, 262144
sp = G.Spectrogram(G.MtmParams(n=n, overlap=0.0, w=2.5, kmax=4))
v, sig = sp.tapers()
tap = torch.from_numpy((v / np.sqrt(n * (1.0 + sig))[:, None]).astype(np.float32)).to(dev)    # weights and 1/N folded in
x = torch.randn(frames * n, device=dev)
out = torch.empty((frames, n // 2 + 1), device=dev)
def torch_mtm(chunk=8192):
    for f0 in range(0, frames, chunk):
        fr = x[f0 * n:(f0 + chunk) * n].view(chunk, 1, n)
        s = torch.fft.rfft(fr * tap[None], dim=-1)
        out[f0:f0 + chunk] = (s.real ** 2 + s.imag ** 2).sum(1)
dt_t = timeit(torch_mtm)
ref = out[:64].clone()
dt_g = timeit(lambda: sp.run(x, out=out))
err = ((out[:64] - ref).abs().max() / ref.max()).item()
print("C3 multitaper N=4096 5 tapers: torch.fft composition %.2f M frames/s, fused kernel %.2f M frames/s (x%.1f); max|d|/max %.1e"
      % (frames / dt_t / 1e6, frames / dt_g / 1e6, dt_t / dt_g, err), flush=True)
# ---- C2: periodogram N=4096 Hanning, overlap 75 %
frames, hop = 262144, 1024
sp = G.Spectrogram(G.FftParams(n=n, overlap=0.75, window_type=G.WINDOWS["hanning"]))
w = torch.from_numpy((sp.window() / np.sqrt(n)).astype(np.float32)).to(dev)
x = torch.randn(frames * hop + (n - hop), device=dev)
xs = torch.cat([torch.zeros(n - hop, device=dev), x])           # the reference's zero history in front
nf = sp.num_frames(x.numel())
out = torch.empty((nf, n // 2 + 1), device=dev)
def torch_fft(chunk=16384):
    fr = xs.unfold(0, n, hop)
    for f0 in range(0, nf, chunk):
        s = torch.fft.rfft(fr[f0:f0 + chunk] * w, dim=-1)
        out[f0:f0 + s.shape[0]] = s.real ** 2 + s.imag ** 2
dt_t = timeit(torch_fft)
ref = out[100:164].clone()
dt_g = timeit(lambda: sp.run(x, out=out))
err = ((out[100:164] - ref).abs().max() / ref.max()).item()
print("C2 periodogram N=4096 overlap 0.75: torch.fft composition %.2f M frames/s, fused kernel %.2f M frames/s (x%.1f); max|d|/max %.1e"
      % (nf / dt_t / 1e6, nf / dt_g / 1e6, dt_t / dt_g, err), flush=True)
