"""64-bit addressing check: a 16 GiB f32 stream (2^20 frames of N=4096, overlap 0, multitaper) and an
8 GiB s16 stream at 75 % overlap (2^22 hops); a few frames spread over the stream, including the
last ones, against the oracle computed from the same samples."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
import glfer_amd as G
from oracle import oracle as O

def check(params, x, probe, name, to_float=lambda a: a):
    sp = G.Spectrogram(params)
    out = sp.run(x)
    torch.cuda.synchronize()
    n, h = sp.n, sp.hop
    worst = 0.0
    for f in probe:
        lo = f * h - (n - h)
        seg = to_float(x[max(lo, 0):f * h + h].cpu().numpy())
        if lo < 0:
            seg = np.concatenate([np.zeros(-lo, np.float32), seg])
        # one frame with its true history: run the oracle on [history | hop] as a zero-overlap frame
        if params.mode == G.MODE_MTM:
            want = O.spectrogram_mtm(seg, n, 0.0, params.w, params.kmax)[0]
        else:
            want = O.spectrogram_fft(seg, n, 0.0, params.window_type)[0]
        got = out[f].cpu().numpy()
        worst = max(worst, np.abs(got - want).max() / want.max())
    print("%s: %d frames, stream %.1f GiB, worst per-frame error %.2e" % (name, out.shape[0], x.numel() * x.element_size() / 2**30, worst))
    assert worst < 1e-5

frames = 1 << 20
x = torch.empty(frames * 4096, dtype=torch.float32, device='cuda')
for s in range(0, x.numel(), 1 << 28):
    e = min(x.numel(), s + (1 << 28))
    x[s:e] = torch.sin(torch.arange(s, e, device='cuda', dtype=torch.float64) * 0.013).float() * 0.4
    x[s:e] += 0.05 * torch.randn(e - s, device='cuda')
check(G.MtmParams(n=4096, overlap=0.0, w=2.5, kmax=4), x, [0, 1, 524287, 524288, frames - 2, frames - 1], "multitaper N=4096 f32")
del x
hops = 1 << 22
x = (torch.randn(hops * 1024, device='cuda') * 3000).clamp_(-32768, 32767).to(torch.int16)
check(G.FftParams(n=4096, window_type=0, overlap=0.75, sample_format=G.SAMPLES_S16), x, [0, 2, 3, 4, 2097151, 2097152, hops - 1], "periodogram N=4096 s16 75%",
      to_float=lambda a: O.pcm_s16_to_float(a))

# the same stream with per-hop mean removal (taken out inside the kernels): the probe frame's hops
# need their own means, so the oracle runs over the frames before it too (true overlap) and the
# last row is compared
def check_mean(params, x, probe, name, to_float):
    sp = G.Spectrogram(params)
    out = sp.run(x)
    torch.cuda.synchronize()
    n, h = sp.n, sp.hop
    back = (n - h + h - 1) // h
    worst = 0.0
    for f in probe:
        f0 = max(f - 2 * back, 0)
        seg = to_float(x[f0 * h:(f + 1) * h].cpu().numpy())
        want = O.spectrogram_fft(seg, n, params.overlap, params.window_type, 0.0, 0, 1, 0)[-1]
        if f0 > 0 or f >= back:
            got = out[f].cpu().numpy()
            worst = max(worst, np.abs(got - want).max() / want.max())
    print("%s: %d frames, worst per-frame error %.2e" % (name, out.shape[0], worst))
    assert worst < 1e-5

check_mean(G.FftParams(n=4096, window_type=0, overlap=0.75, sample_format=G.SAMPLES_S16, sub_mean=1), x,
           [40, 2097151, 2097152, hops - 1], "periodogram N=4096 s16 75% with mean removal", lambda a: O.pcm_s16_to_float(a))
print("ok")
