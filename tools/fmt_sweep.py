"""Kernel throughput by sample format (f32 / s16 / u8) at the C2 and C3 shapes, device-resident input."""
import sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G

def timeit(sp, x, out):
    for _ in range(3): sp.run(x, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8): sp.run(x, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 8

for name, mk, frames in (("C2 periodogram n=4096 overlap 0.75", lambda f: G.FftParams(n=4096, overlap=0.75, window_type=0, sample_format=f), 1048576),
                         ("C3 multitaper n=4096 5 tapers", lambda f: G.MtmParams(n=4096, overlap=0.0, w=2.5, kmax=4, sample_format=f), 262144)):
    xf = None
    for label, fmt in (("f32", G.SAMPLES_F32), ("s16", G.SAMPLES_S16), ("u8", G.SAMPLES_U8)):
        sp = G.Spectrogram(mk(fmt))
        ns = frames * sp.hop + (4096 - sp.hop)
        if xf is None:
            xf = torch.randn(ns, device='cuda') * 0.2
        out = torch.empty((sp.num_frames(ns), sp.bins), device='cuda')
        x = {"f32": lambda: xf, "s16": lambda: (xf * 32767).clamp(-32768, 32767).to(torch.int16),
             "u8": lambda: (xf * 127 + 128).clamp(0, 255).to(torch.uint8)}[label]()
        dt = timeit(sp, x, out)
        print("%s, %s: %.1f M frames/s" % (name, label, out.shape[0] / dt / 1e6), flush=True)
