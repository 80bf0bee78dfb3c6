import sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G
for n in (32768, 65536):
    sp = G.Spectrogram(G.FftParams(n=n, window_type=7, overlap=0.5))
    frames = (1 << 29) // n
    x = torch.randn(frames * sp.hop + n, device='cuda')
    out = torch.empty((frames, sp.bins), device='cuda')
    for _ in range(3): sp.run(x, nframes=frames, out=out)
    torch.cuda.synchronize()
