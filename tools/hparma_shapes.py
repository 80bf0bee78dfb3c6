"""HP-ARMA rates by matrix shape: the kernel with the shape as compile-time constants against the one that reads it from its parameters
(GLFER_HPARMA_GENERIC=1).   python tools/hparma_shapes.py"""
import os
import sys
import time
sys.path.insert(0, '.')
import torch
import glfer_amd as G

for n, t, p_e in ((4096, 128, 32), (4096, 96, 16), (1024, 96, 16), (2048, 64, 8)):
    sp = G.Spectrogram(G.HparmaParams(n=n, overlap=0.0, t=t, p_e=p_e))
    frames = 32768
    x = torch.randn(frames * n, device='cuda') * 0.2
    out = torch.empty((frames, sp.bins), device='cuda')
    res = []
    for generic in ("0", "1"):
        os.environ["GLFER_HPARMA_GENERIC"] = generic
        best = 1e9
        for rep in range(3):
            sp.run(x, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sp.run(x, out=out)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        res.append(frames / best / 1e6)
    print("N %5d t %3d p_e %2d: %.2f M frames/s   (shape from the parameters: %.2f)" % (n, t, p_e, res[0], res[1]))
