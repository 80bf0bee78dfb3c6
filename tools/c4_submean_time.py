"""GPU box: C4 (N = 16384, 9 tapers, overlap 0) without and with per-hop mean removal (the reference's default)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import glfer_amd as lib  # noqa: E402

n, frames = 16384, 32768
x = (torch.randn(frames * n, device="cuda") * 0.2 + 0.05).contiguous()
for sub_mean in (0, 1):
    sp = lib.Spectrogram(lib.MtmParams(n=n, overlap=0.0, w=4.5, kmax=8, sub_mean=sub_mean))
    out = sp.run(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        sp.run(x, out=out) if "out" in sp.run.__code__.co_varnames else sp.run(x)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"C4 sub_mean={sub_mean}: {frames / ms / 1e3:.2f} M frames/s ({ms:.3f} ms)")
