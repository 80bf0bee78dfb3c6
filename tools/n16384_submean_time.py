"""GPU box: N = 16384, 9 tapers, per-hop mean removal inside spectro16w's multitaper form against the copy pre-pass
(GLFER_MEAN_PREPASS=1), at overlap 0, 50, 75 %."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import glfer_amd as lib  # noqa: E402

n, frames = 16384, 32768
for ovl in (0.0, 0.5, 0.75):
    hop = int(n * (1 - ovl))
    x = (torch.randn(frames * hop + n, device="cuda") * 0.2 + 0.05).contiguous()
    for label, sub_mean, pre in (("no mean removal", 0, "0"), ("means inside the kernel", 1, "0"), ("copy pre-pass", 1, "1")):
        os.environ["GLFER_MEAN_PREPASS"] = pre
        sp = lib.Spectrogram(lib.MtmParams(n=n, overlap=ovl, w=4.5, kmax=8, sub_mean=sub_mean))
        sp.run(x, nframes=frames)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            sp.run(x, nframes=frames)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        print(f"N=16384 9 tapers overlap {ovl}: {label}: {frames / ms / 1e3:.2f} M frames/s")
os.environ.pop("GLFER_MEAN_PREPASS", None)
