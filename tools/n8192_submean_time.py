"""GPU box: N = 8192 multitaper with per-hop mean removal -- spectro16h's multitaper form behind the copy pre-pass (default
route) against spectro16w's form with the means taken inside (GLFER_FORM=w)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import glfer_amd as lib  # noqa: E402

n, frames = 8192, 65536
for ovl in (0.0, 0.5):
    hop = int(n * (1 - ovl))
    x = (torch.randn(frames * hop + n, device="cuda") * 0.2 + 0.05).contiguous()
    for kmax, w in ((4, 2.5), (7, 4.0)):
        for sub_mean in (0, 1):
            for form in ("", "w"):
                if form:
                    os.environ["GLFER_FORM"] = form
                else:
                    os.environ.pop("GLFER_FORM", None)
                sp = lib.Spectrogram(lib.MtmParams(n=n, overlap=ovl, w=w, kmax=kmax, sub_mean=sub_mean))
                sp.run(x, nframes=frames)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5):
                    sp.run(x, nframes=frames)
                b.record()
                torch.cuda.synchronize()
                ms = a.elapsed_time(b) / 5
                print(f"N=8192 overlap {ovl} {kmax + 1} tapers sub_mean={sub_mean} form={'default (h)' if not form else 'w'}: {frames / ms / 1e3:.2f} M frames/s")
os.environ.pop("GLFER_FORM", None)
