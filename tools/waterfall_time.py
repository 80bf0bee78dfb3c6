"""GPU box: glfer_hip_waterfall_device with averaging -- the averages taken inside the map kernel
(default) against the staged form (GLFER_WATERFALL_FUSED=0).  python3 tools/waterfall_time.py [rows] [bins]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import glfer_amd as lib  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 2049
g = torch.Generator(device="cuda")
g.manual_seed(1)
psd = (torch.rand((rows, bins), device="cuda", generator=g) ** 4 * 1e-3 + 1e-9).contiguous()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


print(f"# {rows} rows of {bins} bins, scale LOG, autoscale; M rows/s")
for name, mode in (("none", 0), ("plain", lib.AVG_PLAIN), ("sumextreme", lib.AVG_SUMEXTREME), ("sumavg", lib.AVG_SUMAVG)):
    for depth in (4, 8):
        if mode == 0 and depth != 4:
            continue
        for want_lev in (True, False):
            line = f"avg {name:10s} depth {depth} levbuf {'yes' if want_lev else 'no '}:"
            for fused in (("1", "0") if mode else ("1",)):
                os.environ["GLFER_WATERFALL_FUSED"] = fused
                d = lib.Display(scale_type=lib.SCALE_LOG, autoscale=1, overlap=0.5, palette=0)
                ms = timed(lambda: lib.waterfall(d, psd, avg_mode=mode, depth=depth, minbin=0, maxbin=bins, want_lev=want_lev))
                line += f"  {'fused ' if fused == '1' else 'staged'} {rows / ms / 1e3:7.1f}"
            print(line)
