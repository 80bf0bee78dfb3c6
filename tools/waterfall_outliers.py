"""GPU box: per-call times of the staged waterfall (GLFER_WATERFALL_FUSED=0: 2 GiB of averaged rows as scratch per call)
-- how often does a call take far longer than the median?  python3 tools/waterfall_outliers.py [calls]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import glfer_amd as lib  # noqa: E402
from glfer_amd import api  # noqa: E402

os.environ["GLFER_WATERFALL_FUSED"] = "0"
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rows, bins = 131072, 2049
g = torch.Generator(device="cuda")
g.manual_seed(1)
psd = (torch.rand((rows, bins), device="cuda", generator=g) ** 4 * 1e-3 + 1e-9).contiguous()
import ctypes as C  # noqa: E402

# outputs allocated once: what is timed is the library's own allocations (its stream-ordered scratch), not torch's
rgb = torch.empty((rows, bins, 3), dtype=torch.uint8, device="cuda")
lev = torch.empty((rows, bins), dtype=torch.int16, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ts = []
for i in range(calls):
    d = lib.Display(scale_type=lib.SCALE_LOG, autoscale=1, overlap=0.5, palette=0)
    mode = (lib.AVG_PLAIN, lib.AVG_SUMEXTREME, lib.AVG_SUMAVG)[i % 3]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = api.lib().glfer_hip_waterfall_device(C.byref(d), int(mode), 4 + 4 * (i % 2), 0, bins, 0, psd.data_ptr(), rows, bins, rgb.data_ptr(),
                                              lev.data_ptr() if i % 2 else None, None, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ts = torch.tensor(ts[3:])
med = ts.median().item()
print(f"{len(ts)} calls: median {med * 1e3:.2f} ms, max {ts.max().item() * 1e3:.1f} ms, calls over 5x the median: {(ts > 5 * med).sum().item()}")
print("slow calls (index after the 3 dropped, ms):", [(int(i), round(ts[i].item() * 1e3, 1)) for i in (ts > 5 * med).nonzero().flatten()])
