"""One form of C2 (+ update_avg_plain depth 4), a few launches: the program rocprofv3 runs for tools/stall_cmd.sh.
python tools/avg_one.py plain|fused|fused_noret|fused_rows|two [frames]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import glfer_amd as G

form = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
dev = torch.device("cuda", 0)
L = G.api.lib()
sp = G.Spectrogram(bench.make_params(G, "fft"))
x = bench.synth_on_device(torch, frames * sp.hop, dev, seed=0)
bins = sp.bins
psd = torch.empty((frames, bins), dtype=torch.float32, device=dev)
avg = torch.empty((frames, bins), dtype=torch.float64, device=dev)
ret = torch.empty((frames, 4), dtype=torch.float64, device=dev)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for _ in range(4):
    if form in ("plain", "two"):
        sp.run(x, out=psd)
    if form == "two":
        assert L.glfer_hip_avg_device(G.AVG_PLAIN, psd.data_ptr(), frames, bins, bins, 4, 0, bins, 0, avg.data_ptr(), ret.data_ptr(), st) == 0
    if form.startswith("fused"):
        assert L.glfer_hip_spectrogram_avg_device(sp._h, C.c_void_p(x.data_ptr()), x.numel(), 0, frames, G.AVG_PLAIN, 4, 0, bins, 0, bins,
                                                  C.c_void_p(psd.data_ptr() if form == "fused_rows" else None), C.c_void_p(avg.data_ptr()),
                                                  C.c_void_p(None if form == "fused_noret" else ret.data_ptr()), st) == 0
torch.cuda.synchronize()
