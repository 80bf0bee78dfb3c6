"""M frames/s of one workload of tools/exact_mean_time.py with the environment as given (knobs read once per process).
    python tools/one_rate.py C1|C2|C3 <sub_mean 0|1|2>"""
import sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G
case, mode = sys.argv[1], int(sys.argv[2])
P = {"C1": (G.FftParams, dict(n=1024, window_type=0, overlap=0.5)), "C2": (G.FftParams, dict(n=4096, window_type=0, overlap=0.75)),
     "C3": (G.MtmParams, dict(n=4096, overlap=0.0, w=2.5, kmax=4))}[case]
sp = G.Spectrogram(P[0](sub_mean=mode, **P[1]))
frames = min((1 << 30) // sp.hop, 1 << 21)
x = torch.randn(frames * sp.hop + (sp.n - sp.hop), device='cuda') * 0.2 + 0.1
out = torch.empty((sp.num_frames(x.numel()), sp.bins), device='cuda')
best = 1e9
for rep in range(3):
    sp.run(x, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        sp.run(x, out=out)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 4)
print("%.1f" % (out.shape[0] / best / 1e6))
