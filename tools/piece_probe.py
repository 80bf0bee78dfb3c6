"""One workload of tools/exact_mean_time.py, a few calls, for a rocprofv3 --kernel-trace --stats run: are a piece's
estimator launches shorter when the piece's hop means were taken just before (its samples still in the Infinity Cache)?
    python tools/piece_probe.py C1|C2|C3 <sub_mean 0|1|2> [GLFER_EXACT_PIECE_MB [GLFER_EXACT_STREAMS]]"""
import os, sys
sys.path.insert(0, '.')
case, mode = sys.argv[1], int(sys.argv[2])
if len(sys.argv) > 3:
    os.environ["GLFER_EXACT_PIECE_MB"] = sys.argv[3]
if len(sys.argv) > 4:
    os.environ["GLFER_EXACT_STREAMS"] = sys.argv[4]
import torch
import glfer_amd as G
P = {"C1": (G.FftParams, dict(n=1024, window_type=0, overlap=0.5)), "C2": (G.FftParams, dict(n=4096, window_type=0, overlap=0.75)),
     "C3": (G.MtmParams, dict(n=4096, overlap=0.0, w=2.5, kmax=4))}[case]
sp = G.Spectrogram(P[0](sub_mean=mode, **P[1]))
frames = min((1 << 30) // sp.hop, 1 << 21)
x = torch.randn(frames * sp.hop + (sp.n - sp.hop), device='cuda') * 0.2 + 0.1
out = torch.empty((sp.num_frames(x.numel()), sp.bins), device='cuda')
for _ in range(6):
    sp.run(x, out=out)
torch.cuda.synchronize()
print("done", case, mode, sys.argv[3:])
