import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import glfer_amd as G
from oracle import oracle as O
from _signals import synth, rel_err
for (n, ovl, t, pe, frames) in ((4096, 0.0, 128, 32, 12), (1024, 0.5, 96, 16, 16), (4096, 0.75, 96, 16, 10)):
    h = O.hop(n, ovl)
    x = synth(frames * h, seed=1)
    want = O.spectrogram_hparma(x, n, ovl, t, pe)
    sp = G.Spectrogram(G.HparmaParams(n=n, overlap=ovl, t=t, p_e=pe))
    xd = torch.from_numpy(x).cuda()
    got = sp.run(xd); torch.cuda.synchronize()
    got = got.cpu().numpy()
    inv_w, inv_g = 1.0 / want[:, :n // 2].astype(np.float64), 1.0 / got[:, :n // 2].astype(np.float64)
    e_inv = [max(rel_err(inv_g[f], inv_w[f])) for f in range(frames)]
    e_psd = [max(rel_err(got[f], want[f])) for f in range(frames)]
    e_bin = [np.abs(got[f] / want[f] - 1).max() for f in range(frames)]
    print(n, ovl, t, pe, "err |A|^2:", ["%.1e" % e for e in e_inv[:6]], "psd peak-norm:", ["%.1e" % e for e in e_psd[:6]], "per-bin rel:", ["%.1e" % e for e in e_bin[:6]])
# throughput
n, t, pe = 4096, 128, 32
frames = 8192
x = torch.from_numpy(synth(frames * n, seed=2)).cuda()
sp = G.Spectrogram(G.HparmaParams(n=n, overlap=0.0, t=t, p_e=pe))
out = sp.run(x); torch.cuda.synchronize()
t0 = time.perf_counter(); out = sp.run(x); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("C5 throughput: %d frames in %.3f s = %.0f frames/s" % (frames, dt, frames / dt))
