import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import glfer_amd as G
from oracle import oracle as O
from _signals import synth, rel_err
n, ovl, t, pe, sm = 1024, 0.5, 96, 16, 1
frames = 10
h = O.hop(n, ovl)
x = synth(frames * h, seed=n + t)
ref = O.hparma_frames(x, n, ovl, t, pe, sub_mean=sm)
sp = G.Spectrogram(G.HparmaParams(n=n, overlap=ovl, t=t, p_e=pe, sub_mean=sm))
got = sp.run(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64)
k = np.arange(n // 2 + 1)
for f, (psd, a, rank) in enumerate(ref):
    inv_g, inv_w = 1.0 / got[f, :n // 2], 1.0 / psd[:n // 2].astype(np.float64)
    A = np.polyval(a[::-1].astype(np.float64), np.exp(-2j * np.pi * k / n))
    ex = (np.abs(A) ** 2 / n)[:n // 2]
    print(f, "rank", rank, "gpu-vs-oracle %.1e" % max(rel_err(inv_g, inv_w)), "gpu-vs-exact(a_oracle) %.1e" % max(rel_err(inv_g, ex)),
          "oracle-vs-exact %.1e" % max(rel_err(inv_w, ex)), "a0..2", a[:3])
