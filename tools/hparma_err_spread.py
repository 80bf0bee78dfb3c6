"""HP-ARMA: the device's distance from the oracle over many seeds, in the norm tests/test_gpu_parity.py::test_hparma_parity uses
(|A(f)|^2/N peak-normalised), next to the oracle's own movement when its input is perturbed by one float ulp.
python3 tools/hparma_err_spread.py  [GLFER_LIB_PATH=<other build>]"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import glfer_amd as lib
import glfer_amd.api as A
if os.environ.get("GLFER_LIB_PATH"): A.LIB_PATH = os.environ["GLFER_LIB_PATH"]
from oracle import oracle
from _signals import rel_err, synth
for n, overlap, t, p_e, sub_mean in ((4096, 0.0, 128, 32, 0), (1024, 0.5, 96, 16, 1), (4096, 0.75, 96, 16, 0), (2048, 0.0, 64, 8, 0)):
    h = oracle.hop(n, overlap)
    dev, own = [], []
    for seed in range(12):
        frames = 10
        x = synth(frames * h, seed=1000 * seed + n + t)
        ref = oracle.hparma_frames(x, n, overlap, t, p_e, sub_mean=sub_mean)
        xp = np.nextafter(x, np.float32(2.0) * np.sign(x).astype(np.float32)).astype(np.float32)
        per = oracle.hparma_frames(xp, n, overlap, t, p_e, sub_mean=sub_mean)
        sp = lib.Spectrogram(lib.HparmaParams(n=n, overlap=overlap, t=t, p_e=p_e, sub_mean=sub_mean))
        got = sp.run(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64)
        for f in range(frames):
            want = ref[f][0].astype(np.float64)
            dev.append(max(rel_err(1.0 / got[f, :n // 2], 1.0 / want[:n // 2])))
            own.append(max(rel_err(1.0 / per[f][0].astype(np.float64)[:n // 2], 1.0 / want[:n // 2])))
    dev, own = np.array(dev), np.array(own)
    print("N %5d t %3d p_e %2d mean %d: device vs oracle  median %.1e  p90 %.1e  max %.1e   |   oracle, input + 1 ulp  median %.1e  p90 %.1e  max %.1e"
          % (n, t, p_e, sub_mean, np.median(dev), np.percentile(dev, 90), dev.max(), np.median(own), np.percentile(own, 90), own.max()), flush=True)
