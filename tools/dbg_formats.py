import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import glfer_amd as G
from oracle import oracle as O
from _signals import synth, rel_err
x = synth(512 * 20, fs=8000.0, seed=5)
s16 = np.round(x * 32767).astype(np.int16)
u8 = np.clip(np.round(x * 127 + 128), 0, 255).astype(np.uint8)
for name, fmt, raw, conv in (("s16", G.SAMPLES_S16, s16, O.pcm_s16_to_float), ("u8", G.SAMPLES_U8, u8, O.pcm_u8_to_float)):
    for n, ovl in ((1024, 0.5), (4096, 0.0), (1024, 0.0)):
        sp = G.Spectrogram(G.FftParams(n=n, window_type=0, overlap=ovl, sample_format=fmt))
        got = sp.run(torch.from_numpy(raw).cuda()).cpu().numpy()
        want = O.spectrogram_fft(conv(raw), n, ovl, 0)
        errs = [max(rel_err(got[f], want[f])) for f in range(got.shape[0])]
        print(name, n, ovl, "frames", got.shape[0], "errs", ["%.1e" % e for e in errs[:12]])
