"""Host-to-host rate of glfer_hip_spectrogram_host (both ends pinned) by chunk size (GLFER_INGEST_CHUNK, frames per chunk of the ring).
    python tools/chunk_probe.py"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, '.')
import numpy as np
import glfer_amd as G

frames = 131072
for name, P, kw in (("C3 mtm N=4096 ovl 0", G.MtmParams, dict(n=4096, overlap=0.0, w=2.5, kmax=4)), ("C2 fft N=4096 ovl 75%", G.FftParams, dict(n=4096, window_type=0, overlap=0.75))):
    sp = G.Spectrogram(P(sample_format=G.SAMPLES_S16, **kw))
    ns = frames * sp.hop + (sp.n - sp.hop)
    pcm = G.pinned_empty((ns,), np.int16)
    pcm[:] = (np.random.default_rng(1).standard_normal(ns) * 6000).clip(-32768, 32767).astype(np.int16)
    nfr = sp.num_frames(ns)
    rows = G.pinned_empty((nfr, sp.bins), np.float32)
    for chunk in (0, 1024, 2048, 4096, 8192, 16384, 32768):
        if chunk:
            os.environ["GLFER_INGEST_CHUNK"] = str(chunk)
        else:
            os.environ.pop("GLFER_INGEST_CHUNK", None)
        best = 1e9
        for r in range(4):
            nf = C.c_size_t(0)
            t0 = time.perf_counter()
            rc = G.api.lib().glfer_hip_spectrogram_host(sp._h, pcm.ctypes.data, pcm.size, rows.ctypes.data, C.byref(nf))
            dt = time.perf_counter() - t0
            assert rc == 0, rc
            if r:
                best = min(best, dt)
        print("%-24s chunk %6s: %6.2f M frames/s  %5.1f GB/s both ways" % (name, chunk or "dflt", nf.value / best / 1e6, nf.value * (2 * sp.hop + 4 * sp.bins) / best / 1e9))
    sp.close()
    del rows, pcm
