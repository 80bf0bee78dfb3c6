"""The host entry on PAGEABLE numpy buffers: as they are (staging copies), and registered in place for the call (hipHostRegister around it).
    python tools/register_e2e.py"""
import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np
import glfer_amd as G

hip = C.CDLL("libamdhip64.so")
L = G.api.lib()
frames = 131072
for name, P, kw in (("C3 mtm N=4096 ovl 0", G.MtmParams, dict(n=4096, overlap=0.0, w=2.5, kmax=4)), ("C2 fft N=4096 ovl 75%", G.FftParams, dict(n=4096, window_type=0, overlap=0.75))):
    sp = G.Spectrogram(P(sample_format=G.SAMPLES_S16, **kw))
    ns = frames * sp.hop + (sp.n - sp.hop)
    pcm = (np.random.default_rng(1).standard_normal(ns) * 6000).clip(-32768, 32767).astype(np.int16)
    nfr = sp.num_frames(ns)

    def call(rows, reg):
        nf = C.c_size_t(0)
        t0 = time.perf_counter()
        if reg:
            assert hip.hipHostRegister(C.c_void_p(pcm.ctypes.data), C.c_size_t(pcm.nbytes), 0) == 0
            assert hip.hipHostRegister(C.c_void_p(rows.ctypes.data), C.c_size_t(rows.nbytes), 0) == 0
        rc = L.glfer_hip_spectrogram_host(sp._h, pcm.ctypes.data, pcm.size, rows.ctypes.data, C.byref(nf))
        if reg:
            hip.hipHostUnregister(C.c_void_p(pcm.ctypes.data))
            hip.hipHostUnregister(C.c_void_p(rows.ctypes.data))
        assert rc == 0 and nf.value == nfr
        return time.perf_counter() - t0

    ref = None
    for reg in (0, 1):
        fresh, warm = [], []
        for r in range(3):
            rows = np.empty((nfr, sp.bins), np.float32)       # fresh pages every time
            fresh.append(call(rows, reg))
            warm.append(call(rows, reg))                      # the same array again: its pages exist
        if ref is None:
            ref = rows.copy()
        else:
            assert np.array_equal(ref, rows)
        print("%-24s %-28s fresh rows %.2f M frames/s   touched rows %.2f M frames/s" % (name, "registered for the call" if reg else "pageable (staging copies)", nfr / min(fresh) / 1e6, nfr / min(warm) / 1e6))
    sp.close()
