"""Where the read-ahead's time goes: one whole-file call against windowed calls of glfer_hip_spectrogram_wav_range."""
import ctypes as C, os, sys, time, wave
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import glfer_amd as G
from _signals import synth
n, hop, hops = 1024, 512, 100000
pcm = np.round(synth(hops * hop, seed=37) * 32767).astype(np.int16)
path = "/tmp/ra_long.wav"
with wave.open(path, "wb") as w:
    w.setnchannels(1); w.setsampwidth(2); w.setframerate(48000); w.writeframes(pcm.tobytes())
sp = G.Spectrogram(G.FftParams(n=n, window_type=0, overlap=0.5, sub_mean=G.SUBMEAN_EXACT, sample_format=G.SAMPLES_S16))
L = G.api.lib()
out = np.empty((hops, 513), np.float32)
nf = C.c_size_t(0)
for rep in range(3):
    t0 = time.perf_counter()
    L.glfer_hip_spectrogram_wav_range(sp._h, path.encode(), 0, hops, out.ctypes.data, C.byref(nf), 0, 1)
    print("whole file: %.1f ms (%d frames)" % ((time.perf_counter() - t0) * 1e3, nf.value))
for win in (8192, 32768):
    t0 = time.perf_counter()
    k = 0
    while k < hops:
        t1 = time.perf_counter()
        L.glfer_hip_spectrogram_wav_range(sp._h, path.encode(), k, win, out.ctypes.data, C.byref(nf), 0, 1)
        if k == 0 or k == win:
            print("  window of %d at %d: %.1f ms" % (win, k, (time.perf_counter() - t1) * 1e3))
        k += nf.value
    print("windows of %d: %.1f ms" % (win, (time.perf_counter() - t0) * 1e3))
