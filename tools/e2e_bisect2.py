"""The host path's rate against the state of the process's device memory: blocks cached by torch (reserved, not live), and the
seconds after a large hipFree."""
import ctypes as C
import sys
import time
sys.path.insert(0, '.')
import numpy as np
import torch
import glfer_amd as G

frames = 131072
torch.cuda.set_device(0)
pcm = G.pinned_empty((frames * 4096,), np.int16)
pcm[:] = (np.random.default_rng(1).standard_normal(pcm.size) * 6000).clip(-32768, 32767).astype(np.int16)
rows = G.pinned_empty((frames, 2049), np.float32)
sp = G.Spectrogram(G.MtmParams(n=4096, overlap=0.0, w=2.5, kmax=4, sample_format=G.SAMPLES_S16), device=0)

def once():
    nf = C.c_size_t(0)
    t0 = time.perf_counter()
    rc = G.api.lib().glfer_hip_spectrogram_host(sp._h, pcm.ctypes.data, pcm.size, rows.ctypes.data, C.byref(nf))
    assert rc == 0
    return frames / (time.perf_counter() - t0) / 1e6

once(); once()
print("baseline                         %.2f %.2f" % (once(), once()))
t = torch.empty(3 << 30, dtype=torch.uint8, device="cuda"); t.zero_(); torch.cuda.synchronize()
print("3 GiB live                       %.2f %.2f" % (once(), once()))
del t
print("3 GiB cached by torch (%.1f GiB reserved)  %s" % (torch.cuda.memory_reserved() / 2**30, " ".join("%.2f" % once() for _ in range(6))))
t0 = time.perf_counter()
torch.cuda.empty_cache()
print("after empty_cache (hipFree of 3 GiB), one call after the other:")
for i in range(12):
    r = once()
    print("   t = %.2f s  %.2f M frames/s" % (time.perf_counter() - t0, r))
    time.sleep(0.15)

print("---- 12 GiB freed at once, then the host path for five seconds")
ts = [torch.empty(3 << 30, dtype=torch.uint8, device="cuda") for _ in range(4)]
for t in ts:
    t.zero_()
torch.cuda.synchronize()
del ts, t
t0 = time.perf_counter()
torch.cuda.empty_cache()
while time.perf_counter() - t0 < 5.0:
    r = once()
    print("   t = %.2f s  %.2f M frames/s" % (time.perf_counter() - t0, r))
    time.sleep(0.25)
