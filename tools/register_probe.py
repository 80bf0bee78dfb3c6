"""How fast hipHostRegister pins a caller's pageable buffer (the alternative to staging copies for pageable ends of the host entries)."""
import ctypes as C, time
import numpy as np
import torch
torch.cuda.init()
hip = C.CDLL("libamdhip64.so")
for mb in (64, 256, 1024):
    a = np.empty(mb << 20, np.uint8)
    a[:] = 1                                  # touched pages
    t0 = time.perf_counter()
    rc = hip.hipHostRegister(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 0)
    t1 = time.perf_counter()
    hip.hipHostUnregister(C.c_void_p(a.ctypes.data))
    t2 = time.perf_counter()
    b = np.empty(mb << 20, np.uint8)          # untouched pages
    t3 = time.perf_counter()
    rc2 = hip.hipHostRegister(C.c_void_p(b.ctypes.data), C.c_size_t(b.nbytes), 0)
    t4 = time.perf_counter()
    hip.hipHostUnregister(C.c_void_p(b.ctypes.data))
    print("%5d MiB: register touched %6.1f ms (%.1f GB/s, rc %d), unregister %6.1f ms; register untouched %6.1f ms (rc %d)"
          % (mb, (t1 - t0) * 1e3, a.nbytes / (t1 - t0) / 1e9, rc, (t2 - t1) * 1e3, (t4 - t3) * 1e3, rc2))
