"""How many sweeps compute_svd (util.c:294-356) runs and how many rotations it applies on bench.py's C5 stream (numpy restatement,
CPU): bench.py's FP64 model counts 8 sweeps of 528 tested pairs and 2 567 applied rotations -- the smallest of the frames below --
where rounds 1-3 assumed "at least 12 sweeps" (sweepmax's floor, which the loop does not wait for)."""
import sys
sys.path.insert(0, 'tests')
import numpy as np
from _signals import synth
N, t, ncol = 4096, 128, 33


def build(x):                                             # hparma.c:89-102 with the row-0 overflow (glfer_hip.cpp's lag map)
    r = np.zeros(t, np.float32)
    for i in range(t):
        prod = (x[i:] * x[:N - i]).astype(np.float32).astype(np.float64)
        r[i] = np.float32(prod.cumsum()[-1] / (N - i))
    flat = -np.ones((t + 1) * ncol, int)
    flat[:t] = np.arange(t)
    for i in range(1, t):
        for j in range(ncol):
            flat[i * ncol + j] = flat[abs(j - i)]
    return r[np.where(flat[:t * ncol] < 0, 0, flat[:t * ncol]).reshape(t, ncol)].astype(np.float32)


def jacobi(A):
    A = A.copy()
    pending, sweeps, applied = 1, 0, []
    while pending > 0 and sweeps <= max(ncol, 12):
        pending, done = ncol * (ncol - 1) // 2, 0
        for j in range(ncol - 1):
            for k in range(j + 1, ncol):
                aj, ak = A[:, j].astype(np.float64), A[:, k].astype(np.float64)
                p, q, r = (aj * ak).sum(), (aj * aj).sum(), (ak * ak).sum()
                if q * r < 2.22e-16 or p * p / (q * r) < 1e-12:
                    pending -= 1
                    continue
                if q < r:
                    cs, sn = 0.0, 1.0
                else:
                    q -= r
                    v = np.sqrt(4 * p * p + q * q)
                    cs = np.sqrt((v + q) / (2 * v))
                    sn = p / (v * cs)
                A[:, j], A[:, k] = (aj * cs + ak * sn).astype(np.float32), (-aj * sn + ak * cs).astype(np.float32)
                done += 1
        applied.append(done)
        sweeps += 1
    return sweeps, applied


x = synth(8 * N, seed=0)
for f in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    s, a = jacobi(build(x[f * N:(f + 1) * N]))
    print("frame %d: %d sweeps, rotations applied per sweep %s, total %d of %d tested" % (f, s, a, sum(a), s * 528))
