#!/usr/bin/env python3
"""Index model of the 4096-point transform with the LAST pass done across the 16 lanes of a DPP
row instead of after a second LDS exchange (design check for spectro16 'D' form, N = 4096).

lanes t = 16b + c load x[t + 256a] into register a; pass 0 = DFT over a (registers); exchange 0
(LDS) hands lane p = 16*ka + i registers b; pass 1 = DFT over b (registers); pass 2 = DFT over c
ACROSS the 16 lanes of a row by four radix-2 DIF stages whose lane pairs are single DPP controls:
logical stride 8 -> row_ror:8 (xor 8), 4 -> row_half_mirror (xor 7), 2 -> quad_perm (xor 2),
1 -> quad_perm (xor 1).  Physical lane i holds logical index l with i = M(l)."""
import numpy as np

N = 4096
W = lambda n, k: np.exp(-2j * np.pi * (k % n) / n)


def phys(l):            # logical -> physical low 4 lane bits
    return (8 if l & 8 else 0) ^ (7 if l & 4 else 0) ^ (2 if l & 2 else 0) ^ (1 if l & 1 else 0)


LOG = [0] * 16          # physical -> logical
for l in range(16):
    LOG[phys(l)] = l
XOR = {8: 8, 4: 7, 2: 2, 1: 1}
brev4 = lambda v: int('{:04b}'.format(v)[::-1], 2)

rng = np.random.default_rng(0)
x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
# pass 0
z = np.array([[x[t + 256 * a] for a in range(16)] for t in range(256)])
F16 = np.array([[W(16, a * k) for a in range(16)] for k in range(16)])
Y0 = z @ F16.T                                            # Y0[t][ka]
# exchange 0 -> lane p = 16*ka + i, logical c = LOG[i], registers b
V = np.zeros((256, 16), complex)
for p in range(256):
    ka, c = p >> 4, LOG[p & 15]
    for b in range(16):
        V[p][b] = Y0[16 * b + c][ka] * W(256, ka * b)     # twiddle 1: table row ka
Y1 = V @ F16.T                                            # Y1[p][kb]
for p in range(256):
    ka, c = p >> 4, LOG[p & 15]
    for kb in range(16):
        Y1[p][kb] *= W(4096, c * (ka + 16 * kb))          # twiddle 2: per lane, per register
# pass 2 across lanes: radix-2 DIF, logical strides 8,4,2,1
cur = Y1.copy()
for h in (8, 4, 2, 1):
    nxt = np.zeros_like(cur)
    for p in range(256):
        i = p & 15
        l = LOG[i]
        partner = (p & ~15) | (i ^ XOR[h])
        assert LOG[partner & 15] == l ^ h
        upper = bool(l & h)
        s = -1.0 if upper else 1.0
        w = W(2 * h, l % h) if upper else 1.0
        nxt[p] = (s * cur[p] + cur[partner]) * w           # lower: mine+partner; upper: (partner-mine)*w
    cur = nxt
X = np.fft.fft(x)
err = 0.0
for p in range(256):
    ka, l = p >> 4, LOG[p & 15]
    for kb in range(16):
        K = ka + 16 * kb + 256 * brev4(l)
        err = max(err, abs(cur[p][kb] - X[K]))
print("max |model - fft| =", err, " (scale", abs(X).max(), ")")
assert err < 1e-9 * abs(X).max()
print("LOG (physical -> logical):", LOG)
