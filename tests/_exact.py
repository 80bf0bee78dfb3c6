"""Float64 'exact-arithmetic' rows for the parity bounds at large N (tests only).

At N >= 8192 the reference's float32 transform with its recurrence twiddles (fft_radix2.c:127-141) is
itself ~1e-5 (peak-normalised) away from exact arithmetic on noise-like frames, so parity is stated as
err(device, oracle) <= max(1e-5, 1.1 x err(oracle, exact)).  'exact' here = the same frames -- the samples
as floats, each hop's mean removed EXACTLY as fft.c:88-95 does it (a float sum sample after sample, a float
quotient, a float subtraction: that is input preparation, reproduced bit for bit) -- then window / tapers,
transform, |X|^2 / N and the taper sum in float64.
"""
import numpy as np


def hop_len(n, overlap):
    return int(n * (1.0 - float(np.float32(overlap))))                       # fft.c:70


def remove_hop_means(x, h):
    """fft.c:86-96 on every whole hop of a float32 stream (a copy)."""
    x = np.array(x, np.float32, copy=True)
    for j in range(len(x) // h):
        seg = x[j * h:(j + 1) * h]
        s = np.cumsum(seg, dtype=np.float32)[-1]                             # sequential float sum
        mean = np.float32(s) / np.float32(h)
        seg -= mean
    return x


def frames64(x, n, overlap, sub_mean=0, history_mode=0):
    """The assembled frames (fft.c:98-113) as float64 rows of the float32 samples."""
    h = hop_len(n, overlap)
    x = remove_hop_means(x, h) if sub_mean else np.asarray(x, np.float32)
    nfr = len(x) // h
    out = np.zeros((nfr, n))
    for f in range(nfr):
        lo = f * h - (n - h)
        if history_mode:
            out[f, n - h:] = x[f * h:(f + 1) * h]
        else:
            a = max(lo, 0)
            out[f, a - lo:] = x[a:f * h + h]
    return out


def periodogram64(x, n, overlap, window32, sub_mean=0, history_mode=0):
    fr = frames64(x, n, overlap, sub_mean, history_mode)
    return np.abs(np.fft.rfft(fr * np.asarray(window32, np.float64), axis=1)) ** 2 / n      # fft.c:203-226


def multitaper64(x, n, overlap, tapers, sig, sub_mean=0, history_mode=0):
    fr = frames64(x, n, overlap, sub_mean, history_mode)
    out = np.zeros((fr.shape[0], n // 2 + 1))
    for j in range(len(sig)):                                                                # mtm.c:189-220
        out += np.abs(np.fft.rfft(fr * tapers[j], axis=1)) ** 2 / n / (1.0 + sig[j])
    return out
