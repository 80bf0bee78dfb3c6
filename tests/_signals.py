"""Deterministic synthetic audio of SURVEY.md 8(d): two tones + Gaussian noise, clipped."""
import numpy as np


def synth(nsamples, fs=48000.0, seed=0, f0=1000.0, f1=7350.5):
    rng = np.random.default_rng(seed)
    n = np.arange(nsamples, dtype=np.float64)
    x = 0.5 * np.sin(2 * np.pi * f0 * n / fs) + 0.25 * np.sin(2 * np.pi * f1 * n / fs)
    x += 0.05 * rng.standard_normal(nsamples)
    return np.clip(x, -1.0, np.nextafter(1.0, 0.0)).astype(np.float32)


def rel_err(got, ref):
    """(max|d|/max|ref|, ||d||2/||ref||2) -- the norms the 1e-5 target is stated in."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    d = got - ref
    return np.abs(d).max() / np.abs(ref).max(), np.sqrt((d * d).sum() / (ref * ref).sum())


# BASELINE.json configs: name -> dict
CONFIGS = {
    "C1": dict(mode="fft", n=1024, overlap=0.5, window="hanning", fs=8000.0),
    "C2": dict(mode="fft", n=4096, overlap=0.75, window="hanning", fs=48000.0),
    "C3": dict(mode="mtm", n=4096, overlap=0.0, nw=2.5, kmax=4, fs=48000.0),
    "C3o": dict(mode="mtm", n=4096, overlap=0.75, nw=2.5, kmax=4, fs=48000.0),
    "C5": dict(mode="hparma", n=4096, overlap=0.0, t=128, p_e=32, fs=48000.0),
    "C4": dict(mode="mtm", n=16384, overlap=0.0, nw=4.5, kmax=8, fs=48000.0),
}
