"""How far the ORACLE itself moves when its input moves by one float ulp (tests only).

HP-ARMA's AR vector is a noise-subspace direction of a nearly rank-deficient matrix and the LMP statistic divides by a
small variance: where the reference's own result moves by s under such input noise, an implementation that does not
replay its every rounding cannot be held below ~s.  The parity bounds of those two estimators are therefore
max(1e-5, c x s) with s MEASURED IN THE TEST by these helpers, never a constant.
"""
import numpy as np

from _signals import rel_err


def ulp_perturbations(x, k, seed=0):
    """k copies of x with every sample moved by -1, 0 or +1 float32 ulp."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(k):
        step = (rng.integers(0, 3, x.size) - 1).astype(np.float32)
        out.append(np.nextafter(x, x + step).astype(np.float32))
    return out


def hparma_inv_rows(oracle, x, n, overlap, t, p_e, sub_mean=0):
    """|A(f)|^2 / N below Nyquist (= 1 / psd, hparma.c:150-154) per frame, float64."""
    return [1.0 / psd.astype(np.float64)[:n // 2] for psd, _, _ in oracle.hparma_frames(x, n, overlap, t, p_e, sub_mean=sub_mean)]


def hparma_spread(oracle, x, n, overlap, t, p_e, sub_mean=0, draws=12, seed=0):
    """Largest peak-normalised movement of the oracle's |A(f)|^2 / N over the stream's frames under `draws` 1-ulp
    perturbations of the input (a frame's own worst case is sampled poorly by a dozen draws: the stream's maximum is the
    estimate of s)."""
    ref = hparma_inv_rows(oracle, x, n, overlap, t, p_e, sub_mean)
    s = 0.0
    for xp in ulp_perturbations(x, draws, seed=seed):
        for f, v in enumerate(hparma_inv_rows(oracle, xp, n, overlap, t, p_e, sub_mean)):
            fin = np.isfinite(ref[f]) & np.isfinite(v)
            s = max(s, rel_err(v[fin], ref[f][fin])[0])
    return s, ref


def hparma_bound(oracle, x, n, overlap, t, p_e, sub_mean=0, draws=12, seed=0):
    """(bound, spread, oracle rows): 1e-5 flat at BASELINE config 5's shape (N = 4096, t = 128, p_e = 32: the device measures
    <= 4.5e-6 there, profiles/r04_hparma_schedule.txt); max(1e-5, 3 x spread) elsewhere.

    Why 3 and not 1.1 (round 5, gpurun_out/r5/t2.log): the spread is the MAXIMUM OF A SAMPLE (a dozen draws x the stream's frames),
    an estimate of the scale of the oracle's movement, not an upper bound of it -- at t = 96, p_e = 16 a frame of the device sits
    at 2.2e-5 where twelve draws of the oracle reached 1.2e-5 and other streams of the same shape reach 1.4e-5 with eight
    (profiles/r04_hparma_schedule.txt: oracle max 1.5e-5 over a longer run).  With 1.1 the test fails on one frame in ~30 for no
    defect of the device; 3 is the factor tests/test_gpu_round2.py has used for the LMP statistic since round 2."""
    s, ref = hparma_spread(oracle, x, n, overlap, t, p_e, sub_mean, draws, seed)
    if (n, t, p_e) == (4096, 128, 32):
        return 1e-5, s, ref
    return max(1e-5, 3.0 * s), s, ref
