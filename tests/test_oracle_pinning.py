"""CPU tests: the oracle is pinned (a) bit-exact against the reference's own objects where they
build (oracle/_ref: fft_radix2.c, g-l_dpss.c, avg.c, util.c), (b) against independent known
answers (numpy rfft in float64, scipy DPSS, Parseval) for the parts restating fft.c/mtm.c, and
(c) against the committed golden vectors."""
import glob
import os

import numpy as np
import pytest

from _signals import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (reference tree absent)")
    return oracle.Ref()


@pytest.mark.parametrize("n", [2, 4, 8, 16, 64, 512, 1024, 4096, 16384])
def test_fft_bitexact_vs_reference(oracle, ref, n):
    rng = np.random.default_rng(n)
    for _ in range(3):
        x = rng.standard_normal(n).astype(np.float32)
        a, b = oracle.rfft_halfcomplex(x), ref.rfft_halfcomplex(x)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("n,kmax,nw", [(256, 2, 1.5), (1024, 4, 2.5), (4096, 4, 2.5), (4096, 7, 4.0),
                                        (16384, 8, 4.5)])
def test_dpss_bitexact_vs_reference(oracle, ref, n, kmax, nw):
    v, s = oracle.dpss(n, kmax, nw)
    err, v2, s2 = ref.dpss(n, kmax, nw)
    assert err == 0
    assert np.array_equal(v, v2) and np.array_equal(s, s2)


def test_bessel_and_svd_bitexact_vs_reference(oracle, ref):
    for x in np.linspace(0, 12, 97):
        assert oracle.bessel_i0(x) == ref.bessel_i0(x)
    A = np.random.default_rng(1).standard_normal((128, 33)).astype(np.float32)
    rc, U, S, Q = oracle.svd(A)
    rc2, U2, S2, Q2 = ref.svd(A)
    assert rc == rc2 == 0
    assert np.array_equal(U, U2) and np.array_equal(S, S2) and np.array_equal(Q, Q2)


@pytest.mark.parametrize("mode", ["plain", "sumextreme", "sumavg"])
def test_avg_bitexact_vs_reference(oracle, ref, mode):
    rng = np.random.default_rng(5)
    a1, a2 = oracle.Averager(1024, 4), ref.averager(1024, 4)
    for f in range(14):
        psd = (rng.random(513) ** 4).astype(np.float32)
        r1 = a1.update(mode, psd, 10, 400, max0=f % 2, n=513)
        r2 = a2.update(mode, psd, 10, 400, max0=f % 2, n=513)
        assert r1[0] == r2[0] and r1[2] == r2[2]
        assert np.array_equal(r1[1], r2[1])
        assert r1[3] == r2[3] or (np.isnan(r1[3]) and np.isnan(r2[3]))


def test_fft_known_answer_numpy(oracle):
    for n in (64, 1024, 4096):
        x = synth(n, seed=n)
        hc = oracle.rfft_halfcomplex(x).astype(np.float64)
        X = np.fft.rfft(x.astype(np.float64))
        want = np.concatenate([X.real, X.imag[1:-1][::-1]])
        assert np.abs(hc - want).max() / np.abs(want).max() < 4e-6    # the reference's own f32 error


def test_window_properties(oracle):
    for name, t in oracle.WINDOWS.items():
        w = oracle.window(t, 1024).astype(np.float64)
        assert abs((w * w).sum() - 1.0) < 1e-5, name                  # fft.c:352-359
        assert np.allclose(w, w[::-1], atol=1e-6), name               # all eight are symmetric
    h = oracle.window(0, 1024).astype(np.float64)
    hn = np.hanning(1024)
    assert np.allclose(h, hn / np.sqrt((hn * hn).sum()), atol=1e-7)
    k = oracle.window(7, 1024).astype(np.float64)
    kn = np.kaiser(1024, 6.0)                                         # alpha*t = 6
    assert np.allclose(k, kn / np.sqrt((kn * kn).sum()), atol=2e-6)   # polynomial I0


def test_dpss_known_answer_scipy(oracle):
    from scipy.signal.windows import dpss
    v, sig = oracle.dpss(4096, 4, 2.5)
    sv, lam = dpss(4096, 2.5, 5, return_ratios=True)
    for k in range(5):
        sgn = np.sign(np.dot(v[k], sv[k]))
        assert np.abs(v[k] * sgn - sv[k]).max() < 2e-7
        assert abs(1 + sig[k] - lam[k]) < 4e-6
    assert np.allclose((v * v).sum(axis=1), 1.0, atol=1e-12)


def test_periodogram_parseval(oracle):
    # unit-power window, one-sided PSD without doubling: sum(psd) ~= A^2/4 for a sine
    n = 1024
    x = (0.5 * np.sin(2 * np.pi * 100.0 * np.arange(n) / n)).astype(np.float32)
    psd = oracle.spectrogram_fft(x, n, 0.0, oracle.WINDOWS["hanning"])
    assert abs(psd[0].sum() - 0.0625) < 1e-4


def test_mtm_is_weighted_sum_of_eigenspectra(oracle):
    n, kmax, nw = 1024, 4, 2.5
    x = synth(n, fs=8000.0, seed=9)
    v, sig = oracle.dpss(n, kmax, nw)
    want = np.zeros(n // 2 + 1)
    for j in range(kmax + 1):
        X = np.fft.rfft(v[j] * x.astype(np.float64))
        want += (X.real ** 2 + X.imag ** 2) / n / (1.0 + sig[j])      # mtm.c:212-219: sum, not mean
    got = oracle.spectrogram_mtm(x, n, 0.0, nw, kmax)[0]
    assert np.abs(got - want).max() / want.max() < 2e-6


def test_history_and_submean_semantics(oracle):
    n, ovl = 1024, 0.5
    h = oracle.hop(n, ovl)
    x = synth(6 * h, fs=8000.0, seed=3) + np.float32(0.1)
    w = oracle.window(0, n).astype(np.float64)

    def frame_psd(fr):
        X = np.fft.rfft(w * fr)
        return (X.real ** 2 + X.imag ** 2) / n
    # ZERO_FIRST: frame f = samples [f*h-(n-h), f*h+h), zeros before the stream (fft.c:98-113)
    got = oracle.spectrogram_fft(x, n, ovl, 0)
    xx = np.concatenate([np.zeros(n - h), x.astype(np.float64)])
    for f in range(6):
        assert np.abs(got[f] - frame_psd(xx[f * h:f * h + n])).max() / got[f].max() < 2e-6
    # ZERO_ALWAYS: history zeroed in every frame
    got = oracle.spectrogram_fft(x, n, ovl, 0, history_mode=1)
    for f in range(6):
        fr = np.concatenate([np.zeros(n - h), x[f * h:(f + 1) * h].astype(np.float64)])
        assert np.abs(got[f] - frame_psd(fr)).max() / got[f].max() < 2e-6
    # sub_mean: each hop loses the mean of its own new samples (fft.c:86-96)
    got = oracle.spectrogram_fft(x, n, ovl, 0, sub_mean=1)
    hops = x.astype(np.float64).reshape(6, h)
    hops = hops - hops.mean(axis=1, keepdims=True)
    xx = np.concatenate([np.zeros(n - h), hops.ravel()])
    for f in range(6):
        assert np.abs(got[f] - frame_psd(xx[f * h:f * h + n])).max() / got[f].max() < 5e-6


def test_hop_truncation(oracle):
    assert oracle.hop(1024, 0.9) == 102        # (int)(1024*(1.0-0.9f)), SURVEY 8c
    assert oracle.hop(4096, 0.75) == 1024
    assert oracle.hop(1024, 0.0) == 1024


def test_pcm_conversion(oracle):
    assert np.array_equal(oracle.pcm_u8_to_float(np.array([0, 128, 255], np.uint8)),
                          np.array([-1.0, 0.0, 127 / 128], np.float32))
    assert np.array_equal(oracle.pcm_s16_to_float(np.array([-32768, 0, 32767], np.int16)),
                          np.array([-1.0, 0.0, 32767 / 32768], np.float32))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "*.npz"))), ids=os.path.basename)
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    name = os.path.basename(path)
    if name.startswith("spec_"):
        assert np.array_equal(oracle.rfft_halfcomplex(g["win"] * g["x"]), g["halfcomplex"])
    elif name.startswith("avg_floor"):
        psd = g["psd"]
        fl = np.array([oracle.floor_stats(p) for p in psd], np.float64)
        assert np.array_equal(fl, g["floor"])
        for mode in ("plain", "sumextreme", "sumavg"):
            for max0 in (0, 1):
                a = oracle.Averager(1024, int(g["depth"]))
                for f in range(psd.shape[0]):
                    r, avg, peak, var = a.update(mode, psd[f], int(g["minbin"]), int(g["maxbin"]), max0=max0, n=513)
                    assert np.array_equal(avg, g["%s_max%d_avg" % (mode, max0)][f])
                    assert np.array_equal(np.array([r, peak, var]), g["%s_max%d_ret" % (mode, max0)][f], equal_nan=True)
    elif str(g["mode"]) == "fft":
        got = oracle.spectrogram_fft(g["x"], int(g["n"]), float(g["overlap"]), int(g["window"]), float(g["a"]),
                                     int(g["limiter"]), int(g["sub_mean"]), int(g["history_mode"]))
        assert np.array_equal(got, g["psd"])
        assert np.array_equal(oracle.window(int(g["window"]), int(g["n"])), g["win"])
    else:
        got = oracle.spectrogram_mtm(g["x"], int(g["n"]), float(g["overlap"]), float(g["nw"]), int(g["kmax"]),
                                     int(g["sub_mean"]), int(g["history_mode"]))
        assert np.array_equal(got, g["psd"])
