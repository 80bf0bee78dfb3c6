"""CPU tests for the oracle rows added in round 2: the file source (wav_read with its trailing
partial block), the LMP estimator and the MTM harmonic F-test.

Pins: wav_read is checked BIT-EXACT against the reference's own wav_fmt.c object (oracle/_ref).
lmp.c and mtm.c need <gtk/gtk.h> (unbuildable here, no stand-in written), so the LMP statistic and
the F-test are pinned by independent float64 restatements of their formulas in numpy, by their
degenerate cases, and by the committed golden vectors (tests/golden/l_*.npz, f_*.npz)."""
import glob
import os
import struct

import numpy as np
import pytest

from _signals import rel_err, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (reference tree absent)")
    r = oracle.Ref()
    if not hasattr(r.r, "wav_read"):
        pytest.skip("oracle/_ref predates wav_fmt.c")
    return r


def _lp64_wav(path, pcm_bytes, bits, rate):
    """A header the reference's open_wav_file() parses correctly on THIS (LP64) host: its struct
    uses u_long, so it is 88 bytes with format at 40, sample_fq at 48 and bit_p_spl at 66
    (wav_fmt.h:34-52).  Only the reference reads this layout; the product and the
    oracle honour the canonical 44-byte one."""
    hd = bytearray(88)
    hd[0:4] = b"RIFF"
    struct.pack_into("<H", hd, 40, 1)          # format = PCM
    struct.pack_into("<H", hd, 42, 1)          # modus = mono
    struct.pack_into("<Q", hd, 48, rate)       # sample_fq
    struct.pack_into("<H", hd, 64, bits // 8)  # byte_p_spl
    struct.pack_into("<H", hd, 66, bits)       # bit_p_spl
    with open(path, "wb") as f:
        f.write(bytes(hd))
        f.write(pcm_bytes)


@pytest.mark.parametrize("bits,extra_bytes,hop", [(16, 2 * 300, 512), (16, 2 * 300 + 1, 512), (8, 77, 512),
                                                  (16, 0, 256), (8, 1, 102), (16, 2 * 5, 1024)])
def test_wav_read_bitexact_vs_reference(oracle, ref, tmp_path, bits, extra_bytes, hop):
    """Whole blocks, then a short last read: the new samples land on the stale tail of the previous
    block -- as the ESTIMATOR left it (here: mean removed in place, like fft.c:93-95) -- and the block
    still counts; an odd trailing byte of a 16-bit file is dropped; a file of whole blocks ends
    without an extra one."""
    rng = np.random.default_rng(bits + extra_bytes)
    nbytes = 7 * hop * (bits // 8) + extra_bytes
    pcm = rng.integers(0, 256, nbytes, dtype=np.uint8)
    path = tmp_path / "lp64.wav"
    _lp64_wav(path, pcm.tobytes(), bits, 8000)

    def remove_mean(blk):
        blk -= np.float32(blk.sum(dtype=np.float32) / np.float32(blk.size))

    for mutate in (None, remove_mean):
        want, speed = ref.wav_blocks(str(path), hop, mutate)
        got = oracle.wav_blocks(pcm, bits, hop, mutate)
        assert speed == 8000
        assert len(got) == len(want) == 7 + (1 if extra_bytes else 0)
        for a, b in zip(got, want):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    if extra_bytes:
        fresh = extra_bytes if bits == 8 else extra_bytes // 2
        last, prev = got[-1], got[-2]
        conv = oracle.pcm_u8_to_float if bits == 8 else oracle.pcm_s16_to_float
        tail_raw = pcm[7 * hop * (bits // 8):]
        tail = conv(tail_raw if bits == 8 else tail_raw[:2 * fresh].view(np.int16))
        assert np.array_equal(last[:fresh], tail)
        # stale part: the previous block after the estimator's mean removal
        stale = prev.copy()
        remove_mean(stale)
        assert np.array_equal(last[fresh:], stale[fresh:])


def test_wav_spectrogram_is_the_block_loop(oracle):
    """go_wav_spectrogram = wav_read blocks through fft_do/mtm_do on the reader's own buffer: the
    whole-block rows equal the plain stream driver's, the extra last row is the frame built from
    (fresh samples | stale tail)."""
    n, ovl = 1024, 0.5
    h = oracle.hop(n, ovl)
    x = synth(9 * h + 200, fs=8000.0, seed=4)
    raw = np.round(x * 30000).astype(np.int16)
    xf = oracle.pcm_s16_to_float(raw)
    for sub_mean in (0, 1):
        got = oracle.wav_spectrogram(raw, 16, "fft", n, ovl, 7, sub_mean=sub_mean)
        body = oracle.spectrogram_fft(xf, n, ovl, 7, sub_mean=sub_mean)
        assert got.shape[0] == body.shape[0] + 1 == 10
        assert np.array_equal(got[:9], body)
        # the last hop as the reference's buffer holds it, appended to the stream
        prev = xf[8 * h:9 * h].copy()
        if sub_mean:
            prev -= np.float32(prev.sum(dtype=np.float32) / np.float32(h))   # differs from fft.c's loop only in summation order
        last_hop = np.concatenate([xf[9 * h:], prev[200:]])
        if not sub_mean:
            ext = np.concatenate([xf[:9 * h], last_hop])
            assert np.array_equal(got[9], oracle.spectrogram_fft(ext, n, ovl, 7)[9])
    m = oracle.wav_spectrogram(raw, 16, "mtm", n, ovl, nw=2.5, kmax=4)
    assert np.array_equal(m[:9], oracle.spectrogram_mtm(xf, n, ovl, 2.5, 4))
    # a file shorter than one block: one frame, fresh samples over the zeroed buffer (calloc, wav_fmt.c:99)
    short = oracle.wav_spectrogram(raw[:100], 16, "fft", n, ovl, 0)
    pad = np.concatenate([xf[:100], np.zeros(h - 100, np.float32)])
    assert short.shape[0] == 1 and np.array_equal(short[0], oracle.spectrogram_fft(pad, n, ovl, 0)[0])


# ---- LMP -------------------------------------------------------------------------------------
def _lmp_numpy(P, nl):
    """lmp.c:132-160 in float64 numpy from the periodogram rows P[f]: ring slot j holds the latest
    frame g <= f with g % nl == j (zeros before any), mean and variance over the slots."""
    frames, nb = P.shape
    out = np.empty((frames, nb), np.float64)
    ring = np.zeros((nl, nb), np.float64)
    for f in range(frames):
        ring[f % nl] = P[f]
        my = ring.sum(axis=0) / nl
        sy = ((ring - my) ** 2).sum(axis=0) / (nl - 1)
        v = np.maximum(my * my - sy, 0.0)
        v = 0.5 * (my - np.sqrt(v))
        with np.errstate(divide="ignore", invalid="ignore"):
            o = -np.sqrt(nl / 2.0) + (nl * my) / (2.0 * np.sqrt(2.0 * nl) * v)
        o = np.where(o <= 1e-3, 1e-3, o)
        o[0] = 1e-3
        out[f] = o
    return out


@pytest.mark.parametrize("n,ovl,nl,sub_mean", [(1024, 0.0, 4, 0), (1024, 0.5, 4, 1), (512, 0.75, 7, 0), (2048, 0.0, 2, 0)])
def test_lmp_known_answer_numpy(oracle, n, ovl, nl, sub_mean):
    """The LMP statistic against a float64 numpy restatement fed with the ORACLE's own periodograms
    (rectangular window: lmp.c:114-116 overwrites prepare_audio's output with the raw frame)."""
    h = oracle.hop(n, ovl)
    x = synth(23 * h, fs=8000.0, seed=n + nl)
    got = oracle.spectrogram_lmp(x, n, ovl, nl, sub_mean=sub_mean).astype(np.float64)
    P = oracle.spectrogram_fft(x, n, ovl, oracle.WINDOWS["rectangular"], sub_mean=sub_mean)
    want = _lmp_numpy(P.astype(np.float64), nl)
    assert got.shape == want.shape
    assert np.all(got[:, 0] == np.float32(1e-3)) and got.min() >= np.float32(1e-3)
    # float32 output of the same double formula
    assert np.allclose(got, want, rtol=2e-6, atol=0)


def test_lmp_degenerate_cases(oracle):
    n, nl = 512, 4
    # silence: my = 0, v_hat = 0 -> 0/0 = NaN, and NaN <= 1e-3 is false: NaN stays (lmp.c:158-159)
    z = oracle.spectrogram_lmp(np.zeros(6 * n, np.float32), n, 0.0, nl)
    assert np.all(z[:, 0] == np.float32(1e-3)) and np.isnan(z[:, 1:]).all()
    # the same frame nl times: variance 0 -> v_hat = 0 -> +inf from the nl-th frame on
    x = np.tile(synth(n, seed=3), 8)
    r = oracle.spectrogram_lmp(x, n, 0.0, nl)
    assert np.isinf(r[nl - 1:, 1:]).all() and np.isfinite(r[:nl - 1, 1:]).all()
    # nl = 1: division by nl - 1 = 0 -> NaN everywhere but bin 0
    one = oracle.spectrogram_lmp(synth(4 * n, seed=5), n, 0.0, 1)
    assert np.isnan(one[:, 1:]).all() and np.all(one[:, 0] == np.float32(1e-3))


# ---- harmonic F-test --------------------------------------------------------------------------
def _ftest_numpy(frame, tapers, k):
    """Thomson's F statistic as mtm.c:165-233 computes it, in float64 with numpy's rfft."""
    U0 = tapers.sum(axis=1)
    s = (U0 * U0).sum()
    hn = (U0[:, None] * tapers).sum(axis=0) / s
    mu = np.fft.rfft(frame * hn)
    Y = np.fft.rfft(tapers * frame[None, :], axis=1)
    den = (np.abs(Y - mu[None, :] * U0[:, None]) ** 2).sum(axis=0)
    num = k * np.abs(mu) ** 2 * s
    return num, den, (np.abs(Y) ** 2).sum(axis=0)


@pytest.mark.parametrize("n,nw,kmax", [(1024, 2.5, 4), (512, 4.0, 7), (2048, 2.0, 2)])
def test_ftest_known_answer_numpy(oracle, n, nw, kmax):
    x = synth(6 * n, fs=8000.0, seed=n)
    psd, ft = oracle.spectrogram_mtm_ftest(x, n, 0.0, nw, kmax, mu_live=1)
    assert np.array_equal(psd, oracle.spectrogram_mtm(x, n, 0.0, nw, kmax))     # the PSD path is untouched
    tapers, _ = oracle.dpss(n, kmax, nw)
    for f in range(6):
        num, den, tot = _ftest_numpy(x[f * n:(f + 1) * n].astype(np.float64), tapers, kmax)
        want = num[:n // 2] / den[:n // 2]
        got = ft[f, :n // 2].astype(np.float64)
        # float32 spectra in the reference: compare where the statistic is well conditioned
        # (the residual |y_j - mu U0_j|^2 is not a small difference of large numbers)
        ok = den[:n // 2] > 1e-2 * tot[:n // 2]
        assert ok.sum() > n // 4
        assert np.allclose(got[ok], want[ok], rtol=2e-3), np.abs(got[ok] / want[ok] - 1).max()
        # Nyquist: its denominator is never accumulated (mtm.c:206 stops below n/2) -> x/0
        assert np.isinf(ft[f, n // 2]) or np.isnan(ft[f, n // 2])
        # the 1000 Hz line of the test signal is detected: F is huge there
        line = int(round(1000.0 / 8000.0 * n))
        assert ft[f, line - 1:line + 2].max() > 50 * np.median(got)


def test_ftest_reference_build_is_dead(oracle):
    """Without FFTW the reference never writes `mu` (mtm.c:173 transforms inbuf_fft in place):
    numerator 0 -> F = 0 wherever the denominator is not 0, NaN where it is."""
    n = 1024
    x = synth(4 * n, seed=9)
    psd, ft = oracle.spectrogram_mtm_ftest(x, n, 0.0, 2.5, 4, mu_live=0)
    assert np.all(ft[:, :n // 2] == 0.0) and np.isnan(ft[:, n // 2]).all()
    assert np.array_equal(psd, oracle.spectrogram_mtm(x, n, 0.0, 2.5, 4))


def test_ftest_tables(oracle):
    n, kmax = 1024, 4
    tapers, _ = oracle.dpss(n, kmax, 2.5)
    U0, hn, s = oracle.ftest_tables(n, kmax, tapers)
    assert np.allclose(U0, tapers.sum(axis=1), rtol=1e-12, atol=1e-12)     # summation order only
    # odd-order tapers are antisymmetric: their U0 (sum) vanishes against the even ones'
    assert abs(U0[1]) < 1e-6 * abs(U0[0]) and abs(U0[3]) < 1e-6 * abs(U0[0])
    assert np.isclose(s, (U0 * U0).sum(), rtol=1e-6)
    assert np.allclose(hn, (U0[:, None] * tapers).sum(axis=0) / (U0 * U0).sum(), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "[lfw]_*.npz"))), ids=os.path.basename)
def test_oracle_reproduces_round2_golden(oracle, path):
    g = np.load(path)
    kind = os.path.basename(path)[0]
    if kind == "l":
        got = oracle.spectrogram_lmp(g["x"], int(g["n"]), float(g["overlap"]), int(g["nl"]), sub_mean=int(g["sub_mean"]))
        assert np.array_equal(got.view(np.uint32), g["out"].view(np.uint32))
    elif kind == "f":
        psd, ft = oracle.spectrogram_mtm_ftest(g["x"], int(g["n"]), float(g["overlap"]), float(g["nw"]), int(g["kmax"]),
                                               mu_live=int(g["mu_live"]))
        assert np.array_equal(ft.view(np.uint32), g["ftest"].view(np.uint32))
        assert np.array_equal(psd.view(np.uint32), g["psd"].view(np.uint32))
    else:
        got = oracle.wav_spectrogram(g["pcm"], int(g["bits"]), str(g["mode"]), int(g["n"]), float(g["overlap"]),
                                     int(g["window"]), sub_mean=int(g["sub_mean"]), nw=float(g["nw"]), kmax=int(g["kmax"]))
        assert np.array_equal(got.view(np.uint32), g["psd"].view(np.uint32))
