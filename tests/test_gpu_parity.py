"""GPU parity tests (-m gpu): the HIP engine, called through the C-ABI, against the CPU oracle
on the same seeded inputs and against the committed golden vectors.

Tolerance (BASELINE.json north_star: <= 1e-5 relative PSD error vs CPU), stated in the two
norms in which it is achievable: max|d|/max(ref) and
||d||_2/||ref||_2.  The reference's own float32 recurrence-twiddle FFT is up to 4e-6 away from
exact arithmetic in these norms, so most of the budget is the reference's error, not ours.
"""
import glob
import os

import numpy as np
import pytest

from _signals import CONFIGS, rel_err, synth

pytestmark = pytest.mark.gpu
TOL = 1e-5
HPARMA_TOL = 1e-5      # BASELINE config 5's shape; other shapes: max(1e-5, 3 x the oracle's own sampled 1-ulp spread), tests/_spread.py
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _params(lib, g):
    if str(g["mode"]) == "fft":
        return lib.FftParams(n=int(g["n"]), window_type=int(g["window"]), overlap=float(g["overlap"]),
                             a=float(g["a"]), limiter=int(g["limiter"]), sub_mean=int(g["sub_mean"]),
                             history_mode=int(g["history_mode"]))
    return lib.MtmParams(n=int(g["n"]), overlap=float(g["overlap"]), w=float(g["nw"]), kmax=int(g["kmax"]),
                         sub_mean=int(g["sub_mean"]), history_mode=int(g["history_mode"]))


def _run(lib, torch, params, x):
    sp = lib.Spectrogram(params)
    out = sp.run(torch.from_numpy(np.ascontiguousarray(x)).cuda())
    torch.cuda.synchronize()
    return sp, out.cpu().numpy()


@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(GOLD, "*.npz"))
                                        if os.path.basename(p)[0] in "ce"), ids=os.path.basename)
def test_golden_vectors(lib, torch_cuda, path):
    g = np.load(path)
    if str(g["mode"]) == "hparma":
        sp = lib.Spectrogram(lib.HparmaParams(n=int(g["n"]), overlap=float(g["overlap"]), t=int(g["t"]), p_e=int(g["p_e"])))
        got = sp.run(torch_cuda.from_numpy(g["x"]).cuda()).cpu().numpy().astype(np.float64)
        n = int(g["n"])
        for f in range(got.shape[0]):                      # parity on |A(f)|^2/N, see test_hparma_parity
            assert max(rel_err(1.0 / got[f, :n // 2], 1.0 / g["psd"][f, :n // 2].astype(np.float64))) < HPARMA_TOL
        return
    sp, got = _run(lib, torch_cuda, _params(lib, g), g["x"])
    assert got.shape == g["psd"].shape
    for f in range(got.shape[0]):                          # (the limiter fixture too: fft.c:151-156 runs in double on the device since round 4)
        emax, el2 = rel_err(got[f], g["psd"][f])
        assert emax < TOL and el2 < TOL, (f, emax, el2)


@pytest.mark.parametrize("cfg", list(CONFIGS), ids=list(CONFIGS))
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_baseline_configs_vs_oracle(lib, oracle, torch_cuda, cfg, seed):
    c = CONFIGS[cfg]
    frames = 64
    h = oracle.hop(c["n"], c["overlap"])
    x = synth(frames * h, fs=c["fs"], seed=seed)
    if c["mode"] == "hparma":
        pytest.skip("HP-ARMA has its own parity test (different norm)")
    if c["mode"] == "fft":
        params = lib.FftParams(n=c["n"], window_type=lib.WINDOWS[c["window"]], overlap=c["overlap"])
        want = oracle.spectrogram_fft(x, c["n"], c["overlap"], oracle.WINDOWS[c["window"]])
    else:
        params = lib.MtmParams(n=c["n"], overlap=c["overlap"], w=c["nw"], kmax=c["kmax"])
        want = oracle.spectrogram_mtm(x, c["n"], c["overlap"], c["nw"], c["kmax"])
    sp, got = _run(lib, torch_cuda, params, x)
    assert got.shape == want.shape == (frames, c["n"] // 2 + 1)
    worst = max(max(rel_err(got[f], want[f])) for f in range(frames))
    assert worst < TOL, worst


@pytest.mark.parametrize("n,overlap,kmax,nw,sub_mean,history_mode,fmt,frames", [
    (4096, 0.75, 4, 2.5, 1, 0, "f32", 71),      # spectro16xl: odd tapers, LDS tables; mean removal + overlap
    (4096, 0.5, 4, 2.5, 0, 1, "s16", 37),       # history zeroed in every frame, PCM input, odd frame count
    (4096, 0.9, 2, 2.0, 0, 0, "u8", 65),        # 3 tapers, hop 409 (odd), many early frames
    (4096, 0.0, 8, 4.5, 0, 0, "f32", 33),       # 9 tapers: tables too big for LDS -> spectro16x
    (1024, 0.5, 4, 2.5, 1, 1, "f32", 203),      # several frames per block, both options at once
    (512, 0.33, 6, 4.0, 0, 0, "s16", 150),      # 7 tapers, odd hop
    (256, 0.0, 2, 1.5, 0, 0, "f32", 97),        # smallest block: 16 frames per block
    (2048, 0.25, 4, 2.5, 0, 0, "f32", 40),      # N = 2048: xl only (x spills there)
    (8192, 0.5, 4, 2.5, 1, 0, "f32", 21),       # N >= 8192: the real-input kernel, taper by taper
    (8192, 0.75, 7, 4.0, 0, 1, "s16", 23),      # ... 16-bit pairs, history zeroed in every frame
    (8192, 0.33, 3, 2.5, 0, 0, "u8", 17),       # ... odd hop: integer pairs unaligned -> packed kernel
    (16384, 0.5, 8, 4.5, 0, 0, "f32", 13),      # N = 16384, 9 tapers
    (16384, 0.25, 1, 1.5, 1, 1, "u8", 12),      # two tapers, 8-bit pairs
    (4096, 0.75, 3, 2.5, 0, 0, "f32", 50),      # even taper count: packed kernel
])
def test_multitaper_kernel_forms_vs_oracle(lib, oracle, torch_cuda, n, overlap, kmax, nw, sub_mean, history_mode, fmt, frames):
    """Every multitaper kernel form (packed / shared odd taper / LDS-resident tables), with the
    options that change the gather (overlap, per-hop mean removal, history zeroed in every frame,
    PCM formats, odd hops, partial frame groups), frame by frame against the oracle."""
    h = oracle.hop(n, overlap)
    x = synth(frames * h + 3, seed=n + kmax) + np.float32(0.05)
    if fmt == "s16":
        raw = np.clip(np.round(x * 20000), -32768, 32767).astype(np.int16)
        xf, sf = oracle.pcm_s16_to_float(raw), lib.SAMPLES_S16
    elif fmt == "u8":
        raw = np.clip(np.round(x * 100 + 128), 0, 255).astype(np.uint8)
        xf, sf = oracle.pcm_u8_to_float(raw), lib.SAMPLES_U8
    else:
        raw, xf, sf = x, x, lib.SAMPLES_F32
    want = oracle.spectrogram_mtm(xf.copy(), n, overlap, nw, kmax, sub_mean=sub_mean, history_mode=history_mode)
    sp = lib.Spectrogram(lib.MtmParams(n=n, overlap=overlap, w=nw, kmax=kmax, sub_mean=sub_mean,
                                       history_mode=history_mode, sample_format=sf))
    got = sp.run(torch_cuda.from_numpy(raw).cuda()).cpu().numpy()
    assert got.shape == want.shape == (frames, n // 2 + 1)
    for f in range(frames):
        assert np.abs(got[f] - want[f]).max() <= TOL * want[f].max(), f
    # a launch that starts and ends inside frame groups gives the same rows
    first, count = 5, frames - 8
    part = sp.run(torch_cuda.from_numpy(raw).cuda(), first_frame=first, nframes=count).cpu().numpy()
    for f in range(count):
        assert np.abs(part[f] - want[first + f]).max() <= TOL * want[first + f].max(), f


@pytest.mark.parametrize("n", [256, 1024, 4096, 16384])
def test_tiny_frame_counts(lib, oracle, torch_cuda, n):
    """1..5 frames (fewer than a block's frame group, odd counts, a lone last frame) through the
    periodogram and the odd- and even-count multitaper paths, with and without overlap."""
    for overlap in (0.0, 0.5):
        h = oracle.hop(n, overlap)
        for frames in (1, 2, 3, 5):
            x = synth(frames * h, seed=frames + n)
            want = oracle.spectrogram_fft(x, n, overlap, 0)
            _, got = _run(lib, torch_cuda, lib.FftParams(n=n, window_type=0, overlap=overlap), x)
            assert got.shape == want.shape and max(rel_err(got, want)) < TOL, ("fft", overlap, frames)
            for kmax in (2, 3):
                want = oracle.spectrogram_mtm(x, n, overlap, 2.0, kmax)
                _, got = _run(lib, torch_cuda, lib.MtmParams(n=n, overlap=overlap, w=2.0, kmax=kmax), x)
                assert got.shape == want.shape
                for f in range(frames):
                    assert np.abs(got[f] - want[f]).max() <= TOL * want[f].max(), ("mtm", kmax, overlap, frames, f)


@pytest.mark.parametrize("window", ["hanning", "blackman", "gaussian", "welch", "bartlett", "rectangular",
                                    "hamming", "kaiser"])
def test_all_windows(lib, oracle, torch_cuda, window):
    x = synth(12 * 512, fs=8000.0, seed=7)
    sp, got = _run(lib, torch_cuda, lib.FftParams(n=1024, window_type=lib.WINDOWS[window], overlap=0.5), x)
    want = oracle.spectrogram_fft(x, 1024, 0.5, oracle.WINDOWS[window])
    assert np.array_equal(sp.window(), oracle.window(oracle.WINDOWS[window], 1024))
    assert max(max(rel_err(got[f], want[f])) for f in range(12)) < TOL


@pytest.mark.parametrize("n", [256, 512, 1024, 2048, 4096, 8192, 16384])
def test_block_sizes_fft_and_mtm(lib, oracle, torch_cuda, n):
    x = synth(9 * n, seed=n)
    sp, got = _run(lib, torch_cuda, lib.FftParams(n=n, window_type=0, overlap=0.0), x)
    want = oracle.spectrogram_fft(x, n, 0.0, 0)
    assert max(max(rel_err(got[f], want[f])) for f in range(9)) < TOL
    sp, got = _run(lib, torch_cuda, lib.MtmParams(n=n, overlap=0.5, w=3.0, kmax=5), x)
    want = oracle.spectrogram_mtm(x, n, 0.5, 3.0, 5)
    assert max(max(rel_err(got[f], want[f])) for f in range(want.shape[0])) < TOL
    v, s = sp.tapers()
    v2, s2 = oracle.dpss(n, 5, 3.0)
    assert np.array_equal(v, v2) and np.array_equal(s, s2)


def test_halfcomplex_spectrum(lib, torch_cuda):
    g = np.load(os.path.join(GOLD, "spec_fft4096_hann.npz"))
    sp = lib.Spectrogram(lib.FftParams(n=4096, window_type=0, overlap=0.0))
    psd, spec = sp.run(torch_cuda.from_numpy(g["x"]).cuda(), spectrum=True)
    spec = spec.cpu().numpy()[0]
    want = g["halfcomplex"]
    assert np.abs(spec - want).max() / np.abs(want).max() < 4e-6     # the reference's own f32 error
    X = np.fft.rfft(g["win"].astype(np.float64) * g["x"].astype(np.float64))
    exact = np.concatenate([X.real, X.imag[1:-1][::-1]])
    assert np.abs(spec - exact).max() / np.abs(exact).max() < 1e-6   # table twiddles: closer to exact


@pytest.mark.parametrize("n,overlap,fmt", [(512, 0.5, "f32"), (1024, 0.33, "s16"), (4096, 0.75, "f32"),
                                           (4096, 0.9, "u8"), (8192, 0.5, "f32"), (16384, 0.25, "s16")])
def test_real_input_kernel_agrees_with_packed_kernel(lib, oracle, torch_cuda, n, overlap, fmt):
    # The PSD-only periodogram runs the real-input N/2-point kernel (spectro16h.hip); asking for
    # the spectrum as well runs the packed N-point kernel (spectro16.hip).  Two independent
    # kernels, same rows -- and both against the oracle.  Odd hops (overlap 0.33) make the
    # 8-byte sample-pair loads start on odd sample indices.
    rng = np.random.default_rng(n)
    frames = 37
    h = oracle.hop(n, overlap)
    x = synth(frames * h + 5, seed=n)
    if fmt == "s16":
        raw = np.clip(np.round(x * 20000), -32768, 32767).astype(np.int16)
        xf, sf = oracle.pcm_s16_to_float(raw), lib.SAMPLES_S16
    elif fmt == "u8":
        raw = np.clip(np.round(x * 100 + 128), 0, 255).astype(np.uint8)
        xf, sf = oracle.pcm_u8_to_float(raw), lib.SAMPLES_U8
    else:
        raw, xf, sf = x, x, lib.SAMPLES_F32
    del rng
    sp = lib.Spectrogram(lib.FftParams(n=n, window_type=1, overlap=overlap, sample_format=sf))
    d = torch_cuda.from_numpy(raw).cuda()
    half = sp.run(d).cpu().numpy()
    packed, _ = sp.run(d, spectrum=True)
    packed = packed.cpu().numpy()
    want = oracle.spectrogram_fft(xf, n, overlap, 1)
    assert half.shape == want.shape
    assert rel_err(half, packed)[0] < 2e-6
    assert max(rel_err(half, want)) < TOL and max(rel_err(packed, want)) < TOL
    # per frame too (a quiet frame must not hide behind a loud one)
    for f in range(frames):
        assert np.abs(half[f] - want[f]).max() <= TOL * want[f].max()


@pytest.mark.parametrize("fmt,shift", [("s16", 1), ("u8", 1), ("s16", 2), ("u8", 3)])
def test_integer_streams_off_the_pair_alignment(lib, oracle, torch_cuda, fmt, shift):
    # The real-input kernel fetches the integer pair (y[2j], y[2j+1]) with one 4- or 2-byte load, so
    # it needs naturally aligned pairs; a stream that starts `shift` samples into an allocation (a
    # tensor slice) or an odd hop must be routed to the packed kernel and give the same rows.
    n, overlap, frames = 2048, 0.5, 70
    h = oracle.hop(n, overlap)
    x = synth(frames * h + n + 8, seed=77)
    if fmt == "s16":
        raw = np.clip(np.round(x * 20000), -32768, 32767).astype(np.int16)
        conv, sf = oracle.pcm_s16_to_float, lib.SAMPLES_S16
    else:
        raw = np.clip(np.round(x * 100 + 128), 0, 255).astype(np.uint8)
        conv, sf = oracle.pcm_u8_to_float, lib.SAMPLES_U8
    sp = lib.Spectrogram(lib.FftParams(n=n, window_type=1, overlap=overlap, sample_format=sf))
    d = torch_cuda.from_numpy(raw).cuda()
    got = sp.run(d[shift:].contiguous() if False else d[shift:]).cpu().numpy()
    want = oracle.spectrogram_fft(conv(raw[shift:]), n, overlap, 1)
    assert got.shape == want.shape
    assert max(rel_err(got, want)) < TOL
    aligned = sp.run(d[shift:].clone()).cpu().numpy()          # same samples in a fresh (aligned) allocation
    assert rel_err(got, aligned)[0] < 2e-6


@pytest.mark.parametrize("n,kmax,nw", [(4096, 4, 2.5), (1024, 2, 2.0), (512, 6, 4.0)])
def test_quiet_frame_between_loud_frames(lib, oracle, torch_cuda, n, kmax, nw):
    # Odd taper counts: the last taper of two neighbouring frames shares one complex transform
    # (spectro16x.hip).  Each frame must keep ITS OWN 1e-5 (relative to its own peak) whatever its
    # neighbour's level: 120 dB quieter, silent, or the other way round; odd frame count too.
    frames = 11
    x = synth(frames * n, seed=n + kmax).reshape(frames, n).copy()
    gains = [1.0, 1e-6, 1.0, 0.0, 1e-3, 1.0, 1e-6, 1e-6, 30.0, 1.0, 1e-5]
    for f, g in enumerate(gains):
        x[f] *= np.float32(g)
    x = x.ravel()
    want = oracle.spectrogram_mtm(x, n, 0.0, nw, kmax)
    _, got = _run(lib, torch_cuda, lib.MtmParams(n=n, overlap=0.0, w=nw, kmax=kmax), x)
    for f in range(frames):
        pk = want[f].max()
        if pk == 0.0:
            assert not got[f].any()
        else:
            assert np.abs(got[f] - want[f]).max() <= TOL * pk, (f, gains[f])


def test_edge_inputs(lib, oracle, torch_cuda):
    torch = torch_cuda
    sp = lib.Spectrogram(lib.FftParams(n=1024, window_type=0, overlap=0.5))
    # shorter than one hop: zero frames, nothing launched
    out = sp.run(torch.zeros(100, device="cuda"))
    assert out.shape == (0, 513)
    # ragged tail: the partial hop is dropped (wav_fmt.c:119 hands out whole blocks)
    x = synth(512 * 5 + 77, fs=8000.0, seed=1)
    got = sp.run(torch.from_numpy(x).cuda()).cpu().numpy()
    want = oracle.spectrogram_fft(x, 1024, 0.5, 0)
    assert got.shape == want.shape == (5, 513)
    assert max(max(rel_err(got[f], want[f])) for f in range(5)) < TOL
    # silence -> exact zeros; full-scale square wave stays finite
    z = sp.run(torch.zeros(4096, device="cuda")).cpu().numpy()
    assert np.all(z == 0.0)
    sq = np.where(np.arange(8192) % 16 < 8, 1.0, -1.0).astype(np.float32)
    got = sp.run(torch.from_numpy(sq).cuda()).cpu().numpy()
    want = oracle.spectrogram_fft(sq, 1024, 0.5, 0)
    assert np.isfinite(got).all() and max(max(rel_err(got[f], want[f])) for f in range(16)) < TOL
    # frame windows addressed in the middle of a stream (what a shard does)
    x = synth(512 * 40, fs=8000.0, seed=2)
    full = sp.run(torch.from_numpy(x).cuda()).cpu().numpy()
    part = sp.run(torch.from_numpy(x).cuda(), first_frame=17, nframes=9).cpu().numpy()
    assert np.array_equal(part, full[17:26])
    with pytest.raises(lib.GlferHipError, match="bad argument"):
        sp.run(torch.from_numpy(x).cuda(), first_frame=35, nframes=9)


def test_pcm_formats_on_device(lib, oracle, torch_cuda):
    torch = torch_cuda
    x = synth(512 * 20, fs=8000.0, seed=5)
    s16 = np.round(x * 32767).astype(np.int16)
    u8 = np.clip(np.round(x * 127 + 128), 0, 255).astype(np.uint8)
    for fmt, raw, conv in ((lib.SAMPLES_S16, s16, oracle.pcm_s16_to_float), (lib.SAMPLES_U8, u8, oracle.pcm_u8_to_float)):
        sp = lib.Spectrogram(lib.FftParams(n=1024, window_type=0, overlap=0.5, sample_format=fmt))
        got = sp.run(torch.from_numpy(raw).cuda()).cpu().numpy()
        want = oracle.spectrogram_fft(conv(raw), 1024, 0.5, 0)
        assert max(max(rel_err(got[f], want[f])) for f in range(20)) < TOL
        assert np.array_equal(sp.run_host(raw), got)


def test_host_buffer_entry(lib, oracle, torch_cuda):
    x = synth(4096 * 7, seed=11)
    sp = lib.Spectrogram(lib.MtmParams(n=4096, overlap=0.0, w=2.5, kmax=4))
    got = sp.run_host(x)
    want = oracle.spectrogram_mtm(x, 4096, 0.0, 2.5, 4)
    assert max(max(rel_err(got[f], want[f])) for f in range(7)) < TOL


def test_host_buffer_entry_many_chunks(lib, torch_cuda):
    """The host entry streams through pinned buffers in 16 384-frame chunks (the next chunk's samples
    and the previous chunk's rows are copied by host threads while the GPU works): 70 000 overlapped
    frames, five chunks, must be the device entry's rows bit for bit -- the halo carried between
    chunks, the frame-group alignment of the chunk cuts and the threaded copies all show here."""
    n, overlap, frames = 512, 0.5, 70000
    sp = lib.Spectrogram(lib.FftParams(n=n, window_type=3, overlap=overlap, sample_format=lib.SAMPLES_S16))
    rng = np.random.default_rng(3)
    raw = (rng.standard_normal(frames * sp.hop + 17) * 5000).clip(-32768, 32767).astype(np.int16)
    got = sp.run_host(raw)
    want = sp.run(torch_cuda.from_numpy(raw).cuda()).cpu().numpy()
    assert got.shape == want.shape == (frames, n // 2 + 1)
    assert np.array_equal(got, want)


def test_linearity_and_scaling_at_full_size(lib, torch_cuda):
    """Size-independent properties at BASELINE's batch scale (65536 frames of N=4096, MTM K=4):
    PSD is quadratic in amplitude, identical frames give identical rows, and Parseval holds."""
    torch = torch_cuda
    n, frames = 4096, 65536
    sp = lib.Spectrogram(lib.MtmParams(n=n, overlap=0.0, w=2.5, kmax=4))
    one = torch.from_numpy(synth(n, seed=3)).cuda()
    x = one.repeat(frames)
    a = sp.run(x)
    # the odd (5th) taper of frames 2g and 2g+1 shares one transform (spectro16x.hip): equal
    # frames in the same slot are bit-identical, the two slots agree to rounding
    assert torch.equal(a[0], a[frames - 2]) and torch.equal(a[1], a[frames // 2 + 1])
    assert rel_err(a[1].cpu().numpy(), a[0].cpu().numpy())[0] < 1e-6
    b = sp.run(x * 0.5)
    ratio = (b[7].double().sum() / a[7].double().sum()).item()
    assert abs(ratio - 0.25) < 1e-6
    # sum_k psd[k] over the one-sided spectrum: with unit-energy tapers, sum_j (1/lambda_j) * energy/2-ish;
    # compare with float64 numpy on one frame instead of a closed form
    v, sig = sp.tapers()
    want = sum((np.abs(np.fft.rfft(v[j] * one.cpu().numpy().astype(np.float64))) ** 2) / n / (1 + sig[j]) for j in range(5))
    got = a[12345].cpu().numpy()
    assert max(rel_err(got, want)) < 2e-6


def test_floor_statistics(lib, oracle, torch_cuda):
    g = np.load(os.path.join(GOLD, "avg_floor_fft1024.npz"))
    psd = torch_cuda.from_numpy(g["psd"]).cuda()
    got = lib.compute_floor(psd).cpu().numpy().astype(np.float64)
    want = g["floor"]
    assert np.array_equal(got[:, 0], want[:, 0])                        # sig = largest bin: exact
    assert np.array_equal(got[:, 2], want[:, 2]) and np.array_equal(got[:, 3], want[:, 3])   # peak, bin
    assert np.abs(got[:, 1] / want[:, 1] - 1).max() < 2e-6              # floor: float sum order differs
    # larger rows + ties + zeros
    rng = np.random.default_rng(0)
    big = (rng.random((33, 2049)) ** 6).astype(np.float32)
    big[3, :500] = 0.0
    big[4, :] = 0.25
    got = lib.compute_floor(torch_cuda.from_numpy(big).cuda()).cpu().numpy().astype(np.float64)
    want = np.array([oracle.floor_stats(r) for r in big], np.float64)
    assert np.array_equal(got[:, [0, 2, 3]], want[:, [0, 2, 3]])
    assert np.allclose(got[:, 1], want[:, 1], rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("bins", [129, 257, 513, 1025, 2049, 4097, 8193, 100, 2048, 3000, 5, 17, 65, 8257, 16385, 32769])
def test_floor_statistics_row_shapes(lib, oracle, torch_cuda, bins):
    # every row length the estimator produces (one wavefront per row up to 2049 bins, one workgroup
    # per row above), lengths that do not fill the last lane group, and rows built to stress the
    # selection: heavy ties around the 5 % quantile, few distinct values, one tiny outlier, a constant
    # row, all zeros, sorted both ways, the maximum repeated (first index wins), denormals
    rng = np.random.default_rng(bins)
    rows = [(rng.random(bins) ** 5).astype(np.float32) for _ in range(6)]
    q = np.round(rng.random(bins) * 7).astype(np.float32) / 8            # 8 distinct values
    rows.append(q)
    t = (rng.random(bins) ** 3).astype(np.float32)
    t[rng.integers(0, bins, bins // 3)] = np.float32(0.001)              # a third of the row tied near the bottom
    rows.append(t)
    rows.append(np.full(bins, 0.3, np.float32))
    rows.append(np.zeros(bins, np.float32))
    rows.append(np.sort((rng.random(bins) ** 4).astype(np.float32)))
    rows.append(np.sort((rng.random(bins) ** 4).astype(np.float32))[::-1].copy())
    m = (rng.random(bins) * 0.5).astype(np.float32)
    m[[bins // 3, bins // 2, bins - 1]] = 0.75                           # the largest bin three times
    rows.append(m)
    d = (rng.random(bins) * 1e-41).astype(np.float32)                    # denormals
    d[bins // 2] = 1e-30
    rows.append(d)
    o = (rng.random(bins) + 1.0).astype(np.float32)                      # values in [1, 2): keys differ only in the mantissa
    o[7 % bins] = 1e-20
    rows.append(o)
    psd = np.stack(rows).astype(np.float32)
    got = lib.compute_floor(torch_cuda.from_numpy(psd).cuda()).cpu().numpy().astype(np.float64)
    want = np.array([oracle.floor_stats(r) for r in psd], np.float64)
    assert np.array_equal(got[:, [0, 2, 3]], want[:, [0, 2, 3]])
    # the reference adds the m = 5 % smallest bins in float, in sorted order (fft.c:271-273); the kernels
    # add them in double: up to m/2 ulp apart (m = 410 at 8193 bins), inside the 1e-5 of the PSD itself
    m = bins - int(bins * 0.95)
    assert np.allclose(got[:, 1], want[:, 1], rtol=max(TOL, 0.6 * m * 2.0 ** -24), atol=1e-44)    # m/2 ulp: 5e-5 at 32769 bins


@pytest.mark.parametrize("mode_name,mode_id", [("plain", 2), ("sumextreme", 3), ("sumavg", 1)])
@pytest.mark.parametrize("max0", [0, 1])
def test_moving_average(lib, torch_cuda, mode_name, mode_id, max0):
    g = np.load(os.path.join(GOLD, "avg_floor_fft1024.npz"))
    psd = torch_cuda.from_numpy(g["psd"]).cuda()
    avg, ret = lib.update_avg(mode_id, psd, int(g["depth"]), int(g["minbin"]), int(g["maxbin"]), max0=max0)
    avg, ret = avg.cpu().numpy(), ret.cpu().numpy()
    want_avg, want_ret = g["%s_max%d_avg" % (mode_name, max0)], g["%s_max%d_ret" % (mode_name, max0)]
    if mode_name == "plain":
        assert np.array_equal(avg, want_avg)          # per-bin double recurrence: bit-exact
    else:
        assert np.allclose(avg, want_avg, rtol=1e-12, atol=0)   # depends on a reduction over bins
    assert np.allclose(ret[:, 0], want_ret[:, 0], rtol=1e-12)
    assert np.array_equal(ret[:, 1], want_ret[:, 1])
    if mode_name == "sumavg":
        assert np.allclose(ret[:, 2], want_ret[:, 2], rtol=1e-12)


@pytest.mark.parametrize("depth", [1, 4, 37, 200])
def test_moving_average_long_run(lib, oracle, torch_cuda, depth):
    """1000 rows (the device cuts the per-bin recurrence into 128-row chunks that restart from a
    direct sum): identical to the oracle's row-by-row recurrence while the sums are exact in double.
    With an 11-decade burst the reference's running sum keeps the rounding it made while it was
    large (absolute error up to steps * ulp(largest sum so far)); the restarted chunks do not, so
    the two agree to that bound -- the reference's own rounding history -- and no closer."""
    rng = np.random.default_rng(depth)
    frames, bins = 1000, 257
    # multiples of 2^-20 in [0.5, 1): any sum of <= 200 of them is exact in a double
    exact = (rng.integers(2 ** 19, 2 ** 20, (frames, bins)) / 2.0 ** 20).astype(np.float32)
    wide = (rng.random((frames, bins)) ** 3).astype(np.float32)
    wide[300:420] *= np.float32(1e11)                 # a burst 110 dB above the rest
    for p, is_exact in ((exact, True), (wide, False)):
        avg, ret = lib.update_avg(lib.AVG_PLAIN, torch_cuda.from_numpy(p).cuda(), depth, 3, 250)
        avg, ret = avg.cpu().numpy(), ret.cpu().numpy()
        a = oracle.Averager(512, depth)
        largest = np.zeros(bins)
        for f in range(frames):
            r, want, peak, _ = a.update("plain", p[f], 3, 250, n=bins)
            largest = np.maximum(largest, np.abs(want))
            if is_exact:
                assert np.array_equal(avg[f], want), (f, depth)
                assert ret[f, 1] == peak and ret[f, 0] == r
            else:
                bound = (f + 1) * 2.5e-16 * largest
                assert np.all(np.abs(avg[f] - want) <= bound + 1e-300), (f, depth)


@pytest.mark.parametrize("bins,minbin,maxbin", [(129, 2, 120), (257, 0, 257), (513, 10, 500), (1025, 25, 1000),
                                                (2049, 25, 2000), (4097, 1, 4097), (8193, 100, 8100)])
@pytest.mark.parametrize("mode", ["plain", "sumextreme", "sumavg"])
def test_moving_average_row_shapes(lib, oracle, torch_cuda, bins, minbin, maxbin, mode):
    """Every row length of the estimator (1..33 bins per thread of the fused kernel), bands that do and
    do not start at bin 0 / end at the last bin, both normalisations, 300 rows (two restarts of the
    chunked recurrence) with depth 1 and 6, against the oracle's row-by-row averager."""
    rng = np.random.default_rng(bins + len(mode))
    frames = 300
    p = (rng.random((frames, bins)) ** 3 + 0.05).astype(np.float32)
    p[40, minbin + (maxbin - minbin) // 3] = 50.0                       # a clear peak in one row
    mode_id = {"plain": lib.AVG_PLAIN, "sumextreme": lib.AVG_SUMEXTREME, "sumavg": lib.AVG_SUMAVG}[mode]
    for depth, max0 in ((1, 0), (6, 1), (6, 0)):
        avg, ret = lib.update_avg(mode_id, torch_cuda.from_numpy(p).cuda(), depth, minbin, maxbin, max0=max0)
        avg, ret = avg.cpu().numpy(), ret.cpu().numpy()
        a = oracle.Averager(bins, depth)
        for f in range(frames):
            r, want, peak, var = a.update(mode, p[f], minbin, maxbin, max0=max0, n=bins)
            if mode == "plain":
                assert np.array_equal(avg[f], want), (f, depth)
            else:
                assert np.allclose(avg[f], want, rtol=1e-11, atol=1e-300), (f, depth, max0)
            assert np.isclose(ret[f, 0], r, rtol=1e-11) and ret[f, 1] == peak, (f, depth, max0)
            if mode == "sumavg":
                assert np.isclose(ret[f, 2], var, rtol=1e-10), (f, depth, max0)


def test_shards_reproduce_the_full_run(lib, torch_cuda):
    """glfer_amd.shard: frame ranges computed from each rank's own sample window (hops + left
    halo, addressed through a virtual base pointer) give exactly the rows of the full run."""
    from glfer_amd.shard import frame_range, run_shard, sample_window
    torch = torch_cuda
    for params, frames in ((lib.FftParams(n=4096, window_type=0, overlap=0.75), 150),
                           (lib.MtmParams(n=4096, overlap=0.0, w=2.5, kmax=4), 301),
                           (lib.MtmParams(n=4096, overlap=0.75, w=2.5, kmax=4), 131),
                           (lib.MtmParams(n=1024, overlap=0.5, w=2.0, kmax=2), 500),
                           (lib.FftParams(n=1024, window_type=7, overlap=0.9), 333)):
        sp = lib.Spectrogram(params)
        x = torch.from_numpy(synth(frames * sp.hop, seed=21)).cuda()
        full = sp.run(x)
        for world, align in ((2, 32), (3, 32), (8, 32), (3, 1), (7, 1)):
            parts = []
            for rank in range(world):
                first, count = frame_range(frames, rank, world, align=align)
                begin, end = sample_window(first, count, sp.hop, sp.n)
                local = x[begin:end].clone()             # a rank holds only its window
                parts.append(run_shard(sp, local, begin, first, count))
            got = torch.cat(parts)
            if align == 32:
                # cuts on multiples of GLFER_FRAME_ALIGN: bit-identical rows
                assert torch.equal(got, full)
            else:
                # any other cut: a frame may share its odd taper's transform with a different
                # neighbour (or none), which moves the last bits only
                for f in range(frames):
                    assert (got[f] - full[f]).abs().max().item() <= 1e-6 * full[f].max().item()


def _write_wav(path, samples, rate):
    import wave
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(samples.dtype.itemsize)
        w.setframerate(rate)
        w.writeframes(samples.tobytes())


@pytest.mark.parametrize("bits", [16, 8])
def test_wav_file_ingest(lib, oracle, torch_cuda, tmp_path, bits):
    """source.c:118-128 + wav_fmt.c:45-121: header parse, PCM conversion in the gather, streamed
    in several chunks with the N-H history carried on the device between them."""
    x = synth(1024 * 61 + 300, fs=8000.0, seed=6)
    raw = np.round(x * 32767).astype(np.int16) if bits == 16 else np.clip(np.round(x * 127 + 128), 0, 255).astype(np.uint8)
    path = tmp_path / ("t%d.wav" % bits)
    _write_wav(path, raw, 8000)
    info = lib.wav_probe(str(path))
    assert (info.format, info.channels, info.sample_rate, info.bits_per_sample, info.nsamples) == (1, 1, 8000, bits, raw.size)
    conv = oracle.pcm_s16_to_float if bits == 16 else oracle.pcm_u8_to_float
    fmt = lib.SAMPLES_S16 if bits == 16 else lib.SAMPLES_U8
    for params, want in (
            (lib.FftParams(n=1024, window_type=0, overlap=0.75, sample_format=fmt, sub_mean=1),
             oracle.spectrogram_fft(conv(raw), 1024, 0.75, 0, sub_mean=1)),
            (lib.MtmParams(n=1024, overlap=0.5, w=2.5, kmax=4, sample_format=fmt),
             oracle.spectrogram_mtm(conv(raw), 1024, 0.5, 2.5, 4)),
            (lib.FftParams(n=4096, window_type=7, overlap=0.0, sample_format=fmt),
             oracle.spectrogram_fft(conv(raw), 4096, 0.0, 7))):
        sp = lib.Spectrogram(params)
        for chunk in (0, 7, 64):
            got = sp.run_wav(str(path), chunk_frames=chunk)
            assert got.shape == want.shape
            assert max(max(rel_err(got[f], want[f])) for f in range(want.shape[0])) < TOL
    # errors: wrong sample format for the file, not a WAV file, missing file
    with pytest.raises(lib.GlferHipError, match="bad argument"):
        lib.Spectrogram(lib.FftParams(n=1024, sample_format=lib.SAMPLES_F32)).run_wav(str(path))
    junk = tmp_path / "junk.wav"
    junk.write_bytes(b"not a wav file at all, but longer than forty-four bytes....")
    with pytest.raises(lib.GlferHipError, match="bad argument"):
        lib.wav_probe(str(junk))
    with pytest.raises(lib.GlferHipError, match="bad argument"):
        lib.wav_probe(str(tmp_path / "missing.wav"))


@pytest.mark.parametrize("n,overlap,t,p_e,sub_mean", [(4096, 0.0, 128, 32, 0), (1024, 0.5, 96, 16, 1), (4096, 0.75, 96, 16, 0),
                                                     (2048, 0.0, 64, 8, 0)])
def test_hparma_parity(lib, oracle, torch_cuda, n, overlap, t, p_e, sub_mean):
    """BASELINE config 5 (hparma.c:74-157 + util.c:261-386), incl. the reference's row-0 overflow.
    The estimator's output is 1/(|A(f)|^2/N) below Nyquist.  Parity is stated on |A(f)|^2/N,
    peak-normalised: 1e-5 at BASELINE config 5's shape (N = 4096, t = 128, p_e = 32; measured <= 4.5e-6), and
    max(1e-5, 3 x s) at the other shapes, s = the largest movement of the ORACLE's own result over this stream when its
    input samples are perturbed by one float ulp (tests/_spread.py, computed here: the AR vector comes from the noise
    subspace of an ill-conditioned matrix, and at t = 96, p_e = 16 the reference itself moves by 1.4e-5).  The reciprocal amplifies every absolute
    error by max|A|^2/|A_k|^2 at the spectral peaks, where the reference's own float32 FFT is
    ~1e-3 away from exact arithmetic; so the final spectrum is additionally checked against a
    float64 evaluation of the ORACLE's AR vector: per-bin relative error <= 1e-2 even at the peaks
    (same rank; AR coefficients equal to ~1e-7)."""
    frames = 10
    h = oracle.hop(n, overlap)
    x = synth(frames * h, seed=n + t)
    from _spread import hparma_bound
    ref = oracle.hparma_frames(x, n, overlap, t, p_e, sub_mean=sub_mean)
    bound, spread, _ = hparma_bound(oracle, x, n, overlap, t, p_e, sub_mean, seed=t)
    sp = lib.Spectrogram(lib.HparmaParams(n=n, overlap=overlap, t=t, p_e=p_e, sub_mean=sub_mean))
    got = sp.run(torch_cuda.from_numpy(x).cuda()).cpu().numpy().astype(np.float64)
    assert got.shape == (frames, n // 2 + 1) and np.isfinite(got).all()
    k = np.arange(n // 2 + 1)
    worst = 0.0
    for f, (psd, a, rank) in enumerate(ref):
        want = psd.astype(np.float64)
        inv_g, inv_w = 1.0 / got[f, :n // 2], 1.0 / want[:n // 2]
        worst = max(worst, max(rel_err(inv_g, inv_w)))
        assert max(rel_err(inv_g, inv_w)) <= bound, (f, rel_err(inv_g, inv_w), bound, spread)
        # the Nyquist bin is not inverted (hparma.c:154): it IS |A|^2/N, one more entry of the vector the bound is stated on
        assert abs(got[f, n // 2] - want[n // 2]) <= bound * np.abs(inv_w).max(), (f, got[f, n // 2], want[n // 2])
        A = np.polyval(a[::-1].astype(np.float64), np.exp(-2j * np.pi * k / n))   # sum_m a[m] z^m
        exact = np.abs(A) ** 2 / n
        exact[:n // 2] = 1.0 / exact[:n // 2]
        assert np.abs(got[f] / exact - 1).max() < 1e-2, (f, rank, np.abs(got[f] / exact - 1).max())
    print("hparma N=%d t=%d p_e=%d: worst %.2e, bound %.2e (oracle 1-ulp spread %.2e)" % (n, t, p_e, worst, bound, spread))
