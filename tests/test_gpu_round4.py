"""Round 4: flat-spectrum parity at N >= 8192 with the measured bound, the limiter at 1e-5, cfg.sub_mean = 1
as the reference's rows (piecewise hop means beside the estimator launches), the boundary's loose ends (scope window
toggled over a file, leaving without close_wav_file, block sizes above 65536 through the shim), the kept scratch's cap."""
import os
import subprocess

import numpy as np
import pytest

from _exact import multitaper64, periodogram64
from _signals import rel_err, synth

pytestmark = pytest.mark.gpu
TOL = 1e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


# ---- item 1: flat-spectrum inputs at N >= 8192 -----------------------------------------------------------------
def flat_stream(shape, nsamples, hop, seed):
    """noise: 0.3 sigma, nothing else; dcnoise: 0.45 + 0.3 sigma (the stream of the case round 3 dropped);
    alternating: hops of noise and hops of a tone + 0.01 sigma in turn."""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal(nsamples)
    if shape == "noise":
        x = 0.3 * g
    elif shape == "dcnoise":
        x = 0.45 + 0.3 * g
    else:
        k = np.arange(nsamples)
        tone = 0.5 * np.sin(2 * np.pi * 1000.0 * k / 48000.0) + 0.01 * g
        x = np.where((k // hop) % 2 == 0, 0.3 * g, tone)
    return x.clip(-0.99, 0.99).astype(np.float32)


ESTIMATORS = {"hanning": ("fft", 0), "kaiser": ("fft", 7), "mtm5": ("mtm", 2.5, 4), "mtm9": ("mtm", 4.5, 8)}


@pytest.mark.parametrize("sub_mean", [0, 1], ids=["nomean", "refmean"])
@pytest.mark.parametrize("overlap", [0.0, 0.75])
@pytest.mark.parametrize("est", list(ESTIMATORS))
@pytest.mark.parametrize("shape", ["noise", "dcnoise", "alternating"])
@pytest.mark.parametrize("n", [8192, 16384])
def test_flat_spectrum_parity_at_large_blocks(lib, oracle, torch_cuda, n, shape, est, overlap, sub_mean):
    """VERDICT r3 item 1.  Every earlier parity input at N >= 8192 was tone-dominated, which flatters a peak-normalised
    error: on noise-like frames the REFERENCE's own float32 transform with its recurrence twiddles (fft_radix2.c:127-141)
    is ~1e-5 from exact arithmetic at N = 16384 (the oracle's FFT is bit-identical to the reference's compiled object,
    tests/test_oracle_pinning.py), while the device's table-twiddle transform stays within 4.2e-7 of exact.  So per frame
        err(device, oracle) <= max(1e-5, 1.1 x err(oracle, float64-exact))
    with both printed; `exact` = the same float32 samples, the hop means removed exactly as fft.c:88-95 does, the rest in
    float64 (tests/_exact.py).  Periodogram (Hanning, Kaiser) and multitaper (5 tapers; 9 tapers = C4's estimator),
    overlap 0 and 75 %, mean removal off and in the reference's order."""
    torch = torch_cuda
    h = oracle.hop(n, overlap)
    frames = 6 if overlap == 0.0 else 12
    x = flat_stream(shape, frames * h, h, seed=n // 64 + len(shape) + 7 * sub_mean)
    e = ESTIMATORS[est]
    if e[0] == "fft":
        want = oracle.spectrogram_fft(x.copy(), n, overlap, e[1], sub_mean=sub_mean)
        exact = periodogram64(x, n, overlap, oracle.window(e[1], n), sub_mean=sub_mean)
        params = lib.FftParams(n=n, window_type=e[1], overlap=overlap, sub_mean=sub_mean)
    else:
        want = oracle.spectrogram_mtm(x.copy(), n, overlap, e[1], e[2], sub_mean=sub_mean)
        taps, sig = lib.make_dpss(n, e[2], e[1])
        exact = multitaper64(x, n, overlap, taps, sig, sub_mean=sub_mean)
        params = lib.MtmParams(n=n, overlap=overlap, w=e[1], kmax=e[2], sub_mean=sub_mean)
    got = lib.Spectrogram(params).run(torch.from_numpy(x).cuda()).cpu().numpy()
    assert got.shape == want.shape == exact.shape == (frames, n // 2 + 1)
    worst = (0.0, 0.0, 0.0)
    for f in range(frames):
        e_dev, e_ref, e_devx = max(rel_err(got[f], want[f])), max(rel_err(want[f], exact[f])), max(rel_err(got[f], exact[f]))
        worst = max(worst, (e_dev, e_ref, e_devx))
        assert e_dev <= max(TOL, 1.1 * e_ref), (n, shape, est, overlap, sub_mean, f, e_dev, e_ref, e_devx)
        assert e_devx <= 1e-6, (n, shape, est, overlap, sub_mean, f, e_devx)        # the device itself: within 1e-6 of exact (observed: 4.2e-7)
    print("flat spectrum N=%d %s %s ovl %.2f mean %d: worst frame err(device, oracle) %.2e  err(oracle, exact) %.2e  err(device, exact) %.2e"
          % (n, shape, est, overlap, sub_mean, *worst))


def test_the_case_round_3_dropped(lib, oracle, torch_cuda):
    """gpurun_out/r3_tests5.log:106: N = 16384, overlap 0, 0.45 DC + 0.3 sigma noise, Hanning, the reference's mean order --
    frame 1 at 1.31e-5 of the oracle against 1e-5.  The same stream (rng seed 12 after the four earlier cases' draws), with
    the cause measured: the ORACLE is 1.31e-5 from exact on that frame (worst 1.59e-5, median 1.0e-5 over the 70 frames),
    the device ~1e-6 -- the difference is the reference's recurrence twiddles, not the device."""
    torch = torch_cuda
    rng = np.random.default_rng(12)
    for n, overlap in ((1024, 0.9), (4096, 0.75), (2048, 0.0), (512, 0.5)):
        rng.standard_normal(70 * int(n * (1.0 - float(np.float32(overlap)))))
    n, frames = 16384, 70
    x = (0.45 + 0.3 * rng.standard_normal(frames * n)).clip(-0.99, 0.99).astype(np.float32)
    want = oracle.spectrogram_fft(x.copy(), n, 0.0, oracle.WINDOWS["hanning"], sub_mean=1)
    exact = periodogram64(x, n, 0.0, oracle.window(oracle.WINDOWS["hanning"], n), sub_mean=1)
    got = lib.Spectrogram(lib.FftParams(n=n, window_type=lib.WINDOWS["hanning"], overlap=0.0, sub_mean=1)).run(torch.from_numpy(x).cuda()).cpu().numpy()
    e_dev = np.array([max(rel_err(got[f], want[f])) for f in range(frames)])
    e_ref = np.array([max(rel_err(want[f], exact[f])) for f in range(frames)])
    e_devx = np.array([max(rel_err(got[f], exact[f])) for f in range(frames)])
    print("dropped case: frame 1 err(device, oracle) %.2e  err(oracle, exact) %.2e  err(device, exact) %.2e; over 70 frames worst %.2e / %.2e / %.2e"
          % (e_dev[1], e_ref[1], e_devx[1], e_dev.max(), e_ref.max(), e_devx.max()))
    assert 1.0e-5 < e_ref[1] < 1.6e-5                 # the reference's own distance from exact on that frame
    assert (e_dev <= np.maximum(TOL, 1.1 * e_ref)).all()
    assert e_devx.max() <= 1e-6


# ---- item 2: the limiter at 1e-5 --------------------------------------------------------------------------------
@pytest.mark.parametrize("n,a,overlap,window", [(4096, 0.001, 0.5, "hamming"), (4096, 0.0, 0.0, "hanning"), (1024, 0.001, 0.75, "kaiser"),
                                                (64, 0.0, 0.5, "hanning"), (16384, 0.001, 0.5, "blackman"), (65536, 0.0, 0.5, "hanning")])
def test_limiter_at_the_usual_tolerance(lib, oracle, torch_cuda, n, a, overlap, window):
    """fft.c:151-156 in the estimator kernels the way prepare_kernel does it (the reference's double log / exp with its
    float ftmp): |y|^0.1 flattens the frame, so the float intrinsics' 1e-7 |log y| per sample showed as 2e-4 of the row
    maximum (round 3's tolerance for this path).  Limiter alone and limiter + RA9MB, the packed kernel (N = 1024, 4096,
    16384), spectro_small (N = 64) and the two-kernel form (N = 65536)."""
    torch = torch_cuda
    h = oracle.hop(n, overlap)
    frames = 9 if n <= 16384 else 3
    x = synth(frames * h, seed=n % 97 + 3)
    want = oracle.spectrogram_fft(x.copy(), n, overlap, oracle.WINDOWS[window], a, 1)
    got = lib.Spectrogram(lib.FftParams(n=n, window_type=lib.WINDOWS[window], overlap=overlap, a=a, limiter=1)).run(torch.from_numpy(x).cuda()).cpu().numpy()
    worst = 0.0
    for f in range(frames):
        e = max(rel_err(got[f], want[f]))
        worst = max(worst, e)
        bound = TOL
        if n > 16384:                                     # (above 16384 the reference's own transform is > 1e-5 from exact: measured bound)
            fr = np.zeros(n)
            fr[max(0, n - (f + 1) * h):] = x[max(0, (f + 1) * h - n):(f + 1) * h]
            y = fr.astype(np.float32) * oracle.window(oracle.WINDOWS[window], n)
            ft = np.log(np.abs(y.astype(np.float64))).astype(np.float32)
            lim = np.where(y > 0, np.exp(ft.astype(np.float64) * 0.1), -np.exp(ft.astype(np.float64) * 0.1)).astype(np.float32)
            exact = np.abs(np.fft.rfft(lim.astype(np.float64))) ** 2 / n
            bound = max(TOL, 1.1 * max(rel_err(want[f], exact)))
        assert e <= bound, (n, a, f, e, bound)
    print("limiter N=%d a=%g: worst %.2e" % (n, a, worst))


# ---- item 3: cfg.sub_mean = 1 is the reference's rows, piece by piece ---------------------------------------------
@pytest.mark.parametrize("case", [("fft", 4096, 0.75, "f32"), ("fft", 1024, 0.5, "s16"), ("fft", 4096, 0.0, "u8"), ("mtm", 4096, 0.0, "f32"),
                                  ("mtm", 4096, 0.75, "s16"), ("fft", 2048, 0.875, "f32"), ("mtm", 1024, 0.0, "f32")], ids=lambda c: "%s-%d-%.3f-%s" % c)
def test_reference_means_piece_by_piece(lib, oracle, torch_cuda, case):
    """The hop means in the reference's order are taken PIECE BY PIECE (side stream) in front of the estimator launches
    (glfer_hip.cpp launch_body_with_reference_means), so that a piece's second read comes out of the Infinity Cache.  Rows
    must not depend on the pieces: one piece (GLFER_EXACT_PIECE_MB=0), 1 MB pieces on one, two and three streams, every
    means kernel (64 / 16 / 4 hops per wavefront, a persistent grid of 8 blocks) -- bit-identical, and the oracle's rows on a
    DC-heavy stream (0.45 + 0.3 sigma: where the order of the sum shows) to 1e-5."""
    torch = torch_cuda
    mode, n, overlap, fmt = case
    h = oracle.hop(n, overlap)
    frames = (3 << 20) // h + 37                                  # ~12 MB of f32 samples: a dozen 1 MB pieces, a ragged last one
    rng = np.random.default_rng(n + frames)
    x = (0.45 + 0.3 * rng.standard_normal(frames * h)).clip(-0.99, 0.99).astype(np.float32)
    if fmt == "s16":
        raw = np.round(x * 32767).astype(np.int16)
        xf, sf = oracle.pcm_s16_to_float(raw), lib.SAMPLES_S16
    elif fmt == "u8":
        raw = np.clip(np.round(x * 127 + 128), 0, 255).astype(np.uint8)
        xf, sf = oracle.pcm_u8_to_float(raw), lib.SAMPLES_U8
    else:
        raw, xf, sf = x, x, lib.SAMPLES_F32
    params = (lib.FftParams(n=n, window_type=0, overlap=overlap, sub_mean=lib.SUBMEAN_EXACT, sample_format=sf) if mode == "fft" else
              lib.MtmParams(n=n, overlap=overlap, w=2.5, kmax=4, sub_mean=lib.SUBMEAN_EXACT, sample_format=sf))
    sp = lib.Spectrogram(params)
    d = torch.from_numpy(raw).cuda()
    knobs = ("GLFER_EXACT_PIECE_MB", "GLFER_EXACT_STREAMS", "GLFER_MEANS_HPW", "GLFER_MEANS_BLOCKS", "GLFER_MEANS_PRODUCERS", "GLFER_FUSED_MIN_FRAMES",
             "GLFER_FUSED_BLOCK_FRAMES", "GLFER_FUSED_LOOK")
    saved = {k: os.environ.get(k) for k in knobs}

    def run(**env):
        for k in knobs:
            os.environ.pop(k, None)
        for k, v in env.items():
            os.environ[k] = str(v)
        out = sp.run(d).cpu().numpy()
        torch.cuda.synchronize()
        return out
    try:
        base = run(GLFER_EXACT_PIECE_MB=0, GLFER_MEANS_HPW=64, GLFER_MEANS_PRODUCERS=0)
        # ... and the fused launch (the periodogram's table form: the means produced by the launch's own first workgroups)
        for env in (dict(), dict(GLFER_FUSED_MIN_FRAMES=64, GLFER_MEANS_PRODUCERS=8), dict(GLFER_FUSED_MIN_FRAMES=64, GLFER_MEANS_PRODUCERS=64),
                    dict(GLFER_FUSED_MIN_FRAMES=64, GLFER_MEANS_PRODUCERS=256),
                    # round 5: the lock-stepped fused launch (consumer workgroups of a few frames walked front by front, producers throttled)
                    dict(GLFER_FUSED_MIN_FRAMES=64, GLFER_MEANS_PRODUCERS=8, GLFER_FUSED_BLOCK_FRAMES=16, GLFER_FUSED_LOOK=64),
                    dict(GLFER_FUSED_MIN_FRAMES=64, GLFER_MEANS_PRODUCERS=64, GLFER_FUSED_BLOCK_FRAMES=8, GLFER_FUSED_LOOK=0),
                    dict(GLFER_FUSED_MIN_FRAMES=64, GLFER_MEANS_PRODUCERS=16, GLFER_FUSED_BLOCK_FRAMES=64, GLFER_FUSED_LOOK=1024),
                    dict(GLFER_EXACT_PIECE_MB=1, GLFER_EXACT_STREAMS=1, GLFER_MEANS_HPW=16),
                    dict(GLFER_EXACT_PIECE_MB=1, GLFER_EXACT_STREAMS=2, GLFER_MEANS_HPW=4, GLFER_MEANS_BLOCKS=8),
                    dict(GLFER_EXACT_PIECE_MB=1, GLFER_EXACT_STREAMS=3, GLFER_MEANS_HPW=16, GLFER_MEANS_BLOCKS=8),
                    dict(GLFER_EXACT_PIECE_MB=2, GLFER_EXACT_STREAMS=3, GLFER_MEANS_HPW=64)):
            got = run(**env)
            assert np.array_equal(got.view(np.uint32), base.view(np.uint32)), env
        part = None
        if frames > 200:                                              # a launch inside the stream, on the frame-group grid
            for k in knobs:
                os.environ.pop(k, None)
            os.environ["GLFER_EXACT_PIECE_MB"] = "1"
            part = sp.run(d, first_frame=64, nframes=frames - 101).cpu().numpy()
            os.environ["GLFER_EXACT_PIECE_MB"] = "0"
            os.environ["GLFER_FUSED_MIN_FRAMES"] = "64"
            part_fused = sp.run(d, first_frame=64, nframes=frames - 101).cpu().numpy()
            assert np.array_equal(part_fused.view(np.uint32), part.view(np.uint32))
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if part is not None:
        cmp = base[64:64 + part.shape[0]]
        assert (np.abs(part - cmp).max(axis=1) <= 2e-6 * np.abs(cmp).max(axis=1)).all()
        assert np.array_equal(part[:frames - 101 - 40].view(np.uint32), cmp[:frames - 101 - 40].view(np.uint32))
    nchk = 150
    want = (oracle.spectrogram_fft(xf[:nchk * h].copy(), n, overlap, 0, sub_mean=1) if mode == "fft" else
            oracle.spectrogram_mtm(xf[:nchk * h].copy(), n, overlap, 2.5, 4, sub_mean=1))
    for f in range(nchk):
        assert max(rel_err(base[f], want[f])) <= TOL, (case, f, rel_err(base[f], want[f]))


def test_reference_means_on_two_caller_streams(lib, torch_cuda):
    """The piecewise form forks onto the plan's side streams and joins the caller's stream again: two callers' streams
    using one plan at once, nothing synchronised in between, must both see their own rows."""
    torch = torch_cuda
    sp = lib.Spectrogram(lib.FftParams(n=4096, window_type=0, overlap=0.75, sub_mean=lib.SUBMEAN_EXACT))
    xs = [torch.randn(1024 * 6000 + 3072, device="cuda") * 0.3 + 0.2 * (i + 1) for i in range(2)]
    saved = os.environ.get("GLFER_EXACT_PIECE_MB")
    os.environ["GLFER_EXACT_PIECE_MB"] = "2"
    try:
        want = [sp.run(x).clone() for x in xs]
        torch.cuda.synchronize()
        outs = [torch.empty_like(w) for w in want]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for rep in range(3):
            for i in range(2):
                with torch.cuda.stream(streams[i]):
                    sp.run(xs[i], out=outs[i])
        torch.cuda.synchronize()
        for i in range(2):
            assert torch.equal(outs[i], want[i]), i
    finally:
        if saved is None:
            os.environ.pop("GLFER_EXACT_PIECE_MB", None)
        else:
            os.environ["GLFER_EXACT_PIECE_MB"] = saved


# ---- the boundary's loose ends -------------------------------------------------------------------------
def _write_wav16(path, pcm, rate=8000):
    import struct
    data = pcm.astype("<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16))
        f.write(b"data" + struct.pack("<I", len(data)) + data)


def _build(tmp_path, name):
    libdir = os.path.join(ROOT, "glfer_amd", "lib")
    exe = tmp_path / name
    subprocess.run(["gcc", "-std=gnu99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", name + ".c"),
                    "-o", str(exe), "-L", libdir, "-lglfer_compat", "-lglfer_hip", "-Wl,-rpath," + libdir], check=True)
    return exe


@pytest.mark.parametrize("scope", [(5, 6, 8, 9), (5, 6, 7, 8), (3, 9, 30, 31), (0, 2, 4, 5)], ids=lambda s: "scope-%d-%d-%d-%d" % s)
def test_scope_window_toggled_over_a_file(oracle, tmp_path, scope):
    """the scope window open for a hop, closed for one or two, open again, at 75 % overlap -- the per-hop
    path ran fewer hops ago than the history reaches back, so the hand-over must CONTINUE from the estimator's
    inbuf_audio (hops [.., 5] + 6, 7 -> frame [5, 6, 7, 8]), not restart from zeros.  Every row against the oracle, the
    hops outside the scope's spans served from the read-ahead."""
    exe = _build(tmp_path, "c_compat_scope_demo")
    n, overlap = 1024, 0.75
    hop = oracle.hop(n, overlap)
    frames = 60
    x = synth(frames * hop, seed=41) * np.float32(0.5) + np.float32(0.2)
    pcm = np.round(x * 32767).astype(np.int16)
    wav = tmp_path / "scope.wav"
    _write_wav16(wav, pcm)
    out = tmp_path / "scope.f32"
    r = subprocess.run([str(exe), str(n), repr(overlap), str(wav), str(out)] + [str(v) for v in scope], check=True, timeout=300,
                       capture_output=True, text=True)
    hops, served = map(int, r.stdout.split())
    per_hop = (scope[1] - scope[0]) + (scope[3] - scope[2])
    assert hops == frames and served == frames - per_hop, (hops, served, r.stderr)
    got = np.fromfile(out, np.float32).reshape(frames, n // 2 + 1)
    want = oracle.wav_spectrogram(pcm, 16, "fft", n, overlap, window_type=0, sub_mean=1, history_mode=0)
    for f in range(frames):
        assert max(rel_err(got[f], want[f])) <= TOL, (scope, f, rel_err(got[f], want[f]))


@pytest.mark.parametrize("quit_hop", [3, 40, 4500])
def test_leaving_without_close_wav_file(tmp_path, quit_hop):
    """/Source/Quit goes straight to gtk_main_quit (g_main.c:115) -- no close_audio, no fft_close.  With
    a read-ahead window being computed on the second host thread the process used to end in std::terminate (a joinable
    std::thread in a static object); it must end with its own exit code.  quit_hop 4500 is inside the second window."""
    exe = _build(tmp_path, "c_compat_scope_demo")
    n, overlap = 1024, 0.5
    pcm = np.round(synth(6000 * 512, seed=43) * 32767).astype(np.int16)
    wav = tmp_path / "quit.wav"
    _write_wav16(wav, pcm)
    r = subprocess.run([str(exe), str(n), repr(overlap), str(wav), str(tmp_path / "q.f32"), "-1", "-1", "-1", "-1", str(quit_hop)],
                       timeout=300, capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr[-400:])
    hops, served = map(int, r.stdout.split())
    assert hops == quit_hop and served == quit_hop


def test_shim_at_block_sizes_above_65536(oracle, tmp_path):
    """fft_init's warm-up launch went through an entry that takes rows of at most 32769 bins and died at
    every n >= 131072 -- the sizes the same round added.  fft_init / fft_do / fft_psd at n = 131072 through the shim (the
    halfcomplex spectrum stops at 32768: fft_do leaves the PSD only), rows within the measured bound of the oracle."""
    exe = _build(tmp_path, "c_compat_wav_demo")
    n, overlap = 131072, 0.5
    hop = n // 2
    frames = 3
    pcm = np.round(synth(frames * hop, fs=8000.0, seed=47) * 32767).astype(np.int16)
    wav = tmp_path / "big.wav"
    _write_wav16(wav, pcm)
    for readahead in (0, 1):
        out = tmp_path / ("big%d.f32" % readahead)
        r = subprocess.run([str(exe), "fft", str(n), repr(overlap), "1", str(readahead), str(wav), str(out)], timeout=600, capture_output=True, text=True)
        assert r.returncode == 0, (readahead, r.returncode, r.stderr[-400:])
        got = np.fromfile(out, np.float32).reshape(frames, n // 2 + 1)
        xf = oracle.pcm_s16_to_float(pcm)
        want = oracle.spectrogram_fft(xf.copy(), n, overlap, 0, sub_mean=1)
        exact = periodogram64(xf, n, overlap, oracle.window(0, n), sub_mean=1)
        for f in range(frames):
            e_ref = max(rel_err(want[f], exact[f]))
            assert max(rel_err(got[f], want[f])) <= max(TOL, 1.1 * e_ref), (readahead, f, e_ref)
            assert max(rel_err(got[f], exact[f])) <= 3e-6, (readahead, f)


def test_scratch_cap_bounds_what_is_kept(lib, torch_cuda):
    """the cap is enforced when a block is handed BACK too -- glfer_hip_scratch_limit(0) keeps nothing
    between calls, whatever was kept before the limit was set."""
    torch = torch_cuda
    L = lib.api.lib()
    saved = os.environ.get("GLFER_MEAN_PREPASS")
    os.environ["GLFER_MEAN_PREPASS"] = "1"
    try:
        sp = lib.Spectrogram(lib.FftParams(n=1024, window_type=1, overlap=0.5, sub_mean=lib.SUBMEAN_FAST))
        frames = (16 << 20) // sp.hop
        x = torch.from_numpy(synth(frames * sp.hop, seed=51)).cuda()
        want = sp.run(x).clone()
        torch.cuda.synchronize()
        assert L.glfer_hip_scratch_held(0) >= 64 << 20                 # the corrected copy came from a kept block
        L.glfer_hip_scratch_limit(0)
        assert L.glfer_hip_scratch_held(0) == 0                        # idle blocks above the new cap go at once
        for _ in range(3):
            got = sp.run(x)
            torch.cuda.synchronize()
            assert L.glfer_hip_scratch_held(0) == 0                    # and a block handed back above the cap is not kept
            assert torch.equal(got, want)
        L.glfer_hip_scratch_limit(32 << 20)
        sp.run(x)
        torch.cuda.synchronize()
        assert L.glfer_hip_scratch_held(0) <= 32 << 20
    finally:
        L.glfer_hip_scratch_limit(16 << 30)
        if saved is None:
            os.environ.pop("GLFER_MEAN_PREPASS", None)
        else:
            os.environ["GLFER_MEAN_PREPASS"] = saved


# ---- item 7: a caller-chosen row pitch for the device entries --------------------------------------------------------
PITCH_CASES = [("fft", 64, 0.5, {}), ("fft", 256, 0.0, {}), ("fft", 1024, 0.5, dict(sub_mean=1)), ("fft", 4096, 0.75, {}),
               ("fft", 4096, 0.75, dict(sub_mean=1)), ("fft", 4096, 0.75, dict(sub_mean=2)), ("fft", 4096, 0.3, {}),
               ("fft", 4096, 0.5, dict(history_mode=1)), ("fft", 16384, 0.5, {}), ("fft", 65536, 0.5, {}), ("fft", 131072, 0.5, {}),
               ("mtm", 1024, 0.5, dict(kmax=4)), ("mtm", 1024, 0.0, dict(kmax=7)), ("mtm", 2048, 0.25, dict(kmax=4)),
               ("mtm", 4096, 0.0, dict(kmax=4)), ("mtm", 4096, 0.0, dict(kmax=4, sub_mean=1)), ("mtm", 4096, 0.75, dict(kmax=3)),
               ("mtm", 8192, 0.5, dict(kmax=4)), ("mtm", 16384, 0.0, dict(kmax=8, w=4.5)), ("hparma", 1024, 0.5, {})]


@pytest.mark.parametrize("case", PITCH_CASES, ids=lambda c: "%s-%d-%.2f-%s" % (c[0], c[1], c[2], "-".join("%s%s" % kv for kv in sorted(c[3].items()))))
def test_rows_at_a_callers_pitch(lib, torch_cuda, case):
    """cfg.psd_pitch: every estimator kernel writes its rows `pitch` floats apart -- the bins of a row
    bit-identical to the dense run's, the floats between rows never written -- whole launches, a launch inside the
    stream, the first frames of a stream (zero history: the packed kernel) and the frames off the frame-group grid
    included."""
    torch = torch_cuda
    mode, n, overlap, kw = case
    bins = n // 2 + 1
    pitch = (bins + 15) // 16 * 16 + (16 if n <= 256 else 0)
    if mode == "fft":
        mk = lambda p: lib.FftParams(n=n, window_type=0, overlap=overlap, psd_pitch=p, **kw)
    elif mode == "mtm":
        rest = {a: b for a, b in kw.items() if a not in ("w", "kmax")}
        mk = lambda p: lib.MtmParams(n=n, overlap=overlap, w=kw.get("w", 2.5), kmax=kw["kmax"], psd_pitch=p, **rest)
    else:
        def mk(p):
            q = lib.HparmaParams(n=n, overlap=overlap, t=96, p_e=16)
            q.psd_pitch = p
            return q
    hop = int(n * (1.0 - float(np.float32(overlap))))
    frames = 5 if n > 16384 else (70 if mode == "hparma" else 333)
    x = torch.from_numpy(synth(frames * hop + 7, seed=n % 89) + np.float32(0.1)).cuda()
    dense = lib.Spectrogram(mk(0)).run(x)
    sp = lib.Spectrogram(mk(pitch))
    assert sp.pitch == pitch and dense.shape == (frames, bins)
    out = torch.full((frames, pitch), float("nan"), device="cuda")
    sentinel = out.view(torch.int32)[0, 0].item()
    sp.run(x, out=out)
    torch.cuda.synchronize()
    assert torch.equal(out[:, :bins].contiguous().view(torch.int32), dense.view(torch.int32))
    assert (out.view(torch.int32)[:, bins:] == sentinel).all()
    if frames > 100:
        part = torch.full((frames - 50, pitch), float("nan"), device="cuda")
        sp.run(x, first_frame=37, nframes=frames - 50, out=part)
        ref = lib.Spectrogram(mk(0)).run(x, first_frame=37, nframes=frames - 50)
        assert torch.equal(part[:, :bins].contiguous().view(torch.int32), ref.view(torch.int32))
        assert (part.view(torch.int32)[:, bins:] == sentinel).all()
    with pytest.raises(lib.GlferHipError):
        lib.Spectrogram(mk(bins - 1))
    if mode != "hparma":
        with pytest.raises(lib.GlferHipError):
            sp.run_host(np.zeros(4 * hop, np.float32))            # host rows are dense


def test_per_column_stages_read_pitched_rows(lib, torch_cuda):
    """compute_floor, update_avg_*, the display map and the one-call waterfall on rows 2112 floats apart: the same
    statistics, averages, pixels and levbuf as on the dense rows (glfer_hip_floor_device_pitched,
    glfer_hip_display.psd_pitch; update_avg takes the pitch as its row stride)."""
    import ctypes as C
    torch = torch_cuda
    L = lib.api.lib()
    n, bins, pitch, frames = 4096, 2049, 2112, 700
    x = torch.from_numpy(synth(frames * 1024 + 3072, seed=9)).cuda()
    dense = lib.Spectrogram(lib.FftParams(n=n, window_type=0, overlap=0.75)).run(x)
    rows = lib.Spectrogram(lib.FftParams(n=n, window_type=0, overlap=0.75, psd_pitch=pitch)).run(x)
    frames = dense.shape[0]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    s_d = lib.compute_floor(dense)
    s_p = torch.empty_like(s_d)
    assert L.glfer_hip_floor_device_pitched(rows.data_ptr(), frames, bins, pitch, s_p.data_ptr(), st) == 0
    assert torch.equal(s_d.view(torch.int32), s_p.view(torch.int32))
    for mode in (lib.AVG_PLAIN, lib.AVG_SUMAVG, lib.AVG_SUMEXTREME):
        a_d, r_d = lib.update_avg(mode, dense, 5, 30, 2000, max0=1)
        a_p = torch.empty_like(a_d)
        r_p = torch.empty_like(r_d)
        assert L.glfer_hip_avg_device(int(mode), rows.data_ptr(), frames, pitch, bins, 5, 30, 2000, 1, a_p.data_ptr(), r_p.data_ptr(), st) == 0
        assert torch.equal(a_d.view(torch.int64), a_p.view(torch.int64)) and torch.equal(r_d.view(torch.int64), r_p.view(torch.int64))
    kw = dict(scale_type=lib.SCALE_LOG, autoscale=1, overlap=0.75, palette=7)
    rgb_d, lev_d, _ = lib.display(lib.Display(**kw), dense, s_d)
    dp = lib.Display(psd_pitch=pitch, **kw)
    rgb_p = torch.empty_like(rgb_d)
    lev_p = torch.empty_like(lev_d)
    assert L.glfer_hip_display_device(C.byref(dp), C.c_void_p(rows.data_ptr()), None, C.c_void_p(s_d.data_ptr()), frames, bins,
                                      C.c_void_p(rgb_p.data_ptr()), C.c_void_p(lev_p.data_ptr()), None, st) == 0
    assert torch.equal(rgb_d, rgb_p) and torch.equal(lev_d, lev_p)
    for mode in (0, lib.AVG_PLAIN, lib.AVG_SUMAVG):
        av = dict(avg_mode=mode, depth=4, minbin=10, maxbin=2000, max0=1)
        w_rgb, w_lev, _ = lib.waterfall(lib.Display(**kw), dense, **av)
        p_rgb = torch.empty_like(w_rgb)
        p_lev = torch.empty_like(w_lev)
        dpp = lib.Display(psd_pitch=pitch, **kw)
        assert L.glfer_hip_waterfall_device(C.byref(dpp), int(mode), 4, 10, 2000, 1, rows.data_ptr(), frames, bins, p_rgb.data_ptr(),
                                            p_lev.data_ptr(), None, st) == 0
        assert torch.equal(w_rgb, p_rgb) and torch.equal(w_lev, p_lev), mode


@pytest.mark.parametrize("n,t,p_e", [(1024, 128, 63), (512, 8, 1), (2048, 64, 2), (1024, 100, 40), (4096, 124, 31), (1024, 66, 16)])
def test_hparma_schedule_over_matrix_shapes(lib, oracle, torch_cuda, n, t, p_e):
    """The static rotation schedule (hparma.hip, round 4) at the edges of its range: 64 columns (the widest it takes), 2 and 3
    columns (one and two steps per sweep), t = 100 / 124 (partly filled row chunks), an odd column count, and t = 66 (not a
    multiple of 4: round 3's walk takes it) -- |A(f)|^2 / N against the oracle within max(1e-5, 3 x the oracle's own sampled 1-ulp
    spread on this stream) (tests/_spread.py)."""
    from _spread import hparma_bound
    h = oracle.hop(n, 0.0)
    frames = 6
    x = synth(frames * h, seed=n + t + p_e)
    ref = oracle.hparma_frames(x, n, 0.0, t, p_e, sub_mean=0)
    bound, spread, _ = hparma_bound(oracle, x, n, 0.0, t, p_e, 0, seed=p_e)
    got = lib.Spectrogram(lib.HparmaParams(n=n, overlap=0.0, t=t, p_e=p_e)).run(torch_cuda.from_numpy(x).cuda()).cpu().numpy().astype(np.float64)
    for f in range(frames):
        want = ref[f][0].astype(np.float64)
        # (two columns put a zero of A(z) at z = 1: the reciprocal at bin 0 is inf in the reference too)
        assert np.array_equal(np.isfinite(got[f]), np.isfinite(want)), (n, t, p_e, f)
        e = max(rel_err(1.0 / got[f, :n // 2], 1.0 / want[:n // 2]))
        assert e <= bound, (n, t, p_e, f, e, bound, spread)


@pytest.mark.parametrize("shape", [(128, 32), (96, 16)])
@pytest.mark.parametrize("fmt", ["f32", "s16", "u8"])
def test_hparma_fixed_shape_kernel_equals_the_general_one(lib, torch_cuda, monkeypatch, fmt, shape):
    """BASELINE config 5's shape (t = 128, p_e = 32) and glfer's defaults (t = 96, p_e = 16, glfer.c:248-249) run a kernel with the shape as
    compile-time constants (straight-line steps);
    GLFER_HPARMA_GENERIC=1 sends it through the kernel that takes the shape from its parameters.  Same statements, same order:
    the rows must be the same bits, for every sample format, over enough frames that every workgroup walks more than one."""
    n, frames = 4096, 3000
    x = synth(frames * n, seed=77)
    if fmt == "s16":
        dev, sf = torch_cuda.from_numpy(np.clip(np.round(x * 20000.0), -32768, 32767).astype(np.int16)).cuda(), lib.SAMPLES_S16
    elif fmt == "u8":
        dev, sf = torch_cuda.from_numpy(np.clip(np.round(x * 100.0 + 128.0), 0, 255).astype(np.uint8)).cuda(), lib.SAMPLES_U8
    else:
        dev, sf = torch_cuda.from_numpy(x).cuda(), lib.SAMPLES_F32
    sp = lib.Spectrogram(lib.HparmaParams(n=n, overlap=0.0, t=shape[0], p_e=shape[1], sample_format=sf))
    monkeypatch.delenv("GLFER_HPARMA_GENERIC", raising=False)
    fixed = sp.run(dev).clone()
    monkeypatch.setenv("GLFER_HPARMA_GENERIC", "1")
    general = sp.run(dev).clone()
    assert fixed.shape == (frames, n // 2 + 1)
    assert torch_cuda.equal(fixed.view(torch_cuda.int32), general.view(torch_cuda.int32))
    # frames handed out from the queue (a launch of more frames than wavefronts in flight: 1 792) against launches short enough to
    # take the fixed assignment: a row does not depend on which wavefront computed it
    monkeypatch.delenv("GLFER_HPARMA_GENERIC", raising=False)
    for first, count in ((0, 1000), (1000, 1792), (2792, 208)):
        piece = sp.run(dev, first_frame=first, nframes=count)
        assert torch_cuda.equal(piece.view(torch_cuda.int32), fixed[first:first + count].view(torch_cuda.int32)), (first, count)



@pytest.mark.parametrize("n,ovl,nl", [(1024, 0.5, 5), (512, 0.0, 7), (2048, 0.75, 16), (256, 0.0, 33), (1024, 0.0, 64), (4096, 0.0, 10)])
def test_lmp_ring_of_any_size(lib, torch_cuda, n, ovl, nl):
    """lmp_av other than 2, 3, 4, 8 (a free entry of glfer's options): launches of >= 64 frames keep a thread's ring in LDS and read a
    row once (lmp_ring_any_kernel); shorter ones go frame by frame (lmp_kernel<0>, 2 nl row reads each).  Same sums in the same slot
    order: a long launch must equal the short launches that tile it bit for bit, wherever they start in the stream -- and the
    reference's formula evaluated in numpy on the device's own periodograms."""
    hop = int(n * (1.0 - ovl))
    frames = 300
    x = synth(frames * hop, fs=8000.0, seed=n + nl)
    dx = torch_cuda.from_numpy(x).cuda()
    sp = lib.Spectrogram(lib.LmpParams(n=n, overlap=ovl, avg=nl))
    full = sp.run(dx)
    assert full.shape == (frames, n // 2 + 1)
    for first, count in ((0, 40), (40, 63), (103, 50), (153, 17), (170, 63), (233, 63), (296, 4)):
        piece = sp.run(dx, first_frame=first, nframes=count)
        assert torch_cuda.equal(piece.view(torch_cuda.int32), full[first:first + count].view(torch_cuda.int32)), (first, count)
    late = sp.run(dx, first_frame=37, nframes=frames - 37)          # a long launch that starts inside the stream, off the ring's period
    assert torch_cuda.equal(late.view(torch_cuda.int32), full[37:].view(torch_cuda.int32))
    per = lib.Spectrogram(lib.FftParams(n=n, window_type=lib.WINDOWS["rectangular"], overlap=ovl))
    P = per.run(dx).cpu().numpy().astype(np.float64)
    ring = np.zeros((nl, P.shape[1]))
    got = full.cpu().numpy()
    for f in range(frames):
        ring[f % nl] = P[f]
        my = ring.sum(axis=0) / nl
        sy = ((ring - my) ** 2).sum(axis=0) / (nl - 1)
        v = 0.5 * (my - np.sqrt(np.maximum(my * my - sy, 0.0)))
        with np.errstate(divide="ignore", invalid="ignore"):
            o = -np.sqrt(nl / 2.0) + (nl * my) / (2.0 * np.sqrt(2.0 * nl) * v)
        o = np.where(o <= 1e-3, 1e-3, o)
        o[0] = 1e-3
        assert np.allclose(got[f], o, rtol=3e-6, atol=0), (f, np.abs(got[f] / o - 1).max())
