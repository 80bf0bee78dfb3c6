"""Round 5 (-m gpu): what VERDICT r4 asked to see in the driver's own record.

* 64-bit addressing: streams of 16 GiB (f32) and 8 GiB (s16), rows past the 2^32-byte marks, probed at the start, the
  middle and the LAST frames against the oracle (was tests/big_stream_check.py, a script).
* the bench-scale launches themselves (bench.py's 2^30-sample streams: C3 262 144 frames, C2 1 048 576 frames, C1
  2 097 152 frames), with and without the reference's mean removal: their first, middle and LAST frames against the
  oracle -- the workgroup that walks the end of the range, the clamped frame slots, the last rows' stores.
"""
import numpy as np
import pytest

from _signals import rel_err, synth as synth_stream

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _oracle_row(oracle, params, lib, seg, n, overlap):
    """The LAST row of the oracle run over `seg` (whole hops, true overlap)."""
    if params.mode == lib.MODE_MTM:
        return oracle.spectrogram_mtm(seg, n, overlap, params.w, params.kmax, sub_mean=1 if params.sub_mean else 0)[-1]
    return oracle.spectrogram_fft(seg, n, overlap, params.window_type, 0.0, 0, 1 if params.sub_mean else 0, 0)[-1]


def _probe(lib, oracle, torch, params, x, probes, to_float=lambda a: a, out=None):
    """Run the whole stream through the C-ABI, then compare the probed frames with the oracle.  A probed frame f is
    recomputed by the oracle from the hops [f - 2 * back, f]: by the last of those frames the zero history of the
    segment's start has left the frame (and every hop the frame touches carries its own mean), so the last row is the
    frame as the reference would compute it in the middle of the stream."""
    sp = lib.Spectrogram(params)
    rows = sp.run(x, out=out)
    torch.cuda.synchronize()
    n, h = sp.n, sp.hop
    assert rows.shape[0] == x.numel() // h
    back = (n - h + h - 1) // h
    worst = 0.0
    for f in probes:
        f0 = max(f - 2 * back, 0)
        seg = np.ascontiguousarray(to_float(x[f0 * h:(f + 1) * h].cpu().numpy()))
        want = _oracle_row(oracle, params, lib, seg, n, params.overlap)
        got = rows[f].cpu().numpy()
        e = max(rel_err(got, want))
        assert e < TOL, (f, e)
        worst = max(worst, e)
    total = rows.shape[0]
    sp.close()
    return worst, total


def test_streams_past_4_gib_are_addressed_with_64_bits(lib, oracle, torch_cuda):
    """16 GiB of f32 samples (2^20 frames of N = 4096, multitaper, overlap 0: 8.6 GB of rows) and 8 GiB of s16 samples at
    75 % overlap (2^22 hops: 34 GB of rows), with and without mean removal: frames either side of every 2^32-byte mark
    that matters and the last ones."""
    torch = torch_cuda
    frames = 1 << 20
    x = torch.empty(frames * 4096, dtype=torch.float32, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    for s in range(0, x.numel(), 1 << 28):
        e = min(x.numel(), s + (1 << 28))
        x[s:e] = torch.sin(torch.arange(s, e, device="cuda", dtype=torch.float64) * 0.013).float() * 0.4
        x[s:e] += 0.05 * torch.randn(e - s, device="cuda", generator=g)
    worst, total = _probe(lib, oracle, torch, lib.MtmParams(n=4096, overlap=0.0, w=2.5, kmax=4), x,
                          [0, 1, 262143, 262144, 524287, 524288, frames - 2, frames - 1])
    print("multitaper N=4096 f32, %d frames, 16 GiB stream: worst probed frame %.2e" % (total, worst))
    del x
    torch.cuda.empty_cache()
    hops = 1 << 22
    x = torch.empty(hops * 1024, dtype=torch.int16, device="cuda")
    for s in range(0, x.numel(), 1 << 28):
        e = min(x.numel(), s + (1 << 28))
        x[s:e] = (torch.randn(e - s, device="cuda", generator=g) * 3000 + 700).clamp_(-32768, 32767).to(torch.int16)
    out = torch.empty((hops, 2049), dtype=torch.float32, device="cuda")
    for sub_mean in (0, 1):
        worst, total = _probe(lib, oracle, torch, lib.FftParams(n=4096, window_type=0, overlap=0.75, sample_format=lib.SAMPLES_S16, sub_mean=sub_mean), x,
                              [0, 2, 3, 4, 40, 524287, 524288, 2097151, 2097152, hops - 2, hops - 1], to_float=oracle.pcm_s16_to_float, out=out)
        print("periodogram N=4096 s16 75%%, sub_mean %d, %d frames, 8 GiB stream: worst probed frame %.2e" % (sub_mean, total, worst))


@pytest.mark.parametrize("workload,sub_mean", [("mtm", 0), ("mtm", 1), ("fft", 0), ("fft", 1), ("fft1k", 0), ("fft1k", 1), ("mtm75", 0), ("mtm16k", 0)])
def test_last_frames_of_the_bench_scale_launch(lib, oracle, torch_cuda, workload, sub_mean):
    """bench.py's own launches -- the same stream generator, seed, length and parameters -- probed at their first, middle and
    last frames (frame 262 143 of the 2^30-sample C3 launch, 1 048 575 of C2's, 2 097 151 of C1's)."""
    import bench
    torch = torch_cuda
    name, n, overlap, nw, kmax, frames, _ = bench.WORKLOADS[workload]
    params = bench.make_params(lib, workload, **(dict(sub_mean=lib.SUBMEAN_EXACT) if sub_mean else {}))
    h = oracle.hop(n, overlap)
    x = bench.synth_on_device(torch, frames * h, torch.device("cuda", 0), seed=0, fs=8000.0 if workload == "fft1k" else 48000.0)
    if sub_mean:
        x += 0.1                                     # (the "+mean" rows of the line run on a stream with a DC level)
    mid = frames // 2
    worst, total = _probe(lib, oracle, torch, params, x, [0, 1, 5, mid - 1, mid, frames - 65, frames - 3, frames - 2, frames - 1])
    assert total == frames
    print("%s sub_mean %d: %d frames, worst probed frame %.2e" % (workload, sub_mean, frames, worst))


# ---- north_star: "|X|^2 + block-average fused in-register"  ------------------------------------------
def _pcm(x, fmt, lib, oracle):
    if fmt == "s16":
        raw = np.clip(np.round(x * 20000), -32768, 32767).astype(np.int16)
        return raw, oracle.pcm_s16_to_float(raw), lib.SAMPLES_S16
    if fmt == "u8":
        raw = np.clip(np.round(x * 100 + 128), 0, 255).astype(np.uint8)
        return raw, oracle.pcm_u8_to_float(raw), lib.SAMPLES_U8
    return x, x, lib.SAMPLES_F32


@pytest.mark.parametrize("sub_mean", [0, 1], ids=["mean-off", "reference-means"])
@pytest.mark.parametrize("n,overlap,fmt,depth,band,wide", [
    (4096, 0.75, "f32", 4, (0, 2049), False),          # the bench row (C2 + update_avg_plain, depth 4)
    (4096, 0.75, "f32", 4, (34, 103), True),           # the reference's default band (glfer.c:278-279: 400-1200 Hz at 48 kHz), avgdata N wide
    (4096, 0.75, "s16", 3, (1, 2048), False),
    (1024, 0.5, "f32", 4, (0, 513), False),
    (2048, 0.0, "u8", 2, (10, 1000), False),
    (512, 0.875, "f32", 4, (3, 250), True),
    (4096, 0.6, "f32", 4, (0, 2049), False),           # a hop that is no whole number of register groups: every frame loaded whole
    (2048, 0.5, "s16", 1, (0, 1025), False),           # depth 1: the "average" of one row (divisor 2, avg.c:138-139)
])
def test_average_inside_the_estimator_launch(lib, oracle, torch_cuda, monkeypatch, n, overlap, fmt, depth, band, wide, sub_mean):
    """glfer_hip_spectrogram_avg_device with the plain average taken inside the periodogram kernel (spectro16h.hip AVG): the
    averaged rows must be the doubles glfer_hip_avg_device makes of the PSD rows, the PSD rows (when asked for) the bits of
    glfer_hip_spectrogram_device, the peak bin equal and the band mean equal to 1e-12 (a sum over lanes in another order);
    with and without the rows stored, from the start of the stream and from its middle; and against the oracle's averager."""
    torch = torch_cuda
    h = oracle.hop(n, overlap)
    frames = 1700
    x = synth_stream(frames * h, seed=n + depth) + np.float32(0.2 * sub_mean)      # (a DC level where the means matter)
    raw, xf, sf = _pcm(x, fmt, lib, oracle)
    # sub_mean = 1: the reference's default (fft.c:186: sub_mean = opt.autoscale) -- the hop means in its summation order, given to the kernel
    sp = lib.Spectrogram(lib.FftParams(n=n, window_type=lib.WINDOWS["hanning"], overlap=overlap, sample_format=sf, sub_mean=sub_mean))
    d = torch.from_numpy(raw).cuda()
    minbin, maxbin = band
    n_out = n if wide else sp.bins
    rows = sp.run(d)
    want_avg, want_ret = lib.update_avg(lib.AVG_PLAIN, rows, depth, minbin, maxbin, n_out=n_out)
    avg, ret, psd = sp.run_avg(d, lib.AVG_PLAIN, depth, minbin, maxbin, n_out=n_out, want_psd=True)
    torch.cuda.synchronize()
    assert torch.equal(psd, rows)
    assert torch.equal(avg, want_avg)
    assert torch.equal(ret[:, 1], want_ret[:, 1]) and torch.equal(ret[:, 3], want_ret[:, 3]) and torch.equal(ret[:, 2], want_ret[:, 2])
    assert torch.allclose(ret[:, 0], want_ret[:, 0], rtol=1e-12, atol=0)
    # rows not stored, return values not wanted: the same averages
    avg2, none_ret, none_psd = sp.run_avg(d, lib.AVG_PLAIN, depth, minbin, maxbin, n_out=n_out, want_psd=False, want_ret=False)
    assert none_ret is None and none_psd is None and torch.equal(avg2, want_avg)
    # a call that starts in the middle of the stream: the averaging state is empty at ITS first frame
    f0, nf = 37, frames - 50
    w_avg, w_ret = lib.update_avg(lib.AVG_PLAIN, rows[f0:f0 + nf].contiguous(), depth, minbin, maxbin, n_out=n_out)
    avg3, ret3, _ = sp.run_avg(d, lib.AVG_PLAIN, depth, minbin, maxbin, n_out=n_out, first_frame=f0, nframes=nf)
    assert torch.equal(avg3, w_avg) and torch.equal(ret3[:, 1], w_ret[:, 1]) and torch.allclose(ret3[:, 0], w_ret[:, 0], rtol=1e-12, atol=0)
    # the two launches (GLFER_AVG_FUSED=0 is read once per process: compare through the other modes, which always take them)
    a_s, r_s, _ = sp.run_avg(d, lib.AVG_SUMEXTREME, depth, minbin, maxbin, n_out=n_out)
    w_s, wr_s = lib.update_avg(lib.AVG_SUMEXTREME, rows, depth, minbin, maxbin, n_out=n_out)
    assert torch.equal(a_s, w_s) and torch.equal(r_s, wr_s)
    # the oracle's row-by-row averager (avg.c:108-159) over the oracle's own PSD rows' device twins
    a = oracle.Averager(n_out, depth)
    rows_h, avg_h, ret_h = rows.cpu().numpy(), avg.cpu().numpy(), ret.cpu().numpy()
    for f in range(40):
        r, want, peak, _ = a.update("plain", rows_h[f], minbin, maxbin, n=n_out)
        assert np.array_equal(avg_h[f], want), f
        assert ret_h[f, 1] == peak and np.isclose(ret_h[f, 0], r, rtol=1e-12), f
    sp.close()


def test_average_inside_the_launch_at_bench_scale(lib, torch_cuda):
    """The bench row's own launch (C2, 262 144 frames, update_avg_plain depth 4 over the whole band): every averaged row equal to
    the two-launch path's, first to last."""
    import bench
    torch = torch_cuda
    frames = 262144
    sp = lib.Spectrogram(bench.make_params(lib, "fft"))
    x = bench.synth_on_device(torch, frames * sp.hop, torch.device("cuda", 0), seed=0)
    rows = sp.run(x)
    want_avg, want_ret = lib.update_avg(lib.AVG_PLAIN, rows, 4, 0, sp.bins)
    avg, ret, _ = sp.run_avg(x, lib.AVG_PLAIN, 4, 0, sp.bins)
    assert torch.equal(avg, want_avg) and torch.equal(ret[:, 1], want_ret[:, 1])
    assert torch.allclose(ret[:, 0], want_ret[:, 0], rtol=1e-12, atol=0)
    sp.close()


# ---- a kept set of workers ------------------------------------------------------------------------------
def _write_wav(path, pcm, rate=48000):
    import struct
    bits = 8 * pcm.dtype.itemsize
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + pcm.nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * bits // 8, bits // 8, bits))
        f.write(b"data" + struct.pack("<I", pcm.nbytes))
        pcm.tofile(f)


@pytest.mark.parametrize("mode,n,overlap,workers", [("mtm", 16384, 0.0, 3), ("fft", 1024, 0.5, 5), ("mtm", 4096, 0.75, 2)])
def test_workers_handle_keeps_plans_and_rings(lib, oracle, torch_cuda, tmp_path, mode, n, overlap, workers):
    """glfer_hip_workers_*: a handle's calls (file and host buffer, several workers sharing this GPU) give the rows of the one-plan
    entries bit for bit, call after call; the phases add up to something sane; the stateless entry -- which now keeps its own
    handles -- gives the same rows, and glfer_hip_scratch_trim(device, 0) takes what it kept back."""
    import ctypes as C
    h = oracle.hop(n, overlap)
    frames = 700 if n == 16384 else 2500
    x = synth_stream(frames * h + 123, seed=n + workers) * 0.8 + np.float32(0.05)
    pcm = np.clip(np.round(x * 30000), -32768, 32767).astype(np.int16)
    path = str(tmp_path / "in.wav")
    _write_wav(path, pcm)
    if mode == "mtm":
        params = lib.MtmParams(n=n, overlap=overlap, w=2.5, kmax=4, sample_format=lib.SAMPLES_S16, sub_mean=1)
    else:
        params = lib.FftParams(n=n, window_type=0, overlap=overlap, sample_format=lib.SAMPLES_S16, sub_mean=1)
    sp = lib.Spectrogram(params)
    want = sp.run_wav(path)
    want_tail = sp.run_wav(path, partial_tail=True)
    sp.close()
    assert want.shape[0] == frames and want_tail.shape[0] == frames + 1
    W = lib.Workers(params, [0] * workers, hint_frames=frames)
    rows = lib.pinned_empty((frames + 1, n // 2 + 1), np.float32)
    for rep in range(3):
        rows[:] = -1.0
        nf, ph = W.run_wav(path, rows[:frames])
        assert nf == frames and np.array_equal(rows[:frames], want), rep
        assert ph["wall_s"] > 0 and ph["chunks"] >= workers and ph["kernel_s"] > 0 and ph["read_s"] > 0 and ph["wall_s"] < 5.0
    nf, _ = W.run_wav(path, rows, partial_tail=True)
    assert nf == frames + 1 and np.array_equal(rows, want_tail)
    pin = lib.pinned_empty((frames * h,), np.int16)
    pin[:] = pcm[:frames * h]
    rows[:] = -1.0
    nf, ph = W.run_host(pin, rows[:frames])
    assert nf == frames and np.array_equal(rows[:frames], want) and ph["read_s"] >= 0
    nf, _ = W.run_host(pcm[:frames * h].copy(), rows[:frames])          # pageable samples: staged
    assert nf == frames and np.array_equal(rows[:frames], want)
    W.close()
    # the stateless entry: twice (the second call finds the workers it kept), then everything handed back
    L = lib.api.lib()
    for rep in range(2):
        got = lib.spectrogram_wav_workers(params, path, [0] * workers)
        assert np.array_equal(got, want), rep
    assert L.glfer_hip_scratch_held(0) > 0
    L.glfer_hip_scratch_trim(0, 0)
    assert L.glfer_hip_scratch_held(0) == 0
    got = lib.spectrogram_wav_workers(params, path, [0] * workers)
    assert np.array_equal(got, want)
    L.glfer_hip_scratch_trim(0, 0)


@pytest.mark.parametrize("mode,n,overlap,sub_mean,workers", [("mtm", 16384, 0.0, 0, 4), ("fft", 4096, 0.75, 1, 3)])
def test_c_program_over_a_workers_handle(lib, oracle, tmp_path, mode, n, overlap, sub_mean, workers):
    """The round-5 entries from a C program (gcc, C99, only include/glfer_hip.h; no Python or ctypes in the data path): ABI check, glfer_hip_wav_probe,
    glfer_hip_workers_create, three calls of glfer_hip_workers_spectrogram_wav into pinned rows -- against the ORACLE's rows for the same file."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "glfer_amd", "lib")
    exe = tmp_path / "c_workers_demo"
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c_workers_demo.c"),
                    "-o", str(exe), "-L", libdir, "-lglfer_hip", "-Wl,-rpath," + libdir], check=True)
    h = oracle.hop(n, overlap)
    frames = 300 if n == 16384 else 1200
    x = synth_stream(frames * h, seed=77) * 0.7 + np.float32(0.1)
    pcm = np.clip(np.round(x * 30000), -32768, 32767).astype(np.int16)
    _write_wav(str(tmp_path / "in.wav"), pcm)
    r = subprocess.run([str(exe), mode, str(n), repr(overlap), str(sub_mean), str(workers), str(tmp_path / "in.wav"), str(tmp_path / "out.f32")],
                       capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    print(r.stdout.strip())
    got = np.fromfile(tmp_path / "out.f32", np.float32).reshape(frames, n // 2 + 1)
    xf = oracle.pcm_s16_to_float(pcm)
    if mode == "mtm":
        want = oracle.spectrogram_mtm(xf, n, overlap, 2.5, 4, sub_mean=sub_mean)
    else:
        want = oracle.spectrogram_fft(xf, n, overlap, oracle.WINDOWS["hanning"], 0.0, 0, sub_mean, 0)
    for f in range(frames):
        assert max(rel_err(got[f], want[f])) < TOL, f


@pytest.mark.parametrize("n,t,p_e,overlap", [(64, 24, 6, 0.0), (128, 64, 12, 0.5), (32768, 128, 32, 0.0), (32, 16, 4, 0.0)])
def test_hparma_block_sizes_outside_256_to_16384(lib, oracle, torch_cuda, n, t, p_e, overlap):
    """g_options.c:386-387 accepts any power of two; HP-ARMA took N = 256 .. 16384 until round 5.  N = 32 .. 128 and 32768 (137 KB of LDS: one
    frame in flight per CU) against the oracle, bound as in test_hparma_parity."""
    from _spread import hparma_bound
    frames = 6
    h = oracle.hop(n, overlap)
    x = synth_stream(frames * h, seed=n + t)
    bound, spread, ref = hparma_bound(oracle, x, n, overlap, t, p_e, 0, draws=6, seed=t)
    got = lib.Spectrogram(lib.HparmaParams(n=n, overlap=overlap, t=t, p_e=p_e)).run(torch_cuda.from_numpy(x).cuda()).cpu().numpy().astype(np.float64)
    assert got.shape == (frames, n // 2 + 1)
    frames_ref = oracle.hparma_frames(x, n, overlap, t, p_e, sub_mean=0)
    k = np.arange(n // 2)
    for f in range(frames):
        fin = np.isfinite(ref[f])
        inv = 1.0 / got[f, :n // 2]
        assert np.array_equal(np.isfinite(inv), fin), f
        e = max(rel_err(inv[fin], ref[f][fin]))
        # |A(f)|^2 / N in float64 from the ORACLE's AR vector: at N = 32768 the reference's own recurrence-twiddle transform of that
        # vector is 1e-4 ... 1e-3 from it (fft_radix2.c:127-141, as for the periodogram: tests/test_gpu_round3.py), so the bound
        # there is the reference's own distance from exact arithmetic (x 1.1), and the device must sit within 1e-5 of exact
        a = frames_ref[f][1].astype(np.float64)
        exact = np.abs(np.polyval(a[::-1], np.exp(-2j * np.pi * k / n))) ** 2 / n
        e_ref = max(rel_err(ref[f][fin], exact[fin]))
        e_dev = max(rel_err(inv[fin], exact[fin]))
        assert e <= max(bound, 1.1 * e_ref), (n, f, e, bound, spread, e_ref)
        if n > 16384:
            assert e_dev <= 1e-5 and e_ref > 1e-5, (n, f, e_dev, e_ref)
