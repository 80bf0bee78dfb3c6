"""GPU parity tests (-m gpu) of the display mapping (main_window_draw, g_main.c:1099-1236):
glfer_hip_display_device against the oracle's restatement.

Integer/byte outputs are compared exactly.  The only device arithmetic that is not the same
IEEE operation as on the CPU is the double-precision log10 (device libm vs glibc, both within
an ulp of the true value): 10*log10(x) is TRUNCATED to a whole dB (levbuf is short), so a last-
bit difference can only show when 10*log10(x) lies within an ulp of an integer.  MISMATCH_MAX
bounds the fraction of such pixels; every other pixel must be identical."""
import os

import numpy as np
import pytest

from _signals import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
MISMATCH_MAX = 1e-4

CASES = {"log_auto_hsv": dict(palette=0, scale_type=2, autoscale=1, overlap=0.5),
         "lin_fixed_thresh": dict(palette=1, scale_type=0, autoscale=0, max_level_db=-25.0, min_level_db=-70.0,
                                  thr_level=20.0),
         "log_fixed_bone": dict(palette=5, scale_type=3, autoscale=0, max_level_db=-20.0, min_level_db=-80.0,
                                thr_level=5.0)}


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _same(got, want, what):
    bad = np.count_nonzero(got != want)
    assert bad <= MISMATCH_MAX * want.size, "%s: %d of %d differ" % (what, bad, want.size)
    return bad


@pytest.mark.parametrize("case", sorted(CASES))
def test_display_golden(lib, torch_cuda, case):
    g = np.load(os.path.join(GOLD, "display_fft1024.npz"))
    d = lib.Display(**CASES[case])
    rgb, lev, levels = lib.display(d, torch_cuda.from_numpy(g["psd"]).cuda(),
                                   torch_cuda.from_numpy(g["stats"]).cuda())
    levels = levels.cpu().numpy()
    want_levels = g[case + "_levels"]
    if CASES[case]["autoscale"]:
        # the recurrence itself is exact IEEE; the dB conversion of the levels is log10
        assert np.abs(levels[:, :2] - want_levels).max() <= 4e-6 * np.abs(want_levels).max()
    else:
        assert np.array_equal(levels[:, :2], want_levels)
    exact_levels = np.array_equal(levels[:, :2], want_levels)
    bad = _same(lev.cpu().numpy(), g[case + "_lev"], "levbuf")
    if exact_levels and bad == 0:
        assert np.array_equal(rgb.cpu().numpy(), g[case + "_rgb"])
    else:
        _same(rgb.cpu().numpy().reshape(-1, 3).view(np.dtype("V3")), g[case + "_rgb"].reshape(-1, 3).view(np.dtype("V3")), "rgb")
    assert d.first_buffer == (0 if CASES[case]["autoscale"] else 1)
    assert (d.display_max_lvl, d.display_min_lvl) == (levels[-1, 2], levels[-1, 3])


def test_display_averaged_source(lib, torch_cuda):
    g = np.load(os.path.join(GOLD, "display_fft1024.npz"))
    d = lib.Display(palette=7, scale_type=2, autoscale=1, overlap=0.5)
    rgb, lev, levels = lib.display(d, torch_cuda.from_numpy(g["avg"]).cuda(), torch_cuda.from_numpy(g["stats"]).cuda())
    _same(lev.cpu().numpy(), g["avg_log_auto_otd_lev"], "levbuf")
    _same(rgb.cpu().numpy().reshape(-1, 3).view(np.dtype("V3")),
          g["avg_log_auto_otd_rgb"].reshape(-1, 3).view(np.dtype("V3")), "rgb")


def test_linear_scale_pixels_are_exact(lib, oracle, torch_cuda):
    # linear scale: no log10 between the PSD and the pixel, so every byte must agree
    rng = np.random.default_rng(11)
    psd = (rng.random((300, 2049)) ** 8).astype(np.float32)
    psd[5, :40] = 0.0
    stats = np.array([oracle.floor_stats(r) for r in psd], np.float32)
    for autoscale, thr, pal in ((1, 0.0, 3), (0, 35.0, 6), (1, 10.0, 2)):
        d = lib.Display(palette=pal, scale_type=1, autoscale=autoscale, overlap=0.75, max_level_db=-3.0,
                        min_level_db=-40.0, thr_level=thr)
        rgb, lev, levels = lib.display(d, torch_cuda.from_numpy(psd).cuda(), torch_cuda.from_numpy(stats).cuda())
        w_rgb, w_lev, w_levels, st = oracle.display(psd, stats, palette_id=pal, scale_log=False,
                                                    autoscale=bool(autoscale), overlap=0.75, max_level_db=-3.0,
                                                    min_level_db=-40.0, thr_level=thr)
        assert np.array_equal(levels.cpu().numpy()[:, :2], w_levels)
        assert np.array_equal(rgb.cpu().numpy(), w_rgb)
        _same(lev.cpu().numpy(), w_lev, "levbuf")
        assert (d.first_buffer, d.display_max_lvl, d.display_min_lvl) == st


def test_autoscale_levels_over_many_chunks(lib, oracle, torch_cuda):
    """20 000 columns: the device walks the level recurrence in 4096-column chunks in parallel (each
    after a warm-up, every seam verified bit for bit and re-walked if needed); the result must be
    the sequential recurrence exactly, whatever the data does at the seams (steps of 6 decades,
    zeros, a constant stretch)."""
    rng = np.random.default_rng(5)
    frames, bins = 20000, 17
    stats = np.abs(rng.standard_normal((frames, 4))).astype(np.float32) + np.float32(1e-3)
    stats[:, 1] *= np.float32(0.01)
    stats[4000:4200] *= np.float32(1e6)              # a burst straddling the first seam
    stats[8190:8195] = 0.0                           # zeros on the second seam
    stats[12000:13000] = np.float32(0.25)            # constant input: the walk reaches a fixed point
    psd = (rng.random((frames, bins)) ** 2).astype(np.float32)
    for first_buffer, state in ((True, (0.0, 0.0)), (False, (3.5, 0.02))):
        d = lib.Display(palette=4, scale_type=0, autoscale=1, overlap=0.5, first_buffer=int(first_buffer))
        d.display_max_lvl, d.display_min_lvl = state
        rgb, lev, levels = lib.display(d, torch_cuda.from_numpy(psd).cuda(), torch_cuda.from_numpy(stats).cuda())
        w_rgb, w_lev, w_levels, st = oracle.display(psd, stats, palette_id=4, scale_log=False, autoscale=True,
                                                    overlap=0.5, first_buffer=first_buffer, state=state)
        assert np.array_equal(levels.cpu().numpy()[:, :2], w_levels)
        assert np.array_equal(rgb.cpu().numpy(), w_rgb)
        assert (d.first_buffer, d.display_max_lvl, d.display_min_lvl) == st


def test_state_carries_across_calls(lib, oracle, torch_cuda):
    rng = np.random.default_rng(12)
    psd = (rng.random((257, 513)) ** 4).astype(np.float32)
    stats = np.array([oracle.floor_stats(r) for r in psd], np.float32)
    P, S = torch_cuda.from_numpy(psd).cuda(), torch_cuda.from_numpy(stats).cuda()
    one = lib.Display(scale_type=0, autoscale=1, overlap=0.5)
    rgb, _, levels = lib.display(one, P, S)
    two = lib.Display(scale_type=0, autoscale=1, overlap=0.5)
    parts = [lib.display(two, P[a:b].contiguous(), S[a:b].contiguous()) for a, b in ((0, 1), (1, 130), (130, 257))]
    assert torch_cuda.equal(torch_cuda.cat([p[0] for p in parts]), rgb)
    assert torch_cuda.equal(torch_cuda.cat([p[2] for p in parts]), levels)
    assert (one.first_buffer, one.display_max_lvl, one.display_min_lvl) == \
           (two.first_buffer, two.display_max_lvl, two.display_min_lvl)


def test_whole_db_boundaries(lib, oracle, torch_cuda):
    # levbuf truncates 10*log10(x) to a whole dB: the kernel takes the logarithm in float and, inside a
    # guard band around every integer, compares x with the point where 10.0*log10(x) crosses it.  Rows packed with values ON and
    # next to the boundaries 10^(k/10) -- the exact float, its neighbours 1, 2, 3, 1000 and 20000 ulp
    # away (inside and just outside the band) -- over the whole float range, plus a log-uniform fill;
    # row lengths 4k, 4k+1, 4k+3 exercise the four-pixel groups and their tail.
    rng = np.random.default_rng(5)
    ks = np.arange(-370, 381)
    base = (10.0 ** (ks / 10.0)).astype(np.float32)
    vals = [base]
    for ulps in (1, 2, 3, 1000, 20000):
        for sign in (1, -1):
            vals.append((base.view(np.int32) + sign * ulps).view(np.float32))
    edge = np.concatenate(vals)
    edge = edge[np.isfinite(edge) & (edge > 0)]
    for n in (2048, 2049, 1027):
        rows = 24
        psd = (10.0 ** rng.uniform(-30, 5, (rows, n))).astype(np.float32)
        flat = psd.reshape(-1)
        idx = rng.permutation(flat.size)[: edge.size]
        flat[idx] = edge
        stats = np.array([oracle.floor_stats(r) for r in psd], np.float32)
        for scale_type, autoscale in ((2, 0), (2, 1), (0, 0)):
            d = lib.Display(palette=2, scale_type=scale_type, autoscale=autoscale, max_level_db=-5.0, min_level_db=-120.0,
                            thr_level=3.0)
            rgb, lev, _ = lib.display(d, torch_cuda.from_numpy(psd).cuda(), torch_cuda.from_numpy(stats).cuda())
            w_rgb, w_lev, _, _ = oracle.display(psd, stats, palette_id=2, scale_log=scale_type >= 2, autoscale=bool(autoscale),
                                                max_level_db=-5.0, min_level_db=-120.0, thr_level=3.0)
            # every bin here is a normal float: the whole-dB steps come from a table of the HOST libm's own
            # crossing points (host_tables.cpp log_thresholds), so levbuf is exact; the pixels may still
            # differ where the autoscale levels' own dB conversion (device log10) moved by an ulp
            assert np.array_equal(lev.cpu().numpy(), w_lev), "levbuf n=%d" % n
            _same(rgb.cpu().numpy().reshape(-1, 3).view(np.dtype("V3")), w_rgb.reshape(-1, 3).view(np.dtype("V3")), "rgb n=%d" % n)


def test_special_values(lib, oracle, torch_cuda):
    # zero, denormal, huge, inf and NaN bins; equal max/min levels (division by zero -> NaN/inf)
    psd = np.array([[0.0, 1e-45, 1e-20, 1.0, 3e38, np.inf, np.nan, 0.5]], np.float32)
    stats = np.array([[1.0, 1.0, 1.0, 3.0]], np.float32)          # sig == floor -> display span 0
    for scale_type, autoscale in ((2, 0), (0, 0), (2, 1), (0, 1)):
        d = lib.Display(palette=4, scale_type=scale_type, autoscale=autoscale, max_level_db=0.0, min_level_db=-100.0)
        rgb, lev, _ = lib.display(d, torch_cuda.from_numpy(psd).cuda(), torch_cuda.from_numpy(stats).cuda())
        w_rgb, w_lev, _, _ = oracle.display(psd, stats, palette_id=4, scale_log=scale_type >= 2,
                                            autoscale=bool(autoscale), max_level_db=0.0, min_level_db=-100.0)
        assert np.array_equal(lev.cpu().numpy(), w_lev), (scale_type, autoscale)
        assert np.array_equal(rgb.cpu().numpy(), w_rgb), (scale_type, autoscale)


def test_waterfall_end_to_end(lib, oracle, torch_cuda):
    # samples -> PSD -> floor -> display, all on the device, against the same chain in the oracle
    n, ovl = 1024, 0.5
    x = synth(200 * 512, fs=8000.0, seed=21)
    sp = lib.Spectrogram(lib.FftParams(n=n, overlap=ovl, window_type=0))
    psd = sp.run(torch_cuda.from_numpy(x).cuda())
    stats = lib.compute_floor(psd)
    d = lib.Display(palette=0, scale_type=2, autoscale=1, overlap=ovl)
    rgb, lev, _ = lib.display(d, psd, stats)
    w_psd = oracle.spectrogram_fft(x, n, ovl, 0)
    w_stats = np.array([oracle.floor_stats(r) for r in w_psd], np.float32)
    w_rgb, w_lev, _, _ = oracle.display(w_psd, w_stats, palette_id=0, scale_log=True, autoscale=True, overlap=ovl)
    # the PSDs agree to 1e-5 relative to the PEAK, far-down bins less tightly; a whole-dB cell
    # changes when such a difference straddles an integer dB.  Bins within 30 dB of the frame's
    # peak (relative PSD difference < 1e-2 there) must agree in all but ~1e-2/4.3 of the cells.
    lev, w_lev = lev.cpu().numpy(), w_lev
    strong = w_psd[:, ::-1] > w_psd.max(axis=1, keepdims=True) * 1e-3
    assert np.abs(lev.astype(int) - w_lev)[strong].max() <= 1
    assert np.count_nonzero((lev != w_lev) & strong) <= 1e-2 * np.count_nonzero(strong)
    # colour index moves by at most one whole dB's worth: 255/(display span in dB) + rounding
    assert np.count_nonzero((rgb.cpu().numpy() != w_rgb).any(axis=2) & (lev == w_lev)) <= 1e-3 * lev.size


@pytest.mark.parametrize("max_db,min_db,thr", [(0.0, -100.0, 0.0), (-5.5, -60.25, 7.0), (0.0, -253.0, 0.0), (0.0, -254.0, 0.0),
                                               (0.0, -255.0, 2.0), (10.0, -300.0, 0.0), (-20.0, -21.0, 0.0), (-20.0, -20.5, 50.0),
                                               (-30.0, -10.0, 0.0), (-30.0, -30.0, 0.0), (120.0, -120.0, 99.0)])
def test_log_scale_colour_table_edges(lib, oracle, torch_cuda, max_db, min_db, thr):
    """On the logarithmic scales a column's colours come from a table over 256 whole-dB levels from
    floor(display_min) - 1 on, valid when its last entry is past display_max; wider spans, reversed
    or equal levels map bin by bin.  Fixed levels (exact on both sides), spans of 1/2 dB, 253, 254,
    255 and 300 dB, fractional levels, thresholds, max < min (the reference then uses max/10) and
    max == min; bins over the whole float range and ON the whole-dB steps; float rows and
    averaged (double) rows; pixels and levbuf must be identical."""
    rng = np.random.default_rng(int(abs(max_db) * 7 + abs(min_db)))
    rows, n = 16, 1027
    psd = (10.0 ** rng.uniform(-37, 38, (rows, n))).astype(np.float32)
    ks = np.arange(-370, 381)
    psd.reshape(-1)[rng.permutation(rows * n)[: ks.size]] = (10.0 ** (ks / 10.0)).astype(np.float32)
    psd[3, :5] = [0.0, 1e-45, np.inf, np.nan, 3e38]
    stats = np.array([oracle.floor_stats(r) for r in psd], np.float32)
    avg = psd.astype(np.float64) * 1.0000001
    avg[5, :4] = [1e-300, 1e300, 1e-15, 1e-15]
    for scale_type in (lib.SCALE_LOG, lib.SCALE_LOG_MAX0):
        for src in (psd, avg):
            d = lib.Display(palette=1, scale_type=scale_type, autoscale=0, max_level_db=max_db, min_level_db=min_db, thr_level=thr)
            rgb, lev, levels = lib.display(d, torch_cuda.from_numpy(src).cuda(), torch_cuda.from_numpy(stats).cuda())
            w_rgb, w_lev, w_levels, _ = oracle.display(src, stats, palette_id=1, scale_log=True, autoscale=False, max_level_db=max_db,
                                                       min_level_db=min_db, thr_level=thr)
            assert np.array_equal(levels.cpu().numpy()[:, :2], w_levels)
            assert np.array_equal(lev.cpu().numpy(), w_lev), src.dtype
            assert np.array_equal(rgb.cpu().numpy(), w_rgb), src.dtype
