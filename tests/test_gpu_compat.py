"""GPU tests of libglfer_compat.so: glfer's own entry points (fft.h:77-83, mtm.h:47-49,
avg.h:38-43), driven hop by hop the way source.c:130-158 and g_main.c:1109-1183 drive them,
checked against the CPU oracle run over the same stream."""
import ctypes as C
import os

import numpy as np
import pytest

from _signals import rel_err, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-5


class FftParams(C.Structure):       # fft.h:52-63
    _fields_ = [("inbuf_audio", C.POINTER(C.c_float)), ("inbuf_fft", C.POINTER(C.c_float)),
                ("outbuf", C.POINTER(C.c_float)), ("n", C.c_int), ("window", C.POINTER(C.c_float)),
                ("window_type", C.c_int), ("overlap", C.c_float), ("a", C.c_float), ("limiter", C.c_int),
                ("sub_mean", C.c_int)]


class MtmParams(C.Structure):       # mtm.h:36-44
    _fields_ = [("fft", FftParams), ("window", C.POINTER(C.POINTER(C.c_double))), ("sig", C.POINTER(C.c_double)),
                ("w", C.c_float), ("kmax", C.c_int)]


class AvgData(C.Structure):         # avg.h:28-36
    _fields_ = [("avgwidth", C.c_int), ("avgdepth", C.c_int), ("effdepth", C.c_int), ("avg", C.POINTER(C.c_double)),
                ("cum", C.POINTER(C.c_double)), ("avgarray", C.POINTER(C.POINTER(C.c_double)))]


@pytest.fixture(scope="module")
def compat(lib):
    import torch
    assert torch.cuda.is_available()
    L = C.CDLL(os.path.join(ROOT, "glfer_amd", "lib", "libglfer_compat.so"))
    L.update_avg_plain.restype = C.c_double
    L.update_avg_sumextreme.restype = C.c_double
    L.update_avg_sumavg.restype = C.c_double
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _set(compat, name, value):
    C.c_int.in_dll(compat, name).value = value


@pytest.mark.parametrize("autoscale", [1, 0])
def test_fft_do_hop_by_hop(compat, oracle, autoscale):
    n, ovl, win = 1024, 0.5, 0          # Hanning
    h = oracle.hop(n, ovl)
    x = synth(12 * h, fs=8000.0, seed=3) + np.float32(0.05)
    _set(compat, "glfer_compat_autoscale", autoscale)        # opt.autoscale -> sub_mean (fft.c:186)
    _set(compat, "glfer_compat_first_buffer", 1)             # g_main.c:990
    p = FftParams(n=n, window_type=win, overlap=ovl, a=0.0, limiter=0)
    compat.fft_init(C.byref(p))
    assert p.sub_mean == autoscale
    assert np.array_equal(np.ctypeslib.as_array(p.window, (n,)), oracle.window(win, n))
    psd = np.empty(n // 2 + 1, np.float32)
    got = []
    for f in range(12):
        hop = x[f * h:(f + 1) * h].copy()
        compat.fft_do(_fp(hop), C.byref(p))
        compat.fft_psd(_fp(psd), None, C.byref(p))
        got.append(psd.copy())
        if autoscale:                                        # the drawer clears it only then (g_main.c:1111-1120)
            _set(compat, "glfer_compat_first_buffer", 0)
    want = oracle.spectrogram_fft(x, n, ovl, win, sub_mean=autoscale, history_mode=0 if autoscale else 1)
    worst = max(max(rel_err(g, w)) for g, w in zip(got, want))
    assert worst < TOL, worst
    # what fft_do leaves in outbuf: the halfcomplex spectrum of the last windowed frame
    hc = np.ctypeslib.as_array(p.outbuf, (n,)).copy()
    chk = np.empty(n // 2 + 1, np.float32)
    chk[0] = hc[0] ** 2 / n
    chk[1:n // 2] = (hc[1:n // 2] ** 2 + hc[n - 1:n // 2:-1] ** 2) / n
    chk[n // 2] = hc[n // 2] ** 2 / n
    assert max(rel_err(chk, want[-1])) < TOL
    compat.fft_close(C.byref(p))
    assert not p.window and not p.inbuf_audio


def test_mtm_do_hop_by_hop(compat, oracle):
    n, ovl, nw, kmax = 4096, 0.75, 2.5, 4
    h = oracle.hop(n, ovl)
    x = synth(9 * h, seed=8)
    _set(compat, "glfer_compat_autoscale", 0)
    _set(compat, "glfer_compat_first_buffer", 1)
    p = MtmParams()
    p.fft.n, p.fft.window_type, p.fft.overlap, p.fft.a, p.fft.limiter = n, 5, ovl, 0.0, 0   # source.c:343-347
    p.w, p.kmax = nw, kmax
    compat.mtm_init(C.byref(p))
    v, sig = oracle.dpss(n, kmax, nw)
    assert np.array_equal(np.array([p.sig[k] for k in range(kmax + 1)]), sig)
    assert all(p.window[i + 1][j] == v[j, i] for i in (0, 1, 2047, 4095) for j in range(kmax + 1))   # [1..n][0..kmax]
    psd = np.empty(n // 2 + 1, np.float32)
    got = []
    for f in range(9):
        hop = x[f * h:(f + 1) * h].copy()
        compat.mtm_do(_fp(hop), _fp(psd), None, C.byref(p))
        got.append(psd.copy())
        _set(compat, "glfer_compat_first_buffer", 0)
    want = oracle.spectrogram_mtm(x, n, ovl, nw, kmax)
    assert max(max(rel_err(g, w)) for g, w in zip(got, want)) < TOL
    compat.mtm_close(C.byref(p))


def test_compute_floor_and_update_avg(compat, oracle):
    g = np.load(os.path.join(ROOT, "tests", "golden", "avg_floor_fft1024.npz"))
    psd = g["psd"]
    s, fl, pk, pb = C.c_float(), C.c_float(), C.c_float(), C.c_uint()
    for f in range(psd.shape[0]):
        row = psd[f].copy()
        compat.compute_floor(_fp(row), 513, C.byref(s), C.byref(fl), C.byref(pk), C.byref(pb))
        w = g["floor"][f]
        assert s.value == w[0] and pk.value == w[2] and pb.value == w[3]
        assert abs(fl.value / w[1] - 1) < 2e-6
    depth, lo, hi = int(g["depth"]), int(g["minbin"]), int(g["maxbin"])
    for mode, fn in (("plain", compat.update_avg_plain), ("sumextreme", compat.update_avg_sumextreme),
                     ("sumavg", compat.update_avg_sumavg)):
        a = AvgData()
        compat.init_avg(C.byref(a))
        compat.alloc_avg(C.byref(a), 1024, depth)             # width = data_block_size (source.c:312)
        peak, var = C.c_int(-1), C.c_double(0)
        for f in range(psd.shape[0]):
            row = psd[f].copy()
            if mode == "plain":
                r = fn(C.byref(a), 513, _fp(row), lo, hi, C.byref(peak))
            elif mode == "sumextreme":
                r = fn(C.byref(a), 513, _fp(row), 1, lo, hi, C.byref(peak))
            else:
                r = fn(C.byref(a), 513, _fp(row), 1, lo, hi, C.byref(peak), C.byref(var))
            want_avg = g["%s_max%d_avg" % (mode, 0 if mode == "plain" else 1)][f]
            want_ret = g["%s_max%d_ret" % (mode, 0 if mode == "plain" else 1)][f]
            got_avg = np.ctypeslib.as_array(a.avg, (513,))
            assert np.allclose(got_avg, want_avg, rtol=1e-11, atol=0)
            assert abs(r / want_ret[0] - 1) < 1e-11 and peak.value == want_ret[1]
            assert a.effdepth == min(f + 1, depth)
        compat.delete_avg(C.byref(a))


class HparmaParams(C.Structure):    # hparma.h:25-32
    _fields_ = [("fft", FftParams), ("t", C.c_int), ("p_e", C.c_int), ("q_e", C.c_int)]


def test_hparma_do_hop_by_hop(compat, oracle):
    n, ovl, t, p_e = 4096, 0.5, 128, 32
    h = oracle.hop(n, ovl)
    x = synth(5 * h, seed=14)
    _set(compat, "glfer_compat_autoscale", 0)
    _set(compat, "glfer_compat_first_buffer", 1)
    p = HparmaParams()
    p.fft.n, p.fft.window_type, p.fft.overlap, p.fft.a, p.fft.limiter = n, 5, ovl, 0.0, 0   # source.c:368-372
    p.t, p.p_e, p.q_e = t, p_e, -1                                                          # source.c:373-375
    compat.hparma_init(C.byref(p))
    psd = np.empty(n // 2 + 1, np.float32)
    want = oracle.spectrogram_hparma(x, n, ovl, t, p_e)
    for f in range(5):
        hop = x[f * h:(f + 1) * h].copy()
        compat.hparma_do(_fp(hop), _fp(psd), None, C.byref(p))
        _set(compat, "glfer_compat_first_buffer", 0)
        # BASELINE config 5's matrix shape (t = 128, p_e = 32, N = 4096): the usual 1e-5 on |A(f)|^2 / N (tests/_spread.py)
        assert max(rel_err(1.0 / psd[:n // 2].astype(np.float64), 1.0 / want[f, :n // 2].astype(np.float64))) < 1e-5
    compat.hparma_close(C.byref(p))


@pytest.mark.parametrize("mode,n,overlap", [("fft", 1024, 0.5), ("mtm", 4096, 0.75)])
def test_c_program_against_the_shim(oracle, tmp_path, mode, n, overlap):
    """A C program that calls the reference's own entry points hop by hop, compiled with gcc and
    linked against libglfer_compat.so (no Python, no ctypes in the data path), against the oracle."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "glfer_amd", "lib")
    exe = tmp_path / "c_compat_demo"
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c_compat_demo.c"), "-o", str(exe), "-L", libdir, "-lglfer_compat",
                    "-lglfer_hip", "-Wl,-rpath," + libdir], check=True)
    h = oracle.hop(n, overlap)
    x = synth(7 * h, seed=31) + np.float32(0.03)
    (tmp_path / "in.f32").write_bytes(x.tobytes())
    subprocess.run([str(exe), mode, str(n), repr(overlap), str(tmp_path / "in.f32"), str(tmp_path / "out.f32")],
                   check=True, timeout=120)
    got = np.fromfile(tmp_path / "out.f32", np.float32).reshape(7, n // 2 + 1)
    if mode == "fft":
        want = oracle.spectrogram_fft(x, n, overlap, 0, sub_mean=1)       # autoscale on: mean removal (fft.c:186)
    else:
        want = oracle.spectrogram_mtm(x, n, overlap, 2.5, 4, sub_mean=1)
    for f in range(7):
        assert np.abs(got[f] - want[f]).max() <= TOL * want[f].max(), f


class LmpParams(C.Structure):       # lmp.h:36-45
    _fields_ = [("fft", FftParams), ("avg", C.c_int), ("window", C.POINTER(C.POINTER(C.c_double))),
                ("sig", C.POINTER(C.c_double)), ("w", C.c_float), ("kmax", C.c_int)]


def test_lmp_do_hop_by_hop(compat, oracle):
    """lmp_init / lmp_do / lmp_close (lmp.c:59-194) as source.c:390-398, 155-157 drives them, over more
    hops than the shim's device history holds (it slides), against the oracle within the statistic's
    conditioning (tests/test_gpu_round2.py::test_lmp_vs_oracle measures it)."""
    n, ovl, nl = 1024, 0.5, 4
    h = oracle.hop(n, ovl)
    frames = 27
    x = synth(frames * h, fs=8000.0, seed=19)
    _set(compat, "glfer_compat_autoscale", 1)
    _set(compat, "glfer_compat_first_buffer", 1)
    p = LmpParams()
    p.fft.n, p.fft.window_type, p.fft.overlap, p.fft.a, p.fft.limiter, p.avg = n, 5, ovl, 0.0, 0, nl
    compat.lmp_init(C.byref(p))
    want = oracle.spectrogram_lmp(x, n, ovl, nl, sub_mean=1)
    # the bound of tests/test_gpu_round2.py::test_lmp_vs_oracle: 3 x the largest movement the oracle's own statistic makes in some
    # frame of this stream when its input moves by one float ulp (the statistic divides by a per-bin variance), never below 1e-5
    from _spread import ulp_perturbations
    spread = 0.0
    for xp in ulp_perturbations(x, 6, seed=n):
        wp = oracle.spectrogram_lmp(xp, n, ovl, nl, sub_mean=1)
        spread = max(spread, float((np.abs(wp.astype(np.float64) - want).max(axis=1) / want.max(axis=1)).max()))
    bound = max(1e-5, 3.0 * spread)
    print("lmp_do: oracle 1-ulp spread %.2e, bound %.2e" % (spread, bound))
    psd = np.empty(n // 2 + 1, np.float32)
    for f in range(frames):
        hop = x[f * h:(f + 1) * h].copy()
        compat.lmp_do(_fp(hop), _fp(psd), None, C.byref(p))
        _set(compat, "glfer_compat_first_buffer", 0)
        assert psd[0] == np.float32(1e-3)
        assert np.abs(psd.astype(np.float64) - want[f]).max() <= bound * want[f].max(), (f, np.abs(psd.astype(np.float64) - want[f]).max() / want[f].max(), bound)
    compat.lmp_close(C.byref(p))


def test_prepare_audio_leaves_the_windowed_frame(compat, oracle):
    """prepare_audio() alone (fft.h:77): inbuf_audio = the assembled frame, inbuf_fft = RA9MB + window
    applied to it (fft.c:98-149) -- what lmp.c:101-120 and the scope (g_scope.c:194-197) read."""
    n, ovl = 1024, 0.75
    h = oracle.hop(n, ovl)
    x = synth(6 * h, fs=8000.0, seed=8)
    _set(compat, "glfer_compat_autoscale", 0)
    _set(compat, "glfer_compat_first_buffer", 1)
    p = FftParams(n=n, window_type=1, overlap=ovl, a=0.001, limiter=0)           # Blackman + RA9MB
    compat.fft_init(C.byref(p))
    w = oracle.window(1, n)
    frame = np.zeros(n, np.float32)
    for f in range(6):
        hop = x[f * h:(f + 1) * h].copy()
        compat.prepare_audio(_fp(hop), C.byref(p))
        _set(compat, "glfer_compat_first_buffer", 0)
        frame = np.concatenate([frame[h:], x[f * h:(f + 1) * h]])
        assert np.array_equal(np.ctypeslib.as_array(p.inbuf_audio, (n,)), frame)
        want = (frame / (np.float32(0.001) + frame * frame)) * w                # float ops, fft.c:127-136
        assert np.array_equal(np.ctypeslib.as_array(p.inbuf_fft, (n,)).view(np.uint32), want.astype(np.float32).view(np.uint32))
    compat.fft_close(C.byref(p))


def test_update_avg_fills_cum_and_avgarray(compat, oracle):
    """avgdata->cum (the sliding sums) and avgdata->avgarray (the per-bin shift registers) after
    every update, against the oracle's state (avg.c:114-127)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "avg_floor_fft1024.npz"))
    psd, depth, lo, hi = g["psd"], int(g["depth"]), int(g["minbin"]), int(g["maxbin"])
    a = AvgData()
    compat.init_avg(C.byref(a))
    compat.alloc_avg(C.byref(a), 1024, depth)
    ref = oracle.Averager(1024, depth)
    peak = C.c_int(-1)
    for f in range(psd.shape[0]):
        row = psd[f].copy()
        compat.update_avg_plain(C.byref(a), 513, _fp(row), lo, hi, C.byref(peak))
        ref.update("plain", psd[f], lo, hi, n=513)
        want_cum = np.ctypeslib.as_array(ref._a.cum, (1024,))[lo:hi]
        assert np.array_equal(np.ctypeslib.as_array(a.cum, (1024,))[lo:hi], want_cum), f
        for b in (lo, (lo + hi) // 2, hi - 1):
            regs = np.ctypeslib.as_array(a.avgarray[b], (depth,))
            hist = [float(psd[g_][b]) if g_ >= 0 else 0.0 for g_ in range(f - depth + 1, f + 1)]
            assert np.array_equal(regs, np.array(hist)), (f, b)
    compat.delete_avg(C.byref(a))
