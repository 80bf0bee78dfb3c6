"""ctypes binding of the parity oracle (oracle/liboracle.so) and, when it has
been built, of the reference's own objects (oracle/_ref/libglfer_ref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under glfer_amd/ may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libglfer_ref.so")

WINDOWS = {"hanning": 0, "blackman": 1, "gaussian": 2, "welch": 3,
           "bartlett": 4, "rectangular": 5, "hamming": 6, "kaiser": 7}

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(ref=True):
    """Compile liboracle.so (and _ref/ when the reference tree is present)."""
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.run(["make", "-C", _HERE] + targets, check=True,
                   stdout=subprocess.DEVNULL)


def _load():
    if not os.path.exists(_LIB):
        build(ref=False)
    lib = C.CDLL(_LIB)
    lib.go_bessel_i0.restype = C.c_double
    lib.go_bessel_i0.argtypes = [C.c_double]
    lib.go_window.argtypes = [C.c_int, C.c_int, _f32p]
    lib.go_hop.restype = C.c_int
    lib.go_hop.argtypes = [C.c_int, C.c_float]
    lib.go_rfft_halfcomplex.argtypes = [_f32p, C.c_size_t]
    lib.go_psd.argtypes = [_f32p, C.c_int, _f32p]
    lib.go_dpss.restype = C.c_int
    lib.go_dpss.argtypes = [C.c_int, C.c_int, C.c_double, _f64p, _f64p]
    lib.go_floor.argtypes = [_f32p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                             C.POINTER(C.c_float), C.POINTER(C.c_uint)]
    lib.go_num_frames.restype = C.c_size_t
    lib.go_num_frames.argtypes = [C.c_size_t, C.c_int, C.c_float]
    lib.go_spectrogram_fft.argtypes = [_f32p, C.c_size_t, C.c_int, C.c_float, C.c_int,
                                       C.c_float, C.c_int, C.c_int, C.c_int, _f32p]
    lib.go_spectrogram_mtm.argtypes = [_f32p, C.c_size_t, C.c_int, C.c_float, C.c_double,
                                       C.c_int, C.c_int, C.c_int, _f32p]
    lib.go_spectrogram_hparma.argtypes = [_f32p, C.c_size_t, C.c_int, C.c_float, C.c_int, C.c_int,
                                          C.c_int, C.c_int, _f32p]
    lib.go_hparma_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, _f32p, C.c_int, _f32p, _f32p, C.POINTER(C.c_int)]
    lib.go_fft_state_init.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_float, C.c_int, C.c_int]
    lib.go_fft_state_free.argtypes = [C.c_void_p]
    lib.go_svd.restype = C.c_int
    lib.go_svd.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p]
    lib.go_wav_spectrogram.restype = C.c_size_t
    lib.go_wav_spectrogram.argtypes = [np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS"), C.c_size_t, C.c_int,
                                       C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                       C.c_double, C.c_int, C.c_size_t, _f32p]
    lib.go_spectrogram_lmp.argtypes = [_f32p, C.c_size_t, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, _f32p]
    lib.go_spectrogram_mtm_ftest.argtypes = [_f32p, C.c_size_t, C.c_int, C.c_float, C.c_double, C.c_int, C.c_int,
                                             C.c_int, C.c_int, _f32p, _f32p]
    lib.go_ftest_tables.argtypes = [C.c_int, C.c_int, _f64p, _f64p, _f32p, C.POINTER(C.c_float)]
    lib.go_pcm_u8_to_float.argtypes = [np.ctypeslib.ndpointer(np.uint8), C.c_size_t, _f32p]
    lib.go_pcm_s16_to_float.argtypes = [np.ctypeslib.ndpointer(np.int16), C.c_size_t, _f32p]
    return lib


_lib = _load()


class _GoAvg(C.Structure):
    _fields_ = [("width", C.c_int), ("depth", C.c_int), ("effdepth", C.c_int),
                ("avg", C.POINTER(C.c_double)), ("cum", C.POINTER(C.c_double)),
                ("ring", C.POINTER(C.c_double))]


_lib.go_avg_alloc.argtypes = [C.POINTER(_GoAvg), C.c_int, C.c_int]
_lib.go_avg_free.argtypes = [C.POINTER(_GoAvg)]
_lib.go_avg_plain.restype = C.c_double
_lib.go_avg_plain.argtypes = [C.POINTER(_GoAvg), C.c_int, _f32p, C.c_int, C.c_int,
                              C.POINTER(C.c_int)]
_lib.go_avg_sumextreme.restype = C.c_double
_lib.go_avg_sumextreme.argtypes = [C.POINTER(_GoAvg), C.c_int, _f32p, C.c_int, C.c_int,
                                   C.c_int, C.POINTER(C.c_int)]
_lib.go_avg_sumavg.restype = C.c_double
_lib.go_avg_sumavg.argtypes = [C.POINTER(_GoAvg), C.c_int, _f32p, C.c_int, C.c_int, C.c_int,
                               C.POINTER(C.c_int), C.POINTER(C.c_double)]


def bessel_i0(x):
    return _lib.go_bessel_i0(float(x))


def window(window_type, n):
    w = np.empty(n, np.float32)
    _lib.go_window(int(window_type), n, w)
    return w


def hop(n, overlap):
    return _lib.go_hop(n, C.c_float(overlap))


def num_frames(nsamples, n, overlap):
    return _lib.go_num_frames(nsamples, n, C.c_float(overlap))


def rfft_halfcomplex(x):
    d = np.ascontiguousarray(x, np.float32).copy()
    _lib.go_rfft_halfcomplex(d, d.size)
    return d


def psd_from_halfcomplex(hc):
    hc = np.ascontiguousarray(hc, np.float32)
    out = np.empty(hc.size // 2 + 1, np.float32)
    _lib.go_psd(hc, hc.size, out)
    return out


def dpss(n, kmax, nw):
    """(tapers[kmax+1][n] float64, sig[kmax+1] float64 = lambda-1)."""
    v = np.empty((kmax + 1, n), np.float64)
    sig = np.empty(kmax + 1, np.float64)
    err = _lib.go_dpss(n, kmax, float(nw), v, sig)
    if err:
        raise RuntimeError("go_dpss: Jacobi did not converge")
    return v, sig


def floor_stats(psd):
    psd = np.ascontiguousarray(psd, np.float32)
    s, f, p, b = C.c_float(), C.c_float(), C.c_float(), C.c_uint()
    _lib.go_floor(psd, psd.size, C.byref(s), C.byref(f), C.byref(p), C.byref(b))
    return s.value, f.value, p.value, b.value


def spectrogram_fft(stream, n, overlap, window_type=0, a=0.0, limiter=0, sub_mean=0,
                    history_mode=0):
    stream = np.ascontiguousarray(stream, np.float32)
    frames = num_frames(stream.size, n, overlap)
    out = np.empty((frames, n // 2 + 1), np.float32)
    _lib.go_spectrogram_fft(stream, stream.size, n, C.c_float(overlap), int(window_type),
                            C.c_float(a), int(limiter), int(sub_mean), int(history_mode), out)
    return out


def spectrogram_mtm(stream, n, overlap, nw, kmax, sub_mean=0, history_mode=0):
    stream = np.ascontiguousarray(stream, np.float32)
    frames = num_frames(stream.size, n, overlap)
    out = np.empty((frames, n // 2 + 1), np.float32)
    _lib.go_spectrogram_mtm(stream, stream.size, n, C.c_float(overlap), float(nw), int(kmax),
                            int(sub_mean), int(history_mode), out)
    return out


def spectrogram_hparma(stream, n, overlap, t, p_e, sub_mean=0, history_mode=0):
    stream = np.ascontiguousarray(stream, np.float32)
    frames = num_frames(stream.size, n, overlap)
    out = np.empty((frames, n // 2 + 1), np.float32)
    _lib.go_spectrogram_hparma(stream, stream.size, n, C.c_float(overlap), int(t), int(p_e),
                               int(sub_mean), int(history_mode), out)
    return out


def spectrogram_lmp(stream, n, overlap, nl, sub_mean=0, history_mode=0):
    """lmp_do (lmp.c:101-181) once per hop: the detection statistic of every frame."""
    stream = np.ascontiguousarray(stream, np.float32)
    frames = num_frames(stream.size, n, overlap)
    out = np.empty((frames, n // 2 + 1), np.float32)
    _lib.go_spectrogram_lmp(stream, stream.size, n, C.c_float(overlap), int(nl), int(sub_mean),
                            int(history_mode), out)
    return out


def spectrogram_mtm_ftest(stream, n, overlap, nw, kmax, sub_mean=0, history_mode=0, mu_live=1):
    """mtm_do with its harmonic F-test side computation (mtm.c:165-174, 203-233): (psd, ftest).
    mu_live=0 restates the reference build without FFTW, where `mu` is never written."""
    stream = np.ascontiguousarray(stream, np.float32)
    frames = num_frames(stream.size, n, overlap)
    psd = np.empty((frames, n // 2 + 1), np.float32)
    ft = np.empty((frames, n // 2 + 1), np.float32)
    _lib.go_spectrogram_mtm_ftest(stream, stream.size, n, C.c_float(overlap), float(nw), int(kmax),
                                  int(sub_mean), int(history_mode), int(mu_live), psd, ft)
    return psd, ft


def ftest_tables(n, kmax, tapers):
    """mtm.c:76-83, 124-136: (U0[kmax+1] float64, hn[n] float32, sum_U0_sqr float32)."""
    U0 = np.empty(kmax + 1, np.float64)
    hn = np.empty(n, np.float32)
    s = C.c_float(0.0)
    _lib.go_ftest_tables(n, kmax, np.ascontiguousarray(tapers, np.float64), U0, hn, C.byref(s))
    return U0, hn, np.float32(s.value)


def wav_spectrogram(pcm, bits, mode, n, overlap, window_type=0, a=0.0, limiter=0, sub_mean=0,
                    history_mode=0, nw=0.0, kmax=0, max_frames=None):
    """The file source: wav_read() blocks (wav_fmt.c:81-121, incl. the trailing partial block over
    the stale tail of the previous one) through fft_do+fft_psd (mode "fft") or mtm_do ("mtm")."""
    raw = np.ascontiguousarray(pcm).view(np.uint8).reshape(-1)
    h = hop(n, overlap)
    bsz = h * bits // 8
    frames = -(-raw.size // bsz)
    if max_frames is not None:
        frames = min(frames, max_frames)
    out = np.empty((frames, n // 2 + 1), np.float32)
    got = _lib.go_wav_spectrogram(raw, raw.size, bits, 1 if mode == "mtm" else 0, n, C.c_float(overlap),
                                  int(window_type), C.c_float(a), int(limiter), int(sub_mean),
                                  int(history_mode), float(nw), int(kmax), frames, out)
    assert got == frames, (got, frames)
    return out


class _GoWav(C.Structure):
    _fields_ = [("pcm", C.c_void_p), ("nbytes", C.c_size_t), ("pos", C.c_size_t), ("bits", C.c_int),
                ("out_len", C.c_int), ("buff", C.POINTER(C.c_float))]


def wav_blocks(pcm, bits, hop_len, mutate=None):
    """Every block wav_read() hands out for these PCM bytes (copies).  mutate(block) stands for
    what the estimator does to the reader's buffer between reads (mean removal, fft.c:93-95)."""
    raw = np.ascontiguousarray(pcm).view(np.uint8).reshape(-1)
    _lib.go_wav_open.argtypes = [C.POINTER(_GoWav), C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    _lib.go_wav_read.argtypes = [C.POINTER(_GoWav), C.POINTER(C.POINTER(C.c_float))]
    _lib.go_wav_close.argtypes = [C.POINTER(_GoWav)]
    w = _GoWav()
    _lib.go_wav_open(C.byref(w), raw.ctypes.data, raw.size, bits, hop_len)
    out = []
    buf = C.POINTER(C.c_float)()
    while _lib.go_wav_read(C.byref(w), C.byref(buf)):
        blk = np.ctypeslib.as_array(buf, shape=(hop_len,))
        out.append(blk.copy())
        if mutate is not None:
            mutate(blk)
    _lib.go_wav_close(C.byref(w))
    return out


class _GoFftState(C.Structure):
    _fields_ = [("n", C.c_int), ("overlap", C.c_float), ("window_type", C.c_int), ("a", C.c_float),
                ("limiter", C.c_int), ("sub_mean", C.c_int), ("window", C.c_void_p),
                ("inbuf_audio", C.c_void_p), ("inbuf_fft", C.c_void_p)]


def hparma_frames(stream, n, overlap, t, p_e, sub_mean=0):
    """Per frame: (psd[n/2+1], AR vector a[p_e+1], rank p) -- hparma_do with its intermediates."""
    stream = np.ascontiguousarray(stream, np.float32)
    h = hop(n, overlap)
    frames = num_frames(stream.size, n, overlap)
    st = _GoFftState()
    _lib.go_fft_state_init(C.byref(st), n, C.c_float(overlap), WINDOWS["rectangular"], C.c_float(0.0), 0, sub_mean)
    out = []
    for f in range(frames):
        hopbuf = stream[f * h:(f + 1) * h].copy()
        psd = np.empty(n // 2 + 1, np.float32)
        a = np.empty(p_e + 1, np.float32)
        rank = C.c_int(0)
        _lib.go_hparma_frame(C.byref(st), t, p_e, hopbuf, 1 if f == 0 else 0, psd, a, C.byref(rank))
        out.append((psd, a, rank.value))
    _lib.go_fft_state_free(C.byref(st))
    return out


def svd(A):
    A = np.ascontiguousarray(A, np.float32).copy()
    nrow, ncol = A.shape
    S = np.empty(ncol, np.float32)
    Q = np.empty((ncol, ncol), np.float32)
    rc = _lib.go_svd(A, nrow, ncol, S, Q)
    return rc, A, S, Q


PALETTES = {"hsv": 0, "thresh": 1, "cool": 2, "hot": 3, "bw": 4, "bone": 5, "copper": 6, "otd": 7}


class _GoDisplayState(C.Structure):
    _fields_ = [("scale_log", C.c_int), ("autoscale", C.c_int), ("overlap", C.c_float),
                ("max_level_db", C.c_float), ("min_level_db", C.c_float), ("thr_level", C.c_float),
                ("first_buffer", C.c_int), ("display_max_lvl", C.c_float),
                ("display_min_lvl", C.c_float)]


def palette(p_n):
    """g_main.c:651-762 set_palette -> uint8 [256][3]."""
    tab = np.zeros(768, np.uint8)
    _lib.go_palette.argtypes = [C.c_int, np.ctypeslib.ndpointer(np.uint8)]
    _lib.go_palette(int(p_n), tab)
    return tab.reshape(256, 3)


def display(psd, stats, palette_id=0, scale_log=True, autoscale=True, overlap=0.0,
            max_level_db=-10.0, min_level_db=-60.0, thr_level=0.0, first_buffer=True,
            state=(0.0, 0.0)):
    """The waterfall loop of g_main.c:1099-1236 over rows of `psd` (float32 PSD, or float64 for
    the averaged spectrum) with compute_floor outputs `stats` [frames][>=2] = (sig, floor, ..).
    Returns rgb uint8 [frames][bins][3], lev int16 [frames][bins], levels float32 [frames][2]
    and the final (first_buffer, display_max_lvl, display_min_lvl)."""
    psd = np.ascontiguousarray(psd)
    frames, n = psd.shape
    tab = np.ascontiguousarray(palette(palette_id).reshape(-1))
    st = _GoDisplayState(int(scale_log), int(autoscale), overlap, max_level_db, min_level_db,
                         thr_level, int(first_buffer), state[0], state[1])
    rgb = np.zeros((frames, n, 3), np.uint8)
    lev = np.zeros((frames, n), np.int16)
    levels = np.zeros((frames, 2), np.float32)
    _lib.go_display_column.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                       C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    is_d = psd.dtype == np.float64
    if not is_d:
        psd = psd.astype(np.float32, copy=False)
    for f in range(frames):
        row = psd[f].ctypes.data
        _lib.go_display_column(C.byref(st), None if is_d else row, row if is_d else None, n,
                               float(stats[f][0]), float(stats[f][1]), tab.ctypes.data,
                               rgb[f].ctypes.data, lev[f].ctypes.data, levels[f].ctypes.data)
    return rgb, lev, levels, (st.first_buffer, st.display_max_lvl, st.display_min_lvl)


def pcm_u8_to_float(b):
    b = np.ascontiguousarray(b, np.uint8)
    out = np.empty(b.size, np.float32)
    _lib.go_pcm_u8_to_float(b, b.size, out)
    return out


def pcm_s16_to_float(s):
    s = np.ascontiguousarray(s, np.int16)
    out = np.empty(s.size, np.float32)
    _lib.go_pcm_s16_to_float(s, s.size, out)
    return out


class Averager:
    """avg.c state machine (go_avg); modes 'plain' | 'sumextreme' | 'sumavg'."""

    def __init__(self, width, depth):
        self._a = _GoAvg()
        _lib.go_avg_alloc(C.byref(self._a), width, depth)
        self.width = width

    def __del__(self):
        try:
            _lib.go_avg_free(C.byref(self._a))
        except Exception:
            pass

    def update(self, mode, psd, minbin, maxbin, max0=0, n=None):
        psd = np.ascontiguousarray(psd, np.float32)
        n = self.width if n is None else n
        peak = C.c_int(-1)
        var = C.c_double(0.0)
        if mode == "plain":
            r = _lib.go_avg_plain(C.byref(self._a), n, psd, minbin, maxbin, C.byref(peak))
        elif mode == "sumextreme":
            r = _lib.go_avg_sumextreme(C.byref(self._a), n, psd, max0, minbin, maxbin,
                                       C.byref(peak))
        elif mode == "sumavg":
            r = _lib.go_avg_sumavg(C.byref(self._a), n, psd, max0, minbin, maxbin,
                                   C.byref(peak), C.byref(var))
        else:
            raise ValueError(mode)
        avg = np.ctypeslib.as_array(self._a.avg, shape=(self.width,))[:n].copy()
        return r, avg, peak.value, var.value


# ---------------------------------------------------------------------------
# The reference's own objects (fft_radix2.c, g-l_dpss.c, avg.c, util.c compiled
# unmodified into oracle/_ref/).  Present in the build container; travels to the
# GPU box as a prebuilt .so; absent in a fresh clone without the reference tree.

def have_ref():
    return os.path.exists(_REF)


class _RefAvg(C.Structure):           # avg.h:28-36
    _fields_ = [("avgwidth", C.c_int), ("avgdepth", C.c_int), ("effdepth", C.c_int),
                ("avg", C.POINTER(C.c_double)), ("cum", C.POINTER(C.c_double)),
                ("avgarray", C.POINTER(C.POINTER(C.c_double)))]


class Ref:
    """Thin binding of the reference functions that build without GTK."""

    def __init__(self):
        if not have_ref():
            raise FileNotFoundError(_REF)
        r = C.CDLL(_REF)
        r.fft_real_radix2_transform.argtypes = [_f32p, C.c_size_t]
        r.bessel_I0.restype = C.c_double
        r.bessel_I0.argtypes = [C.c_double]
        r.dmatrix.restype = C.POINTER(C.POINTER(C.c_double))
        r.dmatrix.argtypes = [C.c_long] * 4
        r.free_dmatrix.argtypes = [C.POINTER(C.POINTER(C.c_double))] + [C.c_long] * 4
        r.dvector.restype = C.POINTER(C.c_double)
        r.dvector.argtypes = [C.c_long] * 2
        r.free_dvector.argtypes = [C.POINTER(C.c_double)] + [C.c_long] * 2
        r.matrix.restype = C.POINTER(C.POINTER(C.c_float))
        r.matrix.argtypes = [C.c_long] * 4
        r.free_matrix.argtypes = [C.POINTER(C.POINTER(C.c_float))] + [C.c_long] * 4
        r.gl_dpss.restype = C.c_int
        r.gl_dpss.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double,
                              C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_double),
                              C.POINTER(C.c_int)]
        r.compute_svd.restype = C.c_int
        r.compute_svd.argtypes = [C.POINTER(C.POINTER(C.c_float)), C.c_int, C.c_int,
                                  C.POINTER(C.c_float), C.POINTER(C.POINTER(C.c_float))]
        r.init_avg.argtypes = [C.POINTER(_RefAvg)]
        r.alloc_avg.argtypes = [C.POINTER(_RefAvg), C.c_int, C.c_int]
        r.delete_avg.argtypes = [C.POINTER(_RefAvg)]
        for name in ("update_avg_plain", "update_avg_sumextreme", "update_avg_sumavg"):
            getattr(r, name).restype = C.c_double
        r.update_avg_plain.argtypes = [C.POINTER(_RefAvg), C.c_int, _f32p, C.c_int, C.c_int,
                                       C.POINTER(C.c_int)]
        r.update_avg_sumextreme.argtypes = [C.POINTER(_RefAvg), C.c_int, _f32p, C.c_int,
                                            C.c_int, C.c_int, C.POINTER(C.c_int)]
        r.update_avg_sumavg.argtypes = [C.POINTER(_RefAvg), C.c_int, _f32p, C.c_int, C.c_int,
                                        C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        if hasattr(r, "wav_read"):               # wav_fmt.c, when the prebuilt _ref/ has it
            r.open_wav_file.restype = C.c_int
            r.open_wav_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
            r.wav_read.argtypes = [C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int)]
            r.close_wav_file.argtypes = []
        self.r = r

    def wav_blocks(self, path, hop_len, mutate=None):
        """open_wav_file + wav_read until n == 0 (wav_fmt.c:45-121, source.c:118-124): copies of
        every block handed out, and the sample rate the header parse reported."""
        speed = C.c_int(0)
        self.r.open_wav_file(os.fsencode(path), hop_len, C.byref(speed))
        out = []
        buf = C.POINTER(C.c_float)()
        n = C.c_int(0)
        while True:
            self.r.wav_read(C.byref(buf), C.byref(n))
            if n.value == 0:
                break
            blk = np.ctypeslib.as_array(buf, shape=(hop_len,))
            out.append(blk.copy())
            if mutate is not None:
                mutate(blk)
        self.r.close_wav_file()
        return out, speed.value

    def rfft_halfcomplex(self, x):
        d = np.ascontiguousarray(x, np.float32).copy()
        self.r.fft_real_radix2_transform(d, d.size)
        return d

    def bessel_i0(self, x):
        return self.r.bessel_I0(float(x))

    def dpss(self, n, kmax, nw):
        v = self.r.dmatrix(1, n, 0, kmax)          # mtm.c:118
        sig = self.r.dvector(0, kmax)              # mtm.c:119
        it = C.c_int(0)
        err = self.r.gl_dpss(n, kmax, n, float(nw), v, sig, C.byref(it))  # mtm.c:73
        tap = np.empty((kmax + 1, n), np.float64)
        for i in range(n):
            row = v[i + 1]
            for k in range(kmax + 1):
                tap[k, i] = row[k]
        s = np.array([sig[k] for k in range(kmax + 1)], np.float64)
        self.r.free_dmatrix(v, 1, n, 0, kmax)
        self.r.free_dvector(sig, 0, kmax)
        return err, tap, s

    def svd(self, A):
        A = np.asarray(A, np.float32)
        nrow, ncol = A.shape
        m = self.r.matrix(0, nrow - 1, 0, ncol - 1)
        q = self.r.matrix(0, ncol - 1, 0, ncol - 1)
        S = (C.c_float * ncol)()
        for i in range(nrow):
            for j in range(ncol):
                m[i][j] = float(A[i, j])
        rc = self.r.compute_svd(m, nrow, ncol, S, q)
        U = np.array([[m[i][j] for j in range(ncol)] for i in range(nrow)], np.float32)
        Q = np.array([[q[i][j] for j in range(ncol)] for i in range(ncol)], np.float32)
        Sv = np.array(list(S), np.float32)
        self.r.free_matrix(m, 0, nrow - 1, 0, ncol - 1)
        self.r.free_matrix(q, 0, ncol - 1, 0, ncol - 1)
        return rc, U, Sv, Q

    def averager(self, width, depth):
        return _RefAverager(self.r, width, depth)


class _RefAverager:
    def __init__(self, r, width, depth):
        self.r = r
        self.a = _RefAvg()
        r.init_avg(C.byref(self.a))
        r.alloc_avg(C.byref(self.a), width, depth)
        self.width = width

    def __del__(self):
        try:
            self.r.delete_avg(C.byref(self.a))
        except Exception:
            pass

    def update(self, mode, psd, minbin, maxbin, max0=0, n=None):
        psd = np.ascontiguousarray(psd, np.float32)
        n = self.width if n is None else n
        peak = C.c_int(-1)
        var = C.c_double(0.0)
        if mode == "plain":
            v = self.r.update_avg_plain(C.byref(self.a), n, psd, minbin, maxbin, C.byref(peak))
        elif mode == "sumextreme":
            v = self.r.update_avg_sumextreme(C.byref(self.a), n, psd, max0, minbin, maxbin,
                                             C.byref(peak))
        else:
            v = self.r.update_avg_sumavg(C.byref(self.a), n, psd, max0, minbin, maxbin,
                                         C.byref(peak), C.byref(var))
        avg = np.ctypeslib.as_array(self.a.avg, shape=(self.width,))[:n].copy()
        return v, avg, peak.value, var.value
