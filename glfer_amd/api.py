"""Host-side binding of the C-ABI in include/glfer_hip.h (glfer_amd/lib/libglfer_hip.so).

Everything computed here runs in the HIP library; torch is used only to own device
memory and streams.  There is no CPU fallback: if the library is missing, or HIP cannot
run, the calls raise.

Names follow the reference's estimator interface (fft.h:77-83, mtm.h:47-49, avg.h:38-43):
`FftParams` / `MtmParams` carry what change_params() copies out of `opt`
(source.c:320-325, 343-350) and `Spectrogram` is the per-hop loop of source.c:130-158
run over a whole stream.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GLFER_LIB_PATH: another build of the library (same-box A/B runs of kernel variants, tools/)
LIB_PATH = os.environ.get("GLFER_LIB_PATH") or os.path.join(_HERE, "lib", "libglfer_hip.so")

MODE_FFT, MODE_MTM, MODE_HPARMA, MODE_LMP = 0, 1, 2, 3
WAV_PARTIAL_TAIL = 1
WINDOWS = {"hanning": 0, "blackman": 1, "gaussian": 2, "welch": 3,
           "bartlett": 4, "rectangular": 5, "hamming": 6, "kaiser": 7}
SAMPLES_F32, SAMPLES_S16, SAMPLES_U8 = 0, 1, 2
HISTORY_ZERO_FIRST, HISTORY_ZERO_ALWAYS = 0, 1
SUBMEAN_OFF, SUBMEAN_EXACT, SUBMEAN_FAST = 0, 1, 2     # cfg.sub_mean: 1 = the reference's rows (fft.c:86-96 in its own summation order)
AVG_SUMAVG, AVG_PLAIN, AVG_SUMEXTREME = 1, 2, 3

# every symbol include/glfer_hip.h declares
EXPORTS = [
    "glfer_hip_plan_create", "glfer_hip_plan_destroy", "glfer_hip_hop", "glfer_hip_bins",
    "glfer_hip_num_tapers", "glfer_hip_num_frames", "glfer_hip_get_window", "glfer_hip_get_tapers",
    "glfer_hip_make_window", "glfer_hip_make_dpss", "glfer_hip_spectrogram_device",
    "glfer_hip_spectrum_device", "glfer_hip_spectrogram_host", "glfer_hip_wav_probe",
    "glfer_hip_spectrogram_wav", "glfer_hip_submean_device",
    "glfer_hip_floor_device",
    "glfer_hip_avg_device", "glfer_hip_palette", "glfer_hip_display_device", "glfer_hip_strerror", "glfer_hip_last_hip_error", "glfer_hip_version",
    "glfer_hip_frame_range", "glfer_hip_prepare_device", "glfer_hip_mtm_ftest_device", "glfer_hip_host_alloc",
    "glfer_hip_host_free", "glfer_hip_spectrogram_host_multi", "glfer_hip_spectrogram_wav_ex",
    "glfer_hip_avg_cum_device", "glfer_hip_waterfall_host", "glfer_hip_waterfall_device",
    "glfer_hip_spectrogram_host_workers",
    # round 3
    "glfer_hip_scratch_trim", "glfer_hip_scratch_held", "glfer_hip_scratch_limit", "glfer_hip_spectrogram_wav_range",
    "glfer_hip_spectrogram_wav_multi", "glfer_hip_spectrogram_wav_workers", "glfer_hip_levels_host",
    "glfer_hip_waterfall_map_device", "glfer_hip_waterfall_host_workers", "glfer_hip_waterfall_wav_workers",
    "glfer_hip_waterfall_wav_multi", "glfer_hip_submean_exact_device",
    # round 4
    "glfer_hip_numa_node_of_bus_id", "glfer_hip_numa_node_cpus", "glfer_hip_floor_device_pitched",
    # round 5
    "glfer_hip_abi_version", "glfer_hip_spectrogram_avg_device", "glfer_hip_workers_create", "glfer_hip_workers_destroy",
    "glfer_hip_workers_spectrogram_wav", "glfer_hip_workers_spectrogram_host",
]


class GlferHipError(RuntimeError):
    pass


class Config(C.Structure):
    """glfer_hip_config (include/glfer_hip.h)."""
    _fields_ = [("mode", C.c_int), ("n", C.c_int), ("overlap", C.c_float),
                ("window_type", C.c_int), ("limiter_a", C.c_float), ("enable_limiter", C.c_int),
                ("sub_mean", C.c_int), ("history_mode", C.c_int), ("mtm_w", C.c_float),
                ("mtm_k", C.c_int), ("sample_format", C.c_int), ("device", C.c_int),
                ("hparma_t", C.c_int), ("hparma_p_e", C.c_int), ("lmp_av", C.c_int), ("psd_pitch", C.c_int)]


class Display(C.Structure):
    """glfer_hip_display (include/glfer_hip.h): the options and the carried state of
    main_window_draw's level tracking and pixel mapping (g_main.c:1099-1236)."""
    _fields_ = [("scale_type", C.c_int), ("autoscale", C.c_int), ("overlap", C.c_float),
                ("max_level_db", C.c_float), ("min_level_db", C.c_float), ("thr_level", C.c_float),
                ("palette", C.c_int), ("first_buffer", C.c_int), ("display_max_lvl", C.c_float),
                ("display_min_lvl", C.c_float), ("psd_pitch", C.c_int)]

    def __init__(self, scale_type=2, autoscale=1, overlap=0.0, max_level_db=-10.0, min_level_db=-60.0,
                 thr_level=0.0, palette=0, first_buffer=1, psd_pitch=0):
        super().__init__(scale_type, autoscale, overlap, max_level_db, min_level_db, thr_level,
                         palette, first_buffer, 0.0, 0.0, psd_pitch)


SCALE_LIN, SCALE_LIN_MAX0, SCALE_LOG, SCALE_LOG_MAX0 = range(4)          # glfer.h:43
PALETTES = {"hsv": 0, "thresh": 1, "cool": 2, "hot": 3, "bw": 4, "bone": 5, "copper": 6, "otd": 7}


class Phases(C.Structure):
    """glfer_hip_phases: where a call through a workers handle spent its time (seconds; the largest value over the workers)."""
    _fields_ = [("setup_s", C.c_double), ("read_s", C.c_double), ("h2d_s", C.c_double), ("kernel_s", C.c_double),
                ("d2h_s", C.c_double), ("wall_s", C.c_double), ("chunks", C.c_uint)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class WavInfo(C.Structure):
    """glfer_wav_info (include/glfer_hip.h): the header fields of wav_fmt.h:34-52 that are used."""
    _fields_ = [("format", C.c_int), ("channels", C.c_int), ("sample_rate", C.c_int),
                ("bits_per_sample", C.c_int), ("data_offset", C.c_size_t), ("nsamples", C.c_size_t),
                ("data_bytes", C.c_size_t)]


_lib = None


def lib():
    """Load libglfer_hip.so (built by __graft_entry__.build()); fail loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (same soname as /opt/rocm's); load it first so that
    # this process ends up with ONE runtime shared by torch tensors/streams and our kernels
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise GlferHipError(
            "HIP extension not built: %s is missing (run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C glfer_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, sz = C.c_void_p, C.c_size_t
    L.glfer_hip_plan_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.glfer_hip_plan_destroy.argtypes = [vp]
    L.glfer_hip_plan_destroy.restype = None
    for f in ("glfer_hip_hop", "glfer_hip_bins", "glfer_hip_num_tapers"):
        getattr(L, f).argtypes = [vp]
    L.glfer_hip_num_frames.argtypes = [vp, sz]
    L.glfer_hip_num_frames.restype = sz
    L.glfer_hip_get_window.argtypes = [vp, vp]
    L.glfer_hip_get_tapers.argtypes = [vp, vp, vp]
    L.glfer_hip_make_window.argtypes = [C.c_int, C.c_int, vp]
    L.glfer_hip_make_dpss.argtypes = [C.c_int, C.c_int, C.c_double, vp, vp]
    L.glfer_hip_spectrogram_device.argtypes = [vp, vp, sz, sz, sz, vp, vp]
    L.glfer_hip_spectrum_device.argtypes = [vp, vp, sz, sz, sz, vp, vp, vp]
    L.glfer_hip_spectrogram_host.argtypes = [vp, vp, sz, vp, C.POINTER(sz)]
    L.glfer_hip_wav_probe.argtypes = [C.c_char_p, C.POINTER(WavInfo)]
    L.glfer_hip_spectrogram_wav.argtypes = [vp, C.c_char_p, vp, sz, C.POINTER(sz), sz]
    L.glfer_hip_submean_device.argtypes = [vp, vp, C.c_int, sz, C.c_int, vp]
    L.glfer_hip_submean_exact_device.argtypes = [vp, vp, C.c_int, sz, C.c_int, vp]
    L.glfer_hip_floor_device.argtypes = [vp, sz, C.c_int, vp, vp]
    L.glfer_hip_floor_device_pitched.argtypes = [vp, sz, C.c_int, C.c_int, vp, vp]
    L.glfer_hip_palette.argtypes = [C.c_int, vp]
    L.glfer_hip_display_device.argtypes = [C.POINTER(Display), vp, vp, vp, sz, C.c_int, vp, vp, vp, vp]
    L.glfer_hip_avg_device.argtypes = [C.c_int, vp, sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, vp, vp, vp]
    L.glfer_hip_frame_range.argtypes = [sz, C.c_uint, C.c_uint, C.POINTER(sz), C.POINTER(sz)]
    L.glfer_hip_frame_range.restype = None
    L.glfer_hip_prepare_device.argtypes = [vp, vp, sz, sz, sz, vp, vp]
    L.glfer_hip_mtm_ftest_device.argtypes = [vp, vp, sz, sz, sz, vp, C.c_int, vp]
    L.glfer_hip_host_alloc.argtypes = [sz]
    L.glfer_hip_host_alloc.restype = vp
    L.glfer_hip_host_free.argtypes = [vp]
    L.glfer_hip_host_free.restype = None
    L.glfer_hip_spectrogram_host_multi.argtypes = [C.POINTER(Config), C.c_uint, vp, sz, vp, C.POINTER(sz)]
    L.glfer_hip_spectrogram_host_workers.argtypes = [C.POINTER(Config), C.POINTER(C.c_int), C.c_int, vp, sz, vp, C.POINTER(sz)]
    L.glfer_hip_spectrogram_wav_ex.argtypes = [vp, C.c_char_p, vp, sz, C.POINTER(sz), sz, C.c_uint]
    L.glfer_hip_avg_cum_device.argtypes = [vp, sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    L.glfer_hip_waterfall_device.argtypes = [C.POINTER(Display), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, sz, C.c_int,
                                             vp, vp, vp, vp]
    L.glfer_hip_waterfall_host.argtypes = [vp, C.POINTER(Display), vp, sz, vp, vp, C.POINTER(sz)]
    cfgp, ip, dispp, szp = C.POINTER(Config), C.POINTER(C.c_int), C.POINTER(Display), C.POINTER(sz)
    L.glfer_hip_spectrogram_wav_range.argtypes = [vp, C.c_char_p, sz, sz, vp, szp, sz, C.c_uint]
    L.glfer_hip_spectrogram_wav_multi.argtypes = [cfgp, C.c_uint, C.c_char_p, vp, sz, szp, C.c_uint]
    L.glfer_hip_spectrogram_wav_workers.argtypes = [cfgp, ip, C.c_int, C.c_char_p, vp, sz, szp, C.c_uint]
    L.glfer_hip_waterfall_host_workers.argtypes = [cfgp, ip, C.c_int, dispp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   vp, sz, vp, vp, szp]
    L.glfer_hip_waterfall_wav_workers.argtypes = [cfgp, ip, C.c_int, dispp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.c_char_p, sz, vp, vp, szp, C.c_uint]
    L.glfer_hip_waterfall_wav_multi.argtypes = [cfgp, C.c_uint, dispp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_char_p, sz, vp, vp, szp, C.c_uint]
    L.glfer_hip_levels_host.argtypes = [dispp, vp, sz, vp, C.c_int]
    L.glfer_hip_waterfall_map_device.argtypes = [dispp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, sz, sz, C.c_int,
                                                 vp, vp, vp, vp]
    L.glfer_hip_numa_node_of_bus_id.argtypes = [C.c_char_p, C.c_char_p]
    L.glfer_hip_numa_node_cpus.argtypes = [C.c_int, C.c_char_p, vp, sz]
    L.glfer_hip_scratch_trim.argtypes = [C.c_int, sz]
    L.glfer_hip_scratch_trim.restype = sz
    L.glfer_hip_scratch_held.argtypes = [C.c_int]
    L.glfer_hip_scratch_held.restype = sz
    L.glfer_hip_scratch_limit.argtypes = [sz]
    L.glfer_hip_scratch_limit.restype = None
    for f in ("glfer_hip_strerror", "glfer_hip_last_hip_error", "glfer_hip_version"):
        getattr(L, f).restype = C.c_char_p
    L.glfer_hip_strerror.argtypes = [C.c_int]
    L.glfer_hip_abi_version.argtypes = []
    L.glfer_hip_workers_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_int), C.c_int, sz, C.POINTER(vp)]
    L.glfer_hip_workers_destroy.argtypes = [vp]
    L.glfer_hip_workers_destroy.restype = None
    L.glfer_hip_workers_spectrogram_wav.argtypes = [vp, C.c_char_p, vp, sz, C.POINTER(sz), C.c_uint, C.POINTER(Phases)]
    L.glfer_hip_workers_spectrogram_host.argtypes = [vp, vp, sz, vp, C.POINTER(sz), C.POINTER(Phases)]
    L.glfer_hip_spectrogram_avg_device.argtypes = [vp, vp, sz, sz, sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        L = lib()
        raise GlferHipError("%s failed: %s (%s)" % (
            what, L.glfer_hip_strerror(rc).decode(), L.glfer_hip_last_hip_error().decode()))


def version():
    return lib().glfer_hip_version().decode()


def make_window(window_type, n):
    """compute_window (fft.c:309-360), host code of the library."""
    w = np.empty(n, np.float32)
    _check(lib().glfer_hip_make_window(int(window_type), n, w.ctypes.data), "make_window")
    return w


def make_dpss(n, kmax, nw):
    """gl_dpss (g-l_dpss.c:288-347), host code of the library: (tapers[kmax+1][n], sig)."""
    v = np.empty((kmax + 1, n), np.float64)
    s = np.empty(kmax + 1, np.float64)
    _check(lib().glfer_hip_make_dpss(n, kmax, float(nw), v.ctypes.data, s.ctypes.data), "make_dpss")
    return v, s


class FftParams:
    """What source.c:320-325 sets before fft_init(): n, window_type, overlap, a, limiter."""

    def __init__(self, n=1024, window_type=7, overlap=0.0, a=0.0, limiter=0, sub_mean=0,
                 history_mode=HISTORY_ZERO_FIRST, sample_format=SAMPLES_F32, psd_pitch=0):
        self.mode = MODE_FFT
        self.psd_pitch = psd_pitch                      # cfg.psd_pitch: floats from one PSD row to the next (0 = dense)
        self.n, self.window_type, self.overlap = n, window_type, overlap
        self.a, self.limiter, self.sub_mean = a, limiter, sub_mean
        self.history_mode, self.sample_format = history_mode, sample_format
        self.w, self.kmax = 0.0, 0


class MtmParams:
    """What source.c:343-350 sets before mtm_init(): fft.{n,overlap}, w (=N*W), kmax."""

    def __init__(self, n=1024, overlap=0.0, w=4.0, kmax=7, sub_mean=0,
                 history_mode=HISTORY_ZERO_FIRST, sample_format=SAMPLES_F32, psd_pitch=0):
        self.mode = MODE_MTM
        self.psd_pitch = psd_pitch
        self.n, self.overlap, self.w, self.kmax = n, overlap, w, kmax
        self.window_type = WINDOWS["rectangular"]       # source.c:344
        self.a, self.limiter, self.sub_mean = 0.0, 0, sub_mean
        self.history_mode, self.sample_format = history_mode, sample_format


class HparmaParams:
    """What source.c:368-376 sets before hparma_init(): fft.{n,overlap}, t, p_e (q_e = -1)."""

    def __init__(self, n=4096, overlap=0.0, t=96, p_e=16, sub_mean=0, history_mode=HISTORY_ZERO_FIRST,
                 sample_format=SAMPLES_F32):
        self.mode = MODE_HPARMA
        self.n, self.overlap, self.t, self.p_e = n, overlap, t, p_e
        self.window_type = WINDOWS["rectangular"]       # source.c:369
        self.a, self.limiter, self.sub_mean = 0.0, 0, sub_mean
        self.history_mode, self.sample_format = history_mode, sample_format
        self.w, self.kmax = 0.0, 0


class LmpParams:
    """What source.c:390-398 sets before lmp_init(): fft.{n,overlap}, avg = opt.lmp_av (window
    rectangular; a and the limiter act on a buffer lmp_do overwrites, lmp.c:114-116)."""

    def __init__(self, n=1024, overlap=0.0, avg=4, sub_mean=0, history_mode=HISTORY_ZERO_FIRST,
                 sample_format=SAMPLES_F32):
        self.mode = MODE_LMP
        self.n, self.overlap, self.avg = n, overlap, avg
        self.window_type = WINDOWS["rectangular"]       # source.c:395
        self.a, self.limiter, self.sub_mean = 0.0, 0, sub_mean
        self.history_mode, self.sample_format = history_mode, sample_format
        self.w, self.kmax = 0.0, 0


_TORCH_DTYPES = None


def _torch():
    import torch
    return torch


def make_config(params, device=0):
    return Config(params.mode, params.n, params.overlap, params.window_type, params.a,
                  params.limiter, params.sub_mean, params.history_mode, params.w, params.kmax,
                  params.sample_format, device, getattr(params, "t", 0), getattr(params, "p_e", 0),
                  getattr(params, "avg", 0), getattr(params, "psd_pitch", 0))


class Spectrogram:
    """A plan (fft_init / mtm_init) bound to one GPU, plus the batched hot path."""

    def __init__(self, params, device=0):
        cfg = make_config(params, device)
        self._h = C.c_void_p()
        self._destroy = lib().glfer_hip_plan_destroy     # held here: module globals are gone by the time __del__ runs at exit
        _check(lib().glfer_hip_plan_create(C.byref(cfg), C.byref(self._h)), "glfer_hip_plan_create")
        self.params, self.device = params, device
        self.n = params.n
        self.hop = lib().glfer_hip_hop(self._h)
        self.bins = lib().glfer_hip_bins(self._h)
        self.pitch = getattr(params, "psd_pitch", 0) or self.bins     # floats from one row of run()'s output to the next
        self.ntapers = lib().glfer_hip_num_tapers(self._h)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._destroy(self._h)
            self._h.value = None

    __del__ = close

    def num_frames(self, nsamples):
        return lib().glfer_hip_num_frames(self._h, nsamples)

    def window(self):
        w = np.empty(self.n, np.float32)
        _check(lib().glfer_hip_get_window(self._h, w.ctypes.data), "get_window")
        return w

    def tapers(self):
        v = np.empty((self.ntapers, self.n), np.float64)
        s = np.empty(self.ntapers, np.float64)
        _check(lib().glfer_hip_get_tapers(self._h, v.ctypes.data, s.ctypes.data), "get_tapers")
        return v, s

    def _sample_dtype(self):
        torch = _torch()
        return {SAMPLES_F32: torch.float32, SAMPLES_S16: torch.int16,
                SAMPLES_U8: torch.uint8}[self.params.sample_format]

    def run(self, stream, first_frame=0, nframes=None, out=None, spectrum=False):
        """stream: 1-D torch tensor on this GPU.  Returns psd [nframes][bins] (and the
        halfcomplex spectra [nframes][n] when spectrum=True), launched on torch's current
        stream."""
        torch = _torch()
        assert stream.is_cuda and stream.dim() == 1 and stream.is_contiguous()
        assert stream.dtype == self._sample_dtype(), (stream.dtype, self._sample_dtype())
        total = self.num_frames(stream.numel())
        if nframes is None:
            nframes = total - first_frame
        if out is None:
            out = torch.empty((nframes, self.pitch), dtype=torch.float32, device=stream.device)
        assert out.is_contiguous() and out.numel() >= nframes * self.pitch
        st = C.c_void_p(torch.cuda.current_stream(stream.device).cuda_stream)
        if spectrum:
            spec = torch.empty((nframes, self.n), dtype=torch.float32, device=stream.device)
            _check(lib().glfer_hip_spectrum_device(self._h, stream.data_ptr(), stream.numel(),
                                                   first_frame, nframes, out.data_ptr(),
                                                   spec.data_ptr(), st), "glfer_hip_spectrum_device")
            return out, spec
        _check(lib().glfer_hip_spectrogram_device(self._h, stream.data_ptr(), stream.numel(),
                                                  first_frame, nframes, out.data_ptr(), st),
               "glfer_hip_spectrogram_device")
        return out                                       # (with cfg.psd_pitch: [nframes][pitch], a row's first `bins` floats are its bins)

    def run_avg(self, stream, avg_mode, depth, minbin, maxbin, max0=0, n_out=None, want_psd=False, want_ret=True,
                first_frame=0, nframes=None):
        """fft_do + fft_psd + update_avg_* in one call (glfer_hip_spectrogram_avg_device): returns (avg [nframes][n_out] float64,
        ret [nframes][4] float64 or None, psd [nframes][bins] float32 or None)."""
        torch = _torch()
        assert stream.is_cuda and stream.dim() == 1 and stream.is_contiguous() and stream.dtype == self._sample_dtype()
        if nframes is None:
            nframes = self.num_frames(stream.numel()) - first_frame
        n_out = n_out or self.bins
        avg = torch.empty((nframes, n_out), dtype=torch.float64, device=stream.device)
        ret = torch.empty((nframes, 4), dtype=torch.float64, device=stream.device) if want_ret else None
        psd = torch.empty((nframes, self.bins), dtype=torch.float32, device=stream.device) if want_psd else None
        st = C.c_void_p(torch.cuda.current_stream(stream.device).cuda_stream)
        _check(lib().glfer_hip_spectrogram_avg_device(self._h, C.c_void_p(stream.data_ptr()), stream.numel(), first_frame, nframes,
                                                      int(avg_mode), int(depth), int(minbin), int(maxbin), int(max0), int(n_out),
                                                      C.c_void_p(psd.data_ptr() if want_psd else None), C.c_void_p(avg.data_ptr()),
                                                      C.c_void_p(ret.data_ptr() if want_ret else None), st),
               "glfer_hip_spectrogram_avg_device")
        return avg, ret, psd

    def run_wav(self, path, chunk_frames=0, max_frames=None, partial_tail=False):
        """Whole WAV file -> numpy psd [frames][bins], streamed through pinned buffers.
        partial_tail: also the reference's extra frame for a trailing partial block
        (wav_fmt.c:102-119; GLFER_WAV_PARTIAL_TAIL)."""
        info = wav_probe(path)
        frames = info.nsamples // self.hop + (1 if partial_tail else 0)
        if max_frames is not None:
            frames = min(frames, max_frames)
        out = np.empty((frames, self.bins), np.float32)
        nf = C.c_size_t(0)
        _check(lib().glfer_hip_spectrogram_wav_ex(self._h, os.fsencode(path), out.ctypes.data, frames,
                                                  C.byref(nf), chunk_frames, WAV_PARTIAL_TAIL if partial_tail else 0),
               "glfer_hip_spectrogram_wav_ex")
        return out[:nf.value]

    def ftest(self, stream, first_frame=0, nframes=None, mu_live=True):
        """The harmonic F statistic of mtm_do (mtm.c:165-174, 203-233) for every frame: a float
        tensor [nframes][bins].  mu_live=False restates the reference build without FFTW."""
        torch = _torch()
        assert stream.is_cuda and stream.dim() == 1 and stream.is_contiguous()
        assert stream.dtype == self._sample_dtype()
        if nframes is None:
            nframes = self.num_frames(stream.numel()) - first_frame
        out = torch.empty((nframes, self.bins), dtype=torch.float32, device=stream.device)
        st = C.c_void_p(torch.cuda.current_stream(stream.device).cuda_stream)
        _check(lib().glfer_hip_mtm_ftest_device(self._h, stream.data_ptr(), stream.numel(), first_frame, nframes,
                                                out.data_ptr(), 1 if mu_live else 0, st), "glfer_hip_mtm_ftest_device")
        return out

    def prepare(self, stream, first_frame=0, nframes=None):
        """prepare_audio's inbuf_fft (fft.c:66-165) for every frame: float tensor [nframes][n]."""
        torch = _torch()
        assert stream.is_cuda and stream.dim() == 1 and stream.is_contiguous()
        assert stream.dtype == self._sample_dtype()
        if nframes is None:
            nframes = self.num_frames(stream.numel()) - first_frame
        out = torch.empty((nframes, self.n), dtype=torch.float32, device=stream.device)
        st = C.c_void_p(torch.cuda.current_stream(stream.device).cuda_stream)
        _check(lib().glfer_hip_prepare_device(self._h, stream.data_ptr(), stream.numel(), first_frame, nframes,
                                              out.data_ptr(), st), "glfer_hip_prepare_device")
        return out

    def waterfall_host(self, samples, disp, want_lev=True):
        """Host samples -> (rgb uint8 [frames][bins][3], lev int16 [frames][bins] | None) through the
        chunked ring, mapping done on the device (glfer_hip_waterfall_host)."""
        want = {SAMPLES_F32: np.float32, SAMPLES_S16: np.int16, SAMPLES_U8: np.uint8}[self.params.sample_format]
        samples = np.ascontiguousarray(samples, want)
        frames = self.num_frames(samples.size)
        rgb = np.empty((frames, self.bins, 3), np.uint8)
        lev = np.empty((frames, self.bins), np.int16) if want_lev else None
        nf = C.c_size_t(0)
        _check(lib().glfer_hip_waterfall_host(self._h, C.byref(disp), samples.ctypes.data, samples.size, rgb.ctypes.data,
                                              lev.ctypes.data if want_lev else None, C.byref(nf)), "glfer_hip_waterfall_host")
        assert nf.value == frames
        return rgb, lev

    def run_host(self, samples, pinned=False):
        """samples: numpy array on the host; returns numpy psd [frames][bins].
        pinned=True: the rows come back in pinned memory (glfer_hip_host_alloc; DMA straight into it,
        3-4x the rate of a pageable array); the memory is released when the array is collected."""
        want = {SAMPLES_F32: np.float32, SAMPLES_S16: np.int16, SAMPLES_U8: np.uint8}[
            self.params.sample_format]
        samples = np.ascontiguousarray(samples, want)
        frames = self.num_frames(samples.size)
        out = pinned_empty((frames, self.bins), np.float32) if pinned and frames else np.empty((frames, self.bins), np.float32)
        nf = C.c_size_t(0)
        _check(lib().glfer_hip_spectrogram_host(self._h, samples.ctypes.data, samples.size,
                                                out.ctypes.data, C.byref(nf)),
               "glfer_hip_spectrogram_host")
        assert nf.value == frames
        return out


def frame_range(total_frames, rank, world):
    """glfer_hip_frame_range: (first, count) of a rank's contiguous frame block."""
    a, b = C.c_size_t(0), C.c_size_t(0)
    lib().glfer_hip_frame_range(total_frames, rank, world, C.byref(a), C.byref(b))
    return a.value, b.value


def spectrogram_host_multi(params, samples, devices, out=None):
    """glfer_hip_spectrogram_host_multi: one stream on the host, its frames dealt out over
    `devices` (a list of HIP device ordinals), numpy psd [frames][bins] back."""
    want = {SAMPLES_F32: np.float32, SAMPLES_S16: np.int16, SAMPLES_U8: np.uint8}[params.sample_format]
    samples = np.ascontiguousarray(samples, want)
    cfg = make_config(params, 0)
    hop = int(params.n * (1.0 - float(np.float32(params.overlap))))
    frames = samples.size // hop
    if out is None:
        out = np.empty((frames, params.n // 2 + 1), np.float32)
    nf = C.c_size_t(0)
    if len(set(devices)) == len(devices):
        mask = 0
        for d in devices:
            mask |= 1 << d
        _check(lib().glfer_hip_spectrogram_host_multi(C.byref(cfg), mask, samples.ctypes.data, samples.size,
                                                      out.ctypes.data, C.byref(nf)), "glfer_hip_spectrogram_host_multi")
    else:                                   # workers sharing a GPU: glfer_hip_spectrogram_host_workers
        devs = (C.c_int * len(devices))(*devices)
        _check(lib().glfer_hip_spectrogram_host_workers(C.byref(cfg), devs, len(devices), samples.ctypes.data, samples.size,
                                                        out.ctypes.data, C.byref(nf)), "glfer_hip_spectrogram_host_workers")
    return out[:nf.value]


def _hop_of(params):
    return int(params.n * (1.0 - float(np.float32(params.overlap))))


def spectrogram_wav_workers(params, path, devices, partial_tail=False, max_frames=None):
    """glfer_hip_spectrogram_wav_workers / _multi: a WAV file's frames dealt out over `devices` (HIP
    ordinals; one may repeat), every worker reading its own part of the file; numpy psd [frames][bins]."""
    info = wav_probe(path)
    cfg = make_config(params, 0)
    frames = info.nsamples // _hop_of(params) + (1 if partial_tail else 0)
    if max_frames is not None:
        frames = min(frames, max_frames)
    out = np.empty((frames, params.n // 2 + 1), np.float32)
    nf = C.c_size_t(0)
    flags = WAV_PARTIAL_TAIL if partial_tail else 0
    if len(set(devices)) == len(devices):
        mask = 0
        for d in devices:
            mask |= 1 << d
        _check(lib().glfer_hip_spectrogram_wav_multi(C.byref(cfg), mask, os.fsencode(path), out.ctypes.data, frames,
                                                     C.byref(nf), flags), "glfer_hip_spectrogram_wav_multi")
    else:
        devs = (C.c_int * len(devices))(*devices)
        _check(lib().glfer_hip_spectrogram_wav_workers(C.byref(cfg), devs, len(devices), os.fsencode(path), out.ctypes.data,
                                                       frames, C.byref(nf), flags), "glfer_hip_spectrogram_wav_workers")
    return out[:nf.value]


def waterfall_workers(params, disp, devices, samples=None, path=None, avg_mode=0, depth=1, minbin=0, maxbin=1, max0=0,
                      want_lev=True, partial_tail=False):
    """glfer_hip_waterfall_host_workers (samples: numpy array) or glfer_hip_waterfall_wav_workers (path):
    (rgb uint8 [frames][bins][3], lev int16 [frames][bins] | None), the columns dealt out over `devices`."""
    cfg = make_config(params, 0)
    hop, bins = _hop_of(params), params.n // 2 + 1
    devs = (C.c_int * len(devices))(*devices)
    nf = C.c_size_t(0)
    if path is None:
        want = {SAMPLES_F32: np.float32, SAMPLES_S16: np.int16, SAMPLES_U8: np.uint8}[params.sample_format]
        samples = np.ascontiguousarray(samples, want)
        frames = samples.size // hop
    else:
        frames = wav_probe(path).nsamples // hop + (1 if partial_tail else 0)
    rgb = np.empty((frames, bins, 3), np.uint8)
    lev = np.empty((frames, bins), np.int16) if want_lev else None
    levp = lev.ctypes.data if want_lev else None
    if path is None:
        _check(lib().glfer_hip_waterfall_host_workers(C.byref(cfg), devs, len(devices), C.byref(disp), int(avg_mode), depth, minbin,
                                                      maxbin, int(max0), samples.ctypes.data, samples.size, rgb.ctypes.data, levp,
                                                      C.byref(nf)), "glfer_hip_waterfall_host_workers")
    else:
        _check(lib().glfer_hip_waterfall_wav_workers(C.byref(cfg), devs, len(devices), C.byref(disp), int(avg_mode), depth, minbin,
                                                     maxbin, int(max0), os.fsencode(path), frames, rgb.ctypes.data, levp,
                                                     C.byref(nf), WAV_PARTIAL_TAIL if partial_tail else 0),
               "glfer_hip_waterfall_wav_workers")
    return rgb[:nf.value], (lev[:nf.value] if want_lev else None)


class Workers:
    """glfer_hip_workers: a kept set of workers (one plan + chunk ring per entry of `devices`) for the file / host-buffer entries."""

    def __init__(self, params, devices, hint_frames=0):
        cfg = make_config(params, devices[0])
        devs = (C.c_int * len(devices))(*devices)
        self._h = C.c_void_p()
        self._destroy = lib().glfer_hip_workers_destroy
        _check(lib().glfer_hip_workers_create(C.byref(cfg), devs, len(devices), hint_frames, C.byref(self._h)), "glfer_hip_workers_create")
        self.params, self.devices = params, list(devices)
        self.bins = params.n // 2 + 1

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._destroy(self._h)
            self._h.value = None

    __del__ = close

    def run_wav(self, path, out, partial_tail=False, phases=True):
        """rows into `out` (a numpy array [frames][bins], ideally pinned); returns (frames, phases dict or None).  phases=False: no
        timing events are recorded (they cost a few per cent with one worker and more with several sharing a GPU)."""
        nf, ph = C.c_size_t(0), Phases()
        _check(lib().glfer_hip_workers_spectrogram_wav(self._h, os.fsencode(path), out.ctypes.data, out.shape[0], C.byref(nf),
                                                       WAV_PARTIAL_TAIL if partial_tail else 0, C.byref(ph) if phases else None),
               "glfer_hip_workers_spectrogram_wav")
        return nf.value, (ph.as_dict() if phases else None)

    def run_host(self, samples, out, phases=True):
        nf, ph = C.c_size_t(0), Phases()
        _check(lib().glfer_hip_workers_spectrogram_host(self._h, samples.ctypes.data, samples.size, out.ctypes.data, C.byref(nf),
                                                        C.byref(ph) if phases else None), "glfer_hip_workers_spectrogram_host")
        return nf.value, (ph.as_dict() if phases else None)


class PinnedArray:
    """A numpy view of pinned host memory from glfer_hip_host_alloc (rows land in it by DMA)."""

    def __init__(self, shape, dtype):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = lib().glfer_hip_host_alloc(self.nbytes)
        if not self.ptr:
            raise GlferHipError("glfer_hip_host_alloc(%d) failed" % self.nbytes)
        buf = (C.c_ubyte * self.nbytes).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            lib().glfer_hip_host_free(self.ptr)
            self.ptr = None


def pinned_empty(shape, dtype):
    """numpy.empty in pinned host memory; freed when the array (and every view of it) is gone."""
    import weakref
    nbytes = max(1, int(np.prod(shape)) * np.dtype(dtype).itemsize)
    ptr = lib().glfer_hip_host_alloc(nbytes)
    if not ptr:
        raise GlferHipError("glfer_hip_host_alloc(%d) failed" % nbytes)
    buf = (C.c_ubyte * nbytes).from_address(ptr)
    weakref.finalize(buf, lib().glfer_hip_host_free, ptr)      # buf is the base of every view
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def wav_probe(path):
    """open_wav_file()'s header parse (wav_fmt.c:45-80) with fixed-width fields."""
    info = WavInfo()
    _check(lib().glfer_hip_wav_probe(os.fsencode(path), C.byref(info)), "glfer_hip_wav_probe")
    return info


def compute_floor(psd):
    """compute_floor (fft.c:240-294) for every row of a device PSD tensor.
    Returns a [frames][4] float tensor: sig, floor, peak value, peak bin."""
    torch = _torch()
    assert psd.is_cuda and psd.dtype == torch.float32 and psd.is_contiguous() and psd.dim() == 2
    out = torch.empty((psd.shape[0], 4), dtype=torch.float32, device=psd.device)
    st = C.c_void_p(torch.cuda.current_stream(psd.device).cuda_stream)
    _check(lib().glfer_hip_floor_device(psd.data_ptr(), psd.shape[0], psd.shape[1], out.data_ptr(), st),
           "glfer_hip_floor_device")
    return out


def palette(p_n):
    """set_palette (g_main.c:651-762) as a uint8 [256][3] numpy array (host table)."""
    import numpy as np
    tab = (C.c_ubyte * 768)()
    _check(lib().glfer_hip_palette(int(p_n), tab), "glfer_hip_palette")
    return np.frombuffer(bytes(tab), np.uint8).reshape(256, 3).copy()


def display(disp, src, stats, want_lev=True, want_levels=True):
    """The column mapping of main_window_draw (g_main.c:1099-1236) for every row of `src`
    (float32 PSD, or float64 averaged spectrum from update_avg) with compute_floor's `stats`.
    `disp` (a Display) carries first_buffer / display_*_lvl across calls and is updated.
    Returns (rgb uint8 [frames][bins][3], lev int16 [frames][bins] | None,
    levels float32 [frames][4] | None)."""
    torch = _torch()
    assert src.is_cuda and src.is_contiguous() and src.dim() == 2
    assert src.dtype in (torch.float32, torch.float64)
    assert stats.is_cuda and stats.dtype == torch.float32 and stats.is_contiguous()
    assert stats.shape == (src.shape[0], 4)
    frames, bins = src.shape
    rgb = torch.empty((frames, bins, 3), dtype=torch.uint8, device=src.device)
    lev = torch.empty((frames, bins), dtype=torch.int16, device=src.device) if want_lev else None
    levels = torch.empty((frames, 4), dtype=torch.float32, device=src.device) if want_levels else None
    st = C.c_void_p(torch.cuda.current_stream(src.device).cuda_stream)
    is_d = src.dtype == torch.float64
    _check(lib().glfer_hip_display_device(
        C.byref(disp), None if is_d else C.c_void_p(src.data_ptr()),
        C.c_void_p(src.data_ptr()) if is_d else None, C.c_void_p(stats.data_ptr()), frames, bins,
        C.c_void_p(rgb.data_ptr()), C.c_void_p(lev.data_ptr()) if want_lev else None,
        C.c_void_p(levels.data_ptr()) if want_levels else None, st), "glfer_hip_display_device")
    return rgb, lev, levels


def waterfall(disp, psd, avg_mode=0, depth=1, minbin=0, maxbin=1, max0=0, want_lev=True, want_stats=False):
    """glfer_hip_waterfall_device: floor statistics, optional moving average, level tracking and the
    pixel map of a batch of PSD rows, tile by tile.  Returns (rgb, lev | None, stats | None)."""
    torch = _torch()
    assert psd.is_cuda and psd.dtype == torch.float32 and psd.is_contiguous() and psd.dim() == 2
    frames, bins = psd.shape
    rgb = torch.empty((frames, bins, 3), dtype=torch.uint8, device=psd.device)
    lev = torch.empty((frames, bins), dtype=torch.int16, device=psd.device) if want_lev else None
    stats = torch.empty((frames, 4), dtype=torch.float32, device=psd.device) if want_stats else None
    st = C.c_void_p(torch.cuda.current_stream(psd.device).cuda_stream)
    _check(lib().glfer_hip_waterfall_device(C.byref(disp), int(avg_mode), depth, minbin, maxbin, int(max0), psd.data_ptr(), frames,
                                            bins, rgb.data_ptr(), lev.data_ptr() if want_lev else None,
                                            stats.data_ptr() if want_stats else None, st), "glfer_hip_waterfall_device")
    return rgb, lev, stats


def avg_cum(psd, depth, minbin, maxbin, n_out=None):
    """avgdata->cum after each frame (avg.c:114-127): float64 [frames][n_out], zeros out of band."""
    torch = _torch()
    assert psd.is_cuda and psd.dtype == torch.float32 and psd.is_contiguous() and psd.dim() == 2
    frames, bins = psd.shape
    n_out = bins if n_out is None else n_out
    cum = torch.zeros((frames, n_out), dtype=torch.float64, device=psd.device)
    st = C.c_void_p(torch.cuda.current_stream(psd.device).cuda_stream)
    _check(lib().glfer_hip_avg_cum_device(psd.data_ptr(), frames, bins, n_out, depth, minbin, maxbin, cum.data_ptr(), st),
           "glfer_hip_avg_cum_device")
    return cum


def update_avg(mode, psd, depth, minbin, maxbin, max0=0, n_out=None):
    """update_avg_{sumavg,plain,sumextreme} (avg.c:108-298) applied to consecutive rows of a
    device PSD tensor, starting from an empty averaging state (alloc_avg, avg.c:38-60).
    Returns (avg [frames][n_out] float64, ret [frames][4] float64 = value, peakbin, variance,
    effdepth)."""
    torch = _torch()
    assert psd.is_cuda and psd.dtype == torch.float32 and psd.is_contiguous() and psd.dim() == 2
    frames, bins = psd.shape
    n_out = bins if n_out is None else n_out
    avg = torch.empty((frames, n_out), dtype=torch.float64, device=psd.device)
    ret = torch.empty((frames, 4), dtype=torch.float64, device=psd.device)
    st = C.c_void_p(torch.cuda.current_stream(psd.device).cuda_stream)
    _check(lib().glfer_hip_avg_device(int(mode), psd.data_ptr(), frames, bins, n_out, depth, minbin,
                                      maxbin, int(max0), avg.data_ptr(), ret.data_ptr(), st),
           "glfer_hip_avg_device")
    return avg, ret
