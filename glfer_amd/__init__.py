"""glfer_amd -- MI355X-native spectrogram / multitaper engine behind glfer's estimator entry
points.  The product is the HIP library in glfer_amd/lib (sources in glfer_amd/csrc, C-ABI in
include/glfer_hip.h); this package is the thin host mirror over it."""
from .api import (Workers, Phases, AVG_PLAIN, AVG_SUMAVG, AVG_SUMEXTREME, HISTORY_ZERO_ALWAYS, HISTORY_ZERO_FIRST,
                  MODE_FFT, MODE_HPARMA, MODE_LMP, MODE_MTM, LmpParams, PinnedArray, pinned_empty, WAV_PARTIAL_TAIL, avg_cum, frame_range, make_config, spectrogram_host_multi, spectrogram_wav_workers, waterfall, waterfall_workers, PALETTES, SCALE_LIN, SCALE_LIN_MAX0, SCALE_LOG, SCALE_LOG_MAX0, SAMPLES_F32, SAMPLES_S16, SAMPLES_U8, SUBMEAN_EXACT, SUBMEAN_FAST, SUBMEAN_OFF, WINDOWS, Display, FftParams,
                  GlferHipError, HparmaParams, MtmParams, Spectrogram, compute_floor, display, make_dpss, make_window, palette,
                  update_avg, version, wav_probe)
