"""glfer_amd -- MI355X-native spectrogram / multitaper engine behind glfer's estimator entry
points.  The product is the HIP library in glfer_amd/lib (sources in glfer_amd/csrc, C-ABI in
include/glfer_hip.h); this package is the thin host mirror over it."""
from .api import (AVG_PLAIN, AVG_SUMAVG, AVG_SUMEXTREME, HISTORY_ZERO_ALWAYS, HISTORY_ZERO_FIRST,
                  MODE_FFT, MODE_HPARMA, MODE_MTM, SAMPLES_F32, SAMPLES_S16, SAMPLES_U8, WINDOWS, FftParams,
                  GlferHipError, HparmaParams, MtmParams, Spectrogram, compute_floor, make_dpss, make_window,
                  update_avg, version, wav_probe)
