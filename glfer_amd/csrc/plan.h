// plan.h -- what the translation units of the C-ABI layer share: the plan (fft_init / mtm_init
// state, fft.c:168-187, mtm.c:88-151), error plumbing and the device guard.  Internal.
#pragma once
#include "../../include/glfer_hip.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace glfer {

int hip_fail(hipError_t e, const char *what);      // records the text for glfer_hip_last_hip_error(), returns GLFER_E_HIP
std::string error_text();                          // this thread's recorded text
void set_error_text(const std::string &text);      // (worker threads hand theirs to the calling thread)

// Every entry point runs on the device its plan (or its data) lives on and leaves the caller's
// current device as it found it (torch reads it through hipGetDevice).
class DeviceGuard {
 public:
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
    err_ = (device == prev_) ? hipSuccess : hipSetDevice(device);
    changed_ = err_ == hipSuccess && device != prev_;
  }
  ~DeviceGuard() {
    if (changed_ && prev_ >= 0) (void)hipSetDevice(prev_);
  }
  hipError_t error() const { return err_; }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;

 private:
  int prev_ = -1;
  bool changed_ = false;
  hipError_t err_ = hipSuccess;
};

// the device a device pointer belongs to (-1: not a device pointer HIP knows)
int device_of(const void *d_ptr);

// Stream-ordered scratch on the current device: small requests from a pool of their own, middle ones
// from the default pool (release threshold raised), 16 MiB and more from blocks the library keeps
// (glfer_hip.cpp: the stream-ordered pool costs milliseconds, and now and then seconds, per GB-sized
// request).  Always given back through scratch_free on the stream that used it.
hipError_t scratch_malloc(void **p, size_t bytes, hipStream_t st);
void scratch_free(void *p, hipStream_t st);
// idle kept blocks of `dev` given back until at most keep_bytes stay / bytes kept now / the per-device cap
size_t scratch_trim(int dev, size_t keep_bytes);
size_t scratch_held(int dev);
void scratch_set_cap(size_t bytes);
bool scratch_keeping();                        // false with GLFER_SCRATCH_CACHE=0: the library keeps nothing between calls
size_t scratch_cap();                          // glfer_hip_scratch_limit / GLFER_SCRATCH_CAP_MB

// Allow `bytes` of dynamic LDS for `kernel` on the current device (hipFuncSetAttribute, once per
// device, kernel and size class).
hipError_t allow_dynamic_lds(const void *kernel, size_t bytes);

// The host-to-host entries' chunk ring (ingest.cpp run_job): two pinned sample buffers, two device
// sample buffers, two device row buffers, two pinned row buffers (+ the waterfall's), two streams.
// Kept with the plan between calls, freed by glfer_hip_plan_destroy.
struct IngestRing {
  unsigned char *h_in[2] = {nullptr, nullptr}, *d_in[2] = {nullptr, nullptr}, *h_out[2] = {nullptr, nullptr};
  unsigned char *d_rgb[2] = {nullptr, nullptr};
  short *h_lev[2] = {nullptr, nullptr}, *d_lev[2] = {nullptr, nullptr};
  float *d_psd[2] = {nullptr, nullptr}, *d_stats[2] = {nullptr, nullptr};
  hipStream_t st[2] = {nullptr, nullptr};
  hipStream_t up = nullptr;                       // every chunk's upload (round 4): uploads queue behind one another, so that chunk c + 1 goes up while chunk c's rows come down
  hipEvent_t ev_up[2] = {nullptr, nullptr};       // chunk b's upload done: its stream's kernels wait for it
  hipEvent_t ev_t[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};   // timing (glfer_hip_phases, made on first use): upload begins / ends, kernels end, download ends
  size_t cap[2][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
  bool busy = false;
};
void ingest_ring_free(IngestRing *r);          // parks the ring as the device's spare, or frees it
IngestRing *ingest_ring_take(int dev);         // the device's parked ring (the caller owns it), or null
void ingest_ring_drop_spare(int dev);          // frees the parked ring (glfer_hip_scratch_trim, glfer_hip_scratch_limit)
void workers_drop_kept();                      // the *_workers entries' kept handles (ingest.cpp): all of them
size_t workers_kept_bytes(int dev);            // their rings' bytes on `dev`
size_t ingest_ring_spare_bytes(int dev);       // pinned host + device bytes of the parked ring (counted by glfer_hip_scratch_held)

}  // namespace glfer

#define HIP_TRY(call)                                          \
  do {                                                         \
    hipError_t e_ = (call);                                    \
    if (e_ != hipSuccess) return glfer::hip_fail(e_, #call);   \
  } while (0)

struct glfer_hip_plan {
  glfer_hip_config cfg;
  int n, hop, keep, bins, ntapers, npairs, lanes;
  int pitch = 0;                    // floats from one PSD row to the next in the device entries (cfg.psd_pitch, or bins)
  int lmp_av = 0;                   // LMP mode: periodograms in the ring (lmp.c:85)
  std::vector<float> window;        // [n] as the reference stores it (unit power)
  std::vector<double> tapers;       // [ntapers][n]
  std::vector<double> sig;          // [ntapers]
  float *d_taps = nullptr;          // [npairs][8][n/16][4] scaled tables (tap_slot)
  float *d_window = nullptr;        // [n] the window itself (FFT mode, not rectangular): prepare_audio's multiply
  float2 *d_tw = nullptr;           // [slots][lanes]
  float *d_htaps = nullptr;         // real-input form (spectro16h.hip): window pairs, [htapers][8][n/32][4]
  int htapers = 0;                  // 1: periodogram window; > 1: the tapers of the multitaper form (n >= 8192)
  float2 *d_htw = nullptr;          //   twiddles of the n/2-point transform
  float2 *d_hrot = nullptr;         //   (cos,sin)(2 pi t/n), t < n/32
  float *d_wtaps = nullptr;         // wavefront-private real-input form (spectro16w.hip): [wtapers][W][8][64][4]
  int wtapers = 0;
  float2 *d_wtw = nullptr;          //   [27][64] twiddles of the 1024-point transform
  float2 *d_wcomb = nullptr;        //   combine / split twiddles per lane
  float2 *d_bigtw = nullptr;        // spectro_big.hip (N >= 32768): [W][16] the register part of the sub-transforms' twiddles
  float *d_xtaps = nullptr;         // odd taper counts (spectro16x.hip): the last taper alone, [4][n/16][4]
  float *d_ltaps = nullptr;         // odd taper counts, LDS-resident half tables (spectro16xl.hip)
  uint16_t *d_lagmap = nullptr;     // HP-ARMA: [t][p_e+1] lag held by each matrix cell
  int *d_rot_sched = nullptr;       // HP-ARMA: [rot_steps][8] the Jacobi sweep as steps of up to eight column-disjoint rotations (j | k << 8, -1 = none)
  int rot_steps = 0, rot_width = 8;
  float2 *d_unit = nullptr;         // HP-ARMA: [n/2+1] exp(-2 pi i k/n)
  // harmonic F-test (mtm.c:124-136): built on first use
  float *d_ftaps_mu_first = nullptr;   // the allocation: [hn][taper 0..ntapers-1][hn], each [2n] alone in slot 0 of the packed layout
  float *d_ftaps = nullptr;         // = d_ftaps_mu_first + 2n: [ntapers+1][2n], taper j, then hn
  float *d_ftaps2 = nullptr;        // the paired form (round 5): [ceil((ntapers+1)/2)][2n] with (hn, taper 0), (taper 1, taper 2) ... as (re, im), halved;
  float *d_ftaps2_nomu = nullptr;   //   then [ceil(ntapers/2)][2n] with (taper 0, taper 1) ... (mu_live = 0); one allocation
  double *d_U0 = nullptr;           // [ntapers]
  std::vector<double> U0;           // [ntapers]
  std::vector<float> hn;            // [n]
  float sum_U0_sqr = 0.0f;
  float spec_unscale = 1.0f;
  bool nonlin = false;
  glfer::IngestRing *ring = nullptr;   // the host entries' chunk ring, kept between calls (ingest.cpp)
  // side streams of the piecewise mean pass (glfer_hip.cpp launch_mean_inkernel): [0] runs the hop means of piece c+1
  // beside piece c's estimator launch, [1] takes every other estimator launch; made on first use
  hipStream_t aux[2] = {nullptr, nullptr};
  std::vector<hipEvent_t> aux_events;  // their events, made once (hipEventCreate per piece cost more than a piece's kernels)
  bool aux_busy = false;               // one call at a time forks onto the side streams; a second one meanwhile stays on its own stream
};

// frames [first, first+nframes) of a device-resident stream (virtual base allowed); psd and/or
// halfcomplex spectra out.  Asynchronous on `st`.  Defined in glfer_hip.cpp.
// tail_fresh >= 0: the LAST frame of the call is the file source's trailing partial block with that
// many fresh samples (wav_fmt.c:102-119); see glfer_hip_spectrogram_wav_ex.
int glfer_run_device(glfer_hip_plan *p, const void *d_stream, size_t nsamples, size_t first, size_t nframes,
                     float *d_psd, float *d_spec, hipStream_t st, long tail_fresh = -1);
