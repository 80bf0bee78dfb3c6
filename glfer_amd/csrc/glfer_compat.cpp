// glfer_compat.cpp -- glfer's own estimator entry points (include/glfer_compat.h), one hop per
// call, computed by the HIP engine through the batch C-ABI of glfer_hip.h.
//
// What stays on the host is bookkeeping the reference also does outside its numerics: sliding
// the N-H history and appending the H new samples (memmove), and copying results into the
// caller's buffers.  Window/taper multiply, FFT, |X|^2, taper sum, floor statistics and the
// moving average all run in the kernels.  Failure = message on stderr + exit(-1), like the
// reference (fft.c:249-252).
#include "../../include/glfer_compat.h"
#include "../../include/glfer_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

extern "C" {
// glfer's own globals (glfer.c:56-57), referenced weakly: present when this library is linked into
// glfer (or any program that defines them), absent -- address 0 -- for a stand-alone user
extern opt_t opt __attribute__((weak));
extern glfer_t glfer __attribute__((weak));
int glfer_compat_autoscale = 1;        // opt.autoscale default, glfer.c:275
int glfer_compat_first_buffer = 1;     // glfer.first_buffer = TRUE at start-up, g_main.c:990
int glfer_compat_get_autoscale(void) { return &opt ? opt.autoscale : glfer_compat_autoscale; }
int glfer_compat_get_first_buffer(void) { return &glfer ? glfer.first_buffer : glfer_compat_first_buffer; }
}

namespace {

[[noreturn]] void die(const char *what, int rc) {
  fprintf(stderr, "glfer_compat: %s: %s (%s)\n", what, glfer_hip_strerror(rc), glfer_hip_last_hip_error());
  exit(-1);
}
void hipck(hipError_t e, const char *what) {
  if (e != hipSuccess) {
    fprintf(stderr, "glfer_compat: %s: %s\n", what, hipGetErrorString(e));
    exit(-1);
  }
}

// device side of one estimator instance: a plan that treats every assembled frame as one
// non-overlapping block (the overlap lives in params->inbuf_audio, as in the reference)
struct Engine {
  glfer_hip_plan *plan = nullptr;
  float *d_frame = nullptr, *d_psd = nullptr, *d_spec = nullptr;
  std::vector<float> psd;          // last PSD, served by fft_psd()
  int n = 0;
};
std::map<const void *, Engine> g_engines;   // keyed by the caller's params struct

Engine &engine_for(const void *key) {
  auto it = g_engines.find(key);
  if (it == g_engines.end()) {
    fprintf(stderr, "glfer_compat: estimator used before *_init()\n");
    exit(-1);
  }
  return it->second;
}

void engine_open(const void *key, const glfer_hip_config &cfg, bool want_spec) {
  Engine e;
  int rc = glfer_hip_plan_create(&cfg, &e.plan);
  if (rc) die("plan_create", rc);
  e.n = cfg.n;
  hipck(hipMalloc((void **)&e.d_frame, (size_t)cfg.n * sizeof(float)), "hipMalloc frame");
  hipck(hipMalloc((void **)&e.d_psd, (size_t)(cfg.n / 2 + 1) * sizeof(float)), "hipMalloc psd");
  if (want_spec) hipck(hipMalloc((void **)&e.d_spec, (size_t)cfg.n * sizeof(float)), "hipMalloc spec");
  e.psd.assign(cfg.n / 2 + 1, 0.0f);
  g_engines[key] = e;
}

void engine_close(const void *key) {
  auto it = g_engines.find(key);
  if (it == g_engines.end()) return;
  glfer_hip_plan_destroy(it->second.plan);
  if (it->second.d_frame) (void)hipFree(it->second.d_frame);
  if (it->second.d_psd) (void)hipFree(it->second.d_psd);
  if (it->second.d_spec) (void)hipFree(it->second.d_spec);
  g_engines.erase(it);
}

// Frame assembly of prepare_audio (fft.c:66-113): optional mean removal of the H new samples
// (kernel K0), slide or zero the N-H history, append.
void assemble(float *audio_buf, fft_params_t *p) {
  const int n = p->n;
  const int h = (int)(n * (1.0 - p->overlap));
  const int keep = n - h;
  if (p->sub_mean) {
    // K0 on the device; the corrected hop goes back into the caller's buffer, which the
    // reference mutates in place (fft.c:93-95)
    static float *d_hop = nullptr;
    static int d_hop_len = 0;
    if (h > d_hop_len) {
      if (d_hop) (void)hipFree(d_hop);
      hipck(hipMalloc((void **)&d_hop, (size_t)h * sizeof(float)), "hipMalloc hop");
      d_hop_len = h;
    }
    hipck(hipMemcpy(d_hop, audio_buf, (size_t)h * sizeof(float), hipMemcpyHostToDevice), "H2D hop");
    int rc = glfer_hip_submean_device(d_hop, d_hop, h, 1, GLFER_SAMPLES_F32, nullptr);
    if (rc) die("submean", rc);
    hipck(hipMemcpy(audio_buf, d_hop, (size_t)h * sizeof(float), hipMemcpyDeviceToHost), "D2H hop");
  }
  if (!glfer_compat_get_first_buffer()) memmove(p->inbuf_audio, p->inbuf_audio + n - keep, (size_t)keep * sizeof(float));
  else memset(p->inbuf_audio, 0, (size_t)keep * sizeof(float));
  memcpy(p->inbuf_audio + keep, audio_buf, (size_t)h * sizeof(float));
}

}  // namespace

extern "C" {

// ---- fft.h ------------------------------------------------------------------------------
void fft_init(fft_params_t *p) {                                   // fft.c:168-187
  p->inbuf_audio = (float *)calloc(p->n, sizeof(float));
  p->inbuf_fft = (float *)calloc(p->n, sizeof(float));
  p->outbuf = p->inbuf_fft;                                        // fft.c:180
  p->window = (float *)malloc((size_t)p->n * sizeof(float));
  p->sub_mean = glfer_compat_get_autoscale();                      // fft.c:186
  glfer_hip_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.mode = GLFER_MODE_FFT;
  cfg.n = p->n;
  cfg.overlap = 0.0f;                    // frames arrive assembled: one block per call
  cfg.window_type = p->window_type;
  cfg.limiter_a = p->a;
  cfg.enable_limiter = p->limiter;
  engine_open(p, cfg, true);
  int rc = glfer_hip_get_window(engine_for(p).plan, p->window);
  if (rc) die("get_window", rc);
}

void prepare_audio(float *audio_buf, fft_params_t *p) {            // fft.c:66-165
  // history / mean handling here, then RA9MB, window and limiter on the device into inbuf_fft
  // (what lmp.c:101-120 and the scope, g_scope.c:194-197, read after this call)
  Engine &e = engine_for(p);
  assemble(audio_buf, p);
  if (!e.d_spec) hipck(hipMalloc((void **)&e.d_spec, (size_t)p->n * sizeof(float)), "hipMalloc prepared frame");
  hipck(hipMemcpy(e.d_frame, p->inbuf_audio, (size_t)p->n * sizeof(float), hipMemcpyHostToDevice), "H2D frame");
  int rc = glfer_hip_prepare_device(e.plan, e.d_frame, (size_t)p->n, 0, 1, e.d_spec, nullptr);
  if (rc) die("prepare_audio", rc);
  hipck(hipMemcpy(p->inbuf_fft, e.d_spec, (size_t)p->n * sizeof(float), hipMemcpyDeviceToHost), "D2H prepared frame");
}

void fft_do(float *audio_buf, fft_params_t *p) {                   // fft.c:190-200
  Engine &e = engine_for(p);
  assemble(audio_buf, p);
  hipck(hipMemcpy(e.d_frame, p->inbuf_audio, (size_t)p->n * sizeof(float), hipMemcpyHostToDevice), "H2D frame");
  int rc = glfer_hip_spectrum_device(e.plan, e.d_frame, (size_t)p->n, 0, 1, e.d_psd, e.d_spec, nullptr);
  if (rc) die("fft_do", rc);
  hipck(hipMemcpy(p->outbuf, e.d_spec, (size_t)p->n * sizeof(float), hipMemcpyDeviceToHost), "D2H spectrum");
  hipck(hipMemcpy(e.psd.data(), e.d_psd, e.psd.size() * sizeof(float), hipMemcpyDeviceToHost), "D2H psd");
}

void fft_psd(float *psd_buf, float *phase_buf, fft_params_t *p) { // fft.c:203-226
  Engine &e = engine_for(p);
  const int n = p->n;
  if (psd_buf) memcpy(psd_buf, e.psd.data(), e.psd.size() * sizeof(float));
  if (phase_buf) {                       // never requested by the reference's callers
    phase_buf[0] = 0;
    for (int i = 1; i < (n + 1) / 2; i++) phase_buf[i] = atan2(p->outbuf[i], p->outbuf[n - i]);
    if (n % 2 == 0) phase_buf[n / 2] = 0;
  }
}

void fft_close(fft_params_t *p) {                                  // fft.c:297-306
  engine_close(p);
  free(p->inbuf_audio); p->inbuf_audio = nullptr;
  free(p->inbuf_fft); p->inbuf_fft = nullptr;
  p->outbuf = nullptr;
  free(p->window); p->window = nullptr;
}

void compute_floor(float *psd_buf, int n, float *sig, float *floor_pwr, float *peak, unsigned int *peak_bin) {
  static float *d_psd = nullptr, *d_st = nullptr;                  // fft.c:240-294; one row of scratch, kept
  static int cap = 0;
  float st[4];
  if (n > cap) {
    if (d_psd) (void)hipFree(d_psd);
    hipck(hipMalloc((void **)&d_psd, (size_t)n * sizeof(float)), "hipMalloc");
    if (!d_st) hipck(hipMalloc((void **)&d_st, 4 * sizeof(float)), "hipMalloc");
    cap = n;
  }
  hipck(hipMemcpy(d_psd, psd_buf, (size_t)n * sizeof(float), hipMemcpyHostToDevice), "H2D psd");
  int rc = glfer_hip_floor_device(d_psd, 1, n, d_st, nullptr);
  if (rc) die("compute_floor", rc);
  hipck(hipMemcpy(st, d_st, sizeof st, hipMemcpyDeviceToHost), "D2H stats");
  *sig = st[0];
  *floor_pwr = st[1];
  *peak = st[2];
  *peak_bin = (unsigned int)st[3];
}

// ---- mtm.h ------------------------------------------------------------------------------
void mtm_init(mtm_params_t *p) {                                   // mtm.c:88-151
  const int n = p->fft.n, kmax = p->kmax;
  p->fft.inbuf_audio = (float *)calloc(n, sizeof(float));
  p->fft.inbuf_fft = (float *)calloc(n, sizeof(float));
  p->fft.outbuf = p->fft.inbuf_fft;
  p->fft.sub_mean = glfer_compat_get_autoscale();                  // mtm.c:111
  glfer_hip_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.mode = GLFER_MODE_MTM;
  cfg.n = n;
  cfg.overlap = 0.0f;
  cfg.mtm_w = p->w;
  cfg.mtm_k = kmax;
  engine_open(p, cfg, false);
  // params->window / sig as the reference lays them out: window[1..n][0..kmax] (mtm.c:118-119)
  std::vector<double> tap((size_t)(kmax + 1) * n);
  p->sig = (double *)malloc((size_t)(kmax + 1) * sizeof(double));
  int rc = glfer_hip_get_tapers(engine_for(p).plan, tap.data(), p->sig);
  if (rc) die("get_tapers", rc);
  double *store = (double *)malloc((size_t)n * (kmax + 1) * sizeof(double));
  double **rows = (double **)malloc((size_t)(n + 1) * sizeof(double *));
  rows[0] = store;                                                 // slot 0 keeps the block for mtm_close
  for (int i = 0; i < n; i++) {
    rows[i + 1] = store + (size_t)i * (kmax + 1);
    for (int j = 0; j <= kmax; j++) rows[i + 1][j] = tap[(size_t)j * n + i];
  }
  p->window = rows;
}

void mtm_do(float *audio_buf, float *psd_buf, float *phase_buf, mtm_params_t *p) {   // mtm.c:154-239
  (void)phase_buf;                                                 // ignored by the reference too
  Engine &e = engine_for(p);
  assemble(audio_buf, &p->fft);
  hipck(hipMemcpy(e.d_frame, p->fft.inbuf_audio, (size_t)p->fft.n * sizeof(float), hipMemcpyHostToDevice), "H2D frame");
  int rc = glfer_hip_spectrogram_device(e.plan, e.d_frame, (size_t)p->fft.n, 0, 1, e.d_psd, nullptr);
  if (rc) die("mtm_do", rc);
  hipck(hipMemcpy(psd_buf, e.d_psd, e.psd.size() * sizeof(float), hipMemcpyDeviceToHost), "D2H psd");
}

void mtm_close(mtm_params_t *p) {                                  // mtm.c:242-265
  engine_close(p);
  free(p->fft.inbuf_audio); p->fft.inbuf_audio = nullptr;
  free(p->fft.inbuf_fft); p->fft.inbuf_fft = nullptr;
  p->fft.outbuf = nullptr;
  if (p->window) { free(p->window[0]); free(p->window); p->window = nullptr; }
  free(p->sig); p->sig = nullptr;
}

// ---- hparma.h ---------------------------------------------------------------------------
void hparma_init(hparma_params_t *p) {                             // hparma.c:45-71
  const int n = p->fft.n;
  p->fft.inbuf_audio = (float *)calloc(n, sizeof(float));
  p->fft.inbuf_fft = (float *)calloc(n, sizeof(float));
  p->fft.outbuf = p->fft.inbuf_fft;
  p->fft.sub_mean = glfer_compat_get_autoscale();                  // hparma.c:62
  if (p->q_e != -1) { fprintf(stderr, "glfer_compat: hparma q_e = %d unsupported (source.c:375 sets -1)\n", p->q_e); exit(-1); }
  glfer_hip_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.mode = GLFER_MODE_HPARMA;
  cfg.n = n;
  cfg.overlap = 0.0f;
  cfg.hparma_t = p->t;
  cfg.hparma_p_e = p->p_e;
  engine_open(p, cfg, false);
}

void hparma_do(float *audio_buf, float *psd_buf, float *phase_buf, hparma_params_t *p) {   // hparma.c:74-157
  (void)phase_buf;
  Engine &e = engine_for(p);
  assemble(audio_buf, &p->fft);
  hipck(hipMemcpy(e.d_frame, p->fft.inbuf_audio, (size_t)p->fft.n * sizeof(float), hipMemcpyHostToDevice), "H2D frame");
  int rc = glfer_hip_spectrogram_device(e.plan, e.d_frame, (size_t)p->fft.n, 0, 1, e.d_psd, nullptr);
  if (rc) die("hparma_do", rc);
  hipck(hipMemcpy(psd_buf, e.d_psd, e.psd.size() * sizeof(float), hipMemcpyDeviceToHost), "D2H psd");
}

void hparma_close(hparma_params_t *p) {                            // hparma.c:160-179
  engine_close(p);
  free(p->fft.inbuf_audio); p->fft.inbuf_audio = nullptr;
  free(p->fft.inbuf_fft); p->fft.inbuf_fft = nullptr;
  p->fft.outbuf = nullptr;
}

// ---- lmp.h ------------------------------------------------------------------------------
namespace {
struct LmpDev {
  float *d_hist = nullptr;      // the last frames, oldest first: 2*nl slots of n samples
  size_t frames = 0;            // frames seen since lmp_init (the ring slot is frames % nl, lmp.c:104)
  int held = 0;                 // frames in d_hist
};
std::map<const void *, LmpDev> g_lmp;
}  // namespace

void lmp_init(lmp_params_t *p) {                                   // lmp.c:59-99
  const int n = p->fft.n;
  p->fft.inbuf_audio = (float *)calloc(n, sizeof(float));
  p->fft.inbuf_fft = (float *)calloc(n, sizeof(float));
  p->fft.outbuf = p->fft.inbuf_fft;
  p->fft.sub_mean = glfer_compat_get_autoscale();                  // lmp.c:82
  glfer_hip_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.mode = GLFER_MODE_LMP;
  cfg.n = n;
  cfg.overlap = 0.0f;                    // frames arrive assembled: one block per call
  cfg.lmp_av = p->avg;                                             // nl = params->avg, lmp.c:85
  engine_open(p, cfg, false);
  LmpDev d;
  hipck(hipMalloc((void **)&d.d_hist, (size_t)2 * p->avg * n * sizeof(float)), "hipMalloc lmp history");
  g_lmp[p] = d;
}

void lmp_do(float *audio_buf, float *psd_buf, float *phase_buf, lmp_params_t *p) {   // lmp.c:101-181
  (void)phase_buf;
  Engine &e = engine_for(p);
  LmpDev &d = g_lmp[p];
  const int n = p->fft.n, nl = p->avg;
  assemble(audio_buf, &p->fft);
  // the engine recomputes the periodograms the ring still holds from the frames themselves, so
  // the last nl assembled frames stay on the device, in time order
  if (d.held == 2 * nl) {
    hipck(hipMemcpy(d.d_hist, d.d_hist + (size_t)(nl + 1) * n, (size_t)(nl - 1) * n * sizeof(float), hipMemcpyDeviceToDevice),
          "slide frames");
    d.held = nl - 1;
  }
  hipck(hipMemcpy(d.d_hist + (size_t)d.held * n, p->fft.inbuf_audio, (size_t)n * sizeof(float), hipMemcpyHostToDevice), "H2D frame");
  d.held++;
  // frame index d.frames of a stream whose sample 0 sits (d.frames + 1 - held) frames before d_hist
  const float *vbase = d.d_hist - ((ptrdiff_t)d.frames + 1 - d.held) * n;
  int rc = glfer_hip_spectrogram_device(e.plan, vbase, (d.frames + 1) * (size_t)n, d.frames, 1, e.d_psd, nullptr);
  if (rc) die("lmp_do", rc);
  hipck(hipMemcpy(psd_buf, e.d_psd, e.psd.size() * sizeof(float), hipMemcpyDeviceToHost), "D2H psd");
  d.frames++;
}

void lmp_close(lmp_params_t *p) {                                  // lmp.c:184-194
  auto it = g_lmp.find(p);
  if (it != g_lmp.end()) {
    (void)hipFree(it->second.d_hist);
    g_lmp.erase(it);
  }
  engine_close(p);
  free(p->fft.inbuf_audio); p->fft.inbuf_audio = nullptr;
  free(p->fft.inbuf_fft); p->fft.inbuf_fft = nullptr;
  p->fft.outbuf = nullptr;
}

// ---- avg.h ------------------------------------------------------------------------------
namespace {
struct AvgDev {
  float *d_rows = nullptr;      // last depth+1 PSD rows, oldest first
  double *d_avg = nullptr, *d_ret = nullptr;
  int rows = 0, bins = 0;
};
std::map<const avg_data_t *, AvgDev> g_avg;
}  // namespace

void init_avg(avg_data_t *a) { a->avgwidth = a->avgdepth = a->effdepth = 0; }   // avg.c:31-36

void alloc_avg(avg_data_t *a, int width, int depth) {              // avg.c:38-60
  a->avgwidth = width;
  a->avgdepth = depth;
  a->effdepth = 0;
  a->avg = (double *)calloc(width, sizeof(double));
  a->cum = (double *)calloc(width, sizeof(double));
  // avgarray[bin][0..depth-1]: the last `depth` values of the bin, oldest first (avg.c:47-52); kept
  // in step below by plain data movement -- the sums themselves come from the device
  a->avgarray = (double **)calloc(width, sizeof(double *));
  for (int i = 0; i < width; i++) a->avgarray[i] = (double *)calloc(depth, sizeof(double));
  AvgDev d;
  hipck(hipMalloc((void **)&d.d_rows, (size_t)2 * (depth + 1) * width * sizeof(float)), "hipMalloc avg rows");
  hipck(hipMemset(d.d_rows, 0, (size_t)2 * (depth + 1) * width * sizeof(float)), "hipMemset avg rows");
  hipck(hipMalloc((void **)&d.d_avg, (size_t)(depth + 1) * width * sizeof(double)), "hipMalloc avg out");
  hipck(hipMalloc((void **)&d.d_ret, (size_t)(depth + 1) * 4 * sizeof(double)), "hipMalloc avg ret");
  g_avg[a] = d;
}

void delete_avg(avg_data_t *a) {                                   // avg.c:62-78
  auto it = g_avg.find(a);
  if (it != g_avg.end()) {
    (void)hipFree(it->second.d_rows);
    (void)hipFree(it->second.d_avg);
    (void)hipFree(it->second.d_ret);
    g_avg.erase(it);
  }
  if (a->avgwidth) {
    free(a->avg);
    free(a->cum);
    if (a->avgarray) {
      for (int i = 0; i < a->avgwidth; i++) free(a->avgarray[i]);
      free(a->avgarray);
    }
  }
  a->avg = a->cum = nullptr;
  a->avgarray = nullptr;
  a->avgwidth = a->avgdepth = a->effdepth = 0;
}

static double avg_step(int mode, avg_data_t *a, int N, float *psd, int max0, int minbin, int maxbin, int *peakbin,
                       double *variance) {
  auto it = g_avg.find(a);
  if (it == g_avg.end()) { fprintf(stderr, "glfer_compat: update_avg before alloc_avg\n"); exit(-1); }
  AvgDev &d = it->second;
  const int depth = a->avgdepth, width = a->avgwidth;
  // Keep the last depth+1 rows contiguous and in time order (row width = avgwidth floats, N
  // of them used).  The buffer holds 2*(depth+1) rows; when it is full the newest `depth`
  // rows move to the front (source and destination do not overlap).
  const int cap = 2 * (depth + 1);
  if (d.rows == cap) {
    hipck(hipMemcpy(d.d_rows, d.d_rows + (size_t)(cap - depth) * width, (size_t)depth * width * sizeof(float),
                    hipMemcpyDeviceToDevice), "slide rows");
    d.rows = depth;
  }
  hipck(hipMemcpy(d.d_rows + (size_t)d.rows * width, psd, (size_t)N * sizeof(float), hipMemcpyHostToDevice), "H2D psd");
  d.rows++;
  // the sliding sum over the last `depth` rows is what a run from an empty state over the
  // last depth+1 rows leaves in its last row (avg.c:116-127)
  const int first = d.rows > depth + 1 ? d.rows - (depth + 1) : 0;
  const int nrows = d.rows - first;
  int rc = glfer_hip_avg_device(mode, d.d_rows + (size_t)first * width, (size_t)nrows, width, width, depth, minbin, maxbin,
                                max0, d.d_avg, d.d_ret, nullptr);
  if (rc) die("update_avg", rc);
  double ret[4];
  hipck(hipMemcpy(a->avg, d.d_avg + (size_t)(nrows - 1) * width, (size_t)N * sizeof(double), hipMemcpyDeviceToHost), "D2H avg");
  // avgdata->cum: the sliding sums of the band (avg.c:114-127), from the device as well (d_avg is reused)
  rc = glfer_hip_avg_cum_device(d.d_rows + (size_t)first * width, (size_t)nrows, width, width, depth, minbin, maxbin, d.d_avg,
                                nullptr);
  if (rc) die("update_avg (sums)", rc);
  hipck(hipMemcpy(a->cum + minbin, d.d_avg + (size_t)(nrows - 1) * width + minbin, (size_t)(maxbin - minbin) * sizeof(double),
                  hipMemcpyDeviceToHost), "D2H cum");
  for (int i = minbin; i < maxbin; i++) {                          // the shift registers: data movement only (avg.c:124-126)
    memmove(a->avgarray[i], a->avgarray[i] + 1, (size_t)(depth - 1) * sizeof(double));
    a->avgarray[i][depth - 1] = psd[i];
  }
  hipck(hipMemcpy(ret, d.d_ret + (size_t)(nrows - 1) * 4, sizeof ret, hipMemcpyDeviceToHost), "D2H ret");
  if (a->effdepth < depth) a->effdepth++;                          // avg.c:138-139
  if (ret[1] >= 0) *peakbin = (int)ret[1];                         // left untouched when nothing exceeds psd[minbin]
  if (variance) *variance = ret[2];
  return ret[0];
}

double update_avg_plain(avg_data_t *a, int N, float *psd, int minbin, int maxbin, int *peakbin) {
  return avg_step(GLFER_AVG_PLAIN, a, N, psd, 0, minbin, maxbin, peakbin, nullptr);
}
double update_avg_sumextreme(avg_data_t *a, int N, float *psd, int max0, int minbin, int maxbin, int *peakbin) {
  return avg_step(GLFER_AVG_SUMEXTREME, a, N, psd, max0, minbin, maxbin, peakbin, nullptr);
}
double update_avg_sumavg(avg_data_t *a, int N, float *psd, int max0, int minbin, int maxbin, int *peakbin,
                         double *variance) {
  return avg_step(GLFER_AVG_SUMAVG, a, N, psd, max0, minbin, maxbin, peakbin, variance);
}

}  // extern "C"
