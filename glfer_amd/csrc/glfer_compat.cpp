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
#include <ctime>
#include <map>
#include <string>
#include <thread>
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
int glfer_compat_readahead = 1;        // 0: every hop of a file source goes through the per-hop path
unsigned long glfer_compat_readahead_served = 0;   // hops served from a device-computed batch (a counter for tests and logs)

// fft.c:47-59 -- the window menu of the options dialog (g_options.c:47-48, 579-583) reads these two data
// symbols from the object this library replaces: same names, same order, same menu paths
static char wn_hanning[] = "/Hanning", wn_blackman[] = "/Blackman", wn_gaussian[] = "/Gaussian", wn_welch[] = "/Welch",
            wn_bartlett[] = "/Bartlett", wn_rectangular[] = "/Rectangular", wn_hamming[] = "/Hamming", wn_kaiser[] = "/Kaiser";
fft_window_t fft_windows[] = {
  {wn_hanning, HANNING_WINDOW},   {wn_blackman, BLACKMAN_WINDOW},       {wn_gaussian, GAUSSIAN_WINDOW}, {wn_welch, WELCH_WINDOW},
  {wn_bartlett, BARTLETT_WINDOW}, {wn_rectangular, RECTANGULAR_WINDOW}, {wn_hamming, HAMMING_WINDOW},   {wn_kaiser, KAISER_WINDOW}};
int num_fft_windows = sizeof(fft_windows) / sizeof(fft_windows[0]);
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
  glfer_hip_config cfg{};          // what the plan was made from (the read-ahead's batch plan starts from it)
  const fft_params_t *fp = nullptr; // the estimator's frame parameters (overlap, sub_mean)
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

void reader_prepare();
void engine_open(const void *key, const glfer_hip_config &cfg, bool want_spec, const fft_params_t *fp) {
  Engine e;
  int rc = glfer_hip_plan_create(&cfg, &e.plan);
  if (rc) die("plan_create", rc);
  e.n = cfg.n;
  e.cfg = cfg;
  e.fp = fp;
  hipck(hipMalloc((void **)&e.d_frame, (size_t)cfg.n * sizeof(float)), "hipMalloc frame");
  hipck(hipMalloc((void **)&e.d_psd, (size_t)(cfg.n / 2 + 1) * sizeof(float)), "hipMalloc psd");
  if (want_spec) hipck(hipMalloc((void **)&e.d_spec, (size_t)cfg.n * sizeof(float)), "hipMalloc spec");
  e.psd.assign(cfg.n / 2 + 1, 0.0f);
  // the library's device code is loaded by its first launch (tens of ms for the whole set of kernels):
  // here, at *_init, not under the first hop
  // (any launch loads the library's one code object; the floor entry takes rows of at most 32769 bins, the
  // estimators go up to N = 1048576 -- a short row does, and a refused warm-up only moves the load under the first hop)
  hipck(hipMemset(e.d_psd, 0, (size_t)(cfg.n / 2 + 1) * sizeof(float)), "hipMemset");
  const int warm_bins = cfg.n / 2 + 1 < 2049 ? cfg.n / 2 + 1 : 2049;
  int wrc = glfer_hip_floor_device(e.d_psd, 1, warm_bins, e.d_frame, nullptr);
  if (wrc) fprintf(stderr, "glfer_compat: warm-up launch: %s (continuing)\n", glfer_hip_strerror(wrc));
  hipck(hipDeviceSynchronize(), "hipDeviceSynchronize");
  g_engines[key] = e;
  reader_prepare();                 // a file is open already: its first window now, not under the first hop
}

void reader_forget(const void *owner);
void reader_bypassed(const float *audio_buf);
void reader_sync_state(const float *audio_buf, fft_params_t *fp);
void engine_close(const void *key) {
  reader_forget(key);
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
    // K0 on the device, the hop summed in the reference's own order (fft.c:88-92: a float accumulated
    // sample after sample -- observable on streams with a DC level, glfer_hip.h GLFER_SUBMEAN_EXACT);
    // the corrected hop goes back into the caller's buffer, which the reference mutates in place
    // (fft.c:93-95)
    static float *d_hop = nullptr;
    static int d_hop_len = 0;
    if (h > d_hop_len) {
      if (d_hop) (void)hipFree(d_hop);
      hipck(hipMalloc((void **)&d_hop, (size_t)h * sizeof(float)), "hipMalloc hop");
      d_hop_len = h;
    }
    hipck(hipMemcpy(d_hop, audio_buf, (size_t)h * sizeof(float), hipMemcpyHostToDevice), "H2D hop");
    int rc = glfer_hip_submean_exact_device(d_hop, d_hop, h, 1, GLFER_SAMPLES_F32, nullptr);
    if (rc) die("submean", rc);
    hipck(hipMemcpy(audio_buf, d_hop, (size_t)h * sizeof(float), hipMemcpyDeviceToHost), "D2H hop");
  }
  if (!glfer_compat_get_first_buffer()) memmove(p->inbuf_audio, p->inbuf_audio + n - keep, (size_t)keep * sizeof(float));
  else memset(p->inbuf_audio, 0, (size_t)keep * sizeof(float));
  memcpy(p->inbuf_audio + keep, audio_buf, (size_t)h * sizeof(float));
}


// ---- the file source (wav_fmt.h:24-26) and the read-ahead behind the per-hop entry points ------------
//
// glfer's loop over a file is wav_read() -> fft_do() / mtm_do() -> main_window_draw(), one hop at a
// time (source.c:112-171).  Served hop by hop that is two or three blocking copies and a launch
// per hop; the batch engine does the same file at the GPU's rate.  So the reader is exported from this
// library as well (same three functions, same buffer semantics as wav_fmt.c:45-141), and when an
// estimator is handed the READER'S OWN buffer, untouched, for the file's hops in order, its rows
// come from a batch the device computed from the file itself (glfer_hip_spectrogram_wav_range, a
// window of frames at a time, the next window computed on a second host thread while this one is
// served; per-hop means in the reference's order, GLFER_SUBMEAN_EXACT).  Anything else -- another
// buffer, samples changed after wav_read, a hop skipped or repeated, the history flag
// (glfer.first_buffer) not following the pattern of the mode, the scope window open (it reads
// inbuf_fft), LMP mode -- goes through the per-hop launch as before, from that hop on.  While hops are
// served the estimator's host-side state (inbuf_audio; the caller's buffer with its mean removed,
// fft.c:93-95) is NOT touched -- nothing reads it -- and it is rebuilt from the file, once, at the hop
// where the per-hop path takes over (reader_sync_state), so the hand-over is exact.
// glfer_compat_readahead = 0 turns the read-ahead off.
struct Window {
  std::vector<float> rows;
  size_t first = 0, count = 0;           // rows holds frames [first, first + count)
};
struct Reader {
  FILE *f = nullptr;
  std::string path;
  glfer_wav_info info{};
  int out_len = 1024;                    // samples per block, wav_fmt.c:42 (FIXME there too)
  unsigned char *buf = nullptr;          // raw block, wav_fmt.c:90-96
  float *buff = nullptr;                 // the block as floats, handed to the caller (wav_fmt.c:99)
  size_t data_left = 0;                  // bytes of the data chunk still to read
  size_t blocks = 0;                     // blocks handed out; buff holds block number blocks - 1
  size_t last_samples = 0;               // fresh samples of the block in buff (< out_len: the trailing partial block)
  unsigned long long stamp = 0;          // checksum of buff as it was handed out
  bool fresh = false;                    // no estimator has taken the block in buff yet
  // read-ahead
  const void *owner = nullptr;           // the estimator the batch belongs to
  glfer_hip_config cfg{};
  glfer_hip_plan *plan = nullptr;
  Window win[2];
  int cur = 0;                           // win[cur] is being served, win[cur ^ 1] is being computed / waits
  std::thread fetcher;
  bool fetching = false;
  int fetch_rc = GLFER_OK;
  size_t taken = 0;                      // blocks the owner has taken, all of them in file order from block 0
  size_t state_hops = 0;                 // the owner's host-side state (inbuf_audio) reflects hops [0, state_hops)
  bool off = false;                      // given up for this file
};
Reader g_rd;

// GLFER_COMPAT_TRACE=1: where the read-ahead's set-up time goes (stderr)
double trace_now() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
void trace_log(const char *what, double t0) {
  static const bool on = [] { const char *e = getenv("GLFER_COMPAT_TRACE"); return e && *e == '1'; }();
  if (on) fprintf(stderr, "glfer_compat: %s %.2f ms\n", what, (trace_now() - t0) * 1e3);
}

// (four independent chains: a 2 KB block costs ~0.1 us)
unsigned long long stamp_of(const float *x, int n) {
  const unsigned *u = reinterpret_cast<const unsigned *>(x);
  unsigned long long a = 0x9e3779b97f4a7c15ull, b = 0xc2b2ae3d27d4eb4full, c = 0x165667b19e3779f9ull, d = 0x27d4eb2f165667c5ull;
  int i = 0;
  for (; i + 4 <= n; i += 4) {
    a = (a + u[i]) * 0x100000001b3ull;
    b = (b + u[i + 1]) * 0x100000001b3ull;
    c = (c + u[i + 2]) * 0x100000001b3ull;
    d = (d + u[i + 3]) * 0x100000001b3ull;
  }
  for (; i < n; i++) a = (a + u[i]) * 0x100000001b3ull;
  return a ^ (b << 1 | b >> 63) ^ (c << 2 | c >> 62) ^ (d << 3 | d >> 61);
}

void reader_join() {
  if (g_rd.fetching) {
    g_rd.fetcher.join();
    g_rd.fetching = false;
  }
}

void reader_drop_batch() {
  reader_join();
  if (g_rd.plan) glfer_hip_plan_destroy(g_rd.plan);
  g_rd.plan = nullptr;
  for (Window &w : g_rd.win) {
    w.rows.clear();
    w.rows.shrink_to_fit();
    w.first = w.count = 0;
  }
  g_rd.owner = nullptr;
}

// an entry point without read-ahead took the reader's block: no batch can follow for this file
void reader_bypassed(const float *audio_buf) {
  if (g_rd.f && audio_buf == g_rd.buff) g_rd.off = true;
}

void reader_forget(const void *owner) {            // the estimator is closed: its batch goes with it
  if (g_rd.owner == owner) {
    reader_drop_batch();
    if (g_rd.taken) g_rd.off = true;                // a new estimator would start without this one's history
  }
}

bool same_cfg(const glfer_hip_config &a, const glfer_hip_config &b) { return memcmp(&a, &b, sizeof a) == 0; }

// frames [first, ...) into w: 64 MiB of rows at a time (at least 256 frames), through a ring of 4096-frame
// chunks (a few MB of pinned memory: allocating the default ring's 100 MB costs more than a 10^5-hop
// file takes).  The first window of a file is one chunk, so that the first column does not wait.
int fetch_window(Window &w, size_t first, size_t bins) {
  size_t nwin = ((size_t)64 << 20) / (bins * sizeof(float));
  if (nwin < 256) nwin = 256;
  const size_t chunk = 4096;
  if (first == 0 && nwin > chunk) nwin = chunk;
  const double tr0 = trace_now();
  w.rows.resize(nwin * bins);
  trace_log("read-ahead: rows.resize", tr0);
  size_t got = 0;
  const double tf0 = trace_now();
  int rc = glfer_hip_spectrogram_wav_range(g_rd.plan, g_rd.path.c_str(), first, nwin, w.rows.data(), &got, chunk, GLFER_WAV_PARTIAL_TAIL);
  trace_log(first == 0 ? "read-ahead: first window" : "read-ahead: window", tf0);
  w.first = first;
  w.count = rc == GLFER_OK ? got : 0;
  return rc;
}

// the batch plan of an estimator over the open file: its own configuration with the real overlap, the
// means in the reference's order, the history mode the flag's pattern stands for, the file's format
glfer_hip_config batch_config(glfer_hip_config want, const fft_params_t *fp) {
  want.overlap = fp->overlap;
  want.sub_mean = fp->sub_mean ? GLFER_SUBMEAN_EXACT : GLFER_SUBMEAN_OFF;
  want.history_mode = fp->sub_mean ? GLFER_HISTORY_ZERO_FIRST : GLFER_HISTORY_ZERO_ALWAYS;
  want.sample_format = g_rd.info.bits_per_sample == 8 ? GLFER_SAMPLES_U8 : GLFER_SAMPLES_S16;
  return want;
}

// blocks the open file holds, the trailing partial one included (wav_fmt.c:102-119)
size_t reader_total_blocks() {
  const size_t block_bytes = (size_t)g_rd.out_len * (g_rd.info.bits_per_sample / 8);
  return block_bytes ? (g_rd.info.data_bytes + block_bytes - 1) / block_bytes : 0;
}

// A host that leaves without close_wav_file -- glfer's /Source/Quit goes straight to gtk_main_quit
// (g_main.c:115), and every exit(-1) in here -- must not reach the static destructor of g_rd with a joinable
// fetcher (std::terminate), nor tear HIP down under a fetcher that is still launching: registered when the first
// fetcher starts, i.e. after the HIP runtime's own handlers, so it runs before them.
void reader_at_exit() {
  if (g_rd.fetching && g_rd.fetcher.joinable()) g_rd.fetcher.join();
  g_rd.fetching = false;
}

void start_prefetch(size_t next, size_t bins) {
  if (next >= reader_total_blocks()) return;       // nothing after the window being served
  static const bool registered = [] { atexit(reader_at_exit); return true; }();
  (void)registered;
  Window *nx = &g_rd.win[g_rd.cur ^ 1];
  g_rd.fetching = true;
  g_rd.fetcher = std::thread([nx, next, bins] { g_rd.fetch_rc = fetch_window(*nx, next, bins); });
}

// "On open, run the batch engine over the file": as soon as a file is open AND one estimator is
// initialised (source.c:193 and change_params, in either order) the batch plan is made and the file's
// first window computed -- set-up (device code loaded by its first launch, pinned ring, pools: ~40 ms)
// that belongs with fft_init / open_wav_file, not under the first column.  Speculative: the first
// *_do checks the configuration again and starts over if another estimator turns up.
void reader_prepare() {
  Reader &r = g_rd;
  if (!glfer_compat_readahead || !r.f || r.blocks != 0 || r.plan || r.off || g_engines.size() != 1) return;
  const Engine &e = g_engines.begin()->second;
  if (!e.fp || e.cfg.mode == GLFER_MODE_LMP) return;
  if ((int)(e.fp->n * (1.0 - e.fp->overlap)) != r.out_len) return;
  const glfer_hip_config want = batch_config(e.cfg, e.fp);
  if (glfer_hip_plan_create(&want, &r.plan) != GLFER_OK) { r.plan = nullptr; return; }
  r.cfg = want;
  r.owner = g_engines.begin()->first;
  const size_t bins = (size_t)e.fp->n / 2 + 1;
  r.cur = 0;
  if (fetch_window(r.win[0], 0, bins) != GLFER_OK || r.win[0].count == 0) { reader_drop_batch(); return; }
  start_prefetch(r.win[0].count, bins);
}

// Row of the hop in the reader's buffer for estimator `owner`, or false (the caller then takes the
// per-hop path).  `want`: the batch configuration this estimator needs (mode, sizes, window, tapers;
// overlap, mean removal, history and sample format are filled in here).  Once per *_do.
bool readahead_row(const float *audio_buf, const void *owner, const fft_params_t *fp, glfer_hip_config want, float *row_out) {
  Reader &r = g_rd;
  if (!r.f || r.off) return false;
  if (audio_buf != r.buff) {
    if (owner == r.owner) r.off = true;            // its history now holds samples that are not the file's
    return false;
  }
  const bool was_fresh = r.fresh;
  r.fresh = false;
  const size_t k = r.blocks - 1;                   // the hop in the buffer
  if (r.owner && owner != r.owner) { r.off = true; return false; }
  if (!was_fresh || k != r.taken) { r.off = true; return false; }          // a hop repeated, or skipped
  r.taken++;
  if (!glfer_compat_readahead) { r.off = true; return false; }
  if (stamp_of(r.buff, r.out_len) != r.stamp) { r.off = true; return false; }   // the caller changed the samples
  const int h = (int)(fp->n * (1.0 - fp->overlap));
  if (h != r.out_len) { r.off = true; return false; }                      // the reader was opened for another hop
  // the history flag must follow the mode's pattern: with autoscale (= sub_mean, fft.c:186) the
  // drawer clears glfer.first_buffer after the first column (g_main.c:1111-1120) -- history from the
  // stream; without, it stays TRUE -- history zeroed in every frame
  const bool zero_always = !fp->sub_mean;
  const bool expect_first = k == 0 || zero_always;
  if ((glfer_compat_get_first_buffer() != 0) != expect_first) { r.off = true; return false; }
  if (&glfer && glfer.scope_window) return false;  // the scope reads inbuf_fft of every hop: per-hop path, this hop
  want = batch_config(want, fp);
  if (!r.plan || !same_cfg(want, r.cfg)) {
    if (r.plan && k != 0) { r.off = true; return false; }                  // the estimator changed under way
    reader_drop_batch();
    const double tp0 = trace_now();
    if (glfer_hip_plan_create(&want, &r.plan) != GLFER_OK) { r.plan = nullptr; r.off = true; return false; }
    trace_log("read-ahead: plan_create", tp0);
    r.cfg = want;
    r.owner = owner;
  }
  const size_t bins = (size_t)fp->n / 2 + 1;
  Window *w = &r.win[r.cur];
  if (k < w->first || k >= w->first + w->count) {
    // not in the window being served: the one computed ahead, if it starts here; else a fetch now
    bool have = false;
    if (r.fetching) {
      reader_join();
      Window &nx = r.win[r.cur ^ 1];
      if (r.fetch_rc == GLFER_OK && k >= nx.first && k < nx.first + nx.count) {
        r.cur ^= 1;
        have = true;
      }
    }
    if (!have && (fetch_window(r.win[r.cur], k, bins) != GLFER_OK || r.win[r.cur].count == 0)) { r.off = true; return false; }
    w = &r.win[r.cur];
    // the window after this one, on a second host thread
    start_prefetch(w->first + w->count, bins);
  }
  memcpy(row_out, w->rows.data() + (k - w->first) * bins, bins * sizeof(float));
  glfer_compat_readahead_served++;
  return true;
}

void assemble(float *audio_buf, fft_params_t *p);

// The per-hop path is about to take the hop in the reader's buffer, and hops before it were served
// from a batch: the estimator's host-side state is rebuilt from the file -- the hops the history
// reaches back over, read again, their means removed (device, the reference's order) and pushed
// through the same frame assembly the per-hop path uses -- and, for a trailing partial block, the
// stale rest of the reader's buffer gets the previous block AS THE ESTIMATOR LEFT IT (fft.c:93-95).
void reader_sync_state(const float *audio_buf, fft_params_t *fp) {
  Reader &r = g_rd;
  if (!r.f || audio_buf != r.buff || r.blocks == 0) return;
  const size_t k = r.blocks - 1;
  if (r.state_hops >= k) { r.state_hops = k + 1; return; }
  const int h = r.out_len, n = fp->n;
  if (h != (int)(n * (1.0 - fp->overlap))) { r.state_hops = k + 1; return; }
  const size_t back = (size_t)((n - h + h - 1) / h);                       // hops the history reaches back over
  size_t j0 = k > back ? k - back : 0;
  const size_t gap_from = r.state_hops;
  if (j0 < r.state_hops) j0 = r.state_hops;
  const int bytes_per = r.info.bits_per_sample / 8;
  const long pos = ftell(r.f);
  std::vector<unsigned char> raw((size_t)h * bytes_per);
  std::vector<float> hop((size_t)h);
  const int saved_first = glfer_compat_get_first_buffer();
  for (size_t j = j0; j < k; j++) {
    if (fseek(r.f, (long)(r.info.data_offset + j * raw.size()), SEEK_SET) != 0 || fread(raw.data(), 1, raw.size(), r.f) != raw.size()) break;
    if (bytes_per == 1) for (int i = 0; i < h; i++) hop[i] = ((float)raw[i] - 128) / 128;
    else for (int i = 0; i < h; i++) hop[i] = (float)((const short *)raw.data())[i] / 32768;
    // the flag as it stood when this hop was taken: TRUE at hop 0 and -- without autoscale -- always
    const int first = (j == 0 || !fp->sub_mean) ? 1 : 0;
    if (&glfer) glfer.first_buffer = first; else glfer_compat_first_buffer = first;
    // a gap between the state the per-hop path left (hops < state_hops) and the oldest hop the history reaches:
    // start from an empty history; WITHOUT a gap (the per-hop path ran fewer than `back` hops ago -- the scope
    // window toggled) inbuf_audio still holds the hops before state_hops and the assembly continues from it
    if (j == j0 && j0 > gap_from && first == 0) memset(fp->inbuf_audio, 0, (size_t)n * sizeof(float));
    assemble(hop.data(), fp);
    if (j + 1 == k && r.last_samples < (size_t)h)
      memcpy(r.buff + r.last_samples, hop.data() + r.last_samples, ((size_t)h - r.last_samples) * sizeof(float));
  }
  if (&glfer) glfer.first_buffer = saved_first; else glfer_compat_first_buffer = saved_first;
  (void)fseek(r.f, pos, SEEK_SET);
  r.state_hops = k + 1;
}

}  // namespace

extern "C" {

// wav_fmt.c:45-79.  The header's chunks are walked (glfer_hip_wav_probe); the return value is a file
// descriptor, as the reference's (source.c:193 keeps it as audio_fd).
int open_wav_file(char *fname, int n, int *speed) {
  Reader &r = g_rd;
  if (r.f) close_wav_file();
  r.out_len = n;
  int rc = glfer_hip_wav_probe(fname, &r.info);
  if (rc != GLFER_OK) {
    fprintf(stderr, "error opening %s (or not a PCM WAV file)\n", fname);  // wav_fmt.c:53-56, 63-69
    exit(-1);
  }
  r.f = fopen(fname, "rb");
  if (!r.f || fseek(r.f, (long)r.info.data_offset, SEEK_SET) != 0) {
    fprintf(stderr, "error opening %s\n", fname);
    exit(-1);
  }
  setvbuf(r.f, nullptr, _IOFBF, (size_t)1 << 18);                      // a block is 1-32 KB: fewer read() calls
  r.path = fname;
  r.data_left = r.info.data_bytes;
  r.blocks = r.taken = r.state_hops = 0;
  r.fresh = false;
  r.off = false;
  if (speed) *speed = r.info.sample_rate;
  reader_prepare();                // an estimator is initialised already: the file's first window now
  return fileno(r.f);
}

// wav_fmt.c:81-121: one block of out_len samples; a short last read converts what it got over the
// stale rest of the buffer; n_out = 0 at the end of the data.
void wav_read(float **buf_out, int *n_out) {
  Reader &r = g_rd;
  if (!r.f) { *n_out = 0; *buf_out = r.buff; return; }
  const int bytes_per = r.info.bits_per_sample / 8;
  const size_t s_bufsize = (size_t)r.out_len * bytes_per;
  if (!r.buf) {
    r.buf = (unsigned char *)calloc(r.out_len, bytes_per);
    r.buff = (float *)calloc(r.out_len, sizeof(float));
  }
  const size_t want = s_bufsize < r.data_left ? s_bufsize : r.data_left;
  const size_t n_read = want ? fread(r.buf, 1, want, r.f) : 0;
  r.data_left -= n_read;
  size_t ns = 0;
  if (bytes_per == 1) {
    ns = n_read;
    for (size_t i = 0; i < ns; i++) r.buff[i] = ((float)r.buf[i] - 128) / 128;               // wav_fmt.c:106-108
  } else {
    ns = n_read / 2;
    const short *b16 = (const short *)r.buf;
    for (size_t i = 0; i < ns; i++) r.buff[i] = (float)b16[i] / 32768;                      // wav_fmt.c:111-114
  }
  *n_out = n_read == 0 ? 0 : 1;
  *buf_out = r.buff;
  if (n_read) {
    r.blocks++;
    r.fresh = true;
    r.last_samples = ns;
    r.stamp = stamp_of(r.buff, r.out_len);
  }
}

void close_wav_file(void) {                                        // wav_fmt.c:123-141
  Reader &r = g_rd;
  reader_drop_batch();
  if (r.f) fclose(r.f);
  r.f = nullptr;
  free(r.buf); r.buf = nullptr;
  free(r.buff); r.buff = nullptr;
  r.blocks = r.taken = r.state_hops = 0;
  r.fresh = r.off = false;
}

}  // extern "C"

namespace {

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
  engine_open(p, cfg, true, p);
  int rc = glfer_hip_get_window(engine_for(p).plan, p->window);
  if (rc) die("get_window", rc);
}

void prepare_audio(float *audio_buf, fft_params_t *p) {            // fft.c:66-165
  // history / mean handling here, then RA9MB, window and limiter on the device into inbuf_fft
  // (what lmp.c:101-120 and the scope, g_scope.c:194-197, read after this call)
  Engine &e = engine_for(p);
  reader_bypassed(audio_buf);
  reader_sync_state(audio_buf, p);
  assemble(audio_buf, p);
  if (!e.d_spec) hipck(hipMalloc((void **)&e.d_spec, (size_t)p->n * sizeof(float)), "hipMalloc prepared frame");
  hipck(hipMemcpy(e.d_frame, p->inbuf_audio, (size_t)p->n * sizeof(float), hipMemcpyHostToDevice), "H2D frame");
  int rc = glfer_hip_prepare_device(e.plan, e.d_frame, (size_t)p->n, 0, 1, e.d_spec, nullptr);
  if (rc) die("prepare_audio", rc);
  hipck(hipMemcpy(p->inbuf_fft, e.d_spec, (size_t)p->n * sizeof(float), hipMemcpyDeviceToHost), "D2H prepared frame");
}

void fft_do(float *audio_buf, fft_params_t *p) {                   // fft.c:190-200
  Engine &e = engine_for(p);
  // a file's hop in the reader's own buffer: the row comes from the batch the device computed from the
  // file (outbuf then keeps the last per-hop spectrum: nothing but fft_psd reads it, and the scope
  // window -- which reads inbuf_fft -- turns the read-ahead off while it is open)
  if (readahead_row(audio_buf, p, p, e.cfg, e.psd.data())) return;
  reader_sync_state(audio_buf, p);
  assemble(audio_buf, p);
  hipck(hipMemcpy(e.d_frame, p->inbuf_audio, (size_t)p->n * sizeof(float), hipMemcpyHostToDevice), "H2D frame");
  if (p->n > 32768) {
    // the halfcomplex spectrum output stops at N = 32768 (glfer_hip.h); above it fft_do leaves the PSD for
    // fft_psd and outbuf as it was -- its one reader is fft_psd's phase branch, which no caller of the
    // reference asks for (source.c:144 passes NULL)
    int rc = glfer_hip_spectrogram_device(e.plan, e.d_frame, (size_t)p->n, 0, 1, e.d_psd, nullptr);
    if (rc) die("fft_do", rc);
    hipck(hipMemcpy(e.psd.data(), e.d_psd, e.psd.size() * sizeof(float), hipMemcpyDeviceToHost), "D2H psd");
    return;
  }
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
  engine_open(p, cfg, false, &p->fft);
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
  if (readahead_row(audio_buf, p, &p->fft, e.cfg, psd_buf)) return;
  reader_sync_state(audio_buf, &p->fft);
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
  engine_open(p, cfg, false, &p->fft);
}

void hparma_do(float *audio_buf, float *psd_buf, float *phase_buf, hparma_params_t *p) {   // hparma.c:74-157
  (void)phase_buf;
  Engine &e = engine_for(p);
  if (readahead_row(audio_buf, p, &p->fft, e.cfg, psd_buf)) return;
  reader_sync_state(audio_buf, &p->fft);
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
  engine_open(p, cfg, false, &p->fft);
  LmpDev d;
  hipck(hipMalloc((void **)&d.d_hist, (size_t)2 * p->avg * n * sizeof(float)), "hipMalloc lmp history");
  g_lmp[p] = d;
}

void lmp_do(float *audio_buf, float *psd_buf, float *phase_buf, lmp_params_t *p) {   // lmp.c:101-181
  (void)phase_buf;
  Engine &e = engine_for(p);
  LmpDev &d = g_lmp[p];
  const int n = p->fft.n, nl = p->avg;
  reader_bypassed(audio_buf);                                      // LMP: per hop only (its ring lives on the device, hop by hop)
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
