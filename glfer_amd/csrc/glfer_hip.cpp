// glfer_hip.cpp -- the C-ABI of include/glfer_hip.h: plans, tables, launches.
//
// Host side of the drop-in boundary.  A plan is what fft_init()/mtm_init() build in the
// reference (fft.c:168-187, mtm.c:88-151): window or DPSS tapers, here also uploaded to
// HBM with the 1/N normalisation (fft.c:212-216), the eigenvalue weights 1/(1+sig_j)
// (mtm.c:214-219) and the 1/2 of the re/im packing folded in, so the kernel's epilogue is
// a single add.  No CPU fallback exists: every compute entry fails with GLFER_E_HIP when
// HIP cannot run it.
#include "plan.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>

#include "host_tables.h"
#include "spectro_params.h"

extern "C" hipError_t glfer_launch_hparma(const SpectroParams *sp, int n, int t, int ncol, const int *rot_sched, int rot_steps, int rot_width, const uint16_t *lagmap,
                                          const float2 *unit, hipStream_t st);
extern "C" hipError_t glfer_launch_floor(const float *psd, size_t nframes, int bins, int pitch, int m, float *stats,
                                         hipStream_t st);
extern "C" hipError_t glfer_launch_avg(int mode, const float *psd, size_t nframes, int bins, int n_out,
                                       int depth, int minbin, int maxbin, int max0, double *avg,
                                       double *ret, hipStream_t st);
extern "C" int glfer_avgmap_applies(size_t walk, int bins, int depth, int minbin, int maxbin);
extern "C" hipError_t glfer_launch_avgmap(int mode, const float *psd, size_t fbeg, size_t nframes, int bins, int depth,
                                          int minbin, int maxbin, int max0, int scale_log, double thr255,
                                          double one_m_thr, const float *levels, const unsigned char *colortab,
                                          const double *log_thr, unsigned char *rgb, short *lev, hipStream_t st);
extern "C" hipError_t glfer_launch_avg_cum(const float *psd, size_t nframes, int bins, int n_out, int depth,
                                           int minbin, int maxbin, double *cum, hipStream_t st);
extern "C" hipError_t glfer_launch_lmp(const float *rows, long long row0, long long first, size_t nframes, int bins,
                                       int nl, float *out, hipStream_t st);
extern "C" hipError_t glfer_launch_ftest(const float *spec, size_t nframes, int n, int ntap, const double *U0,
                                         float sum_U0_sqr, int mu_live, float *ftest, hipStream_t st);
extern "C" hipError_t glfer_launch_submean_tail_ex(const void *raw_last, const float *prev, float *out, int H, int fresh, int exact,
                                                   int fmt, hipStream_t st);
extern "C" hipError_t glfer_launch_hop_means_seq(const void *in, float *means, int H, long long nhops, int fmt, hipStream_t st);
extern "C" hipError_t glfer_launch_hop_means_tiled(const void *in, float *means, int H, long long nhops, int fmt, int hpw, unsigned blocks,
                                                   hipStream_t st);
extern "C" hipError_t glfer_launch_prepare(const SpectroParams *p, int n, const float *window, float *out,
                                           hipStream_t st);

static thread_local std::string g_hip_err;

namespace glfer {
int hip_fail(hipError_t e, const char *what) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  g_hip_err = buf;
  return GLFER_E_HIP;
}

std::string error_text() { return g_hip_err; }
void set_error_text(const std::string &text) { g_hip_err = text; }

int device_of(const void *d_ptr) {
  hipPointerAttribute_t at;
  if (!d_ptr || hipPointerGetAttributes(&at, d_ptr) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged ? at.device : -1;
}
// ---- scratch.  Three sources, by size:
//   * under 1 MiB (statistics, tables, level states): a stream-ordered pool of their own, so that
//     they never carve a piece out of a large free block of the default pool;
//   * 1 MiB .. 16 MiB: the device's default stream-ordered pool, release threshold raised (by default
//     it hands everything back at the next synchronisation);
//   * 16 MiB and more (averaged rows, a mean-corrected copy of a stream, the big-block scratch,
//     LMP / F-test spectra: hundreds of MB to GB): blocks this library keeps (plain hipMalloc, size
//     classes a sixteenth of a power of two apart, at most kBigBlocks per device, never returned
//     before the process ends).  The stream-ordered pool is not made for these: tools/pooltail on this
//     stack -- 2.2 GiB malloc + touch + free + sync per call, sizes alternating by 64 KB: median 2.0 ms
//     a call and ONE CALL IN THREE over 10 ms, up to 4 s (a waterfall call with 2 GiB of averaged
//     rows: 2.3 ms, one in ~100 taking 4.4 s); the same through one kept block: 17 us, never more than
//     24.  A kept block is handed out under a mutex; giving it back records an event on the stream
//     that used it, and the next taker on ANOTHER stream waits for that event first (stream order
//     preserved, nothing synchronised on the host).  When all kept blocks are out (concurrent calls)
//     or would not fit, the request falls through to the pool.  GLFER_SCRATCH_CACHE=0 turns it off.
namespace {
constexpr int kBigBlocks = 6;
constexpr size_t kBigMin = (size_t)16 << 20;
struct BigBlock {
  void *p = nullptr;
  size_t cap = 0;
  bool out = false;
  hipEvent_t freed = nullptr;
  hipStream_t last = nullptr;
  bool used = false;                                 // an event has been recorded on it
  int dev = -1;                                      // the device it was allocated on
};
struct DeviceScratch {
  bool init = false;
  hipMemPool_t small_pool = nullptr;
  BigBlock big[kBigBlocks];
};
std::mutex g_scratch_mu;
DeviceScratch g_scratch[64];
// Bytes of kept blocks per device above which IDLE blocks are given back before a new one is made
// (GLFER_SCRATCH_CAP_MB; default 16 GiB of a 288 GB card; glfer_hip_scratch_limit sets it at run time).
// A single request larger than the cap is still served -- by a kept block that is the only one.
size_t g_scratch_cap = [] {
  const char *e = getenv("GLFER_SCRATCH_CAP_MB");
  return e && *e ? (size_t)strtoull(e, nullptr, 10) << 20 : (size_t)16 << 30;
}();
// Frees one idle kept block (after the work recorded on it has completed).  Caller holds the mutex
// and has the block's device current.
void drop_block(BigBlock &b) {
  if (b.used) (void)hipEventSynchronize(b.freed);
  (void)hipFree(b.p);
  (void)hipEventDestroy(b.freed);
  b = BigBlock();
}
// Idle kept blocks of a device, smallest first, until at most `keep_bytes` stay (blocks that are out
// with a call stay and count).  Returns the bytes freed.
size_t trim_locked(DeviceScratch &ds, size_t keep_bytes) {
  size_t held = 0, freed = 0;
  for (BigBlock &b : ds.big)
    if (b.p) held += b.cap;
  while (held > keep_bytes) {
    BigBlock *victim = nullptr;
    for (BigBlock &b : ds.big)
      if (b.p && !b.out && (!victim || b.cap < victim->cap)) victim = &b;
    if (!victim) break;
    held -= victim->cap;
    freed += victim->cap;
    drop_block(*victim);
  }
  return freed;
}
bool scratch_cache_on() {
  static const bool on = [] {
    const char *e = getenv("GLFER_SCRATCH_CACHE");
    return !(e && atoi(e) == 0 && *e);
  }();
  return on;
}
}  // namespace

bool scratch_keeping() { return scratch_cache_on(); }
size_t scratch_cap() {
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  return g_scratch_cap;
}

hipError_t scratch_malloc(void **p, size_t bytes, hipStream_t st) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipMallocAsync(p, bytes, st);
  std::unique_lock<std::mutex> lock(g_scratch_mu);
  DeviceScratch &ds = g_scratch[dev];
  if (!ds.init) {
    ds.init = true;
    hipMemPool_t pool = nullptr;
    uint64_t keep = (uint64_t)8 << 30;
    if (hipDeviceGetDefaultMemPool(&pool, dev) != hipSuccess ||
        hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep) != hipSuccess)
      (void)hipGetLastError();                   // the pool keeps its default: slower, not wrong
    hipMemPoolProps props = {};
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = dev;
    uint64_t keep_small = (uint64_t)64 << 20;
    if (hipMemPoolCreate(&ds.small_pool, &props) != hipSuccess ||
        hipMemPoolSetAttribute(ds.small_pool, hipMemPoolAttrReleaseThreshold, &keep_small) != hipSuccess) {
      (void)hipGetLastError();                   // no second pool: everything from the default one
      ds.small_pool = nullptr;
    }
  }
  if (bytes < ((size_t)1 << 20) && ds.small_pool) {
    hipMemPool_t from = ds.small_pool;
    lock.unlock();
    return hipMallocFromPoolAsync(p, bytes, from, st);
  }
  if (bytes >= kBigMin && scratch_cache_on()) {
    size_t cls = (size_t)1 << 20;                    // size classes: a batch one row longer finds the block of the last call
    while (cls * 32 <= bytes) cls <<= 1;
    const size_t want = (bytes + cls - 1) / cls * cls;
    BigBlock *best = nullptr, *empty = nullptr;
    for (BigBlock &b : ds.big) {
      if (!b.p) {
        if (!empty) empty = &b;
      } else if (!b.out && b.cap >= want && (!best || b.cap < best->cap)) {
        best = &b;
      }
    }
    if (!best && !empty) {                           // every slot taken by smaller blocks: the smallest idle one makes room
      BigBlock *victim = nullptr;
      for (BigBlock &b : ds.big)
        if (!b.out && b.cap < want && (!victim || b.cap < victim->cap)) victim = &b;
      if (victim) {
        drop_block(*victim);
        empty = victim;
      }
    }
    if (!best && empty) {                            // a new kept block (synchronous hipMalloc: once per size class)
      // the byte cap: idle blocks go first (none of them fits -- `best` would have been one)
      size_t held = 0;
      for (BigBlock &b : ds.big)
        if (b.p) held += b.cap;
      if (held + want > g_scratch_cap) trim_locked(ds, g_scratch_cap > want ? g_scratch_cap - want : 0);
      void *q = nullptr;
      hipEvent_t ev = nullptr;
      hipError_t me = hipMalloc(&q, want);
      if (me != hipSuccess) {                        // out of memory: every idle kept block goes back, then once more
        (void)hipGetLastError();
        q = nullptr;
        if (trim_locked(ds, 0) > 0) me = hipMalloc(&q, want);
        for (BigBlock &b : ds.big)                   // the slot may have moved: any empty one will do
          if (!b.p) { empty = &b; break; }
      }
      if (me == hipSuccess && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
        empty->p = q;
        empty->cap = want;
        empty->freed = ev;
        empty->used = false;
        empty->dev = dev;
        best = empty;
      } else {
        (void)hipGetLastError();
        if (q) (void)hipFree(q);
      }
    }
    if (best) {
      best->out = true;
      hipError_t e = hipSuccess;
      if (best->used && best->last != st) e = hipStreamWaitEvent(st, best->freed, 0);
      if (e != hipSuccess) {
        best->out = false;
        return e;
      }
      *p = best->p;
      return hipSuccess;
    }
  }
  lock.unlock();
  return hipMallocAsync(p, bytes, st);
}

// gives back what scratch_malloc handed out (stream-ordered either way)
void scratch_free(void *q, hipStream_t st) {
  if (!q) return;
  {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    for (DeviceScratch &ds : g_scratch)
      for (BigBlock &b : ds.big)
        if (b.p == q) {
          // the event is recorded on the stream that used the block -- a stream of the block's own
          // device, which need not be the caller's current one
          int cur = -1;
          const bool moved = hipGetDevice(&cur) == hipSuccess && cur != b.dev && b.dev >= 0 && hipSetDevice(b.dev) == hipSuccess;
          if (hipEventRecord(b.freed, st) == hipSuccess) {
            b.used = true;
            b.last = st;
          } else {
            // no event to order the next taker behind: wait here for the work on `st`, after which
            // the block is free for any stream (used = false: nothing to wait for)
            (void)hipGetLastError();
            if (hipStreamSynchronize(st) != hipSuccess) {
              (void)hipGetLastError();               // not even that: the block leaves the cache for good
              (void)hipDeviceSynchronize();
              (void)hipFree(b.p);
              (void)hipEventDestroy(b.freed);
              b = BigBlock();
              if (moved) (void)hipSetDevice(cur);
              return;
            }
            b.used = false;
            b.last = nullptr;
          }
          b.out = false;
          // the cap bounds what is KEPT, not only what is newly made: a block handed back while the
          // device holds more than the cap -- a reused block larger than a cap set since, glfer_hip_scratch_limit(0)
          // -- goes back now (drop_block waits for the event just recorded, then hipFree)
          if (b.dev >= 0 && b.dev < 64) trim_locked(g_scratch[b.dev], g_scratch_cap);
          if (moved) (void)hipSetDevice(cur);
          return;
        }
  }
  (void)hipFreeAsync(q, st);
}

// glfer_hip_scratch_trim / glfer_hip_scratch_limit (include/glfer_hip.h)
size_t scratch_trim(int dev, size_t keep_bytes) {
  if (dev < 0 || dev >= 64) return 0;
  DeviceGuard guard(dev);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  return trim_locked(g_scratch[dev], keep_bytes);
}
size_t scratch_held(int dev) {
  if (dev < 0 || dev >= 64) return 0;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  size_t held = 0;
  for (BigBlock &b : g_scratch[dev].big)
    if (b.p) held += b.cap;
  return held;
}
void scratch_set_cap(size_t bytes) {
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  g_scratch_cap = bytes;
  // idle blocks above the new cap go back at once, on every device that holds any
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) cur = -1;
  for (int dev = 0; dev < 64; dev++) {
    size_t held = 0;
    for (BigBlock &b : g_scratch[dev].big)
      if (b.p) held += b.cap;
    if (held <= bytes) continue;
    if (dev != cur && hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); continue; }
    trim_locked(g_scratch[dev], bytes);
    if (dev != cur && cur >= 0) (void)hipSetDevice(cur);
  }
}

// Dynamic LDS above the default limit has to be allowed per kernel -- and per DEVICE: a process that
// drives several GPUs (glfer_hip_spectrogram_host_multi) launches the same kernel on each.  One call
// per (device, kernel) and size class, not one per launch.
hipError_t allow_dynamic_lds(const void *kernel, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void *>, size_t> allowed;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  size_t &have = allowed[{dev, kernel}];
  if (bytes <= have) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}

}  // namespace glfer
using glfer::DeviceGuard;
using glfer::hip_fail;

// the device the data of a plan-less entry lives on: the pointer's, else the current one
static int data_device(const void *d_ptr) {
  int dev = glfer::device_of(d_ptr);
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
  return dev;
}

static bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// Device layout of the taper tables, the order the kernel's lanes read them in: sample i of a
// frame belongs to lane t = i mod T at register m = i / T (T = n/16 lanes per frame); a lane
// fetches registers m, m+1 of both tapers of a pair with one 16-byte load:
//   taps[pair][m/2][t][4] = { taper 2p @m, taper 2p+1 @m, taper 2p @m+1, taper 2p+1 @m+1 }
static size_t tap_slot(int n, int pair, int i, int which) {
  if (n < 256) return ((size_t)pair * n + i) * 2 + which;       // spectro_small.hip: [pair][i][2]
  const int T = n / 16, t = i % T, m = i / T;
  return (size_t)pair * n * 2 + ((size_t)(m / 2) * T + t) * 4 + (size_t)(m & 1) * 2 + which;
}

extern "C" {

const char *glfer_hip_version(void) { return "glfer_hip 0.9 (gfx950; 16 points/lane Stockham radix-16 with LDS exchange, inter-pass twiddles folded into the butterflies: packed pairs, real-input and wavefront-private forms, shared odd taper, register reuse across overlapped frames, mean removal in the reference's summation order (cfg.sub_mean = 1) or inside the kernels; N = 8..1048576; periodogram, multitaper + F-test, HP-ARMA with column-disjoint Jacobi rotations side by side and frames from a queue, LMP; rows at a caller's pitch; wavefront floor, fused average, display map with the average taken inside it; chunk ring for ingest with uploads and downloads at once, WAV files and waterfalls over several GPUs, read-ahead behind the per-hop shim, kept scratch blocks with a cap, workers bound to their GPU's NUMA node)"; }

int glfer_hip_abi_version(void) { return GLFER_HIP_ABI; }

int glfer_hip_palette(int palette, unsigned char colortab[768]) {
  if (!colortab) return GLFER_E_ARG;
  glfer::make_palette(palette, colortab);
  return GLFER_OK;
}

}  // extern "C"

// update_avg_* inside the mapping (avg_fused_kernel<.., MAP>): the PSD batch the columns belong to,
// their first row in it, and update_avg's arguments
struct MapAverages {
  int mode, depth, minbin, maxbin, max0;
  const float *d_batch;
  size_t first;
};

// The mapping part of main_window_draw for `nframes` columns: level tracking, then the pixel map of the
// PSD rows, of averaged rows, or (fused) of the averages taken on the way.
// d_levels_in: the columns' levels already known (the walk was done elsewhere -- over ALL columns of a
// waterfall whose rows are spread over several GPUs, glfer_hip_levels_host): no walk, d_stats unused,
// the carried state in *d untouched.
static int display_columns(glfer_hip_display *d, const float *d_psd, const double *d_avg, const MapAverages *fused,
                           const float *d_stats, size_t nframes, int bins, unsigned char *d_rgb, short *d_lev,
                           float *d_levels, void *hip_stream, const float *d_levels_in = nullptr, int psd_pitch = 0) {
  if (!d || (!d_stats && !d_levels_in) || !d_rgb || bins < 1) return GLFER_E_ARG;
  if (psd_pitch == 0) psd_pitch = bins;                  // floats from one row of d_psd to the next (cfg.psd_pitch)
  if (psd_pitch < bins || (fused && psd_pitch != bins)) return GLFER_E_ARG;
  if ((d_psd != nullptr) + (d_avg != nullptr) + (fused != nullptr) != 1) return GLFER_E_ARG;
  if (d->scale_type < GLFER_SCALE_LIN || d->scale_type > GLFER_SCALE_LOG_MAX0) return GLFER_E_ARG;
  if (nframes == 0) return GLFER_OK;
  hipStream_t st = (hipStream_t)hip_stream;
  DeviceGuard guard(data_device(d_rgb));
  HIP_TRY(guard.error());
  const int scale_log = d->scale_type == GLFER_SCALE_LOG || d->scale_type == GLFER_SCALE_LOG_MAX0;

  // one stream-ordered allocation: the palette, the table of the dB steps, the levels rows (when
  // the caller does not want them) and the chunk states of the autoscale walk
  const size_t lev_floats = (d_levels || d_levels_in) ? 0 : nframes * 4;
  const size_t st_floats = (d->autoscale && !d_levels_in) ? glfer_levels_scratch_floats(nframes) : 0;
  const size_t thr_bytes = (2 * glfer::kLogThrK + 1) * sizeof(double);
  unsigned char *scratch = nullptr;
  HIP_TRY(glfer::scratch_malloc((void **)&scratch, 768 + thr_bytes + (lev_floats + st_floats) * sizeof(float), st));
  unsigned char *d_tab = scratch;
  double *d_thr = reinterpret_cast<double *>(scratch + 768);
  float *fs = reinterpret_cast<float *>(scratch + 768 + thr_bytes);
  float *levels = d_levels_in ? const_cast<float *>(d_levels_in) : (d_levels ? d_levels : fs);
  float *chunk_state = st_floats ? fs + lev_floats : nullptr;
  int rc = GLFER_OK;
  auto fail = [&](hipError_t err) { rc = hip_fail(err, "glfer_hip_display_device"); };
  unsigned char tab[768];
  glfer::make_palette(d->palette, tab);
  // pageable source: hipMemcpyAsync stages it before returning, so `tab` may go out of scope
  hipError_t e = hipMemcpyAsync(d_tab, tab, 768, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_thr, glfer::log_thresholds(), thr_bytes, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) fail(e);

  if (rc == GLFER_OK && !d_levels_in) {
    if (d->autoscale) {
      e = glfer_launch_levels(d_stats, nframes, scale_log, 1, d->first_buffer, d->overlap, d->display_max_lvl,
                              d->display_min_lvl, levels, chunk_state, st);
    } else {                                                                   // g_main.c:1125-1139
      float mx = pow(10.0, d->max_level_db / 10.0);
      float mn = pow(10.0, d->min_level_db / 10.0);
      mn = (mx > mn ? mn : mx / 10.0);
      const float dmax = scale_log ? (float)(10.0 * log10(mx)) : mx;
      const float dmin = scale_log ? (float)(10.0 * log10(mn)) : mn;
      e = glfer_launch_levels_fixed(nframes, dmax, dmin, mx, mn, levels, st);
    }
    if (e != hipSuccess) fail(e);
  }
  if (rc == GLFER_OK) {
    const float thr_level = d->thr_level / 100.0;                              // g_main.c:1099
    if (fused)
      e = glfer_launch_avgmap(fused->mode, fused->d_batch, fused->first, fused->first + nframes, bins, fused->depth,
                              fused->minbin, fused->maxbin, fused->max0, scale_log, 255.0 * thr_level,
                              1.0 - thr_level, levels, d_tab, d_thr, d_rgb, d_lev, st);
    else
      e = glfer_launch_map(d_psd, d_avg, nframes, bins, psd_pitch, scale_log, 255.0 * thr_level, 1.0 - thr_level, levels,
                           d_tab, d_thr, d_rgb, d_lev, st);
    if (e != hipSuccess) fail(e);
  }
  float last[4] = {0, 0, 0, 0};
  if (rc == GLFER_OK && !d_levels_in) {
    e = hipMemcpyAsync(last, levels + (nframes - 1) * 4, sizeof last, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) fail(e);
  }
  glfer::scratch_free(scratch, st);
  e = hipStreamSynchronize(st);              // the carried state comes back to the host
  if (e != hipSuccess && rc == GLFER_OK) fail(e);
  if (rc == GLFER_OK && !d_levels_in) {
    d->display_max_lvl = last[2];
    d->display_min_lvl = last[3];
    if (d->autoscale) d->first_buffer = 0;                                     // g_main.c:1120
  }
  return rc;
}

extern "C" {

int glfer_hip_display_device(glfer_hip_display *d, const float *d_psd, const double *d_avg, const float *d_stats,
                             size_t nframes, int bins, unsigned char *d_rgb, short *d_lev, float *d_levels,
                             void *hip_stream) {
  if (!d || (d_psd == nullptr) == (d_avg == nullptr)) return GLFER_E_ARG;
  return display_columns(d, d_psd, d_avg, nullptr, d_stats, nframes, bins, d_rgb, d_lev, d_levels, hip_stream, nullptr, d->psd_pitch);
}

// compute_floor + update_avg_* + the display mapping of main_window_draw (g_main.c:1109-1236) for a
// batch of PSD rows in one call.  The level tracking is a chain over the columns (g_main.c:1122-1123)
// fed by every column's floor statistics, so a row is read twice -- once for its statistics (and its
// moving sums), once to be mapped -- and the chain walk costs ~0.1 us per column of LATENCY however
// few columns it is given (its chunks warm up over the 4096 columns before them, display.hip).
// Cache-sized tiles (rows still in the 256 MiB Infinity Cache when the map reads them) were
// measured: 15 M rows/s against 117 M stage by stage over the whole batch
// (profiles/r02_aux_sweep.txt) -- the chain's latency per tile swamps the saved HBM read.  So the
// stages run over tiles that only bound the scratch (averaged rows: 8 B/bin, 4 GiB per tile).
}  // extern "C"

// Rows [row0, row0 + nframes) of the batch d_psd.  row0 > 0 / d_levels_in: the second phase of a
// waterfall whose columns are spread over several GPUs (glfer_hip_waterfall_map_device) -- the moving
// sums of the first rows reach back into the batch's rows before row0, the levels are given.
static int waterfall_columns(glfer_hip_display *d, int avg_mode, int depth, int minbin, int maxbin, int max0,
                             const float *d_batch, size_t row0, size_t nframes, int bins, unsigned char *d_rgb, short *d_lev,
                             float *d_stats, void *hip_stream, const float *d_levels_in, int pitch = 0) {
  if (pitch == 0) pitch = bins;                          // floats from one PSD row to the next (cfg.psd_pitch)
  const float *d_psd = d_batch ? d_batch + row0 * (size_t)pitch : nullptr;
  if (!d || !d_psd || !d_rgb || bins < 1 || bins > 32769 || pitch < bins) return GLFER_E_ARG;
  const bool averaging = avg_mode != 0;
  if (averaging && (avg_mode < GLFER_AVG_SUMAVG || avg_mode > GLFER_AVG_SUMEXTREME || depth < 1 || minbin < 0 ||
                    maxbin <= minbin || maxbin > bins))
    return GLFER_E_ARG;
  if (nframes == 0) return GLFER_OK;
  hipStream_t st = (hipStream_t)hip_stream;
  DeviceGuard guard(data_device(d_psd));
  HIP_TRY(guard.error());
  // Tiles only bound the scratch: the level walk costs its ~0.13 ms of latency per display call
  // however few columns it gets, so fewer, larger tiles are faster (two 65536-row tiles: 116 M rows/s
  // against 147 M stage by stage).  The averages are taken inside the mapping kernel where that form
  // applies (no averaged rows in memory at all: GLFER_WATERFALL_FUSED=0 forces the staged form, for
  // A/B runs and tests); staged, averaged rows are 8 B per bin: 4 GiB of them per tile.  Otherwise
  // only the 16 B of statistics per row are scratch.
  bool fused = averaging && pitch == bins;               // (the average-in-the-map kernel walks dense rows)
  if (const char *e = getenv("GLFER_WATERFALL_FUSED")) fused = fused && atoi(e) != 0;
  size_t tile = (averaging && !fused) ? std::max<size_t>(16384, ((size_t)4 << 30) / ((size_t)bins * sizeof(double))) : (size_t)1 << 22;
  if (const char *e = getenv("GLFER_WATERFALL_TILE")) {    // rows per tile, for tests of the tile seams and for tuning
    const long v = atol(e);
    if (v >= 64) tile = (size_t)v;
  }
  tile = std::min(tile, nframes);
  tile = (nframes + (nframes + tile - 1) / tile - 1) / ((nframes + tile - 1) / tile);   // equal tiles: no short last one
  // the form's chunking depends on the number of rows it walks, and the last tile may be a few rows
  // shorter than the others: the fused form is taken only if it applies to BOTH lengths
  const size_t last_tile = nframes - (nframes - 1) / tile * tile;
  if (fused) fused = glfer_avgmap_applies(tile, bins, depth, minbin, maxbin) != 0 &&
                     glfer_avgmap_applies(last_tile, bins, depth, minbin, maxbin) != 0;
  const size_t back = averaging ? (size_t)depth : 0;       // rows re-read in front of a tile to restart the sliding sums
  float *stats = d_stats;
  double *avg = nullptr, *ret = nullptr;
  if (!stats && !d_levels_in) HIP_TRY(glfer::scratch_malloc((void **)&stats, tile * 4 * sizeof(float), st));
  int rc = GLFER_OK;
  if (averaging && !fused) {
    hipError_t e = glfer::scratch_malloc((void **)&avg, (tile + back) * (size_t)bins * sizeof(double), st);
    if (e == hipSuccess) e = glfer::scratch_malloc((void **)&ret, (tile + back) * 4 * sizeof(double), st);
    if (e != hipSuccess) rc = hip_fail(e, "hipMallocAsync(waterfall tile)");
  }
  for (size_t f0 = 0; rc == GLFER_OK && f0 < nframes; f0 += tile) {
    const size_t nf = std::min(tile, nframes - f0);
    float *tstats = d_stats ? d_stats + f0 * 4 : stats;
    const float *tlevels = d_levels_in ? d_levels_in + f0 * 4 : nullptr;
    unsigned char *trgb = d_rgb + f0 * (size_t)bins * 3;
    short *tlev = d_lev ? d_lev + f0 * (size_t)bins : nullptr;
    if (tstats) rc = glfer_hip_floor_device_pitched(d_psd + f0 * (size_t)pitch, nf, bins, pitch, tstats, st);
    if (rc != GLFER_OK) break;
    if (fused) {
      // the sliding sums of the tile's first rows reach back into the rows before it by themselves
      const MapAverages ma{avg_mode, depth, minbin, maxbin, max0 ? 1 : 0, d_batch, row0 + f0};
      rc = display_columns(d, nullptr, nullptr, &ma, tstats, nf, bins, trgb, tlev, nullptr, st, tlevels);
      continue;
    }
    const double *src_avg = nullptr;
    if (averaging) {
      // the sums of the tile's first rows reach `depth` rows back: run from there (from an empty
      // state at row 0 of the batch, as update_avg does after alloc_avg) and use the tile's rows
      const size_t lead = std::min(back, row0 + f0);
      // (update_avg's `bins` is its rows' stride; the band is minbin..maxbin, the averaged rows are dense)
      rc = glfer_hip_avg_device(avg_mode, d_batch + (row0 + f0 - lead) * (size_t)pitch, nf + lead, pitch, bins, depth, minbin, maxbin, max0,
                                avg, ret, st);
      src_avg = avg + lead * (size_t)bins;
    }
    if (rc == GLFER_OK)
      rc = display_columns(d, averaging ? nullptr : d_psd + f0 * (size_t)pitch, src_avg, nullptr, tstats, nf, bins, trgb, tlev,
                           nullptr, st, tlevels, pitch);
  }
  if (avg) glfer::scratch_free(avg, st);
  if (ret) glfer::scratch_free(ret, st);
  if (!d_stats && stats) glfer::scratch_free(stats, st);
  return rc;
}

extern "C" {

int glfer_hip_waterfall_device(glfer_hip_display *d, int avg_mode, int depth, int minbin, int maxbin, int max0,
                               const float *d_psd, size_t nframes, int bins, unsigned char *d_rgb, short *d_lev,
                               float *d_stats, void *hip_stream) {
  return waterfall_columns(d, avg_mode, depth, minbin, maxbin, max0, d_psd, 0, nframes, bins, d_rgb, d_lev, d_stats, hip_stream,
                           nullptr, d ? d->psd_pitch : 0);
}

// The two halves of glfer_hip_waterfall_device for columns that live on several GPUs.  The level
// tracking (g_main.c:1111-1124) is ONE chain over all columns, fed by 16 bytes of floor statistics
// per column; everything else is per column.  So: every GPU computes its rows and their statistics,
// the statistics meet on the host, ONE walk over them (here: on `device`, the walk of
// glfer_hip_display_device) gives every column its levels, and every GPU maps its own rows with its
// slice of the levels.  Nothing but 16 + 16 bytes per column crosses between GPUs, through the host.
int glfer_hip_levels_host(glfer_hip_display *d, const float *h_stats, size_t nframes, float *h_levels, int device) {
  if (!d || !h_stats || !h_levels) return GLFER_E_ARG;
  if (d->scale_type < GLFER_SCALE_LIN || d->scale_type > GLFER_SCALE_LOG_MAX0) return GLFER_E_ARG;
  if (nframes == 0) return GLFER_OK;
  DeviceGuard guard(device);
  HIP_TRY(guard.error());
  const int scale_log = d->scale_type == GLFER_SCALE_LOG || d->scale_type == GLFER_SCALE_LOG_MAX0;
  const size_t st_floats = d->autoscale ? glfer_levels_scratch_floats(nframes) : 0;
  float *buf = nullptr;
  HIP_TRY(glfer::scratch_malloc((void **)&buf, (nframes * 8 + st_floats) * sizeof(float), nullptr));
  float *d_stats = buf, *levels = buf + nframes * 4, *chunk_state = st_floats ? buf + nframes * 8 : nullptr;
  int rc = GLFER_OK;
  hipError_t e = hipMemcpyAsync(d_stats, h_stats, nframes * 4 * sizeof(float), hipMemcpyHostToDevice, nullptr);
  if (e == hipSuccess) {
    if (d->autoscale) {
      e = glfer_launch_levels(d_stats, nframes, scale_log, 1, d->first_buffer, d->overlap, d->display_max_lvl,
                              d->display_min_lvl, levels, chunk_state, nullptr);
    } else {                                                                   // g_main.c:1125-1139
      float mx = pow(10.0, d->max_level_db / 10.0);
      float mn = pow(10.0, d->min_level_db / 10.0);
      mn = (mx > mn ? mn : mx / 10.0);
      const float dmax = scale_log ? (float)(10.0 * log10(mx)) : mx;
      const float dmin = scale_log ? (float)(10.0 * log10(mn)) : mn;
      e = glfer_launch_levels_fixed(nframes, dmax, dmin, mx, mn, levels, nullptr);
    }
  }
  if (e == hipSuccess) e = hipMemcpyAsync(h_levels, levels, nframes * 4 * sizeof(float), hipMemcpyDeviceToHost, nullptr);
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  glfer::scratch_free(buf, nullptr);
  if (e != hipSuccess) rc = hip_fail(e, "glfer_hip_levels_host");
  if (rc == GLFER_OK) {
    d->display_max_lvl = h_levels[(nframes - 1) * 4 + 2];
    d->display_min_lvl = h_levels[(nframes - 1) * 4 + 3];
    if (d->autoscale) d->first_buffer = 0;                                     // g_main.c:1120
  }
  return rc;
}

int glfer_hip_waterfall_map_device(const glfer_hip_display *d, int avg_mode, int depth, int minbin, int maxbin, int max0,
                                   const float *d_batch, size_t first, size_t nframes, int bins, const float *d_levels,
                                   unsigned char *d_rgb, short *d_lev, void *hip_stream) {
  if (!d || !d_levels) return GLFER_E_ARG;
  glfer_hip_display copy = *d;               // the carried state is not touched: the walk was glfer_hip_levels_host's
  return waterfall_columns(&copy, avg_mode, depth, minbin, maxbin, max0, d_batch, first, nframes, bins, d_rgb, d_lev, nullptr,
                           hip_stream, d_levels, d->psd_pitch);
}

size_t glfer_hip_scratch_trim(int device, size_t keep_bytes) {
  size_t ring = 0;
  if (keep_bytes == 0 && device >= 0 && device < 64) {       // "give everything back": the parked ingest ring too
    DeviceGuard guard(device);
    if (guard.error() == hipSuccess) {
      ring = glfer::workers_kept_bytes(device);
      glfer::workers_drop_kept();                              // (their plans park one ring per device on the way out: dropped next)
      ring += glfer::ingest_ring_spare_bytes(device);
      glfer::ingest_ring_drop_spare(device);
    }
  }
  return ring + glfer::scratch_trim(device, keep_bytes);      // (what glfer_hip_scratch_held counted and is gone)
}
size_t glfer_hip_scratch_held(int device) { return glfer::scratch_held(device) + glfer::ingest_ring_spare_bytes(device) + glfer::workers_kept_bytes(device); }
void glfer_hip_scratch_limit(size_t bytes) {
  glfer::scratch_set_cap(bytes);
  if (bytes == 0) glfer::workers_drop_kept();
  // a parked chunk ring larger than the new cap goes back too (it is pinned host + device memory the host did not ask to keep)
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) cur = -1;
  for (int dev = 0; dev < 64; dev++)
    if (glfer::ingest_ring_spare_bytes(dev) > bytes) {
      DeviceGuard guard(dev);
      if (guard.error() == hipSuccess) glfer::ingest_ring_drop_spare(dev);
    }
}

const char *glfer_hip_strerror(int code) {
  switch (code) {
    case GLFER_OK: return "ok";
    case GLFER_E_ARG: return "bad argument or unsupported configuration";
    case GLFER_E_HIP: return "HIP runtime error";
    case GLFER_E_NOMEM: return "out of memory";
    case GLFER_E_NUMERIC: return "DPSS eigen-solve did not converge";
  }
  return "unknown error";
}

const char *glfer_hip_last_hip_error(void) { return g_hip_err.c_str(); }

int glfer_hip_plan_create(const glfer_hip_config *cfg, glfer_hip_plan **out) {
  if (!cfg || !out) return GLFER_E_ARG;
  *out = nullptr;
  const int n = cfg->n;
  // any power of two the user can type (g_options.c:386-387): 8 .. 128 through spectro_small.hip,
  // 256 .. 16384 through the 16-points-per-lane kernels, 32768 through spectro16w.hip alone, 65536 .. 1048576
  // through spectro_big.hip (sub-transforms and combine as two kernels around a scratch in HBM)
  if (!is_pow2(n) || n < 8 || n > (1 << 20)) return GLFER_E_ARG;
  // HP-ARMA: the frame, its autocorrelation's zero tail and the matrix live in one wavefront's LDS -- any power of two from 32 up to what
  // fits 160 KB (N = 32768 with t = 128, p_e = 32: 137 KB, one frame in flight per CU); round 5 (it was 256 .. 16384)
  if (cfg->mode == GLFER_MODE_HPARMA && (n < 32 || n > 32768)) return GLFER_E_ARG;
  const bool small = n < 256, huge = n > 16384;
  if (!(cfg->overlap >= 0.0f) || !(cfg->overlap < 1.0f)) return GLFER_E_ARG;   // g_options.c:1030
  if (cfg->mode != GLFER_MODE_FFT && cfg->mode != GLFER_MODE_MTM && cfg->mode != GLFER_MODE_HPARMA &&
      cfg->mode != GLFER_MODE_LMP)
    return GLFER_E_ARG;
  if (cfg->mode == GLFER_MODE_LMP && (cfg->lmp_av < 1 || cfg->lmp_av > 4096)) return GLFER_E_ARG;
  if (cfg->mode == GLFER_MODE_HPARMA) {
    const int t = cfg->hparma_t, ncol = cfg->hparma_p_e + 1;
    if (t < 2 || ncol < 2 || ncol > t || t > n || ncol > 256 || t > 65535) return GLFER_E_ARG;   // p_e+1 <= t (hparma.c:107)
    const size_t xlen = (size_t)n + (t <= 128 ? 128 : 0);          // (glfer_launch_hparma: the frame and its zero tail)
    const size_t big = xlen > (size_t)t * ncol ? xlen : (size_t)t * ncol;
    if ((big + (size_t)ncol * ncol + t + 2 * ncol) * sizeof(float) > 160 * 1024) return GLFER_E_ARG;
  }
  if (cfg->sample_format < 0 || cfg->sample_format > 2) return GLFER_E_ARG;
  if (cfg->mode == GLFER_MODE_MTM && (cfg->mtm_k < 0 || cfg->mtm_k > 31 || !(cfg->mtm_w > 0.0f)))
    return GLFER_E_ARG;
  if (cfg->mode == GLFER_MODE_FFT && (cfg->window_type < 0 || cfg->window_type > 7)) return GLFER_E_ARG;

  glfer_hip_plan *p = new (std::nothrow) glfer_hip_plan();
  if (!p) return GLFER_E_NOMEM;
  p->cfg = *cfg;
  p->n = n;
  p->hop = (int)(n * (1.0 - cfg->overlap));                        // fft.c:70
  p->keep = n - p->hop;                                            // fft.c:71
  p->bins = n / 2 + 1;
  p->lanes = n / 64;
  if (p->hop <= 0) { delete p; return GLFER_E_ARG; }
  // cfg.psd_pitch: floats from one PSD row to the next in the DEVICE entries' d_psd (0 = dense, N/2+1).  The LMP statistic
  // and the host / file entries (their rows go home through a dense ring) keep dense rows.
  p->pitch = cfg->psd_pitch ? cfg->psd_pitch : p->bins;
  if (cfg->psd_pitch < 0 || p->pitch < p->bins || (cfg->psd_pitch && cfg->mode == GLFER_MODE_LMP)) { delete p; return GLFER_E_ARG; }

  // --- host tables
  p->window.assign(n, 1.0f);
  std::vector<float> taps;
  // LMP (lmp.c:101-181): the periodogram of the raw assembled frame -- lmp.c:114-116 overwrites what
  // prepare_audio left in inbuf_fft, so window (rectangular anyway, source.c:395), a and limiter
  // have no effect -- followed by the per-bin statistic over the last lmp_av periodograms
  const bool lmp = cfg->mode == GLFER_MODE_LMP;
  if (lmp) {
    p->cfg.window_type = GLFER_WIN_RECTANGULAR;
    p->cfg.limiter_a = 0.0f;
    p->cfg.enable_limiter = 0;
    p->lmp_av = cfg->lmp_av;
  }
  if (cfg->mode == GLFER_MODE_FFT || lmp) {
    p->ntapers = 1;
    p->npairs = 1;
    glfer::make_window(p->cfg.window_type, n, p->window.data());
    p->nonlin = (p->cfg.limiter_a > 0.0f) || (p->cfg.enable_limiter == 1);
    // psd = |X|^2/N (fft.c:212-216); the pair packing contributes |Z_k|^2+|Z_{N-k}|^2 = 2|X_k|^2
    // (N = 32768 runs the real-input form only: its inputs carry sqrt(1/(4N)), see the w tables below)
    const double scale = std::sqrt(1.0 / ((huge ? 4.0 : 2.0) * n));
    p->spec_unscale = (float)scale;
    // device layout: taps[pair][i] = (taper 2*pair, taper 2*pair+1)[i], interleaved
    taps.assign((size_t)2 * n, 0.0f);
    const bool rect = (p->cfg.window_type == GLFER_WIN_RECTANGULAR);
    for (int i = 0; i < n; i++) {
      const double w = rect ? 1.0 : (double)p->window[i];          // fft.c:132,139: no multiply when rectangular
      taps[tap_slot(n, 0, i, 0)] = p->nonlin ? (float)w : (float)(w * scale);
    }
  } else if (cfg->mode == GLFER_MODE_HPARMA) {
    p->ntapers = 0;
    p->npairs = 0;
    taps.assign(4, 0.0f);
  } else {
    p->ntapers = cfg->mtm_k + 1;                                   // mtm.c:189: j = 0..kmax inclusive
    p->npairs = (p->ntapers + 1) / 2;
    p->tapers.resize((size_t)p->ntapers * n);
    p->sig.resize(p->ntapers);
    // gl_dpss (g-l_dpss.c:288-347) is the expensive part of mtm_init -- 0.1-0.3 s at N = 16384 with 9 tapers -- and the
    // *_multi / *_workers entries make a plan per worker and call: the last few results are kept (same doubles, bit for bit)
    {
      struct Kept { int n, kmax; float w; std::vector<double> tapers, sig; };
      static std::mutex mu;
      static std::vector<Kept> kept;
      bool hit = false;
      {
        std::lock_guard<std::mutex> lock(mu);
        for (const Kept &k : kept)
          if (k.n == n && k.kmax == cfg->mtm_k && k.w == cfg->mtm_w) {
            p->tapers = k.tapers;
            p->sig = k.sig;
            hit = true;
            break;
          }
      }
      if (!hit) {
        if (!glfer::make_dpss(n, cfg->mtm_k, (double)cfg->mtm_w, p->tapers.data(), p->sig.data())) {
          delete p;
          return GLFER_E_NUMERIC;
        }
        std::lock_guard<std::mutex> lock(mu);
        if (kept.size() >= 4) kept.erase(kept.begin());
        kept.push_back(Kept{n, cfg->mtm_k, cfg->mtm_w, p->tapers, p->sig});
      }
    }
    taps.assign((size_t)2 * p->npairs * n, 0.0f);
    for (int j = 0; j < p->ntapers && !huge; j++) {
      // psd += |FFT(v_j x)|^2 / N / (1+sig_j)   (mtm.c:212-219), and the 1/2 of the packing
      const double scale = std::sqrt(1.0 / (2.0 * n * (1.0 + p->sig[j])));
      for (int i = 0; i < n; i++)
        taps[tap_slot(n, j / 2, i, j & 1)] = (float)(p->tapers[(size_t)j * n + i] * scale);
    }
    p->spec_unscale = 1.0f;
  }
  int logn = 0;
  while ((1 << logn) < n) logn++;
  p->lanes = n / 16;
  std::vector<float> tw(2, 0.0f);
  if (!small && !huge) {
    tw.assign((size_t)2 * glfer::make_twiddles16(logn, nullptr) * p->lanes, 0.0f);
    glfer::make_twiddles16(logn, tw.data());
  }
  if (huge) taps.assign(4, 0.0f);                                  // no packed form at this size

  // --- real-input form of the linear periodogram (spectro16h.hip): z[i] = y[2i] + i*y[2i+1],
  // lane t of n/32 holds points i = t + (n/32)*m; |X|^2/N needs the inputs scaled by sqrt(1/(4N))
  std::vector<float> htaps, htw, hrot;
  // Multitaper at N >= 8192 takes the same form, one taper after the other on the frame held in
  // registers: the packed N-point form would need the whole N-point exchange buffer (139 KB at
  // N = 16384: one workgroup per CU), the real-input form half of it.  Each taper carries its weight:
  // sqrt(1 / (4N (1 + sig_j))).
  const bool h_periodogram = (cfg->mode == GLFER_MODE_FFT || lmp) && !p->nonlin && n >= 512 && !huge;
  const bool h_multitaper = cfg->mode == GLFER_MODE_MTM && n >= 8192 && !huge;
  if (h_periodogram || h_multitaper) {
    const int th = n / 32;
    const int nwin = h_multitaper ? p->ntapers : 1;
    const bool rect = h_periodogram && (p->cfg.window_type == GLFER_WIN_RECTANGULAR);
    htaps.resize((size_t)nwin * n);
    for (int j = 0; j < nwin; j++) {
      const double scale = std::sqrt(1.0 / (4.0 * n * (h_multitaper ? 1.0 + p->sig[j] : 1.0)));
      for (int m = 0; m < 16; m++)
        for (int t = 0; t < th; t++)
          for (int e = 0; e < 2; e++) {
            const int i = 2 * (t + th * m) + e;
            const double w = h_multitaper ? p->tapers[(size_t)j * n + i] : (rect ? 1.0 : (double)p->window[i]);
            htaps[(size_t)j * n + ((size_t)(m / 2) * th + t) * 4 + (size_t)(m & 1) * 2 + e] = (float)(w * scale);
          }
    }
    p->htapers = nwin;
    htw.resize((size_t)2 * glfer::make_twiddles16(logn - 1, nullptr) * th);
    glfer::make_twiddles16(logn - 1, htw.data());
    hrot.resize((size_t)2 * th);
    for (int t = 0; t < th; t++) {
      const double ang = 2.0 * 3.14159265358979323846 * t / n;
      hrot[2 * t] = (float)std::cos(ang);
      hrot[2 * t + 1] = (float)std::sin(ang);
    }
  }

  // --- the same real-input form with wavefront-private 1024-point sub-transforms (spectro16w.hip),
  // N = 2048 .. 16384: z index n = W*(t + 64 m) + w for wavefront w, lane t, register m
  std::vector<float> wtaps, wtw, wcomb;
  const bool w_form = (((cfg->mode == GLFER_MODE_FFT || lmp) && (!p->nonlin || huge)) || cfg->mode == GLFER_MODE_MTM) && n >= 2048;
  if (w_form) {
    const int M = n / 2, W = M / 1024, LF = 64 * W, IPL = LF <= 512 ? 512 / LF : 1;
    const bool mt = cfg->mode == GLFER_MODE_MTM;
    const int nwin = mt ? p->ntapers : 1;
    const bool rect = !mt && (p->cfg.window_type == GLFER_WIN_RECTANGULAR);
    wtaps.resize((size_t)nwin * n);
    for (int j = 0; j < nwin; j++) {
      const double scale = std::sqrt(1.0 / (4.0 * n * (mt ? 1.0 + p->sig[j] : 1.0)));
      for (int w = 0; w < W; w++)
        for (int m = 0; m < 16; m++)
          for (int t = 0; t < 64; t++)
            for (int e = 0; e < 2; e++) {
              const int i = 2 * (W * (t + 64 * m) + w) + e;
              const double v = mt ? p->tapers[(size_t)j * n + i] : (rect ? 1.0 : (double)p->window[i]);
              // (RA9MB / limiter, N = 32768 only: the plain window, the scale follows the limiter as post_scale)
              wtaps[((((size_t)j * W + w) * 8 + m / 2) * 64 + t) * 4 + (size_t)(m & 1) * 2 + e] = (float)(p->nonlin ? v : v * scale);
            }
    }
    p->wtapers = nwin;
    wtw.resize((size_t)2 * glfer::make_twiddles16(10, nullptr) * 64);
    glfer::make_twiddles16(10, wtw.data());
    wcomb.resize(n <= 32768 ? (size_t)2 * IPL * W * LF : 2);      // (N = 65536 computes its twiddles in the kernels)
    const double two_pi = 2.0 * 3.14159265358979323846;
    for (int i = 0; i < IPL && n <= 32768; i++)
      for (int ww = 0; ww < W; ww++)
        for (int u = 0; u < LF; u++) {
          const long long k1 = u + (long long)LF * i;
          const double ang = ww == 0 ? two_pi * (double)k1 / n : two_pi * (double)((ww * k1) % M) / M;
          wcomb[2 * ((size_t)(i * W + ww) * LF + u)] = (float)std::cos(ang);
          wcomb[2 * ((size_t)(i * W + ww) * LF + u) + 1] = (float)std::sin(ang);
        }
  }

  // --- odd taper count (spectro16x.hip): the last taper shares a transform with the next frame's;
  // |Y|^2 = |E|^2/4 there (no mirror-sum doubling), so its scale carries 1/4 instead of 1/2
  std::vector<float> xtaps;
  if (cfg->mode == GLFER_MODE_MTM && (p->ntapers & 1) && p->ntapers >= 3 && !small && !huge) {
    const int T = n / 16, j = p->ntapers - 1;
    const double scale = std::sqrt(1.0 / (4.0 * n * (1.0 + p->sig[j])));
    xtaps.resize((size_t)n);
    for (int i = 0; i < n; i++) {
      const int t = i % T, m = i / T;
      xtaps[((size_t)(m / 4) * T + t) * 4 + (size_t)(m & 3)] = (float)(p->tapers[(size_t)j * n + i] * scale);
    }
  }

  // --- the same, with half tables that stay in LDS (spectro16xl.hip): DPSS tapers are symmetric
  // (even order) or antisymmetric (odd order) about the frame centre, so samples n < N/2 suffice.
  // Built only if the tapers computed above have that symmetry to 1e-7 of their peak and the
  // tables leave room for two blocks per CU.
  std::vector<float> ltaps;
  if (!xtaps.empty() && n <= 4096) {
    const int T = n / 16, nfull = p->npairs - 1;
    bool sym = true;
    for (int j = 0; j < p->ntapers && sym; j++) {
      const double *v = &p->tapers[(size_t)j * n];
      double peak = 0.0, dev = 0.0;
      for (int i = 0; i < n / 2; i++) {
        peak = std::max(peak, std::fabs(v[i]));
        dev = std::max(dev, std::fabs(v[n - 1 - i] - ((j & 1) ? -v[i] : v[i])));
      }
      sym = dev <= 1e-7 * peak;
    }
    const size_t lds = ((size_t)(n + n / 16) * (T >= 256 ? 1 : 256 / T) + 16 * 17 + 8 + (size_t)nfull * 8 * T + 4 * T) * 8;
    if (sym && lds <= 80 * 1024) {
      ltaps.resize((size_t)nfull * 8 * T * 2 + (size_t)8 * T);
      for (int j = 0; j < p->ntapers; j++) {
        const bool last = j == p->ntapers - 1;
        const double scale = std::sqrt(1.0 / ((last ? 4.0 : 2.0) * n * (1.0 + p->sig[j])));
        for (int m = 0; m < 8; m++)
          for (int t = 0; t < T; t++) {
            const float val = (float)(p->tapers[(size_t)j * n + t + T * m] * scale);
            if (last) ltaps[(size_t)nfull * 8 * T * 2 + (size_t)m * T + t] = val;
            else ltaps[(((size_t)(j / 2) * 8 + m) * T + t) * 2 + (j & 1)] = val;
          }
      }
    }
  }

  // --- HP-ARMA tables: which lag each cell of the t x (p_e+1) matrix holds after the
  // reference's fill (hparma.c:89-102).  r_xx is matrix(0,t,0,p_e) (hparma.c:64): its rows
  // are contiguous (util.c:153-160), lags 0..t-1 are written into row 0 past its p_e+1
  // columns, and the Toeplitz loop then copies cells that it may already have rewritten.
  // Simulated here with lag labels instead of values.
  std::vector<uint16_t> lagmap;
  std::vector<float> unit;
  std::vector<int> rot_sched;                                      // [steps][8]: j | k << 8, or -1 (hparma.hip, round 4)
  if (cfg->mode == GLFER_MODE_HPARMA) {
    const int t = cfg->hparma_t, ncol = cfg->hparma_p_e + 1;
    // The Jacobi sweep of compute_svd (util.c:301-355) as a STATIC SCHEDULE of steps of up to eight rotations that share
    // no column.  Two rotations that share a column must keep the order of the reference's row-cyclic walk -- nothing
    // else orders them (a rotation touches its two columns only, its skip tests are its own) -- so the walk is a DAG in
    // which (j, k) waits for the last earlier rotation on column j and the last on column k; list scheduling by longest
    // remaining path fills steps of eight: 80 steps for 33 columns (66 would be full steps; anti-diagonal by anti-diagonal
    // it is 94).
    // GLFER_HPARMA_WIDTH=16: sixteen rotations per step, each over 4 lanes (A/B runs; 63 steps for 33 columns -- the critical path)
    const int width = [] { const char *e = getenv("GLFER_HPARMA_WIDTH"); return e && atoi(e) == 16 ? 16 : 8; }();
    p->rot_width = width;
    if (ncol >= 2 && ncol <= 64) {
      std::vector<std::pair<int, int>> rots;
      for (int j = 0; j < ncol - 1; j++)
        for (int k = j + 1; k < ncol; k++) rots.push_back({j, k});
      const int R = (int)rots.size();
      std::vector<std::vector<int>> succ(R);
      std::vector<int> indeg(R, 0), lastc(ncol, -1), lp(R, 1);
      for (int i = 0; i < R; i++) {
        const int cs[2] = {rots[i].first, rots[i].second};
        int seen = -1;
        for (int c : cs) {
          if (lastc[c] >= 0 && lastc[c] != seen) {
            succ[lastc[c]].push_back(i);
            indeg[i]++;
            seen = lastc[c];
          }
          lastc[c] = i;
        }
      }
      for (int i = R - 1; i >= 0; i--)
        for (int s : succ[i]) lp[i] = std::max(lp[i], 1 + lp[s]);
      std::vector<int> ready;
      for (int i = 0; i < R; i++)
        if (!indeg[i]) ready.push_back(i);
      int done = 0;
      while (done < R) {
        std::sort(ready.begin(), ready.end(), [&](int a, int b) { return lp[a] != lp[b] ? lp[a] > lp[b] : a < b; });
        const int take = std::min<int>(width, (int)ready.size());
        std::vector<int> cur(ready.begin(), ready.begin() + take);
        ready.erase(ready.begin(), ready.begin() + take);
        for (int g = 0; g < width; g++) rot_sched.push_back(g < take ? (rots[cur[g]].first | rots[cur[g]].second << 8) : -1);
        for (int i : cur) {
          done++;
          for (int s : succ[i])
            if (--indeg[s] == 0) ready.push_back(s);
        }
      }
    }
    std::vector<int> flat((size_t)(t + 1) * ncol, -1);
    for (int i = 0; i < t; i++) flat[i] = i;                       // r_xx[0][i] = r(i)
    for (int i = 1; i < t; i++)
      for (int j = 0; j < ncol; j++) flat[(size_t)i * ncol + j] = flat[std::abs(j - i)];
    lagmap.resize((size_t)t * ncol);
    for (size_t i = 0; i < lagmap.size(); i++) lagmap[i] = (uint16_t)(flat[i] < 0 ? 0 : flat[i]);
    unit.resize((size_t)2 * (n / 2 + 1));
    for (int k = 0; k <= n / 2; k++) {
      const double ang = -2.0 * 3.14159265358979323846 * k / n;
      unit[2 * k] = (float)std::cos(ang);
      unit[2 * k + 1] = (float)std::sin(ang);
    }
  }

  // --- device tables
  DeviceGuard guard(cfg->device);
  hipError_t e = guard.error();
  if (e == hipSuccess) e = hipMalloc((void **)&p->d_taps, taps.size() * sizeof(float));
  if (e == hipSuccess && cfg->mode == GLFER_MODE_FFT && cfg->window_type != GLFER_WIN_RECTANGULAR) {
    e = hipMalloc((void **)&p->d_window, (size_t)n * sizeof(float));            // prepare_audio's multiply, fft.c:139-146
    if (e == hipSuccess) e = hipMemcpy(p->d_window, p->window.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) e = hipMalloc((void **)&p->d_tw, tw.size() * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(p->d_taps, taps.data(), taps.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(p->d_tw, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess && !htaps.empty()) {
    e = hipMalloc((void **)&p->d_htaps, htaps.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_htw, htw.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_hrot, hrot.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->d_htaps, htaps.data(), htaps.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_htw, htw.data(), htw.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_hrot, hrot.data(), hrot.size() * sizeof(float), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && !wtaps.empty()) {
    e = hipMalloc((void **)&p->d_wtaps, wtaps.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_wtw, wtw.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_wcomb, wcomb.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->d_wtaps, wtaps.data(), wtaps.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_wtw, wtw.data(), wtw.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_wcomb, wcomb.data(), wcomb.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && huge) {
      // spectro_big.hip: W_M^(w k1), k1 = t + 64 m, is exp(-2 pi i w t / M) (one sincos per lane) times this table's entry
      const int M = n / 2, W = M / 1024;
      std::vector<float> bt((size_t)W * 16 * 2);
      for (int w = 0; w < W; w++)
        for (int m = 0; m < 16; m++) {
          const double ang = -2.0 * 3.14159265358979323846 * (double)(((long long)64 * w * m) % M) / M;
          bt[2 * ((size_t)w * 16 + m)] = (float)std::cos(ang);
          bt[2 * ((size_t)w * 16 + m) + 1] = (float)std::sin(ang);
        }
      e = hipMalloc((void **)&p->d_bigtw, bt.size() * sizeof(float));
      if (e == hipSuccess) e = hipMemcpy(p->d_bigtw, bt.data(), bt.size() * sizeof(float), hipMemcpyHostToDevice);
    }
  }
  if (e == hipSuccess && !xtaps.empty()) {
    e = hipMalloc((void **)&p->d_xtaps, xtaps.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->d_xtaps, xtaps.data(), xtaps.size() * sizeof(float), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && !ltaps.empty()) {
    e = hipMalloc((void **)&p->d_ltaps, ltaps.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->d_ltaps, ltaps.data(), ltaps.size() * sizeof(float), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && !lagmap.empty()) {
    e = hipMalloc((void **)&p->d_lagmap, lagmap.size() * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_unit, unit.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->d_lagmap, lagmap.data(), lagmap.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && !rot_sched.empty()) {
      e = hipMalloc((void **)&p->d_rot_sched, rot_sched.size() * sizeof(int));
      if (e == hipSuccess) e = hipMemcpy(p->d_rot_sched, rot_sched.data(), rot_sched.size() * sizeof(int), hipMemcpyHostToDevice);
      p->rot_steps = (int)(rot_sched.size() / p->rot_width);
    }
    if (e == hipSuccess) e = hipMemcpy(p->d_unit, unit.data(), unit.size() * sizeof(float), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    int rc = hip_fail(e, "plan_create");
    glfer_hip_plan_destroy(p);
    return rc;
  }
  *out = p;
  return GLFER_OK;
}

void glfer_hip_plan_destroy(glfer_hip_plan *p) {
  if (!p) return;
  DeviceGuard guard(p->cfg.device);
  if (p->d_taps) (void)hipFree(p->d_taps);
  if (p->d_window) (void)hipFree(p->d_window);
  if (p->d_ftaps_mu_first) (void)hipFree(p->d_ftaps_mu_first);
  if (p->d_ftaps2) (void)hipFree(p->d_ftaps2);
  if (p->d_U0) (void)hipFree(p->d_U0);
  if (p->d_tw) (void)hipFree(p->d_tw);
  if (p->d_htaps) (void)hipFree(p->d_htaps);
  if (p->d_htw) (void)hipFree(p->d_htw);
  if (p->d_hrot) (void)hipFree(p->d_hrot);
  if (p->d_xtaps) (void)hipFree(p->d_xtaps);
  if (p->d_wtaps) (void)hipFree(p->d_wtaps);
  if (p->d_wtw) (void)hipFree(p->d_wtw);
  if (p->d_wcomb) (void)hipFree(p->d_wcomb);
  if (p->d_bigtw) (void)hipFree(p->d_bigtw);
  if (p->d_ltaps) (void)hipFree(p->d_ltaps);
  if (p->d_lagmap) (void)hipFree(p->d_lagmap);
  if (p->d_rot_sched) (void)hipFree(p->d_rot_sched);
  if (p->d_unit) (void)hipFree(p->d_unit);
  glfer::ingest_ring_free(p->ring);
  for (hipStream_t a : p->aux)
    if (a) (void)hipStreamDestroy(a);
  for (hipEvent_t ev : p->aux_events) (void)hipEventDestroy(ev);
  delete p;
}

int glfer_hip_hop(const glfer_hip_plan *p) { return p ? p->hop : GLFER_E_ARG; }
int glfer_hip_bins(const glfer_hip_plan *p) { return p ? p->bins : GLFER_E_ARG; }
int glfer_hip_num_tapers(const glfer_hip_plan *p) { return p ? p->ntapers : GLFER_E_ARG; }
size_t glfer_hip_num_frames(const glfer_hip_plan *p, size_t nsamples) {
  return p ? nsamples / (size_t)p->hop : 0;
}

int glfer_hip_get_window(const glfer_hip_plan *p, float *w) {
  if (!p || !w) return GLFER_E_ARG;
  memcpy(w, p->window.data(), (size_t)p->n * sizeof(float));
  return GLFER_OK;
}

int glfer_hip_get_tapers(const glfer_hip_plan *p, double *tapers, double *sig) {
  if (!p || p->cfg.mode != GLFER_MODE_MTM) return GLFER_E_ARG;
  if (tapers) memcpy(tapers, p->tapers.data(), p->tapers.size() * sizeof(double));
  if (sig) memcpy(sig, p->sig.data(), p->sig.size() * sizeof(double));
  return GLFER_OK;
}

int glfer_hip_make_window(int window_type, int n, float *window) {
  if (!window || n < 2 || window_type < 0 || window_type > 7) return GLFER_E_ARG;
  glfer::make_window(window_type, n, window);
  return GLFER_OK;
}

int glfer_hip_make_dpss(int n, int kmax, double nw, double *tapers, double *sig) {
  if (!tapers || !sig || n < 2 || kmax < 0 || kmax > 31 || !(nw > 0.0)) return GLFER_E_ARG;
  return glfer::make_dpss(n, kmax, nw, tapers, sig) ? GLFER_OK : GLFER_E_NUMERIC;
}

extern "C" hipError_t glfer_launch_spectro_small(const SpectroParams *p, int n, const float *staps, hipStream_t st);
extern "C" hipError_t glfer_launch_spectro16w_n15(const SpectroParams *p, hipStream_t st);
extern "C" hipError_t glfer_launch_spectro_big(const SpectroParams *p, int n, hipStream_t st);

static hipError_t launch_packed(const SpectroParams &sp, int n, hipStream_t st) {
  if (n < 256) return glfer_launch_spectro_small(&sp, n, sp.taps, st);   // the same role below the 16-points-per-lane range
  switch (n) {
    case 256: return glfer_launch_spectro16_n8(&sp, st);
    case 512: return glfer_launch_spectro16_n9(&sp, st);
    case 1024: return glfer_launch_spectro16_n10(&sp, st);
    case 2048: return glfer_launch_spectro16_n11(&sp, st);
    case 4096: return glfer_launch_spectro16_n12(&sp, st);
    case 8192: return glfer_launch_spectro16_n13(&sp, st);
    case 16384: return glfer_launch_spectro16_n14(&sp, st);
  }
  return hipErrorInvalidValue;
}

static hipError_t launch_real_input(const SpectroParams &sp, int n, hipStream_t st) {
  switch (n) {
    case 512: return glfer_launch_spectro16h_n9(&sp, st);
    case 1024: return glfer_launch_spectro16h_n10(&sp, st);
    case 2048: return glfer_launch_spectro16h_n11(&sp, st);
    case 4096: return glfer_launch_spectro16h_n12(&sp, st);
    case 8192: return glfer_launch_spectro16h_n13(&sp, st);
    case 16384: return glfer_launch_spectro16h_n14(&sp, st);
  }
  return hipErrorInvalidValue;
}

// measured defaults: where spectro16w.hip is the faster form (profiles/r02_form_size_sweep.txt: the
// multitaper at N = 16384, every taper count; spectro16h.hip keeps the periodogram -- its coalesced
// sample loads beat the sub-transforms' strided ones when a frame is transformed only once -- and
// N = 8192)
#define GLFER_W_PERIODOGRAM(n) (false)
#define GLFER_W_MULTITAPER(n, tapers) ((n) >= 16384)

static int form_override();

static hipError_t launch_wave_private(const SpectroParams &sp, int n, hipStream_t st) {
  switch (n) {
    case 2048: return glfer_launch_spectro16w_n11(&sp, st);
    case 4096: return glfer_launch_spectro16w_n12(&sp, st);
    case 8192: return glfer_launch_spectro16w_n13(&sp, st);
    case 16384: return glfer_launch_spectro16w_n14(&sp, st);
    // N = 32768: the two-kernel form is the faster one (profiles/r02_big_blocks.txt); spectro16w.hip's single-kernel
    // form keeps the halfcomplex spectrum output and GLFER_FORM=w
    case 32768: return (sp.spec || form_override() == 'w') ? glfer_launch_spectro16w_n15(&sp, st) : glfer_launch_spectro_big(&sp, n, st);
    case 65536: return glfer_launch_spectro_big(&sp, n, st);
  }
  if (n >= 131072 && n <= (1 << 20)) return glfer_launch_spectro_big(&sp, n, st);   // two-level combine (round 3)
  return hipErrorInvalidValue;
}

// Which form takes a configuration is a measured choice (profiles/r02_*): GLFER_FORM=h|w|x in the
// environment overrides it for A/B runs on one box (h: spectro16h, w: spectro16w, x: the packed /
// shared-odd-taper forms).
static int form_override() {
  const char *e = getenv("GLFER_FORM");          // read per launch: tests and A/B runs switch it in-process
  return e && *e ? (int)*e : 0;
}

static hipError_t launch_shared_odd(const SpectroParams &sp, int n, hipStream_t st) {
  if (n == 4096) return glfer_launch_spectro16y_n12(&sp, st);   // two frames interleaved per wavefront
  if (sp.ltaps) {                      // taper half tables resident in LDS
    switch (n) {
      case 256: return glfer_launch_spectro16xl_n8(&sp, st);
      case 512: return glfer_launch_spectro16xl_n9(&sp, st);
      case 1024: return glfer_launch_spectro16xl_n10(&sp, st);
      case 2048: return glfer_launch_spectro16xl_n11(&sp, st);
    }
  }
  switch (n) {
    case 256: return glfer_launch_spectro16x_n8(&sp, st);
    case 512: return glfer_launch_spectro16x_n9(&sp, st);
    case 1024: return glfer_launch_spectro16x_n10(&sp, st);
  }
  return hipErrorInvalidValue;
}

// Kernel choice.  spectro16.hip (two tapers packed per N-point transform) does everything; two
// specialisations take the frames that lie wholly inside the stream when they apply:
//   * one taper, PSD only      -> spectro16h.hip, real-input N/2-point transform
//   * odd taper count >= 3     -> last taper shared by two frames: spectro16y.hip (N = 4096, the two
//                                 frames interleaved per wavefront), spectro16xl.hip (taper half
//                                 tables in LDS, when they fit), spectro16x.hip otherwise
// The first ceil(R/H) frames of a stream reach back before sample 0 (zero history, fft.c:103-108);
// they stay with spectro16.hip, which has the range-checked gather for that.
enum BodyRoute { ROUTE_PACKED, ROUTE_REAL_INPUT, ROUTE_SHARED_ODD, ROUTE_WAVE_PRIVATE };
static BodyRoute body_route(const SpectroParams &sp, int n) {
  // spectro16h.hip fetches y[2j], y[2j+1] with one load: integer samples must then sit on naturally
  // aligned pairs (even hop, so every frame starts on an even sample, and an aligned stream)
  const unsigned esz = sp.fmt == GLFER_FMT_F32 ? 4u : (sp.fmt == GLFER_FMT_S16 ? 2u : 1u);
  const bool pairs_aligned = sp.fmt == GLFER_FMT_F32 ||
                             ((sp.H & 1) == 0 && (reinterpret_cast<uintptr_t>(sp.stream) & (2u * esz - 1u)) == 0);
  const int force = form_override();
  bool real_input = sp.htaps && (sp.npairs == 1 || sp.htapers > 1) && n >= 512 && pairs_aligned;
  // built where it fits 3 waves/SIMD without spilling (N = 2048 and N >= 8192 do not: they stay packed)
  bool shared_odd = sp.xtaps && sp.npairs >= 2 && (sp.ltaps || n <= 1024 || n == 4096);
  // wavefront-private sub-transforms: the periodogram and the large-block multitaper by default
  // (GLFER_WAVE_PRIVATE_DEFAULT below), any multitaper at N >= 2048 on request
  bool wave_private = sp.wtaps && n >= 2048 && pairs_aligned &&
                      (force == 'w' || (force == 0 && (sp.npairs == 1 ? GLFER_W_PERIODOGRAM(n) : GLFER_W_MULTITAPER(n, sp.wtapers))));
  if (force == 'h') wave_private = false;
  if (force == 'x') wave_private = real_input = false;
  if (wave_private) real_input = true, shared_odd = false;
  if (sp.spec || sp.nonlin || !(real_input || shared_odd)) return ROUTE_PACKED;
  if (wave_private) return ROUTE_WAVE_PRIVATE;
  return real_input ? ROUTE_REAL_INPUT : ROUTE_SHARED_ODD;
}
// the forms that remove the hop means themselves (SpectroParams::mean_inkernel): the packed kernel,
// spectro16h.hip's periodogram form, spectro16y.hip, spectro16w.hip's multitaper form
static bool route_takes_mean(BodyRoute r, const SpectroParams &sp, int n) {
  if (n < 256 || n > 16384 || sp.spec || sp.nonlin || sp.history_mode) return false;
  if ((16 * sp.H) % n) return false;
  const int k16 = 16 * sp.H / n;
  switch (r) {
    case ROUTE_PACKED: return k16 == 4 || k16 == 8 || k16 == 16;
    case ROUTE_REAL_INPUT: return (sp.npairs == 1 && sp.htapers <= 1 && (k16 == 2 || k16 == 4 || k16 == 8 || k16 == 16)) ||
                                  (sp.htapers > 1 && n >= 8192 && k16 == 16);      // spectro16h.hip's multitaper form, hop = frame (its 50 / 75 % forms exist and lose to the copy: two wavefronts per SIMD against three)
    case ROUTE_SHARED_ODD: return (n == 4096 || n <= 1024) && (k16 == 4 || k16 == 8 || k16 == 16);   // y; x / xl while a frame sits in one wavefront
    case ROUTE_WAVE_PRIVATE: return sp.wtapers > 1 && (k16 == 8 || k16 == 16);     // spectro16w.hip's multitaper form (at 75 % overlap the copy is a quarter of a frame and wins: 7.50 against 7.36 M frames/s)
    default: return false;
  }
}

// ... and of those, the forms that subtract GIVEN means (round 3): spectro16h.hip's periodogram form, the packed
// kernel, spectro16y.hip
static bool route_takes_table(BodyRoute r, const SpectroParams &sp, int n) {
  switch (r) {
    // (round 4: every form that removes the means itself also takes them GIVEN -- spectro16w.hip's multitaper form (C4 with the
    // reference's order used to go through the corrected copy: 6.7 against 8.1 M frames/s), spectro16h.hip's multitaper form,
    // spectro16x / xl's frame loader)
    case ROUTE_PACKED: return true;
    case ROUTE_REAL_INPUT: return true;
    case ROUTE_SHARED_ODD: return true;
    case ROUTE_WAVE_PRIVATE: return true;
    default: return false;
  }
}

static hipError_t launch_by_n(const SpectroParams &sp, int n, hipStream_t st) {
  if (n > 16384) return launch_wave_private(sp, n, st);        // its general form takes every case itself
  const BodyRoute route = body_route(sp, n);
  const bool real_input = route == ROUTE_REAL_INPUT || route == ROUTE_WAVE_PRIVATE, wave_private = route == ROUTE_WAVE_PRIVATE;
  if (sp.mean_inkernel && !route_takes_mean(route, sp, n)) return hipErrorInvalidValue;
  if (route == ROUTE_PACKED) return launch_packed(sp, n, st);
  const long long first_inside = ((long long)sp.R + sp.H - 1) / sp.H;          // first frame f with f*H >= R
  if (sp.mean_inkernel && sp.frame0 < first_inside) return hipErrorInvalidValue;   // (the caller sends the head frames another way)
  // spectro16x.hip works on groups of G consecutive frames (frame f shares its last transform with
  // frame f + G/2).  Groups are aligned to GLOBAL frame indices and only whole groups go to it, so a
  // frame's result does not depend on how the stream was cut into launches, chunks or shards, as
  // long as the cuts fall on multiples of GLFER_FRAME_ALIGN (shard.py and the WAV reader see to it).
  const int lanes = n / 16;
  const long long G = real_input ? 1 : 2 * (lanes >= 256 ? 1 : 256 / lanes);
  const long long lo = sp.frame0, hi = sp.frame0 + sp.nframes;
  long long b0 = lo > first_inside ? lo : first_inside;
  b0 = (b0 + G - 1) / G * G;
  long long b1 = hi / G * G;
  if (sp.mean_inkernel && (b0 != lo || b1 != hi)) return hipErrorInvalidValue;   // (the caller cuts at the frame groups)
  if (b0 >= b1) return launch_packed(sp, n, st);
  auto sub = [&](long long from, long long to) {
    SpectroParams q = sp;
    q.frame0 = from;
    q.nframes = (int)(to - from);
    q.psd = sp.psd + (size_t)(from - lo) * (size_t)sp.pitch;
    q.mean_inkernel = 0;                           // (head and tail frames: never with mean_inkernel, see above)
    q.means = nullptr;
    return q;
  };
  if (b0 > lo) {
    const SpectroParams head = sub(lo, b0);
    hipError_t e = launch_packed(head, n, st);
    if (e != hipSuccess) return e;
  }
  if (hi > b1) {
    const SpectroParams tail = sub(b1, hi);
    hipError_t e = launch_packed(tail, n, st);
    if (e != hipSuccess) return e;
  }
  SpectroParams body = sub(b0, b1);
  body.mean_inkernel = sp.mean_inkernel;
  body.means = sp.means;
  if (wave_private) return launch_wave_private(body, n, st);
  return real_input ? launch_real_input(body, n, st) : launch_shared_odd(body, n, st);
}

}  // extern "C"

// Fills the kernel argument block from a plan.
static void fill_params(const glfer_hip_plan *p, SpectroParams &sp) {
  memset(&sp, 0, sizeof sp);
  sp.H = p->hop;
  sp.R = p->keep;
  sp.npairs = p->npairs;
  sp.history_mode = p->cfg.history_mode ? 1 : 0;
  sp.pitch = p->pitch;
  sp.fmt = p->cfg.sample_format;
  sp.nonlin = p->nonlin ? 1 : 0;
  sp.limiter = (p->cfg.enable_limiter == 1);
  sp.a = p->cfg.limiter_a;
  sp.post_scale = p->spec_unscale;
  sp.spec_unscale = p->spec_unscale;
  sp.taps = p->d_taps;
  sp.tw = p->d_tw;
  sp.htaps = p->d_htaps;
  sp.htapers = p->htapers;
  sp.htw = p->d_htw;
  sp.hrot = p->d_hrot;
  sp.xtaps = p->d_xtaps;
  sp.ltaps = p->d_ltaps;
  sp.wtaps = p->d_wtaps;
  sp.wtapers = p->wtapers;
  sp.wtw = p->d_wtw;
  sp.wcomb = p->d_wcomb;
  sp.bigtw = p->d_bigtw;
}

// cfg.sub_mean: any non-zero value but GLFER_SUBMEAN_FAST asks for the reference's rows, i.e. the hop means in
// the reference's own summation order (1 = GLFER_SUBMEAN_EXACT is what fft_init stores: sub_mean = opt.autoscale)
static bool reference_means(const glfer_hip_plan *p) { return p->cfg.sub_mean != 0 && p->cfg.sub_mean != GLFER_SUBMEAN_FAST; }

// K0 (fft.c:86-96) for the hops that frames [first, first+nframes) touch: the mean of each hop's
// NEW samples is removed before the hop enters the frame history, so every sample is corrected by
// the mean of the hop it arrived in.  ZERO_ALWAYS frames see only their own hop; otherwise a frame
// reaches back ceil(R/H) whole hops -- which is why a shard's or a chunk's halo is whole hops
// (glfer_hip.h, "Cutting a stream").  The corrected float copy lives in stream-ordered scratch
// (hipMallocAsync on `st`: nothing is shared between calls, the call stays asynchronous) and the
// kernels get its VIRTUAL base, so frame indices stay global and the frame groups of the
// shared-odd-taper kernels stay aligned to the stream, not to the launch.
//
// tail_fresh >= 0: the last hop is the file source's trailing partial block.  The caller has laid
// its fresh samples over a copy of the previous hop's RAW samples; what the reference's buffer
// holds there is the previous hop AFTER its mean removal (wav_fmt.c:102-119 over fft.c:93-95), so
// the last hop is rebuilt from the corrected previous hop before its own mean is taken.
static int submean_scratch(const glfer_hip_plan *p, SpectroParams &sp, size_t first, size_t nframes, hipStream_t st,
                           float **scratch_out, long tail_fresh = -1) {
  // ZERO_ALWAYS frames use only their own hop, and no kernel loads the history it would zero
  // (round 3: odd_taper.hpp::load_frame16, spectro16h / spectro16w's HIST gathers start their
  // descriptor at the frame's own hop), so the copy starts there too -- a piece cut for that mode
  // carries no history below its first hop (glfer_hip.h, "Cutting a stream", rule 2).
  const size_t hops_back = sp.history_mode ? 0 : (size_t)((p->keep + p->hop - 1) / p->hop);
  size_t hop_lo = (first > hops_back) ? first - hops_back : 0;
  const size_t last = first + nframes - 1;
  if (tail_fresh >= 0 && last > 0 && hop_lo > last - 1) hop_lo = last - 1;   // the stale part needs the hop before
  const size_t nhops = first + nframes - hop_lo;
  float *scratch = nullptr;
  HIP_TRY(glfer::scratch_malloc((void **)&scratch, nhops * (size_t)p->hop * sizeof(float), st));
  const size_t esz = sp.fmt == GLFER_FMT_F32 ? 4 : (sp.fmt == GLFER_FMT_S16 ? 2 : 1);
  const char *src = (const char *)sp.stream + hop_lo * (size_t)p->hop * esz;
  // GLFER_SUBMEAN_EXACT: the hop means first, accumulated sample after sample as fft.c:88-92 does
  // (submean_seq.hip: one more read of the stream), then the copy with those means
  const bool exact = reference_means(p);
  float *means = nullptr;
  hipError_t e = hipSuccess;
  if (exact) {
    e = glfer::scratch_malloc((void **)&means, nhops * sizeof(float), st);
    if (e == hipSuccess) e = glfer_launch_hop_means_seq(src, means, p->hop, (long long)nhops, sp.fmt, st);
  }
  if (e == hipSuccess) e = glfer_launch_submean(src, scratch, p->hop, (long long)nhops, sp.fmt, st, means);
  if (e == hipSuccess && tail_fresh >= 0)
    e = glfer_launch_submean_tail_ex(src + (nhops - 1) * (size_t)p->hop * esz, nhops > 1 ? scratch + (nhops - 2) * (size_t)p->hop : nullptr,
                                     scratch + (nhops - 1) * (size_t)p->hop, p->hop, (int)tail_fresh, exact ? 1 : 0, sp.fmt, st);
  if (means) glfer::scratch_free(means, st);
  if (e != hipSuccess) {
    glfer::scratch_free(scratch, st);
    return hip_fail(e, "glfer_launch_submean");
  }
  sp.stream = reinterpret_cast<const char *>(scratch) - hop_lo * (size_t)p->hop * sizeof(float);
  sp.fmt = GLFER_FMT_F32;
  *scratch_out = scratch;
  return GLFER_OK;
}

// Can spectro16h.hip take the hop means out itself for this plan and call?  The periodogram (one
// window, PSD only, no RA9MB/limiter), history from the stream, a hop of 2, 4, 8 or all 16 of a
// lane's 16 sample registers (overlap 87.5 / 75 / 50 / 0 %), samples it can fetch in pairs, no
// trailing partial block, and the form not forced elsewhere (GLFER_FORM).  GLFER_MEAN_PREPASS=1
// keeps the pre-pass (A/B runs, and the tests that compare the two).
static bool mean_inkernel_ok(const glfer_hip_plan *p, const SpectroParams &sp, const float *d_spec, long tail_fresh) {
  if (p->nonlin || d_spec || tail_fresh >= 0 || sp.history_mode) return false;
  if (p->cfg.mode != GLFER_MODE_FFT && p->cfg.mode != GLFER_MODE_LMP && p->cfg.mode != GLFER_MODE_MTM) return false;
  const char *e = getenv("GLFER_MEAN_PREPASS");
  if (e && *e == '1') return false;
  const BodyRoute r = body_route(sp, p->n);
  if (!route_takes_mean(r, sp, p->n)) return false;
  // GLFER_SUBMEAN_EXACT: the kernels sum a hop in another order than fft.c:88-92, so they are handed the means
  // (SpectroParams::means) -- the forms that take a table: the periodogram, the packed kernel, spectro16y
  if (reference_means(p) && !route_takes_table(r, sp, p->n)) return false;
  return true;
}

// The hop means in the reference's own order (fft.c:88-92, submean_seq.hip) for hops [hop_lo, hop_lo + nhops) of the raw
// stream, into means[0 .. nhops): the tiled kernel where the hop and the stream's alignment allow it -- 16 or 4 hops per
// wavefront, whichever still gives the pass a few thousand wavefronts -- else 64 hops per wavefront.
static hipError_t launch_reference_means(const glfer_hip_plan *p, const SpectroParams &sp, size_t hop_lo, size_t nhops, float *means,
                                         unsigned blocks, hipStream_t st) {
  const size_t esz = sp.fmt == GLFER_FMT_F32 ? 4 : (sp.fmt == GLFER_FMT_S16 ? 2 : 1);
  const char *src = (const char *)sp.stream + hop_lo * (size_t)p->hop * esz;
  const int forced = [] { const char *e = getenv("GLFER_MEANS_HPW"); return e && *e ? atoi(e) : 0; }();
  int hpw = forced;
  // (measured, profiles/r04_piecewise_means.txt: on a whole 2^30-sample stream the tiled form with 16 hops per wavefront and
  // 16-byte loads is the faster one at H = 1024 -- C2 285 against 267 M frames/s --, level at H = 512, behind at H = 4096)
  if (!hpw) hpw = nhops >= 262144 ? (p->hop <= 2048 ? 16 : 64) : (nhops >= 32768 || p->hop % 1024 != 0 ? 16 : 4);
  if (hpw == 4 && p->hop % 1024 != 0) hpw = 16;
  if (hpw == 16 && p->hop % 256 != 0) hpw = 64;
  if ((hpw == 16 || hpw == 4) && (reinterpret_cast<uintptr_t>(src) & (4 * esz - 1)) == 0)
    return glfer_launch_hop_means_tiled(src, means, p->hop, (long long)nhops, sp.fmt, hpw, blocks, st);
  return glfer_launch_hop_means_seq(src, means, p->hop, (long long)nhops, sp.fmt, st);
}

// Frames [b0, b1) of the body (they lie inside the stream and on the kernel's frame groups) with GIVEN hop means
// (cfg.sub_mean = GLFER_SUBMEAN_EXACT: the reference's rows).  Round 3 took the means of the whole range in one launch and
// then ran the estimator: the stream came from HBM twice (C1 772 against 1 108, C2 230 against 277, C3 64 against 75 M
// frames/s).  Round 4: PIECE BY PIECE -- the means of piece c+1 (side stream) run beside the estimator launch of piece c,
// and a piece is small enough (samples + rows of two pieces under the 256 MiB Infinity Cache) that the estimator's read of
// it is served on-die: the stream leaves HBM once.  Pieces end on multiples of GLFER_FRAME_ALIGN frames, so every row is
// the one-launch row bit for bit.  GLFER_EXACT_PIECE_MB (samples per piece; 0 = one piece), GLFER_EXACT_STREAMS (1: means and
// estimator in turn on the caller's stream; 2: means on a side stream; 3: estimator launches alternate between the
// caller's stream and a second side stream as well), GLFER_MEANS_BLOCKS (grid of the tiled means kernel beside an
// estimator launch) are the knobs tools/exact_mean_time.py sweeps.
static int launch_body_with_reference_means(glfer_hip_plan *p, const SpectroParams &bs, size_t b0, size_t b1, hipStream_t st) {
  const long piece_mb = [] { const char *e = getenv("GLFER_EXACT_PIECE_MB"); return e && *e ? atol(e) : 0L; }();   // (read per call: the sweep sets them between calls)
  const int nstreams_env = [] {                                                  // 1..3: the plan has two side streams (aux[2]) and the join below knows three
    const char *e = getenv("GLFER_EXACT_STREAMS");
    const int v = e && *e ? atoi(e) : 2;
    return v < 1 ? 1 : (v > 3 ? 3 : v);
  }();
  const unsigned means_blocks = [] { const char *e = getenv("GLFER_MEANS_BLOCKS"); return e && *e ? (unsigned)atol(e) : 0u; }();
  const size_t hops_back = (size_t)((p->keep + p->hop - 1) / p->hop);
  const size_t hop_lo = b0 - hops_back, nhops = b1 - hop_lo;                  // (b0 >= first_inside >= hops_back)
  const size_t esz = bs.fmt == GLFER_FMT_F32 ? 4 : (bs.fmt == GLFER_FMT_S16 ? 2 : 1);
  // frames per piece: a multiple of 64 frames (the frame groups of every form and GLFER_FRAME_ALIGN)
  size_t piece = b1 - b0;
  if (piece_mb > 0) {
    piece = ((size_t)piece_mb << 20) / ((size_t)p->hop * esz);
    piece = std::max<size_t>(piece / 64 * 64, 64);
  }
  const size_t npieces = (b1 - b0 + piece - 1) / piece;
  int nstreams = npieces > 1 ? nstreams_env : 1;
  float *means = nullptr;
  hipError_t e = glfer::scratch_malloc((void **)&means, nhops * sizeof(float), st);
  if (e != hipSuccess) return hip_fail(e, "scratch (hop means)");
  SpectroParams q = bs;
  q.means = means - hop_lo;                                                    // indexed by GLOBAL hop (= frame) index
  // THE FUSED LAUNCH (round 4, the periodogram's table form, N <= 8192; an EXPERIMENT, off by default): the hop means are
  // produced INSIDE the estimator's launch -- its first workgroups run the 64-hops-side-by-side chains of submean_seq.hip, the
  // others transform and wait for the chunks of means their frames need (spectro16h.hip produce_hop_means) -- so that the two
  // kinds of workgroup are co-resident whatever the streams' scheduling does and the means' read of the stream could ride in
  // the HBM bandwidth the periodogram kernel leaves idle.  Measured (profiles/r04_piecewise_means.txt): correct (rows
  // bit-identical), and NOT faster -- C1 753 M frames/s with 512 producer workgroups against 787 with the separate launch: the
  // fused launch moves its 12.9 GB at the same 4.6 TB/s the periodogram kernel reaches alone; the chip's rate for this 2 : 1
  // read : write mix, not idle time, is what the second read of the stream costs.
  // GLFER_MEANS_PRODUCERS: producer workgroups (a multiple of 8; 0 = the separate launch, the default).
  {
    const long producers = [] { const char *e = getenv("GLFER_MEANS_PRODUCERS"); return e && *e ? atol(e) : 0L; }();
    const bool periodogram_table = body_route(bs, p->n) == ROUTE_REAL_INPUT && bs.npairs == 1 && bs.htapers <= 1 && p->n <= 8192;
    // (short launches: the producers' lead over the first consumers is a bubble of nhops / producers' rate -- keep the two launches)
    const long min_frames = [] { const char *e = getenv("GLFER_FUSED_MIN_FRAMES"); return e && *e ? atol(e) : 65536L; }();   // (tests lower it)
    if (producers > 0 && periodogram_table && npieces == 1 && (long)(b1 - b0) >= min_frames) {
      // GLFER_FUSED_BLOCK_FRAMES > 0 (round 5): the lock-stepped form -- consumer workgroups of so many frames walked front by front,
      // a flag per 64-hop group, the producers GLFER_FUSED_LOOK hops (default 1024) beyond what their front's resident consumers span
      const long block_frames = [] { const char *e = getenv("GLFER_FUSED_BLOCK_FRAMES"); return e && *e ? atol(e) : 0L; }();
      const long look = [] { const char *e = getenv("GLFER_FUSED_LOOK"); return e && *e ? atol(e) : 1024L; }();
      const bool lockstep = block_frames > 0;
      const int chunk = lockstep ? 64 : 4096;
      const size_t nchunks = (nhops + chunk - 1) / chunk;
      unsigned *ready = nullptr;
      e = glfer::scratch_malloc((void **)&ready, (nchunks + 8) * sizeof(unsigned), st);
      if (e == hipSuccess) e = hipMemsetAsync(ready, 0, (nchunks + 8) * sizeof(unsigned), st);
      if (e == hipSuccess) {
        q.nprod = (int)(producers / 8 * 8);
        q.prod_chunk = chunk;
        q.means_out = means - hop_lo;
        q.means_ready = ready;
        q.prod_hop0 = (long long)hop_lo;
        q.prod_nhops = (long long)nhops;
        if (lockstep) {
          q.prod_front_frames = -1;                                            // (the launcher sizes the fronts from its grid)
          q.prod_block_frames = (int)block_frames;
          q.prod_look = (int)look;
          q.front_done = ready + nchunks;
        }
        e = launch_by_n(q, p->n, st);
      }
      if (ready) glfer::scratch_free(ready, st);
      glfer::scratch_free(means, st);
      return e == hipSuccess ? GLFER_OK : hip_fail(e, "estimator launch (hop means produced in the launch)");
    }
  }
  static std::mutex aux_mu;
  bool own_aux = false;
  if (nstreams > 1) {                                                          // the side streams and their events, once per plan
    std::lock_guard<std::mutex> lock(aux_mu);
    if (p->aux_busy) {
      nstreams = 1;                                                            // another call of this plan is on them: stay on the caller's stream
    } else {
      for (int i = 0; i < nstreams - 1 && e == hipSuccess; i++)
        if (!p->aux[i]) e = hipStreamCreateWithFlags(&p->aux[i], hipStreamNonBlocking);
      while (e == hipSuccess && p->aux_events.size() < 2 * npieces + 1) {
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) p->aux_events.push_back(ev);
      }
      if (e != hipSuccess) {
        (void)hipGetLastError();
        nstreams = 1;
        e = hipSuccess;
      } else {
        p->aux_busy = own_aux = true;
      }
    }
  }
  if (nstreams <= 1) {
    for (size_t c = 0; c < npieces && e == hipSuccess; c++) {
      const size_t f0 = b0 + c * piece, f1 = std::min(b1, f0 + piece);
      const size_t h0 = c == 0 ? hop_lo : f0;
      e = launch_reference_means(p, bs, h0, f1 - h0, means + (h0 - hop_lo), 0, st);
      q.frame0 = (long long)f0;
      q.nframes = (int)(f1 - f0);
      q.psd = bs.psd + (f0 - b0) * (size_t)p->pitch;
      if (e == hipSuccess) e = launch_by_n(q, p->n, st);
    }
    glfer::scratch_free(means, st);
    return e == hipSuccess ? GLFER_OK : hip_fail(e, "estimator launch (given hop means)");
  }
  // means(c) on aux[0]; estimator(c) on the caller's stream (or alternately aux[1]) after means(c); means(c + 2) not before
  // estimator(c) is done, so that at most two pieces are between their two reads at any time.  (An event may be recorded
  // again once every wait on its previous record has been ENQUEUED -- a wait captures the record it follows -- so the
  // plan's events serve call after call.)
  hipStream_t sm = p->aux[0];
  hipEvent_t *ev_m = p->aux_events.data(), *ev_e = ev_m + npieces, ev_fork = p->aux_events[2 * npieces];
  e = hipEventRecord(ev_fork, st);                                             // the samples (and the scratch) are ordered on the caller's stream
  if (e == hipSuccess) e = hipStreamWaitEvent(sm, ev_fork, 0);
  if (e == hipSuccess && nstreams > 2) e = hipStreamWaitEvent(p->aux[1], ev_fork, 0);
  size_t launched = 0;
  for (size_t c = 0; c < npieces && e == hipSuccess; c++) {
    const size_t f0 = b0 + c * piece, f1 = std::min(b1, f0 + piece);
    const size_t h0 = c == 0 ? hop_lo : f0;
    if (c >= 2) e = hipStreamWaitEvent(sm, ev_e[c - 2], 0);
    if (e == hipSuccess) e = launch_reference_means(p, bs, h0, f1 - h0, means + (h0 - hop_lo), c == 0 ? 0 : means_blocks, sm);
    if (e == hipSuccess) e = hipEventRecord(ev_m[c], sm);
    hipStream_t se = (nstreams > 2 && (c & 1)) ? p->aux[1] : st;
    if (e == hipSuccess) e = hipStreamWaitEvent(se, ev_m[c], 0);
    q.frame0 = (long long)f0;
    q.nframes = (int)(f1 - f0);
    q.psd = bs.psd + (f0 - b0) * (size_t)p->pitch;
    if (e == hipSuccess) e = launch_by_n(q, p->n, se);
    if (e == hipSuccess) e = hipEventRecord(ev_e[c], se);
    if (e == hipSuccess) launched = c + 1;
  }
  // join: the caller's stream goes on after everything the side streams were given
  if (nstreams > 2)
    for (size_t c = launched >= 2 ? launched - 2 : 0; c < launched; c++)
      if (c & 1) (void)hipStreamWaitEvent(st, ev_e[c], 0);
  if (e != hipSuccess) {                                                       // a launch failed midway: the side streams may hold work nothing waits for
    (void)hipStreamSynchronize(sm);
    if (nstreams > 2) (void)hipStreamSynchronize(p->aux[1]);
  }
  glfer::scratch_free(means, st);
  if (own_aux) {
    std::lock_guard<std::mutex> lock(aux_mu);
    p->aux_busy = false;
  }
  return e == hipSuccess ? GLFER_OK : hip_fail(e, "estimator launch (given hop means, piecewise)");
}

// Periodograms of frames [first, first + nframes) with the mean removal (fft.c:86-96) done inside the
// periodogram kernel: the frames that lie inside the stream read the RAW stream and no corrected
// copy is written for them; the first ceil(R/H) frames of a stream (zero history: the packed
// kernel) keep the copy, a few hops long.  sp: fill_params + the raw stream.
static int launch_mean_inkernel(glfer_hip_plan *p, const SpectroParams &sp, size_t first, size_t nframes, float *d_psd,
                                hipStream_t st) {
  const size_t first_inside = (size_t)((p->keep + p->hop - 1) / p->hop);
  // the shared-odd-taper kernels take whole, globally aligned groups of frames (launch_by_n): pairs at N = 4096
  const size_t lanes = (size_t)p->n / 16;
  const size_t G = body_route(sp, p->n) == ROUTE_SHARED_ODD ? 2 * (lanes >= 256 ? 1 : 256 / lanes) : 1;
  const size_t end = first + nframes;
  size_t b0 = std::max(first, first_inside);
  b0 = (b0 + G - 1) / G * G;
  size_t b1 = end / G * G;
  if (b0 >= b1) b0 = b1 = end;                                   // no body: everything through the copy
  int rc = GLFER_OK;
  // frames [from, to) through the corrected copy (the stream's first frames, a lone frame off the pair grid)
  auto by_copy = [&](size_t from, size_t to) {
    if (rc != GLFER_OK || from >= to) return;
    SpectroParams hs = sp;
    hs.frame0 = (long long)from;
    hs.nframes = (int)(to - from);
    hs.psd = d_psd + (from - first) * (size_t)p->pitch;
    float *scratch = nullptr;
    rc = submean_scratch(p, hs, from, to - from, st, &scratch);
    if (rc == GLFER_OK) {
      hipError_t e = launch_by_n(hs, p->n, st);
      if (e != hipSuccess) rc = hip_fail(e, "estimator launch (frames through the corrected copy)");
    }
    if (scratch) glfer::scratch_free(scratch, st);
  };
  by_copy(first, std::min(b0, end));
  if (rc == GLFER_OK && b1 > b0) {
    SpectroParams bs = sp;
    bs.frame0 = (long long)b0;
    bs.nframes = (int)(b1 - b0);
    bs.psd = d_psd + (b0 - first) * (size_t)p->pitch;
    bs.spec = nullptr;
    bs.mean_inkernel = 1;
    if (!reference_means(p)) {
      hipError_t e = launch_by_n(bs, p->n, st);
      if (e != hipSuccess) rc = hip_fail(e, "estimator launch (mean removal in the kernel)");
    } else {
      rc = launch_body_with_reference_means(p, bs, b0, b1, st);
    }
  }
  by_copy(std::max(b1, std::min(b0, end)), end);
  return rc;
}

int glfer_run_device(glfer_hip_plan *p, const void *d_stream, size_t nsamples, size_t first, size_t nframes,
                     float *d_psd, float *d_spec, hipStream_t st, long tail_fresh) {
  if (!p || !d_stream || (!d_psd && nframes)) return GLFER_E_ARG;
  if (nframes == 0) return GLFER_OK;
  if ((first + nframes) > nsamples / (size_t)p->hop) return GLFER_E_ARG;   // frame past the stream
  if (nframes > 0x7fffffffu) return GLFER_E_ARG;
  if (d_spec && p->cfg.mode != GLFER_MODE_FFT) return GLFER_E_ARG;
  DeviceGuard guard(p->cfg.device);
  HIP_TRY(guard.error());

  SpectroParams sp;
  fill_params(p, sp);
  sp.stream = d_stream;
  sp.frame0 = (long long)first;
  sp.nframes = (int)nframes;
  sp.psd = d_psd;
  sp.spec = d_spec;

  float *scratch = nullptr, *rows = nullptr;
  int rc = GLFER_OK;
  if (tail_fresh >= 0 && (tail_fresh >= (long)p->hop || p->cfg.mode == GLFER_MODE_LMP)) return GLFER_E_ARG;
  const bool inkernel = p->cfg.sub_mean && mean_inkernel_ok(p, sp, d_spec, tail_fresh);
  if (inkernel && p->cfg.mode != GLFER_MODE_LMP) return launch_mean_inkernel(p, sp, first, nframes, d_psd, st);
  if (p->cfg.sub_mean && !inkernel) rc = submean_scratch(p, sp, first, nframes, st, &scratch, tail_fresh);
  if (rc == GLFER_OK && p->cfg.mode == GLFER_MODE_LMP) {
    // lmp.c:101-181: periodograms of the frames the ring holds when frame first+nframes-1 is done
    // (lmp_av - 1 frames before `first`, recomputed rather than carried), then the statistic
    const size_t back = std::min<size_t>((size_t)p->lmp_av - 1, first);
    const size_t nrows = nframes + back;
    hipError_t e = glfer::scratch_malloc((void **)&rows, nrows * (size_t)p->bins * sizeof(float), st);
    if (e != hipSuccess) rc = hip_fail(e, "hipMallocAsync(lmp rows)");
    if (rc == GLFER_OK && inkernel) {
      rc = launch_mean_inkernel(p, sp, first - back, nrows, rows, st);
      if (rc == GLFER_OK) {
        e = glfer_launch_lmp(rows, (long long)(first - back), (long long)first, nframes, p->bins, p->lmp_av, d_psd, st);
        if (e != hipSuccess) rc = hip_fail(e, "lmp launch");
      }
      glfer::scratch_free(rows, st);
      return rc;
    }
    if (rc == GLFER_OK && back && p->cfg.sub_mean) {
      // the extra frames reach further back than the hops corrected above
      glfer::scratch_free(scratch, st);
      scratch = nullptr;
      sp.stream = d_stream;
      sp.fmt = p->cfg.sample_format;
      rc = submean_scratch(p, sp, first - back, nrows, st, &scratch);
    }
    if (rc == GLFER_OK) {
      sp.frame0 = (long long)(first - back);
      sp.nframes = (int)nrows;
      sp.psd = rows;
      e = launch_by_n(sp, p->n, st);
      if (e == hipSuccess)
        e = glfer_launch_lmp(rows, (long long)(first - back), (long long)first, nframes, p->bins, p->lmp_av, d_psd, st);
      if (e != hipSuccess) rc = hip_fail(e, "lmp launch");
    }
  } else if (rc == GLFER_OK) {
    hipError_t e = p->cfg.mode == GLFER_MODE_HPARMA
                       ? glfer_launch_hparma(&sp, p->n, p->cfg.hparma_t, p->cfg.hparma_p_e + 1, p->d_rot_sched, p->rot_steps, p->rot_width, p->d_lagmap, p->d_unit, st)
                       : launch_by_n(sp, p->n, st);
    if (e != hipSuccess) rc = hip_fail(e, "estimator launch");
  }
  if (rows) glfer::scratch_free(rows, st);
  if (scratch) glfer::scratch_free(scratch, st);
  return rc;
}

extern "C" {

int glfer_hip_spectrogram_device(glfer_hip_plan *p, const void *d_stream, size_t nsamples, size_t first,
                                 size_t nframes, float *d_psd, void *hip_stream) {
  return glfer_run_device(p, d_stream, nsamples, first, nframes, d_psd, nullptr, (hipStream_t)hip_stream);
}

int glfer_hip_spectrum_device(glfer_hip_plan *p, const void *d_stream, size_t nsamples, size_t first,
                              size_t nframes, float *d_psd, float *d_spec, void *hip_stream) {
  if (!d_spec) return GLFER_E_ARG;
  return glfer_run_device(p, d_stream, nsamples, first, nframes, d_psd, d_spec, (hipStream_t)hip_stream);
}

// prepare_audio (fft.c:66-165) for a batch of frames: what it leaves in params->inbuf_fft
// (lmp.c:101-120 and g_scope.c:194-197 read it).  d_frames: [nframes][N] floats.
int glfer_hip_prepare_device(glfer_hip_plan *p, const void *d_stream, size_t nsamples, size_t first,
                             size_t nframes, float *d_frames, void *hip_stream) {
  if (!p || !d_stream || (!d_frames && nframes)) return GLFER_E_ARG;
  if (nframes == 0) return GLFER_OK;
  if ((first + nframes) > nsamples / (size_t)p->hop || nframes > 0x7fffffffu) return GLFER_E_ARG;
  hipStream_t st = (hipStream_t)hip_stream;
  DeviceGuard guard(p->cfg.device);
  HIP_TRY(guard.error());
  SpectroParams sp;
  fill_params(p, sp);
  sp.stream = d_stream;
  sp.frame0 = (long long)first;
  sp.nframes = (int)nframes;
  float *scratch = nullptr;
  int rc = GLFER_OK;
  if (p->cfg.sub_mean) rc = submean_scratch(p, sp, first, nframes, st, &scratch);
  if (rc == GLFER_OK) {
    // MTM / HP-ARMA / LMP run prepare_audio with a rectangular window (source.c:344,369,395); a and
    // the limiter still act on inbuf_fft there, but those estimators overwrite it, so the shims ask
    // for this only in FFT mode
    hipError_t e = glfer_launch_prepare(&sp, p->n, p->d_window, d_frames, st);
    if (e != hipSuccess) rc = hip_fail(e, "glfer_launch_prepare");
  }
  if (scratch) glfer::scratch_free(scratch, st);
  return rc;
}

// The harmonic F-test of mtm_do (mtm.c:165-174, 203-233) as an optional output of the multitaper
// path.  The tapered spectra y_j(f) and mu(f) come from the estimator's own spectrum output (one
// single-taper launch of spectro16_kernel per taper and one for hn), the statistic from a per-bin
// epilogue (stats_kernels.hip).  Frames are processed in groups that keep the spectra scratch
// under ~256 MiB.
int glfer_hip_mtm_ftest_device(glfer_hip_plan *p, const void *d_stream, size_t nsamples, size_t first,
                               size_t nframes, float *d_ftest, int mu_live, void *hip_stream) {
  if (!p || !d_stream || (!d_ftest && nframes) || p->cfg.mode != GLFER_MODE_MTM) return GLFER_E_ARG;
  if (p->n > 16384) return GLFER_E_ARG;              // needs the packed form's spectrum output
  if (nframes == 0) return GLFER_OK;
  if ((first + nframes) > nsamples / (size_t)p->hop || nframes > 0x7fffffffu) return GLFER_E_ARG;
  hipStream_t st = (hipStream_t)hip_stream;
  DeviceGuard guard(p->cfg.device);
  HIP_TRY(guard.error());
  const int n = p->n, T = p->ntapers;
  if (!p->d_ftaps) {                     // tables of mtm.c:76-83, 124-136, once per plan
    p->U0.resize(T);
    p->hn.resize(n);
    glfer::make_ftest_tables(n, T - 1, p->tapers.data(), p->U0.data(), p->hn.data(), &p->sum_U0_sqr);
    // [hn][taper 0..T-1][hn]: the one-launch form starts with hn (mu first), the spectrum-by-spectrum form ends with it
    std::vector<float> taps((size_t)(T + 2) * 2 * n, 0.0f);
    for (int j = -1; j <= T; j++)
      for (int i = 0; i < n; i++)
        taps[(size_t)(j + 1) * 2 * n + tap_slot(n, 0, i, 0)] = (j >= 0 && j < T) ? (float)p->tapers[(size_t)j * n + i] : p->hn[i];
    float *d = nullptr;
    double *du = nullptr;
    HIP_TRY(hipMalloc((void **)&d, taps.size() * sizeof(float)));
    hipError_t e = hipMalloc((void **)&du, (size_t)T * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(d, taps.data(), taps.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(du, p->U0.data(), (size_t)T * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(d);
      (void)hipFree(du);
      return hip_fail(e, "ftest tables");
    }
    p->d_ftaps_mu_first = d;
    p->d_ftaps = d + (size_t)2 * n;
    p->d_U0 = du;
    // the paired form's tables: two real sequences per N-point transform (re / im), each halved (X_a = (Z[k] + conj Z[N-k]) / 2)
    const int r_mu = (T + 2) / 2, r_nomu = (T + 1) / 2;
    std::vector<float> pt((size_t)(r_mu + r_nomu) * 2 * n, 0.0f);
    auto seq = [&](int s_, int i, bool with_mu) -> float {          // sequence s_ of the call: hn first when mu is live, then the tapers
      const int j = with_mu ? s_ - 1 : s_;
      if (j >= T) return 0.0f;
      return 0.5f * (j < 0 ? p->hn[i] : (float)p->tapers[(size_t)j * n + i]);
    };
    for (int r = 0; r < r_mu + r_nomu; r++) {
      const bool with_mu = r < r_mu;
      const int rr = with_mu ? r : r - r_mu;
      for (int i = 0; i < n; i++) {
        pt[(size_t)r * 2 * n + tap_slot(n, 0, i, 0)] = seq(2 * rr, i, with_mu);
        pt[(size_t)r * 2 * n + tap_slot(n, 0, i, 1)] = seq(2 * rr + 1, i, with_mu);
      }
    }
    float *d2 = nullptr;
    hipError_t e2 = hipMalloc((void **)&d2, pt.size() * sizeof(float));
    if (e2 == hipSuccess) e2 = hipMemcpy(d2, pt.data(), pt.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e2 != hipSuccess) {
      (void)hipFree(d2);
      return hip_fail(e2, "ftest tables (paired)");
    }
    p->d_ftaps2 = d2;
    p->d_ftaps2_nomu = d2 + (size_t)r_mu * 2 * n;
  }
  SpectroParams sp;
  fill_params(p, sp);
  sp.pitch = p->bins;                    // (the statistic and this entry's scratch rows are dense)
  sp.stream = d_stream;
  sp.npairs = 1;
  sp.nonlin = 0;
  sp.post_scale = sp.spec_unscale = 1.0f;
  sp.htaps = nullptr;                    // the packed kernel: it is the one with the spectrum output
  sp.xtaps = sp.ltaps = nullptr;
  float *scratch = nullptr;
  int rc = GLFER_OK;
  if (p->cfg.sub_mean) {
    sp.frame0 = (long long)first;
    rc = submean_scratch(p, sp, first, nframes, st, &scratch);
  }
  if (rc == GLFER_OK && n >= 256) {
    // one launch: every round of spectro16_kernel's FT form transforms the frame under one taper and
    // keeps what the statistic needs in registers (no spectrum goes through HBM)
    SpectroParams q = sp;
    q.frame0 = (long long)first;
    q.nframes = (int)nframes;
    q.psd = nullptr;
    q.spec = nullptr;
    q.ftest = d_ftest;
    q.ft_U0 = p->d_U0;
    q.ft_sum_U0_sqr = p->sum_U0_sqr;
    q.ft_mu_live = mu_live ? 1 : 0;
    q.npairs = mu_live ? T + 1 : T;                    // rounds: hn first (mu), then the tapers
    q.taps = mu_live ? p->d_ftaps_mu_first : p->d_ftaps;
    // round 5: two sequences per transform, separated through the mirror bins (GLFER_FTEST_PAIRED=0: one per transform, for A/B runs and tests)
    // Measured (gpurun_out/r5/ftest_rate.txt): N = 4096, 5 tapers 39.2 against 33.5 M frames/s; N = 1024, 8 tapers 94.3 against 105.9 -- the
    // mirror exchange (two barriers, 16 LDS writes, 9 reads a round) costs a short transform more than it saves: paired from N = 2048.
    const char *pe = getenv("GLFER_FTEST_PAIRED");
    if (pe && *pe ? *pe != '0' : n >= 2048) {
      q.ft_nseq = mu_live ? T + 1 : T;
      q.npairs = (q.ft_nseq + 1) / 2;
      q.taps = mu_live ? p->d_ftaps2 : p->d_ftaps2_nomu;
    }
    hipError_t e = launch_packed(q, n, st);
    if (e != hipSuccess) rc = hip_fail(e, "ftest launch");
    if (scratch) glfer::scratch_free(scratch, st);
    return rc;
  }
  size_t group = ((size_t)256 << 20) / ((size_t)(T + 1) * n * sizeof(float));
  group = std::max<size_t>(1, std::min<size_t>(group, 32768));
  float *spec = nullptr, *dummy = nullptr;
  if (rc == GLFER_OK) {
    const size_t g = std::min(group, nframes);
    hipError_t e = glfer::scratch_malloc((void **)&spec, (size_t)(T + 1) * g * n * sizeof(float), st);
    if (e == hipSuccess) e = glfer::scratch_malloc((void **)&dummy, g * (size_t)p->bins * sizeof(float), st);
    if (e != hipSuccess) rc = hip_fail(e, "hipMallocAsync(ftest spectra)");
  }
  for (size_t done = 0; rc == GLFER_OK && done < nframes; done += group) {
    const size_t g = std::min(group, nframes - done);
    hipError_t e = hipSuccess;
    for (int j = 0; j <= T && e == hipSuccess; j++) {
      if (j == T && !mu_live) break;     // mu is never written in the reference build (mtm.c:173)
      SpectroParams q = sp;
      q.frame0 = (long long)(first + done);
      q.nframes = (int)g;
      q.taps = p->d_ftaps + (size_t)j * 2 * n;
      q.psd = dummy;
      q.spec = spec + (size_t)j * g * n;
      e = launch_packed(q, n, st);
    }
    if (e == hipSuccess)
      e = glfer_launch_ftest(spec, g, n, T, p->d_U0, p->sum_U0_sqr, mu_live ? 1 : 0, d_ftest + done * (size_t)p->bins, st);
    if (e != hipSuccess) rc = hip_fail(e, "ftest launch");
  }
  if (spec) glfer::scratch_free(spec, st);
  if (dummy) glfer::scratch_free(dummy, st);
  if (scratch) glfer::scratch_free(scratch, st);
  return rc;
}

int glfer_hip_submean_device(const void *d_in, float *d_out, int hop, size_t nhops, int sample_format,
                             void *hip_stream) {
  if (!d_in || !d_out || hop < 1 || sample_format < 0 || sample_format > 2) return GLFER_E_ARG;
  DeviceGuard guard(data_device(d_out));
  HIP_TRY(guard.error());
  HIP_TRY(glfer_launch_submean(d_in, d_out, hop, (long long)nhops, sample_format, (hipStream_t)hip_stream, nullptr));
  return GLFER_OK;
}

// The same with every hop summed in the reference's own order (fft.c:88-92; submean_seq.hip).
int glfer_hip_submean_exact_device(const void *d_in, float *d_out, int hop, size_t nhops, int sample_format,
                                   void *hip_stream) {
  if (!d_in || !d_out || hop < 1 || sample_format < 0 || sample_format > 2) return GLFER_E_ARG;
  if (nhops == 0) return GLFER_OK;
  hipStream_t st = (hipStream_t)hip_stream;
  DeviceGuard guard(data_device(d_out));
  HIP_TRY(guard.error());
  float *means = nullptr;
  HIP_TRY(glfer::scratch_malloc((void **)&means, nhops * sizeof(float), st));
  hipError_t e = glfer_launch_hop_means_seq(d_in, means, hop, (long long)nhops, sample_format, st);
  if (e == hipSuccess) e = glfer_launch_submean(d_in, d_out, hop, (long long)nhops, sample_format, st, means);
  glfer::scratch_free(means, st);
  return e == hipSuccess ? GLFER_OK : hip_fail(e, "glfer_hip_submean_exact_device");
}

int glfer_hip_floor_device_pitched(const float *d_psd, size_t nframes, int bins, int pitch, float *d_stats, void *hip_stream) {
  if (!d_psd || !d_stats || bins < 1 || bins > 32769 || pitch < bins) return GLFER_E_ARG;
  const int m = bins - (int)(bins * 0.95);                       // fft.c:271: i = N2*0.95 .. N2-1
  DeviceGuard guard(data_device(d_psd));
  HIP_TRY(guard.error());
  HIP_TRY(glfer_launch_floor(d_psd, nframes, bins, pitch, m, d_stats, (hipStream_t)hip_stream));
  return GLFER_OK;
}

int glfer_hip_floor_device(const float *d_psd, size_t nframes, int bins, float *d_stats, void *hip_stream) {
  return glfer_hip_floor_device_pitched(d_psd, nframes, bins, bins, d_stats, hip_stream);
}

int glfer_hip_avg_device(int avg_mode, const float *d_psd, size_t nframes, int bins, int n_out, int depth,
                         int minbin, int maxbin, int max0, double *d_avg, double *d_ret, void *hip_stream) {
  if (!d_psd || !d_avg || !d_ret) return GLFER_E_ARG;
  if (avg_mode < GLFER_AVG_SUMAVG || avg_mode > GLFER_AVG_SUMEXTREME) return GLFER_E_ARG;
  if (depth < 1 || minbin < 0 || maxbin <= minbin || maxbin > bins || maxbin > n_out || n_out < 1)
    return GLFER_E_ARG;
  DeviceGuard guard(data_device(d_psd));
  HIP_TRY(guard.error());
  HIP_TRY(glfer_launch_avg(avg_mode, d_psd, nframes, bins, n_out, depth, minbin, maxbin, max0 ? 1 : 0, d_avg,
                           d_ret, (hipStream_t)hip_stream));
  return GLFER_OK;
}

// fft_do + fft_psd + update_avg_* for a batch of frames in ONE call (source.c:141-158 followed by g_main.c:1153-1183).  Where the
// periodogram's real-input kernel applies -- FFT mode, N = 512 .. 4096, no RA9MB / limiter, no mean removal, history from the
// stream, dense rows -- and the average is the plain one over a window of at most four frames (glfer.c:295-296: the default
// depth is 4), the average is taken INSIDE the estimator launch on the PSD values in registers (spectro16h.hip AVG): no PSD row
// goes to memory unless d_psd asks for it.  The frames up to the first one with a full window behind it inside the stream (at
// most ceil((N-H)/H) + depth - 1 of them) and every other configuration take the two launches (rows, then avg_fused_kernel).
int glfer_hip_spectrogram_avg_device(glfer_hip_plan *p, const void *d_stream, size_t nsamples, size_t first, size_t nframes,
                                     int avg_mode, int depth, int minbin, int maxbin, int max0, int n_out, float *d_psd,
                                     double *d_avg, double *d_ret, void *hip_stream) {
  if (!p || !d_stream || (!d_avg && nframes)) return GLFER_E_ARG;
  if (p->cfg.mode == GLFER_MODE_HPARMA || p->pitch != p->bins) return GLFER_E_ARG;       // (HP-ARMA rows are not averaged by the reference's callers; rows dense)
  if (avg_mode < GLFER_AVG_SUMAVG || avg_mode > GLFER_AVG_SUMEXTREME) return GLFER_E_ARG;
  if (depth < 1 || minbin < 0 || maxbin <= minbin || maxbin > p->bins || n_out < p->bins) return GLFER_E_ARG;
  if (nframes == 0) return GLFER_OK;
  if ((first + nframes) > nsamples / (size_t)p->hop || nframes > 0x7fffffffu) return GLFER_E_ARG;
  hipStream_t st = (hipStream_t)hip_stream;
  DeviceGuard guard(p->cfg.device);
  HIP_TRY(guard.error());
  const size_t bins = (size_t)p->bins;
  int rc = GLFER_OK;
  // frames [from, to): the rows (to d_psd, or scratch), then update_avg over them with the state empty at `from`
  auto two_launches = [&](size_t from, size_t to) {
    if (rc != GLFER_OK || from >= to) return;
    const size_t nf = to - from;
    float *rows = d_psd ? d_psd + (from - first) * bins : nullptr;
    double *ret = d_ret ? d_ret + (from - first) * 4 : nullptr;
    hipError_t e = hipSuccess;
    if (!d_psd) e = glfer::scratch_malloc((void **)&rows, nf * bins * sizeof(float), st);
    if (e == hipSuccess && !d_ret) e = glfer::scratch_malloc((void **)&ret, nf * 4 * sizeof(double), st);
    if (e != hipSuccess) rc = hip_fail(e, "scratch (rows to average)");
    if (rc == GLFER_OK) rc = glfer_run_device(p, d_stream, nsamples, from, nf, rows, nullptr, st);
    if (rc == GLFER_OK) {
      e = glfer_launch_avg(avg_mode, rows, nf, (int)bins, n_out, depth, minbin, maxbin, max0 ? 1 : 0, d_avg + (from - first) * (size_t)n_out, ret, st);
      if (e != hipSuccess) rc = hip_fail(e, "update_avg launch");
    }
    if (!d_psd && rows) glfer::scratch_free(rows, st);
    if (!d_ret && ret) glfer::scratch_free(ret, st);
  };
  SpectroParams sp;
  fill_params(p, sp);
  sp.stream = d_stream;
  sp.frame0 = (long long)first;
  sp.nframes = (int)nframes;
  static const bool fused_off = [] { const char *e = getenv("GLFER_AVG_FUSED"); return e && *e == '0'; }();   // (A/B runs and the tests that compare the two)
  // mean removal: off, or the reference's own (cfg.sub_mean = 1: the means are taken first, in its summation order, and given to the kernel)
  // where the hop is 2, 4, 8 or 16 sixteenths of the block; the in-kernel sums (GLFER_SUBMEAN_FAST) take the two launches
  const bool with_means = p->cfg.sub_mean != 0;
  const bool fused = !fused_off && avg_mode == GLFER_AVG_PLAIN && depth <= 4 && p->cfg.mode == GLFER_MODE_FFT && !p->nonlin &&
                     !p->cfg.history_mode && p->n >= 512 && p->n <= 4096 && n_out <= 2 * p->n && body_route(sp, p->n) == ROUTE_REAL_INPUT &&
                     (!with_means || (reference_means(p) && route_takes_mean(ROUTE_REAL_INPUT, sp, p->n) && !getenv("GLFER_MEAN_PREPASS")));
  // the first frame every one of whose depth-1 predecessors lies inside the stream AND inside this call's averaging state
  const size_t first_inside = (size_t)((p->keep + p->hop - 1) / p->hop), end = first + nframes;
  const size_t b0 = std::max(first, first_inside) + (size_t)(depth - 1);
  if (!fused || b0 + 256 > end) {                         // (short calls: the lead frames of every slot would outweigh the rest)
    two_launches(first, end);
    return rc;
  }
  // The head's state starts empty at `first`, like the whole call's.  Its rows are also what the body's first slots would need
  // in front of them -- they recompute them instead (frames >= b0 - (depth-1) >= first_inside are all computable).
  two_launches(first, b0);
  const size_t piece = (size_t)1 << 24;                     // frames per launch: keeps a workgroup's rows under the 4 GiB of a buffer descriptor
  // (the kernel always forms the return values -- one code path, no branch per bin: without d_ret they land in scratch)
  double *ret_scratch = nullptr;
  if (!d_ret) {
    hipError_t e = glfer::scratch_malloc((void **)&ret_scratch, std::min(piece, end - b0) * 4 * sizeof(double), st);
    if (e != hipSuccess) return hip_fail(e, "scratch (return values)");
  }
  // the hop means of everything the body touches: its frames, the depth-1 frames a slot recomputes in front of them, their history
  float *means = nullptr;
  const size_t hops_back = (size_t)((p->keep + p->hop - 1) / p->hop), hop_lo = b0 - (size_t)(depth - 1) - hops_back;
  if (with_means) {
    hipError_t e = glfer::scratch_malloc((void **)&means, (end - hop_lo) * sizeof(float), st);
    if (e == hipSuccess) e = launch_reference_means(p, sp, hop_lo, end - hop_lo, means, 0, st);
    if (e != hipSuccess) {
      if (means) glfer::scratch_free(means, st);
      if (ret_scratch) glfer::scratch_free(ret_scratch, st);
      return hip_fail(e, "hop means (average inside the kernel)");
    }
  }
  for (size_t f0 = b0; rc == GLFER_OK && f0 < end; f0 += piece) {
    const size_t nf = std::min(piece, end - f0);
    SpectroParams q = sp;
    if (with_means) {
      q.mean_inkernel = 1;
      q.means = means - hop_lo;                              // indexed by GLOBAL hop (= frame) index
    }
    q.frame0 = (long long)f0;
    q.nframes = (int)nf;
    q.psd = d_psd ? d_psd + (f0 - first) * bins : nullptr;
    q.avg = d_avg + (f0 - first) * (size_t)n_out;
    q.avg_ret = d_ret ? d_ret + (f0 - first) * 4 : ret_scratch;
    q.avg_depth = depth;
    q.avg_minbin = minbin;
    q.avg_maxbin = maxbin;
    q.avg_nout = n_out;
    hipError_t e = launch_real_input(q, p->n, st);
    if (e != hipSuccess) rc = hip_fail(e, "estimator launch (average inside the kernel)");
  }
  if (means) glfer::scratch_free(means, st);
  if (ret_scratch) glfer::scratch_free(ret_scratch, st);
  return rc;
}

// the sliding sums alone (avgdata->cum after each frame, avg.c:114-127); bins outside
// [minbin, maxbin) are left untouched
int glfer_hip_avg_cum_device(const float *d_psd, size_t nframes, int bins, int n_out, int depth, int minbin,
                             int maxbin, double *d_cum, void *hip_stream) {
  if (!d_psd || !d_cum) return GLFER_E_ARG;
  if (depth < 1 || minbin < 0 || maxbin <= minbin || maxbin > bins || maxbin > n_out || n_out < 1) return GLFER_E_ARG;
  DeviceGuard guard(data_device(d_psd));
  HIP_TRY(guard.error());
  HIP_TRY(glfer_launch_avg_cum(d_psd, nframes, bins, n_out, depth, minbin, maxbin, d_cum, (hipStream_t)hip_stream));
  return GLFER_OK;
}

}  // extern "C"
