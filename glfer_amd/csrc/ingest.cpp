// ingest.cpp -- host samples in, host rows out: the file / buffer source of source.c:118-158 in
// front of the batch estimator, and the multi-GPU entry.
//
// A job covers a contiguous frame range of one stream on one GPU.  It runs in chunks of whole frames
// through a two-deep ring: two pinned sample buffers, two device sample buffers, two device row
// buffers, two pinned row buffers and TWO STREAMS, chunk c on stream c mod 2, so chunk c+1's
// upload runs while chunk c computes and downloads.  Every chunk's sample buffer starts with the
// history it needs (the N-H overlap rounded up to whole hops, plus lmp_av-1 hops in LMP mode),
// taken from the tail of the previous chunk's pinned buffer, and the kernels are handed the
// stream's VIRTUAL base address, so frame indices -- and the zero history of the very first frames
// -- come out as in a one-shot run.  Rows go straight into the caller's memory when it is pinned
// (glfer_hip_host_alloc), otherwise through the pinned row buffers and a host copy.
#include "plan.h"

#include <algorithm>
#include <cctype>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <chrono>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>
#include <unistd.h>

using glfer::DeviceGuard;
using glfer::hip_fail;

namespace glfer {
// ONE idle ring per device outlives its plan (round 4): the *_multi / *_workers entries make a plan per worker and per call, and a
// ring is ~0.5 GB of pinned and device memory that takes 30-40 ms to allocate -- half of what a 1-hour WAV costs end to end.  A
// destroyed plan parks its ring here (if the slot is empty and the ring under the cap), a new plan's first job takes it.
// glfer_hip_scratch_trim(device, 0) frees it with the kept scratch.
static std::mutex g_spare_mu;
static IngestRing *g_spare_ring[64] = {nullptr};
static size_t ring_bytes(const IngestRing *r) {
  size_t b = 0;
  for (int i = 0; i < 2; i++)
    for (int k = 0; k < 8; k++) b += r->cap[i][k];
  return b;
}
IngestRing *ingest_ring_take(int dev) {
  if (dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(g_spare_mu);
  IngestRing *r = g_spare_ring[dev];
  g_spare_ring[dev] = nullptr;
  return r;
}
static void ingest_ring_destroy(IngestRing *r);
void ingest_ring_drop_spare(int dev) {            // with `dev` current
  IngestRing *r = ingest_ring_take(dev);
  if (r) ingest_ring_destroy(r);
}
size_t ingest_ring_spare_bytes(int dev) {
  if (dev < 0 || dev >= 64) return 0;
  std::lock_guard<std::mutex> lock(g_spare_mu);
  return g_spare_ring[dev] ? ring_bytes(g_spare_ring[dev]) : 0;
}
void ingest_ring_free(IngestRing *r) {          // with the ring's device current (plan_destroy, run_job)
  if (!r) return;
  int dev = -1;
  // parked only if the host lets the library keep memory at all (GLFER_SCRATCH_CACHE, glfer_hip_scratch_limit: a parked ring is
  // unswappable pinned memory plus device memory) and the ring is under both caps
  if (scratch_keeping() && ring_bytes(r) <= scratch_cap() &&
      hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !r->busy && ring_bytes(r) <= ((size_t)2 << 30)) {
    if (r->up) (void)hipStreamSynchronize(r->up);
    for (int i = 0; i < 2; i++)
      if (r->st[i]) (void)hipStreamSynchronize(r->st[i]);
    std::lock_guard<std::mutex> lock(g_spare_mu);
    if (!g_spare_ring[dev]) {
      g_spare_ring[dev] = r;
      return;
    }
  }
  ingest_ring_destroy(r);
}
static void ingest_ring_destroy(IngestRing *r) {
  if (!r) return;
  if (r->up) {
    (void)hipStreamSynchronize(r->up);
    (void)hipStreamDestroy(r->up);
  }
  for (int i = 0; i < 2; i++) {
    if (r->st[i]) (void)hipStreamSynchronize(r->st[i]);
    if (r->ev_up[i]) (void)hipEventDestroy(r->ev_up[i]);
    for (hipEvent_t ev : r->ev_t[i])
      if (ev) (void)hipEventDestroy(ev);
    if (r->h_in[i]) (void)hipHostFree(r->h_in[i]);
    if (r->h_out[i]) (void)hipHostFree(r->h_out[i]);
    if (r->h_lev[i]) (void)hipHostFree(r->h_lev[i]);
    if (r->d_in[i]) (void)hipFree(r->d_in[i]);
    if (r->d_psd[i]) (void)hipFree(r->d_psd[i]);
    if (r->d_stats[i]) (void)hipFree(r->d_stats[i]);
    if (r->d_rgb[i]) (void)hipFree(r->d_rgb[i]);
    if (r->d_lev[i]) (void)hipFree(r->d_lev[i]);
    if (r->st[i]) (void)hipStreamDestroy(r->st[i]);
  }
  delete r;
}
}  // namespace glfer

namespace {

bool numa_bind_on() {
  static const bool on = [] { const char *e = getenv("GLFER_NUMA_BIND"); return !(e && *e == '0'); }();
  return on;
}

// binds the calling thread to the CPUs of `device`'s NUMA node for its lifetime (nothing if the node is unknown, the intersection
// with the thread's own mask is empty, or GLFER_NUMA_BIND=0), and gives the thread its mask back
struct NodeBinding {
  cpu_set_t saved;
  bool bound = false;
  explicit NodeBinding(int device) {
    cpu_set_t want;
    if (device < 0 || !numa_bind_on() || sched_getaffinity(0, sizeof saved, &saved) != 0) return;
    char bus[32] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    unsigned char mask[CPU_SETSIZE / 8];
    const int node = glfer_hip_numa_node_of_bus_id(bus, nullptr);
    if (node < 0 || glfer_hip_numa_node_cpus(node, nullptr, mask, sizeof mask) <= 0) return;
    CPU_ZERO(&want);
    int n = 0;
    for (int c = 0; c < CPU_SETSIZE; c++)
      if ((mask[c >> 3] >> (c & 7) & 1) && CPU_ISSET(c, &saved)) {
        CPU_SET(c, &want);
        n++;
      }
    bound = n > 0 && sched_setaffinity(0, sizeof want, &want) == 0;
  }
  ~NodeBinding() {
    if (bound) (void)sched_setaffinity(0, sizeof saved, &saved);
  }
  NodeBinding(const NodeBinding &) = delete;
  NodeBinding &operator=(const NodeBinding &) = delete;
};

std::mutex g_ring_mu;

size_t sample_bytes(int fmt) { return fmt == GLFER_SAMPLES_F32 ? 4 : (fmt == GLFER_SAMPLES_S16 ? 2 : 1); }

// memcpy spread over a few threads: the destination is usually fresh pageable memory, where the
// page faults, not the copy, set the pace
// A worker's own reader threads, kept by its glfer_hip_workers handle (round 5).  read_wide used to start its threads per chunk: sixteen
// chunks x up to sixteen threads a call, each with a fresh stack on a process's first call -- 31 ms of a 1-hour WAV's first call
// against 10 ms afterwards.  The threads are made by the worker's own thread while it is bound to its GPU's NUMA node, so they read
// into the pinned ring from that node's cores.
class ReaderPool {
 public:
  explicit ReaderPool(unsigned n) {
    for (unsigned i = 0; i < n; i++) {
      try {
        th_.emplace_back([this] { loop(); });
      } catch (...) {
        break;
      }
    }
  }
  ~ReaderPool() {
    {
      std::lock_guard<std::mutex> lock(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  unsigned size() const { return (unsigned)th_.size(); }
  // fn(i) for i in [0, count): on the pool's threads and on the caller; returns when every piece is done
  void run(unsigned count, const std::function<void(unsigned)> &fn) {
    if (count == 0) return;
    {
      std::lock_guard<std::mutex> lock(mu_);
      fn_ = &fn;
      count_ = count;
      next_.store(0, std::memory_order_relaxed);
      left_.store(count, std::memory_order_relaxed);
      gen_++;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lock(mu_);
    done_.wait(lock, [this] { return left_.load(std::memory_order_acquire) == 0 && busy_ == 0; });
    fn_ = nullptr;
  }

 private:
  void work() {
    for (;;) {
      const unsigned i = next_.fetch_add(1, std::memory_order_relaxed);
      if (i >= count_) return;
      (*fn_)(i);
      left_.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  void loop() {
    unsigned long long seen = 0;
    std::unique_lock<std::mutex> lock(mu_);
    for (;;) {
      cv_.wait(lock, [&] { return stop_ || gen_ != seen; });
      if (stop_) return;
      seen = gen_;
      busy_++;
      lock.unlock();
      work();
      lock.lock();
      busy_--;
      if (left_.load(std::memory_order_acquire) == 0 && busy_ == 0) done_.notify_all();
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(unsigned)> *fn_ = nullptr;
  unsigned count_ = 0, busy_ = 0;
  unsigned long long gen_ = 0;
  std::atomic<unsigned> next_{0}, left_{0};
  bool stop_ = false;
};
static thread_local ReaderPool *tl_reader_pool = nullptr;      // the calling worker's pool, if its handle has one

void copy_wide(void *dst, const void *src, size_t bytes);
void copy_wide_pooled(void *dst, const void *src, size_t bytes) {
  ReaderPool *pool = tl_reader_pool;
  const size_t kMin = (size_t)2 << 20;
  const size_t want = pool ? std::max<size_t>(1, std::min<size_t>(pool->size() + 1, bytes / kMin)) : 1;
  if (want <= 1) { copy_wide(dst, src, bytes); return; }
  const size_t per = (((bytes + want - 1) / want) + 4095) & ~(size_t)4095;
  const unsigned pieces = (unsigned)((bytes + per - 1) / per);
  pool->run(pieces, [&](unsigned i) {
    const size_t off = (size_t)i * per, len = std::min(per, bytes - off);
    memcpy((char *)dst + off, (const char *)src + off, len);
  });
}
void copy_wide(void *dst, const void *src, size_t bytes) {
  const size_t kMin = (size_t)8 << 20;
  unsigned nt = bytes < 2 * kMin ? 1u : (unsigned)std::min<size_t>(8, bytes / kMin);
  const unsigned hw = std::thread::hardware_concurrency();
  if (hw && nt > hw) nt = hw;
  if (nt <= 1) { memcpy(dst, src, bytes); return; }
  std::vector<std::thread> th;
  const size_t per = ((bytes / nt) + 4095) & ~(size_t)4095;
  size_t done = 0;                               // bytes handed to threads so far
  try {
    for (unsigned i = 0; i + 1 < nt && done + per < bytes; i++) {
      const size_t off = done;
      th.emplace_back([=] { memcpy((char *)dst + off, (const char *)src + off, per); });
      done += per;
    }
  } catch (...) {                                // no more threads to be had: this one copies the rest
  }
  memcpy((char *)dst + done, (const char *)src + done, bytes - done);
  for (auto &t : th) t.join();
}

// pread spread over a few threads (round 4): a file in the page cache is read at memcpy speed per thread, and BASELINE config 4
// as worded (a 1-hour WAV, 346 MB) was bound by ONE thread doing that -- 0.11 s of wall time around 1.2 ms of kernels
// Threads one reader may use at once (set by the *_workers entries: the host's cores are shared by the workers).
static std::atomic<unsigned> g_read_threads{8};


bool read_wide(int fd, void *dst, size_t offset, size_t bytes) {
  auto read_all = [fd](char *d, size_t off, size_t n) {
    while (n) {
      const ssize_t got = pread(fd, d, n, (off_t)off);
      if (got <= 0) return false;
      d += got;
      off += (size_t)got;
      n -= (size_t)got;
    }
    return true;
  };
  // (round 5: pieces of 2 MiB and up, as many threads as the caller's share of the cores -- a chunk of a few tens of MB read by
  // ONE to five threads at ~8 GB/s each was what a 1-hour WAV's 346 MB waited for: 19.7 ms wall around 1.2 ms of kernels)
  const size_t kMin = (size_t)2 << 20;
  if (ReaderPool *pool = tl_reader_pool) {                       // the worker's kept threads: no thread is made here
    const size_t want = std::max<size_t>(1, std::min<size_t>(pool->size() + 1, bytes / kMin));
    if (want > 1) {
      const size_t per = (((bytes + want - 1) / want) + 4095) & ~(size_t)4095;
      const unsigned pieces = (unsigned)((bytes + per - 1) / per);
      std::vector<char> ok(pieces, 1);
      pool->run(pieces, [&](unsigned i) {
        const size_t off = (size_t)i * per, len = std::min(per, bytes - off);
        ok[i] = read_all((char *)dst + off, offset + off, len) ? 1 : 0;
      });
      for (char c : ok)
        if (!c) return false;
      return true;
    }
    return read_all((char *)dst, offset, bytes);
  }
  unsigned nt = bytes < 2 * kMin ? 1u : (unsigned)std::min<size_t>(g_read_threads.load(std::memory_order_relaxed), bytes / kMin);
  const unsigned hw = std::thread::hardware_concurrency();
  if (hw && nt > hw) nt = hw;
  if (nt <= 1) return read_all((char *)dst, offset, bytes);
  const size_t per = ((bytes / nt) + 4095) & ~(size_t)4095;
  std::vector<std::thread> th;
  std::vector<char> ok(nt, 1);
  size_t done = 0;
  try {
    for (unsigned i = 0; i + 1 < nt && done + per < bytes; i++) {
      const size_t off = done;
      char *flag = &ok[i];
      th.emplace_back([=] { *flag = read_all((char *)dst + off, offset + off, per) ? 1 : 0; });
      done += per;
    }
  } catch (...) {                                // no more threads to be had: this one reads the rest
  }
  const bool mine = read_all((char *)dst + done, offset + done, bytes - done);
  for (auto &t : th) t.join();
  for (char c : ok)
    if (!c) return false;
  return mine;
}

bool is_pinned_host(const void *p) {
  hipPointerAttribute_t at;
  if (!p || hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

// What a job writes per frame.  PSD: bins floats.  Waterfall: the display mapping of
// main_window_draw (g_main.c:1099-1236) applied on the device -- bins*3 RGB bytes (+ bins shorts of
// levbuf) -- so 3-5 bytes per bin cross PCIe instead of 4, and the caller gets pixels.
struct Sink {
  float *h_psd = nullptr;
  glfer_hip_display *disp = nullptr;     // waterfall mode when set
  unsigned char *h_rgb = nullptr;
  short *h_lev = nullptr;
  // device sink (the first phase of a waterfall over several workers): the rows stay on the device,
  // [frames][bins] at d_rows, with their compute_floor statistics [frames][4] at d_stats; nothing
  // comes back to the host
  float *d_rows = nullptr;
  float *d_stats = nullptr;
};

struct Job {
  glfer_hip_plan *p = nullptr;
  size_t frame_lo = 0, frames = 0;       // global frame range [frame_lo, frame_lo + frames)
  long tail_fresh = -1;                  // >= 0: the last frame is the file's trailing partial block with this many fresh samples
  size_t chunk_frames = 0;               // 0 = default
  // fills dst with whole hops [hop_index, hop_index + nhops) of the stream, raw sample format;
  // returns the hops delivered (fewer only at the end of a file)
  std::function<size_t(unsigned char *dst, size_t hop_index, size_t nhops)> read;
  const unsigned char *pinned_src = nullptr;   // the whole stream in pinned host memory (hop 0 at this address): uploaded
                                               // from where it lies, no staging copy
  Sink sink;                             // rows of frame f go to index f - frame_lo
  glfer_hip_phases *phases = nullptr;    // optional: where this job's time went (glfer_hip_workers_*)
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Frames per chunk of a job of `frames` frames: the default is 16384 frames and at most 256 MiB of samples (tools/chunk_probe.py: long
// streams run at the link's rate there); a SHORT job -- a worker's share of a 1-hour WAV is 1 300 frames of 32 KB -- is cut into at
// least sixteen chunks of at least ~2 MiB, so that reading chunk c + 1, uploading it, and bringing chunk c - 1's rows home overlap
// instead of happening once each, one after the other.
size_t pick_chunk(const glfer_hip_plan *p, size_t frames, size_t asked) {
  const size_t esz = sample_bytes(p->cfg.sample_format), hop = (size_t)p->hop;
  size_t chunk = asked;
  if (chunk == 0) {
    chunk = 16384;
    if (const char *ev = getenv("GLFER_INGEST_CHUNK")) {       // (tools/chunk_probe.py)
      const long v = atol(ev);
      if (v > 0) chunk = (size_t)v;
    } else {
      const size_t part = (frames + 15) / 16, floor_frames = std::max<size_t>(1, ((size_t)2 << 20) / (hop * esz));
      if (part < chunk) chunk = std::max(part, floor_frames);
    }
    const size_t cap = ((size_t)256 << 20) / (hop * esz);
    if (chunk > cap) chunk = cap;
  }
  if (chunk < 1) chunk = 1;
  return (chunk + GLFER_FRAME_ALIGN - 1) / GLFER_FRAME_ALIGN * GLFER_FRAME_ALIGN;   // see launch_by_n
}

int run_job(const Job &job, size_t *frames_done) {
  glfer_hip_plan *p = job.p;
  *frames_done = 0;
  // rows go home through a dense ring: a pitched plan (cfg.psd_pitch) or display belongs to the device entries
  if (p->pitch != p->bins || (job.sink.disp && job.sink.disp->psd_pitch)) return GLFER_E_ARG;
  if (job.frames == 0) return GLFER_OK;
  const size_t esz = sample_bytes(p->cfg.sample_format);
  const size_t hop = (size_t)p->hop, bins = (size_t)p->bins;
  // history in front of every chunk: whole hops covering the N-H overlap (per-hop means need
  // complete hops), plus the frames the LMP ring reaches back
  const size_t halo_hops = (size_t)((p->keep + p->hop - 1) / p->hop) +
                           (p->cfg.mode == GLFER_MODE_LMP ? (size_t)p->lmp_av - 1 : 0);
  const double t_job = now_s();
  const size_t chunk = pick_chunk(p, job.frames, job.chunk_frames);
  // chunk boundaries sit on GLOBAL multiples of GLFER_FRAME_ALIGN: the first chunk of a job that
  // starts off the grid is shortened to reach it
  const bool waterfall = job.sink.disp != nullptr;
  const bool dev_sink = job.sink.d_rows != nullptr;
  const size_t row_bytes = waterfall ? bins * 3 : bins * sizeof(float);

  DeviceGuard guard(p->cfg.device);
  HIP_TRY(guard.error());
  const bool direct_out = waterfall ? is_pinned_host(job.sink.h_rgb) && (!job.sink.h_lev || is_pinned_host(job.sink.h_lev))
                                    : is_pinned_host(job.sink.h_psd);
  // The ring's buffers and streams belong to the PLAN and stay with it between calls (a call used to
  // spend ~17 ms allocating and freeing pinned and device memory around ~10 ms of work: the
  // read-ahead of the per-hop shims walks a file in many such calls); a second job running on the
  // same plan at the same time gets a ring of its own for the call.
  glfer::IngestRing *ring = nullptr;
  bool own_ring = false;
  {
    std::lock_guard<std::mutex> lock(g_ring_mu);
    if (!p->ring) p->ring = glfer::ingest_ring_take(p->cfg.device);      // the device's parked ring, if any (its buffers grow on demand)
    if (!p->ring) p->ring = new glfer::IngestRing();
    if (!p->ring->busy) {
      p->ring->busy = true;
      ring = p->ring;
    }
  }
  if (!ring) {
    ring = new glfer::IngestRing();
    ring->busy = true;
    own_ring = true;
  }
  glfer::IngestRing &R = *ring;
  unsigned char **h_in = R.h_in, **d_in = R.d_in, **h_out = R.h_out, **d_rgb = R.d_rgb;
  short **h_lev = R.h_lev, **d_lev = R.d_lev;
  float **d_psd = R.d_psd, **d_stats = R.d_stats;
  hipStream_t *st = R.st;
  const size_t in_bytes = (halo_hops + chunk + 1) * hop * esz;       // + 1: a trailing partial block rides on the last chunk
  const size_t rows_cap = chunk + 1;
  hipError_t e = hipSuccess;
  // a buffer of at least `bytes`: the one the ring has, or a new one (kind: 0 pinned host, 1 device)
  std::unique_ptr<NodeBinding> near_gpu;
  auto ensure = [&](void **ptr, size_t *cap, size_t bytes, int kind) {
    if (e != hipSuccess || (*ptr && *cap >= bytes)) return;
    if (*ptr) (void)(kind == 0 ? hipHostFree(*ptr) : hipFree(*ptr));
    *ptr = nullptr;
    *cap = 0;
    if (kind == 0 && !near_gpu) near_gpu.reset(new NodeBinding(p->cfg.device));   // pinned staging beside the GPU (see glfer_hip_host_alloc)
    e = kind == 0 ? hipHostMalloc(ptr, bytes, hipHostMallocDefault) : hipMalloc(ptr, bytes);
    if (e == hipSuccess) *cap = bytes;
  };
  // Uploads have a stream of their own.  With chunk c's upload, kernels and download all on stream c % 2 -- rounds 2-3 -- the two
  // streams ran in step: both uploading, then both downloading, the link used one way at a time (25 + 25 GB/s where this box
  // does 57 one way and 48 + 48 both ways: tools/pcie_probe.py shows the same for a bare ring of copies).  Queued behind one another
  // the uploads stagger the chunks by themselves: chunk c + 1 goes up while chunk c's rows come down.
  if (!R.up) e = hipStreamCreateWithFlags(&R.up, hipStreamNonBlocking);
  for (int b = 0; b < 2 && e == hipSuccess; b++) {
    if (!R.ev_up[b]) e = hipEventCreateWithFlags(&R.ev_up[b], hipEventDisableTiming);
    if (e != hipSuccess) break;
    if (!st[b]) e = hipStreamCreateWithFlags(&st[b], hipStreamNonBlocking);
    if (!job.pinned_src) ensure((void **)&h_in[b], &R.cap[b][0], in_bytes, 0);
    ensure((void **)&d_in[b], &R.cap[b][1], in_bytes, 1);
    if (!dev_sink) ensure((void **)&d_psd[b], &R.cap[b][2], rows_cap * bins * sizeof(float), 1);
    if (!direct_out && !dev_sink) ensure((void **)&h_out[b], &R.cap[b][3], rows_cap * row_bytes, 0);
    if (waterfall) {
      ensure((void **)&d_stats[b], &R.cap[b][4], rows_cap * 4 * sizeof(float), 1);
      ensure((void **)&d_rgb[b], &R.cap[b][5], rows_cap * bins * 3, 1);
      if (job.sink.h_lev) {
        ensure((void **)&d_lev[b], &R.cap[b][6], rows_cap * bins * sizeof(short), 1);
        if (!direct_out) ensure((void **)&h_lev[b], &R.cap[b][7], rows_cap * bins * sizeof(short), 0);
      }
    }
  }
  near_gpu.reset();                                // (the thread's own mask back before any work)
  const bool timed = job.phases != nullptr;
  for (int b = 0; b < 2 && timed && e == hipSuccess; b++)
    for (int k = 0; k < 4 && e == hipSuccess; k++)
      if (!R.ev_t[b][k]) e = hipEventCreate(&R.ev_t[b][k]);
  int rc = (e == hipSuccess) ? GLFER_OK : hip_fail(e, "ingest: allocate");
  double ph_read = 0.0, ph_h2d = 0.0, ph_kernel = 0.0, ph_d2h = 0.0;
  unsigned ph_chunks = 0;
  const double t_loop = now_s();

  struct Pending { size_t first = 0, nf = 0; bool live = false; } pend[2];
  // hands chunk b's rows to the caller (after its stream has drained)
  auto drain = [&](int b) -> int {
    if (!pend[b].live) return GLFER_OK;
    hipError_t err = hipStreamSynchronize(st[b]);
    pend[b].live = false;
    if (err != hipSuccess) return hip_fail(err, "ingest: chunk");
    if (timed) {
      float ms = 0.0f;
      if (hipEventElapsedTime(&ms, R.ev_t[b][0], R.ev_t[b][1]) == hipSuccess) ph_h2d += ms * 1e-3;
      if (hipEventElapsedTime(&ms, R.ev_t[b][1], R.ev_t[b][2]) == hipSuccess) ph_kernel += ms * 1e-3;   // (from the upload's end: includes any wait for the stream's previous chunk)
      if (hipEventElapsedTime(&ms, R.ev_t[b][2], R.ev_t[b][3]) == hipSuccess) ph_d2h += ms * 1e-3;
      (void)hipGetLastError();
      ph_chunks++;
    }
    if (!direct_out && !dev_sink) {
      const size_t off = pend[b].first - job.frame_lo, nf = pend[b].nf;
      if (waterfall) {
        copy_wide(job.sink.h_rgb + off * bins * 3, h_out[b], nf * bins * 3);
        if (job.sink.h_lev) copy_wide(job.sink.h_lev + off * bins, h_lev[b], nf * bins * sizeof(short));
      } else {
        copy_wide(job.sink.h_psd + off * bins, h_out[b], nf * bins * sizeof(float));
      }
    }
    *frames_done = pend[b].first - job.frame_lo + pend[b].nf;
    return GLFER_OK;
  };

  const size_t frame_hi = job.frame_lo + job.frames;
  size_t cf = job.frame_lo;                        // first frame of the chunk being prepared
  size_t prev_lo_hop = 0, prev_hops = 0;           // what the previous chunk's pinned buffer holds
  int b = 0;
  bool ended = false;                              // the reader ran dry (a file shorter than announced)
  while (rc == GLFER_OK && cf < frame_hi && !ended) {
    size_t nf = std::min(chunk - cf % chunk, frame_hi - cf);
    if (job.tail_fresh >= 0 && cf + nf + 1 == frame_hi) nf++;       // never leave the partial block alone in a chunk
    const size_t lo_hop = cf > halo_hops ? cf - halo_hops : 0;       // first hop the chunk's buffer holds
    // Buffer set b: its pinned SAMPLE buffer is free as soon as chunk c-2's upload is done (round 5: the loop used to wait here for
    // the whole of chunk c-2 -- kernels and download too -- before it read chunk c, so reading and the link took turns: a 1-hour WAV
    // ran 13.4 ms with 8 ms of reading and 7 ms of copies each way); everything else of the set is ordered on its stream, and the
    // chunk's rows are handed over (drain) after the read, just before the set's next work is queued.
    if (pend[b].live && !job.pinned_src) {
      e = hipEventSynchronize(R.ev_up[b]);
      if (e != hipSuccess) { rc = hip_fail(e, "ingest: upload"); break; }
    }
    if (job.pinned_src) {
      rc = drain(b);
      if (rc) break;
      // the caller's stream is pinned: history, hops and all go up from where they lie
      const size_t up = cf - lo_hop + nf;
      if (timed) (void)hipEventRecord(R.ev_t[b][0], R.up);
      e = hipMemcpyAsync(d_in[b], job.pinned_src + lo_hop * hop * esz, up * hop * esz, hipMemcpyHostToDevice, R.up);
      if (timed) (void)hipEventRecord(R.ev_t[b][1], R.up);
      if (e == hipSuccess) e = hipEventRecord(R.ev_up[b], R.up);
      if (e == hipSuccess) e = hipStreamWaitEvent(st[b], R.ev_up[b], 0);
      if (e != hipSuccess) { rc = hip_fail(e, "ingest: upload"); break; }
    }
    // history: from the previous chunk's pinned buffer where it has it, else from the reader
    size_t have = 0;                               // hops of [lo_hop, cf) copied so far
    const int pb = b ^ 1;
    const double t_read = now_s();
    if (job.pinned_src) {
      have = cf - lo_hop;
    } else if (prev_hops && lo_hop >= prev_lo_hop && lo_hop < prev_lo_hop + prev_hops) {
      const size_t from = lo_hop - prev_lo_hop;
      have = std::min(cf - lo_hop, prev_hops - from);
      memcpy(h_in[b], h_in[pb] + from * hop * esz, have * hop * esz);
    }
    if (!job.pinned_src && lo_hop + have < cf) {
      const size_t want = cf - lo_hop - have;
      if (job.read(h_in[b] + have * hop * esz, lo_hop + have, want) != want) { rc = GLFER_E_ARG; break; }
    }
    if (!job.pinned_src) {
      const size_t got = job.read(h_in[b] + (cf - lo_hop) * hop * esz, cf, nf);
      if (got < nf) { nf = got; ended = true; }
    }
    ph_read += now_s() - t_read;
    if (!job.pinned_src) {
      rc = drain(b);                               // chunk c-2's rows are home (usually long since): the set's device buffers and its pinned row buffer are free
      if (rc) break;
    }
    if (nf == 0) break;
    const bool has_tail = job.tail_fresh >= 0 && cf + nf == frame_hi;
    const size_t up_hops = cf - lo_hop + nf;
    if (!job.pinned_src) {
      if (timed) (void)hipEventRecord(R.ev_t[b][0], R.up);
      e = hipMemcpyAsync(d_in[b], h_in[b], up_hops * hop * esz, hipMemcpyHostToDevice, R.up);
      if (timed) (void)hipEventRecord(R.ev_t[b][1], R.up);
      if (e == hipSuccess) e = hipEventRecord(R.ev_up[b], R.up);
      if (e == hipSuccess) e = hipStreamWaitEvent(st[b], R.ev_up[b], 0);
      if (e != hipSuccess) { rc = hip_fail(e, "ingest: upload"); break; }
    }
    const unsigned char *vbase = d_in[b] - lo_hop * hop * esz;       // virtual address of stream sample 0
    float *rows = dev_sink ? job.sink.d_rows + (cf - job.frame_lo) * bins : d_psd[b];
    rc = glfer_run_device(p, vbase, (cf + nf) * hop, cf, nf, rows, nullptr, st[b], has_tail ? job.tail_fresh : -1);
    if (rc) break;
    if (timed) (void)hipEventRecord(R.ev_t[b][2], st[b]);
    if (dev_sink) {
      if (job.sink.d_stats) rc = glfer_hip_floor_device(rows, nf, (int)bins, job.sink.d_stats + (cf - job.frame_lo) * 4, st[b]);
      if (rc) break;
    } else if (waterfall) {
      rc = glfer_hip_floor_device(d_psd[b], nf, (int)bins, d_stats[b], st[b]);
      if (rc) break;
      // the level tracking carries its state from column to column (g_main.c:1081, 1111-1124): the
      // display call reads it back, so the previous chunk's display must have finished -- its
      // stream is drained by that call itself (glfer_hip_display_device synchronises)
      rc = glfer_hip_display_device(job.sink.disp, d_psd[b], nullptr, d_stats[b], nf, (int)bins, d_rgb[b], d_lev[b], nullptr, st[b]);
      if (rc) break;
      unsigned char *dst = direct_out ? job.sink.h_rgb + (cf - job.frame_lo) * bins * 3 : h_out[b];
      e = hipMemcpyAsync(dst, d_rgb[b], nf * bins * 3, hipMemcpyDeviceToHost, st[b]);
      if (e == hipSuccess && job.sink.h_lev) {
        short *ldst = direct_out ? job.sink.h_lev + (cf - job.frame_lo) * bins : h_lev[b];
        e = hipMemcpyAsync(ldst, d_lev[b], nf * bins * sizeof(short), hipMemcpyDeviceToHost, st[b]);
      }
    } else {
      float *dst = direct_out ? job.sink.h_psd + (cf - job.frame_lo) * bins : reinterpret_cast<float *>(h_out[b]);
      e = hipMemcpyAsync(dst, d_psd[b], nf * bins * sizeof(float), hipMemcpyDeviceToHost, st[b]);
    }
    if (e != hipSuccess) { rc = hip_fail(e, "ingest: download"); break; }
    if (timed) (void)hipEventRecord(R.ev_t[b][3], st[b]);
    pend[b].first = cf;
    pend[b].nf = nf;
    pend[b].live = true;
    prev_lo_hop = lo_hop;
    prev_hops = up_hops;
    cf += nf;
    b ^= 1;
  }
  // the two chunks still in flight, oldest first
  if (rc == GLFER_OK) rc = drain(b);
  if (rc == GLFER_OK) rc = drain(b ^ 1);
  if (R.up) (void)hipStreamSynchronize(R.up);
  for (int i = 0; i < 2; i++)
    if (st[i]) (void)hipStreamSynchronize(st[i]);
  if (own_ring) {
    glfer::ingest_ring_free(ring);
  } else {
    std::lock_guard<std::mutex> lock(g_ring_mu);
    ring->busy = false;
  }
  if (timed) {
    glfer_hip_phases &ph = *job.phases;
    ph.setup_s += t_loop - t_job;
    ph.read_s += ph_read;
    ph.h2d_s += ph_h2d;
    ph.kernel_s += ph_kernel;
    ph.d2h_s += ph_d2h;
    ph.wall_s += now_s() - t_job;
    ph.chunks += ph_chunks;
  }
  return rc;
}

// The plan's ring sized in advance for PSD jobs in chunks of `chunk` frames whose rows go home by DMA (glfer_hip_workers_create: a
// handle's first call must not spend 30-40 ms per worker allocating pinned and device buffers).  The same buffers run_job would make.
int reserve_psd_ring(glfer_hip_plan *p, size_t chunk) {
  DeviceGuard guard(p->cfg.device);
  HIP_TRY(guard.error());
  const size_t esz = sample_bytes(p->cfg.sample_format), hop = (size_t)p->hop, bins = (size_t)p->bins;
  const size_t halo_hops = (size_t)((p->keep + p->hop - 1) / p->hop) + (p->cfg.mode == GLFER_MODE_LMP ? (size_t)p->lmp_av - 1 : 0);
  const size_t in_bytes = (halo_hops + chunk + 1) * hop * esz, rows_cap = chunk + 1;
  {
    // (the ring is claimed under the lock and filled outside it: eight workers reserve theirs side by side)
    std::lock_guard<std::mutex> lock(g_ring_mu);
    if (!p->ring) p->ring = glfer::ingest_ring_take(p->cfg.device);
    if (!p->ring) p->ring = new glfer::IngestRing();
    if (p->ring->busy) return GLFER_OK;
    p->ring->busy = true;
  }
  glfer::IngestRing &R = *p->ring;
  struct Release {
    glfer::IngestRing &r;
    ~Release() {
      std::lock_guard<std::mutex> lock(g_ring_mu);
      r.busy = false;
    }
  } release{R};
  NodeBinding near_gpu(p->cfg.device);
  hipError_t e = hipSuccess;
  auto ensure = [&](void **ptr, size_t *cap, size_t bytes, int kind) {
    if (e != hipSuccess || (*ptr && *cap >= bytes)) return;
    if (*ptr) (void)(kind == 0 ? hipHostFree(*ptr) : hipFree(*ptr));
    *ptr = nullptr;
    *cap = 0;
    e = kind == 0 ? hipHostMalloc(ptr, bytes, hipHostMallocDefault) : hipMalloc(ptr, bytes);
    if (e == hipSuccess) *cap = bytes;
  };
  if (!R.up) e = hipStreamCreateWithFlags(&R.up, hipStreamNonBlocking);
  for (int b = 0; b < 2 && e == hipSuccess; b++) {
    if (!R.ev_up[b]) e = hipEventCreateWithFlags(&R.ev_up[b], hipEventDisableTiming);
    if (e == hipSuccess && !R.st[b]) e = hipStreamCreateWithFlags(&R.st[b], hipStreamNonBlocking);
    for (int k = 0; k < 4 && e == hipSuccess; k++)
      if (!R.ev_t[b][k]) e = hipEventCreate(&R.ev_t[b][k]);
    const bool fresh = !R.h_in[b] || R.cap[b][0] < in_bytes;
    ensure((void **)&R.h_in[b], &R.cap[b][0], in_bytes, 0);
    ensure((void **)&R.d_in[b], &R.cap[b][1], in_bytes, 1);
    ensure((void **)&R.d_psd[b], &R.cap[b][2], rows_cap * bins * sizeof(float), 1);
    // the pinned buffer's pages are mapped for the CPU on first touch: touched here, not by the first call's reader; and one copy each
    // way through the set's streams (the copy engines' first transfers after an idle spell run at a fraction of their rate)
    if (e == hipSuccess && fresh) {
      memset(R.h_in[b], 0, in_bytes);
      e = hipMemcpyAsync(R.d_in[b], R.h_in[b], in_bytes, hipMemcpyHostToDevice, R.up);
      if (e == hipSuccess) e = hipStreamSynchronize(R.up);
      if (e == hipSuccess) e = hipMemcpyAsync(R.h_in[b], R.d_in[b], in_bytes, hipMemcpyDeviceToHost, R.st[b]);
      if (e == hipSuccess) e = hipStreamSynchronize(R.st[b]);
    }
  }
  return e == hipSuccess ? GLFER_OK : hip_fail(e, "workers: reserve ring");
}

// reader over a host array of whole hops
std::function<size_t(unsigned char *, size_t, size_t)> array_reader(const void *h_stream, size_t hop_bytes) {
  return [=](unsigned char *dst, size_t hop_index, size_t nhops) {
    copy_wide_pooled(dst, (const unsigned char *)h_stream + hop_index * hop_bytes, nhops * hop_bytes);
    return nhops;
  };
}

// The frame range of rank r of `world` (the arithmetic of glfer_amd/shard.py frame_range): the
// stream is dealt out in units of GLFER_FRAME_ALIGN frames, the remainder units to the low ranks.
void frame_range(size_t total, unsigned rank, unsigned world, size_t *first, size_t *count) {
  const size_t align = GLFER_FRAME_ALIGN;
  const size_t units = (total + align - 1) / align;
  const size_t base = units / world, rem = units % world;
  const size_t u_first = rank * base + std::min<size_t>(rank, rem);
  const size_t u_count = base + (rank < rem ? 1 : 0);
  const size_t lo = std::min(total, u_first * align), hi = std::min(total, (u_first + u_count) * align);
  *first = lo;
  *count = hi - lo;
}

}  // namespace

extern "C" {

// The NUMA node of a PCI function: <sysfs_root>/bus/pci/devices/<bus id, lower case>/numa_node (sysfs_root NULL = "/sys").
// -1: unknown (no such file, an unparsable one, or the kernel's own -1 on a one-node host).
int glfer_hip_numa_node_of_bus_id(const char *bus_id, const char *sysfs_root) {
  if (!bus_id || !*bus_id) return -1;
  std::string id(bus_id);
  for (char &c : id) c = (char)tolower((unsigned char)c);
  if (id.find("..") != std::string::npos || id.find('/') != std::string::npos) return -1;
  if (id.size() == 7) id = "0000:" + id;                                  // "c1:00.0": the domain left out
  const std::string path = std::string(sysfs_root ? sysfs_root : "/sys") + "/bus/pci/devices/" + id + "/numa_node";
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  return node < 0 ? -1 : node;
}

// The CPUs of a node: <sysfs_root>/devices/system/node/node<N>/cpulist ("0-15,128-143") as a bit mask (bit c of
// mask[c / 8]); returns the number of CPUs set, -1 when the list cannot be read or parsed.
int glfer_hip_numa_node_cpus(int node, const char *sysfs_root, unsigned char *mask, size_t mask_bytes) {
  if (node < 0 || !mask || !mask_bytes) return -1;
  const std::string path = std::string(sysfs_root ? sysfs_root : "/sys") + "/devices/system/node/node" + std::to_string(node) + "/cpulist";
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return -1;
  char text[4096];
  const size_t got = fread(text, 1, sizeof text - 1, f);
  fclose(f);
  text[got] = 0;
  memset(mask, 0, mask_bytes);
  int count = 0;
  const char *q = text;
  while (*q) {
    while (*q == ',' || *q == ' ' || *q == '\n') q++;
    if (!*q) break;
    char *end = nullptr;
    const long a = strtol(q, &end, 10);
    if (end == q || a < 0) return -1;
    long b = a;
    q = end;
    if (*q == '-') {
      b = strtol(q + 1, &end, 10);
      if (end == q + 1 || b < a) return -1;
      q = end;
    }
    for (long c = a; c <= b; c++)
      if ((size_t)c < mask_bytes * 8 && !(mask[c >> 3] >> (c & 7) & 1)) {
        mask[c >> 3] |= (unsigned char)(1u << (c & 7));
        count++;
      }
  }
  return count;
}

void glfer_hip_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

void glfer_hip_frame_range(size_t total_frames, unsigned rank, unsigned world, size_t *first, size_t *count) {
  size_t f = 0, c = 0;
  if (world && rank < world) frame_range(total_frames, rank, world, &f, &c);
  if (first) *first = f;
  if (count) *count = c;
}

int glfer_hip_spectrogram_host(glfer_hip_plan *p, const void *h_stream, size_t nsamples, float *h_psd,
                               size_t *nframes_out) {
  if (!p || !h_stream || !nframes_out) return GLFER_E_ARG;
  if (p->pitch != p->bins) return GLFER_E_ARG;                 // host rows are dense (cfg.psd_pitch: the device entries)
  const size_t frames = nsamples / (size_t)p->hop;
  *nframes_out = frames;
  if (frames == 0) return GLFER_OK;
  if (!h_psd) return GLFER_E_ARG;
  const size_t esz = sample_bytes(p->cfg.sample_format);
  const size_t hop_bytes = (size_t)p->hop * esz;
  if (frames * hop_bytes + frames * (size_t)p->bins * sizeof(float) < ((size_t)4 << 20)) {
    // a few megabytes: one copy each way costs less than setting the pipeline up
    DeviceGuard guard(p->cfg.device);
    HIP_TRY(guard.error());
    const size_t used = frames * (size_t)p->hop;
    void *d_in = nullptr;
    float *d_out = nullptr;
    int rc = GLFER_OK;
    hipError_t e = hipMalloc(&d_in, used * esz);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, frames * (size_t)p->bins * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_in, h_stream, used * esz, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = hip_fail(e, "spectrogram_host: staging");
    if (rc == GLFER_OK) rc = glfer_run_device(p, d_in, used, 0, frames, d_out, nullptr, nullptr);
    if (rc == GLFER_OK) {
      e = hipMemcpy(h_psd, d_out, frames * (size_t)p->bins * sizeof(float), hipMemcpyDeviceToHost);
      if (e != hipSuccess) rc = hip_fail(e, "spectrogram_host: copy back");
    }
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    return rc;
  }
  Job job;
  job.p = p;
  job.frames = frames;
  job.read = array_reader(h_stream, hop_bytes);
  if (is_pinned_host(h_stream)) job.pinned_src = (const unsigned char *)h_stream;
  job.sink.h_psd = h_psd;
  return run_job(job, nframes_out);
}

// Host samples -> RGB waterfall columns (and levbuf): estimator, compute_floor and the display
// mapping on the device, 3 (+2) bytes per bin back over PCIe.
int glfer_hip_waterfall_host(glfer_hip_plan *p, glfer_hip_display *disp, const void *h_stream, size_t nsamples,
                             unsigned char *h_rgb, short *h_lev, size_t *nframes_out) {
  if (!p || !disp || !h_stream || !nframes_out) return GLFER_E_ARG;
  if (disp->scale_type < GLFER_SCALE_LIN || disp->scale_type > GLFER_SCALE_LOG_MAX0) return GLFER_E_ARG;
  const size_t frames = nsamples / (size_t)p->hop;
  *nframes_out = frames;
  if (frames == 0) return GLFER_OK;
  if (!h_rgb) return GLFER_E_ARG;
  Job job;
  job.p = p;
  job.frames = frames;
  job.read = array_reader(h_stream, (size_t)p->hop * sample_bytes(p->cfg.sample_format));
  if (is_pinned_host(h_stream)) job.pinned_src = (const unsigned char *)h_stream;
  job.sink.disp = disp;
  job.sink.h_rgb = h_rgb;
  job.sink.h_lev = h_lev;
  return run_job(job, nframes_out);
}

}  // extern "C"

// ---- sources: a stream of whole hops in the raw sample format, readable by several workers at once ----
namespace {

typedef std::function<size_t(unsigned char *, size_t, size_t)> HopReader;

struct Source {
  size_t frames = 0;                       // frames the stream yields (the trailing partial block counted when taken)
  long tail_fresh = -1;                    // >= 0: the LAST frame is a file's trailing partial block with this many fresh samples
  std::function<HopReader()> open;         // a reader of its own per worker (a file handle each); empty reader = failure
  const unsigned char *pinned_src = nullptr;
};

Source array_source(const void *h_stream, size_t frames, size_t hop_bytes, bool pinned) {
  Source s;
  s.frames = frames;
  s.open = [=] { return array_reader(h_stream, hop_bytes); };
  if (pinned) s.pinned_src = (const unsigned char *)h_stream;
  return s;
}

unsigned rd_u16(const unsigned char *b) { return b[0] | (b[1] << 8); }
unsigned rd_u32(const unsigned char *b) { return b[0] | (b[1] << 8) | (b[2] << 16) | ((unsigned)b[3] << 24); }

// The block reader of wav_read (wav_fmt.c:81-121) over hop indices: whole blocks from the data chunk;
// with_tail: the short last read counts as one more block (n_read != 0) whose fresh samples lie over
// the STALE rest of the reader's buffer -- here the RAW samples of the block before (zeros, the
// calloc of wav_fmt.c:99, for a file shorter than one block); with per-hop mean removal the device
// replaces the stale part by the corrected previous hop (submean_tail_kernel).
struct WavLayout {
  size_t data_offset = 0, hop_bytes = 0, esz = 1, whole = 0, fresh = 0;
  bool with_tail = false;
};
HopReader wav_reader(const std::string &path, const WavLayout &w) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return HopReader();
  std::shared_ptr<FILE> fp(f, fclose);
  return [fp, w](unsigned char *dst, size_t hop_index, size_t nhops) -> size_t {
    FILE *f = fp.get();
    const size_t hop_bytes = w.hop_bytes, whole = w.whole;
    const size_t full = hop_index + nhops <= whole ? nhops : (hop_index < whole ? whole - hop_index : 0);
    // the whole blocks by pread, several threads for a chunk of tens of MB (the stdio stream is only used for the ragged end below)
    if (!read_wide(fileno(f), dst, w.data_offset + hop_index * hop_bytes, full * hop_bytes)) return 0;
    if (full == nhops) return nhops;
    if (fseek(f, (long)(w.data_offset + (hop_index + full) * hop_bytes), SEEK_SET) != 0) return full;
    if (!w.with_tail || hop_index + nhops != whole + 1) return full;
    unsigned char *last = dst + full * hop_bytes;
    if (whole == 0) {
      memset(last, w.esz == 1 ? 0x80 : 0, hop_bytes);               // sample value 0.0 (u8: 128)
    } else if (full > 0) {
      memcpy(last, last - hop_bytes, hop_bytes);
    } else {
      if (fseek(f, (long)(w.data_offset + (whole - 1) * hop_bytes), SEEK_SET) != 0) return full;
      if (fread(last, 1, hop_bytes, f) != hop_bytes) return full;
      if (fseek(f, (long)(w.data_offset + whole * hop_bytes), SEEK_SET) != 0) return full;
    }
    if (w.fresh && fread(last, 1, w.fresh * w.esz, f) != w.fresh * w.esz) return full;
    return nhops;
  };
}

// probe + layout of a WAV file for an estimator of hop `hop` and sample format `fmt`
int wav_source(const char *path, int hop, int fmt, int mode, unsigned flags, size_t max_frames, Source *src) {
  if (!path || (flags & ~(unsigned)GLFER_WAV_PARTIAL_TAIL) || hop <= 0) return GLFER_E_ARG;
  glfer_wav_info wi;
  int rc = glfer_hip_wav_probe(path, &wi);
  if (rc) return rc;
  if (fmt != (wi.bits_per_sample == 8 ? GLFER_SAMPLES_U8 : GLFER_SAMPLES_S16)) return GLFER_E_ARG;
  WavLayout w;
  w.esz = (size_t)wi.bits_per_sample / 8;
  w.hop_bytes = (size_t)hop * w.esz;
  w.data_offset = wi.data_offset;
  w.whole = wi.data_bytes / w.hop_bytes;                           // full blocks, wav_fmt.c:102
  // wav_fmt.c:102-119: a short last read still counts as a block (n_read != 0) and converts
  // n_read (8 bit) or n_read/2 (16 bit: an odd last byte is dropped) samples over the stale rest
  const size_t rest = wi.data_bytes - w.whole * w.hop_bytes;
  const bool tail = (flags & GLFER_WAV_PARTIAL_TAIL) && rest > 0 && mode != GLFER_MODE_LMP;
  w.fresh = rest / w.esz;
  size_t frames = w.whole + (tail ? 1 : 0);
  w.with_tail = tail;
  if (frames > max_frames) { frames = max_frames; w.with_tail = false; }
  src->frames = frames;
  // (a partial block with no fresh sample at all -- one odd byte of a 16-bit file -- is the previous
  // block over again: tail_fresh = 0)
  src->tail_fresh = w.with_tail ? (long)w.fresh : -1;
  const std::string p(path);
  src->open = [p, w] { return wav_reader(p, w); };
  return GLFER_OK;
}

int check_devices(const int *devices, int nworkers) {
  if (!devices || nworkers < 1 || nworkers > 64) return GLFER_E_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return hip_fail(hipGetLastError(), "hipGetDeviceCount");
  for (int i = 0; i < nworkers; i++)
    if (devices[i] < 0 || devices[i] >= ndev) return GLFER_E_ARG;
  return GLFER_OK;
}

// ---- host-side placement of a GPU's worker ("scaling risk is host-side only -- PCIe, NUMA placement of
// pinned buffers, thread wake-ups").  A worker thread is bound to the CPUs of its GPU's NUMA node BEFORE it makes its plan
// and its pinned ring (hipHostMalloc places pinned memory where the allocating thread runs; the rows' host copy is spread
// over threads this one starts, which inherit the binding, so the caller's row range is first touched on that node
// too).  The node is the GPU's PCI function's: /sys/bus/pci/devices/<bus id>/numa_node, its CPUs from
// /sys/devices/system/node/node<N>/cpulist, intersected with what this process may use (a 1-GPU box hands a job 16 of
// the host's cores).  Nothing is bound when the node is unknown (-1: a one-node host), the intersection is empty or
// GLFER_NUMA_BIND=0.
// runs work(r) for r = 0 .. world-1, one host thread each (this thread takes worker 0); the first
// failure's code and text are what the caller sees.  devices (may be null): worker r's GPU, for the placement above.
int on_workers(unsigned world, const std::function<int(unsigned)> &work, const int *devices = nullptr) {
  std::vector<int> rcs(world, GLFER_OK);
  std::vector<std::string> msgs(world);
  auto run = [&](unsigned r) {
    NodeBinding here(devices && world > 1 ? devices[r] : -1);     // (worker 0 is the caller's own thread: its mask comes back)
    rcs[r] = work(r);
    if (rcs[r]) msgs[r] = glfer::error_text();
  };
  std::vector<std::thread> th;
  for (unsigned r = 1; r < world; r++) {
    try {
      th.emplace_back(run, r);
    } catch (...) {
      run(r);                                    // no thread to be had: this one does that share too
    }
  }
  run(0);
  for (auto &t : th) t.join();
  for (unsigned r = 0; r < world; r++)
    if (rcs[r]) {
      glfer::set_error_text(msgs[r]);
      return rcs[r];
    }
  return GLFER_OK;
}

// ---- multi-GPU: source.c:130-158 over one stream, the frame range dealt out over the GPUs of the
// node.  One host thread per GPU, each with its own plan, streams, reader and pinned ring (run_job);
// the ranges come from the same arithmetic as glfer_amd/shard.py; rows land in disjoint ranges of
// h_psd; nothing is exchanged between GPUs.
// One worker (host thread + plan + two streams + pinned ring) per entry of devices[]; an ordinal may
// appear more than once -- several workers then share that GPU, which is how a one-GPU box exercises
// the whole multi-worker path (frame offsets, halos from the middle of the stream, disjoint rows).
// plans (may be null): worker r's plan, kept by a glfer_hip_workers handle -- its tables and its ring are there already; without, a plan
// per worker is made and destroyed inside the call (its ring parks with the device if the slot is free).
// phases (may be null): per field the LARGEST value over the workers (they run side by side), the chunks summed.
int psd_workers(const glfer_hip_config *cfg, const int *devices, int nworkers, const Source &src, float *h_psd,
                glfer_hip_plan *const *plans = nullptr, glfer_hip_phases *phases = nullptr,
                const std::vector<std::unique_ptr<ReaderPool>> *pools = nullptr) {
  const unsigned world = (unsigned)nworkers;
  const size_t frames = src.frames, bins = (size_t)cfg->n / 2 + 1;
  // the host's cores are shared by the workers' readers (a 1-GPU box hands a job 16)
  unsigned avail = std::thread::hardware_concurrency();
  {
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) avail = (unsigned)CPU_COUNT(&set);
  }
  g_read_threads.store(std::max(1u, std::min(16u, (avail ? avail : 8u) / world)), std::memory_order_relaxed);
  std::vector<glfer_hip_phases> per(phases ? world : 0);
  const double t0 = now_s();
  int rc_all = on_workers(world, [&](unsigned r) -> int {
    size_t first = 0, count = 0;
    frame_range(frames, r, world, &first, &count);
    if (count == 0) return GLFER_OK;
    glfer_hip_plan *plan = plans ? plans[r] : nullptr;
    int rc = GLFER_OK;
    if (!plan) {
      glfer_hip_config c = *cfg;
      c.device = devices[r];
      rc = glfer_hip_plan_create(&c, &plan);
    }
    if (rc == GLFER_OK) {
      Job job;
      job.p = plan;
      job.frame_lo = first;
      job.frames = count;
      job.tail_fresh = first + count == frames ? src.tail_fresh : -1;   // the partial block is the stream's last frame
      job.read = src.open();
      job.pinned_src = src.pinned_src;
      job.sink.h_psd = h_psd + first * bins;
      if (phases) {
        memset(&per[r], 0, sizeof per[r]);
        job.phases = &per[r];
      }
      size_t done = 0;
      tl_reader_pool = pools && r < pools->size() ? (*pools)[r].get() : nullptr;     // this worker's kept reader threads
      rc = job.read ? run_job(job, &done) : GLFER_E_ARG;
      tl_reader_pool = nullptr;
      if (rc == GLFER_OK && done != count) rc = GLFER_E_HIP;
    }
    std::string msg = rc ? glfer::error_text() : std::string();
    if (!plans) glfer_hip_plan_destroy(plan);
    if (rc) glfer::set_error_text(msg);
    return rc;
  }, devices);
  if (phases) {
    memset(phases, 0, sizeof *phases);
    for (const glfer_hip_phases &w : per) {
      phases->setup_s = std::max(phases->setup_s, w.setup_s);
      phases->read_s = std::max(phases->read_s, w.read_s);
      phases->h2d_s = std::max(phases->h2d_s, w.h2d_s);
      phases->kernel_s = std::max(phases->kernel_s, w.kernel_s);
      phases->d2h_s = std::max(phases->d2h_s, w.d2h_s);
      phases->chunks += w.chunks;
    }
    phases->wall_s = now_s() - t0;
  }
  return rc_all;
}

// The waterfall of main_window_draw (g_main.c:1099-1236) over several workers.  The level tracking is
// one chain over ALL columns (g_main.c:1111-1124), fed by the 16 bytes of compute_floor statistics
// per column; everything else is per column.  Three phases: (1) every worker computes its frames' PSD
// rows -- they STAY on its GPU -- and their statistics; with a moving average of depth D it
// recomputes the D rows in front of its range instead of receiving them; (2) the
// statistics meet on the host and one walk gives every column its levels (glfer_hip_levels_host);
// (3) every worker maps its own rows with its slice of the levels and sends the pixels home.  What
// crosses between GPUs, through the host: 16 + 16 bytes per column.
struct AvgArgs { int mode, depth, minbin, maxbin, max0; };
int waterfall_workers(const glfer_hip_config *cfg, const int *devices, int nworkers, const Source &src,
                      glfer_hip_display *disp, const AvgArgs &av, unsigned char *h_rgb, short *h_lev) {
  const unsigned world = (unsigned)nworkers;
  const size_t frames = src.frames, bins = (size_t)cfg->n / 2 + 1;
  struct Worker {
    glfer_hip_plan *plan = nullptr;
    size_t first = 0, count = 0, lead = 0;
    float *d_rows = nullptr, *d_stats = nullptr;
  };
  std::vector<Worker> ws(world);
  std::vector<float> stats(frames * 4), levels(frames * 4);
  auto release = [&] {
    for (unsigned r = 0; r < world; r++) {
      DeviceGuard g(devices[r]);
      if (ws[r].d_rows) (void)hipFree(ws[r].d_rows);
      if (ws[r].d_stats) (void)hipFree(ws[r].d_stats);
      glfer_hip_plan_destroy(ws[r].plan);
      ws[r] = Worker();
    }
  };
  // (1) rows + statistics, resident
  int rc = on_workers(world, [&](unsigned r) -> int {
    Worker &w = ws[r];
    frame_range(frames, r, world, &w.first, &w.count);
    if (w.count == 0) return GLFER_OK;
    w.lead = av.mode ? std::min(w.first, (size_t)av.depth) : 0;
    glfer_hip_config c = *cfg;
    c.device = devices[r];
    int rc = glfer_hip_plan_create(&c, &w.plan);
    if (rc) return rc;
    DeviceGuard g(devices[r]);
    HIP_TRY(g.error());
    const size_t nrows = w.lead + w.count;
    hipError_t e = hipMalloc((void **)&w.d_rows, nrows * bins * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&w.d_stats, nrows * 4 * sizeof(float));
    if (e != hipSuccess) return hip_fail(e, "waterfall workers: rows");
    Job job;
    job.p = w.plan;
    job.frame_lo = w.first - w.lead;
    job.frames = nrows;
    job.tail_fresh = w.first + w.count == frames ? src.tail_fresh : -1;
    job.read = src.open();
    job.pinned_src = src.pinned_src;
    job.sink.d_rows = w.d_rows;
    job.sink.d_stats = w.d_stats;
    size_t done = 0;
    rc = job.read ? run_job(job, &done) : GLFER_E_ARG;
    if (rc == GLFER_OK && done != nrows) rc = GLFER_E_HIP;
    if (rc) return rc;
    e = hipMemcpy(stats.data() + w.first * 4, w.d_stats + w.lead * 4, w.count * 4 * sizeof(float), hipMemcpyDeviceToHost);
    return e == hipSuccess ? GLFER_OK : hip_fail(e, "waterfall workers: statistics");
  }, devices);
  // (2) one walk over all columns
  if (rc == GLFER_OK) rc = glfer_hip_levels_host(disp, stats.data(), frames, levels.data(), devices[0]);
  // (3) the pixel map, each worker its own columns
  if (rc == GLFER_OK)
    rc = on_workers(world, [&](unsigned r) -> int {
      Worker &w = ws[r];
      if (w.count == 0) return GLFER_OK;
      DeviceGuard g(devices[r]);
      HIP_TRY(g.error());
      float *d_levels = nullptr;
      unsigned char *d_rgb = nullptr;
      short *d_lev = nullptr;
      int rc = GLFER_OK;
      hipError_t e = hipMalloc((void **)&d_levels, w.count * 4 * sizeof(float));
      if (e == hipSuccess) e = hipMalloc((void **)&d_rgb, w.count * bins * 3);
      if (e == hipSuccess && h_lev) e = hipMalloc((void **)&d_lev, w.count * bins * sizeof(short));
      if (e == hipSuccess) e = hipMemcpy(d_levels, levels.data() + w.first * 4, w.count * 4 * sizeof(float), hipMemcpyHostToDevice);
      if (e != hipSuccess) rc = hip_fail(e, "waterfall workers: map buffers");
      if (rc == GLFER_OK)
        rc = glfer_hip_waterfall_map_device(disp, av.mode, av.depth, av.minbin, av.maxbin, av.max0, w.d_rows, w.lead, w.count,
                                            (int)bins, d_levels, d_rgb, d_lev, nullptr);
      if (rc == GLFER_OK) {
        e = hipMemcpy(h_rgb + w.first * bins * 3, d_rgb, w.count * bins * 3, hipMemcpyDeviceToHost);
        if (e == hipSuccess && h_lev) e = hipMemcpy(h_lev + w.first * bins, d_lev, w.count * bins * sizeof(short), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_fail(e, "waterfall workers: pixels");
      }
      if (d_levels) (void)hipFree(d_levels);
      if (d_rgb) (void)hipFree(d_rgb);
      if (d_lev) (void)hipFree(d_lev);
      return rc;
    }, devices);
  std::string msg = rc ? glfer::error_text() : std::string();
  release();
  if (rc) glfer::set_error_text(msg);
  return rc;
}

}  // namespace

// A kept set of workers (include/glfer_hip.h): per worker a plan on its GPU -- tables, and the chunk ring that stays with the plan.
struct glfer_hip_workers {
  glfer_hip_config cfg;
  std::vector<int> devices;
  std::vector<glfer_hip_plan *> plans;
  std::vector<std::unique_ptr<ReaderPool>> pools;  // worker r's reader threads (made by a thread bound to its GPU's NUMA node)
  std::mutex mu;                                   // one call at a time
};

namespace {

bool same_cfg(const glfer_hip_config &a, const glfer_hip_config &b) {   // (field by field: the struct has padding; cfg.device is per worker)
  return a.mode == b.mode && a.n == b.n && a.overlap == b.overlap && a.window_type == b.window_type && a.limiter_a == b.limiter_a &&
         a.enable_limiter == b.enable_limiter && a.sub_mean == b.sub_mean && a.history_mode == b.history_mode && a.mtm_w == b.mtm_w &&
         a.mtm_k == b.mtm_k && a.sample_format == b.sample_format && a.hparma_t == b.hparma_t && a.hparma_p_e == b.hparma_p_e &&
         a.lmp_av == b.lmp_av && a.psd_pitch == b.psd_pitch;
}

int make_workers(const glfer_hip_config *cfg, const int *devices, int nworkers, size_t hint_frames, glfer_hip_workers **out) {
  std::unique_ptr<glfer_hip_workers> w(new (std::nothrow) glfer_hip_workers());
  if (!w) return GLFER_E_NOMEM;
  w->cfg = *cfg;
  w->devices.assign(devices, devices + nworkers);
  w->plans.assign((size_t)nworkers, nullptr);
  w->pools.resize((size_t)nworkers);
  unsigned avail = std::thread::hardware_concurrency();
  {
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) avail = (unsigned)CPU_COUNT(&set);
  }
  const unsigned per_worker = std::max(1u, std::min(16u, (avail ? avail : 8u) / (unsigned)nworkers));   // the caller's thread is one of them
  const int hop = (int)(cfg->n * (1.0 - cfg->overlap));
  if (hop <= 0) return GLFER_E_ARG;
  // every worker makes its own plan and ring, side by side, bound to its GPU's NUMA node (the first plan's DPSS tapers are reused by the others)
  glfer_hip_config c0 = *cfg;
  c0.device = devices[0];
  int rc = glfer_hip_plan_create(&c0, &w->plans[0]);
  if (rc == GLFER_OK)
    rc = on_workers((unsigned)nworkers, [&](unsigned r) -> int {
      int rcw = GLFER_OK;
      if (r > 0) {
        glfer_hip_config c = *cfg;
        c.device = devices[r];
        rcw = glfer_hip_plan_create(&c, &w->plans[r]);
      }
      if (rcw == GLFER_OK && per_worker > 1) {
        try {
          w->pools[r].reset(new ReaderPool(per_worker - 1));     // (made here: this thread is bound to the GPU's NUMA node)
        } catch (...) {
        }
      }
      if (rcw == GLFER_OK && hint_frames != (size_t)-1) {
        size_t first = 0, count = 0;
        frame_range(hint_frames ? hint_frames : (size_t)16384 * (size_t)nworkers, r, (unsigned)nworkers, &first, &count);
        if (count) rcw = reserve_psd_ring(w->plans[r], pick_chunk(w->plans[r], count, 0));
      }
      return rcw;
    }, devices);
  if (rc != GLFER_OK) {
    std::string msg = glfer::error_text();
    for (glfer_hip_plan *p : w->plans) glfer_hip_plan_destroy(p);
    glfer::set_error_text(msg);
    return rc;
  }
  *out = w.release();
  return GLFER_OK;
}

void free_workers(glfer_hip_workers *w) {
  if (!w) return;
  for (glfer_hip_plan *p : w->plans) glfer_hip_plan_destroy(p);
  delete w;
}

// the stateless entries' own handles: the two most recently used
std::mutex g_kept_mu;
std::vector<std::shared_ptr<glfer_hip_workers>> g_kept;

std::shared_ptr<glfer_hip_workers> kept_workers(const glfer_hip_config *cfg, const int *devices, int nworkers, int *rc) {
  *rc = GLFER_OK;
  if (!glfer::scratch_keeping() || glfer::scratch_cap() == 0) return nullptr;
  {
    std::lock_guard<std::mutex> lock(g_kept_mu);
    for (size_t i = 0; i < g_kept.size(); i++) {
      glfer_hip_workers &k = *g_kept[i];
      if ((int)k.devices.size() == nworkers && std::equal(k.devices.begin(), k.devices.end(), devices) && same_cfg(k.cfg, *cfg)) {
        std::shared_ptr<glfer_hip_workers> hit = g_kept[i];
        g_kept.erase(g_kept.begin() + (long)i);
        g_kept.push_back(hit);                     // most recently used last
        return hit;
      }
    }
  }
  glfer_hip_workers *raw = nullptr;
  *rc = make_workers(cfg, devices, nworkers, (size_t)-1, &raw);      // (rings grow on the first call: the job's size is not known here)
  if (*rc != GLFER_OK) return nullptr;
  std::shared_ptr<glfer_hip_workers> made(raw, free_workers);
  std::lock_guard<std::mutex> lock(g_kept_mu);
  if (g_kept.size() >= 2) g_kept.erase(g_kept.begin());
  g_kept.push_back(made);
  return made;
}

// a stateless entry's call: through a kept handle when one can be had (and is idle), else with plans of its own
int psd_workers_kept(const glfer_hip_config *cfg, const int *devices, int nworkers, const Source &src, float *h_psd) {
  int rc = GLFER_OK;
  std::shared_ptr<glfer_hip_workers> k = kept_workers(cfg, devices, nworkers, &rc);
  if (rc != GLFER_OK) return rc;
  if (k) {
    std::unique_lock<std::mutex> busy(k->mu, std::try_to_lock);
    if (busy.owns_lock()) return psd_workers(cfg, devices, nworkers, src, h_psd, k->plans.data(), nullptr, &k->pools);
  }
  return psd_workers(cfg, devices, nworkers, src, h_psd);
}

}  // namespace

namespace glfer {
// the stateless entries' kept handles: dropped by glfer_hip_scratch_trim(device, 0) / glfer_hip_scratch_limit(0); their rings' bytes on
// `dev` are part of what glfer_hip_scratch_held reports
void workers_drop_kept() {
  std::vector<std::shared_ptr<glfer_hip_workers>> gone;
  {
    std::lock_guard<std::mutex> lock(g_kept_mu);
    gone.swap(g_kept);
  }
}                                                    // (a handle still in a running call lives until that call ends: shared_ptr)
size_t workers_kept_bytes(int dev) {
  std::lock_guard<std::mutex> lock(g_kept_mu);
  size_t b = 0;
  for (const auto &k : g_kept)
    for (const glfer_hip_plan *p : k->plans)
      if (p && p->cfg.device == dev && p->ring) b += ring_bytes(p->ring);
  return b;
}
}  // namespace glfer

namespace {

int mask_to_devices(unsigned device_mask, int devs[32]) {
  int n = 0;
  for (int d = 0; d < 32; d++)
    if (device_mask & (1u << d)) devs[n++] = d;
  return n;
}

bool avg_args_ok(const AvgArgs &av, size_t bins) {
  if (av.mode == 0) return true;
  return av.mode >= GLFER_AVG_SUMAVG && av.mode <= GLFER_AVG_SUMEXTREME && av.depth >= 1 && av.minbin >= 0 && av.maxbin > av.minbin &&
         (size_t)av.maxbin <= bins;
}

}  // namespace

extern "C" {

// Pinned host memory for samples and rows.  Allocated with the calling thread on the CPUs of the current device's NUMA node (pinning
// places the pages where the allocating thread runs; DMA to the other socket's memory crosses the sockets' link), the thread's own
// mask restored afterwards; GLFER_NUMA_BIND=0 or an unknown node: allocated where the thread is.
void *glfer_hip_host_alloc(size_t bytes) {
  void *p = nullptr;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    dev = -1;
  }
  NodeBinding here(dev);
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}


int glfer_hip_spectrogram_host_workers(const glfer_hip_config *cfg, const int *devices, int nworkers,
                                       const void *h_stream, size_t nsamples, float *h_psd, size_t *nframes_out) {
  if (!cfg || !h_stream || !nframes_out) return GLFER_E_ARG;
  int rc = check_devices(devices, nworkers);
  if (rc) return rc;
  // hop and bins as every plan will compute them (fft.c:70)
  const int hop = (int)(cfg->n * (1.0 - cfg->overlap));
  if (hop <= 0) return GLFER_E_ARG;
  const size_t frames = nsamples / (size_t)hop;
  *nframes_out = frames;
  if (frames == 0) return GLFER_OK;
  if (!h_psd) return GLFER_E_ARG;
  bool pinned_in = false;
  {
    DeviceGuard g0(devices[0]);                                  // (pointer attributes need a current device)
    pinned_in = g0.error() == hipSuccess && is_pinned_host(h_stream);
  }
  return psd_workers_kept(cfg, devices, nworkers, array_source(h_stream, frames, (size_t)hop * sample_bytes(cfg->sample_format), pinned_in),
                          h_psd);
}

// The GPUs named by a bit mask, one worker each.
int glfer_hip_spectrogram_host_multi(const glfer_hip_config *cfg, unsigned device_mask, const void *h_stream,
                                     size_t nsamples, float *h_psd, size_t *nframes_out) {
  if (device_mask == 0) return GLFER_E_ARG;
  int devs[32];
  const int n = mask_to_devices(device_mask, devs);
  return glfer_hip_spectrogram_host_workers(cfg, devs, n, h_stream, nsamples, h_psd, nframes_out);
}

int glfer_hip_waterfall_host_workers(const glfer_hip_config *cfg, const int *devices, int nworkers, glfer_hip_display *disp,
                                     int avg_mode, int depth, int minbin, int maxbin, int max0, const void *h_stream,
                                     size_t nsamples, unsigned char *h_rgb, short *h_lev, size_t *nframes_out) {
  if (!cfg || !disp || !h_stream || !nframes_out) return GLFER_E_ARG;
  if (disp->scale_type < GLFER_SCALE_LIN || disp->scale_type > GLFER_SCALE_LOG_MAX0) return GLFER_E_ARG;
  int rc = check_devices(devices, nworkers);
  if (rc) return rc;
  const int hop = (int)(cfg->n * (1.0 - cfg->overlap));
  if (hop <= 0) return GLFER_E_ARG;
  const AvgArgs av{avg_mode, depth, minbin, maxbin, max0};
  if (!avg_args_ok(av, (size_t)cfg->n / 2 + 1)) return GLFER_E_ARG;
  const size_t frames = nsamples / (size_t)hop;
  *nframes_out = frames;
  if (frames == 0) return GLFER_OK;
  if (!h_rgb) return GLFER_E_ARG;
  bool pinned_in = false;
  {
    DeviceGuard g0(devices[0]);
    pinned_in = g0.error() == hipSuccess && is_pinned_host(h_stream);
  }
  return waterfall_workers(cfg, devices, nworkers, array_source(h_stream, frames, (size_t)hop * sample_bytes(cfg->sample_format), pinned_in),
                           disp, av, h_rgb, h_lev);
}

// ---- ingest (wav_fmt.c:45-121, source.c:118-128) ------------------------------------------

// The reference takes the header as one fixed 44-byte struct (wav_fmt.h:34-52: "fmt " at byte 12, "data"
// at byte 36, samples from byte 44) and reads samples until read() returns 0.  Files written by anything
// but the simplest tools carry other chunks ("LIST", "fact", "bext" ...) before or after "data", which
// that layout would play as samples: the chunks are walked here (id + 32-bit little-endian size, padded
// to even), "fmt " gives the format fields, "data" the offset and length.  A file whose chunks cannot
// be walked (no "WAVE" tag, no "data" chunk, sizes running past the end) is read the reference's way.
int glfer_hip_wav_probe(const char *path, glfer_wav_info *info) {
  if (!path || !info) return GLFER_E_ARG;
  FILE *f = fopen(path, "rb");
  if (!f) return GLFER_E_ARG;                                    // wav_fmt.c:53-56 exits; we report
  unsigned char hd[44];
  const size_t got = fread(hd, 1, sizeof hd, f);
  long end = 0;
  if (fseek(f, 0, SEEK_END) == 0) end = ftell(f);
  if (got < sizeof hd || memcmp(hd, "RIFF", 4) != 0) {           // "input file less than 20 bytes long" /
    fclose(f);                                                   // "input file not in WAV format", wav_fmt.c:61-64
    return GLFER_E_ARG;
  }
  // the reference's fixed layout, the fallback
  const unsigned char *fmt = hd + 20;                            // wav_fmt.h:42-47
  size_t data_offset = 44, data_bytes = end > 44 ? (size_t)(end - 44) : 0;
  unsigned char fm[16];
  if (memcmp(hd + 8, "WAVE", 4) == 0) {
    long pos = 12;
    bool have_fmt = false, have_data = false;
    size_t d_off = 0, d_len = 0;
    while (pos + 8 <= end) {
      unsigned char ck[8];
      if (fseek(f, pos, SEEK_SET) != 0 || fread(ck, 1, 8, f) != 8) break;
      const size_t len = rd_u32(ck + 4);
      if (memcmp(ck, "fmt ", 4) == 0 && len >= 16 && !have_fmt) {
        if (fread(fm, 1, 16, f) != 16) break;
        have_fmt = true;
      } else if (memcmp(ck, "data", 4) == 0) {
        d_off = (size_t)pos + 8;
        const size_t left = (size_t)end - d_off;
        // a recorder that was cut off leaves 0 or 0xffffffff here: the data then runs to the end of the file
        d_len = (len == 0 || len == 0xffffffffu || len > left) ? left : len;
        if (d_len < left) {
          // Bytes after the declared data: another chunk ("LIST" ...: id of four printable characters and
          // a size that fits the file) is not samples; anything else is -- a header that was not
          // updated, a stray byte: the reference reads on to the end of the file (wav_fmt.c:102)
          const long after = (long)(d_off + d_len + (d_len & 1));
          unsigned char nx[8];
          bool chunk = false;
          if (after + 8 <= end && fseek(f, after, SEEK_SET) == 0 && fread(nx, 1, 8, f) == 8) {
            chunk = true;
            for (int i = 0; i < 4; i++) chunk = chunk && nx[i] >= 0x20 && nx[i] < 0x7f;
            chunk = chunk && (long)rd_u32(nx + 4) <= end - after - 8;
          }
          if (!chunk) d_len = left;
        }
        have_data = true;
        break;
      }
      pos += 8 + (long)len + (long)(len & 1);
    }
    if (have_fmt && have_data) {
      fmt = fm;
      data_offset = d_off;
      data_bytes = d_len;
    }
  }
  fclose(f);
  info->format = (int)rd_u16(fmt + 0);                           // wav_fmt.h:42
  info->channels = (int)rd_u16(fmt + 2);                         // wav_fmt.h:43
  info->sample_rate = (int)rd_u32(fmt + 4);                      // wav_fmt.h:44  -> *speed, wav_fmt.c:70
  info->bits_per_sample = (int)rd_u16(fmt + 14);                 // wav_fmt.h:47  -> bits, wav_fmt.c:71
  info->data_offset = data_offset;
  if (info->format != 1) return GLFER_E_ARG;                     // "input is not a PCM WAV file", wav_fmt.c:68-69
  if (info->bits_per_sample != 8 && info->bits_per_sample != 16) return GLFER_E_ARG;   // wav_fmt.c:87-96 handles only these
  info->data_bytes = data_bytes;
  info->nsamples = data_bytes / (size_t)(info->bits_per_sample / 8);
  return GLFER_OK;
}

// One GPU, frames [first_frame, first_frame + max_frames) of the file (clipped to what it holds): rows to
// h_psd[0 ..).  The read-ahead of the per-hop shims (glfer_compat.cpp) walks a file in such windows.
int glfer_hip_spectrogram_wav_range(glfer_hip_plan *p, const char *path, size_t first_frame, size_t max_frames, float *h_psd,
                                    size_t *nframes_out, size_t chunk_frames, unsigned flags) {
  if (!p || !path || !nframes_out) return GLFER_E_ARG;
  Source src;
  int rc = wav_source(path, p->hop, p->cfg.sample_format, p->cfg.mode, flags, (size_t)-1, &src);
  if (rc) return rc;
  size_t count = first_frame < src.frames ? src.frames - first_frame : 0;
  bool with_tail = src.tail_fresh >= 0;
  if (count > max_frames) { count = max_frames; with_tail = false; }
  *nframes_out = count;
  if (count == 0) return GLFER_OK;
  if (!h_psd) return GLFER_E_ARG;
  Job job;
  job.p = p;
  job.frame_lo = first_frame;
  job.frames = count;
  job.chunk_frames = chunk_frames;
  job.tail_fresh = with_tail ? src.tail_fresh : -1;
  job.sink.h_psd = h_psd;
  job.read = src.open();
  if (!job.read) return GLFER_E_ARG;
  return run_job(job, nframes_out);
}

int glfer_hip_spectrogram_wav_ex(glfer_hip_plan *p, const char *path, float *h_psd, size_t max_frames,
                                 size_t *nframes_out, size_t chunk_frames, unsigned flags) {
  return glfer_hip_spectrogram_wav_range(p, path, 0, max_frames, h_psd, nframes_out, chunk_frames, flags);
}

int glfer_hip_spectrogram_wav(glfer_hip_plan *p, const char *path, float *h_psd, size_t max_frames,
                              size_t *nframes_out, size_t chunk_frames) {
  return glfer_hip_spectrogram_wav_ex(p, path, h_psd, max_frames, nframes_out, chunk_frames, 0);
}

// BASELINE config 4 as worded -- "1-hour 48 kHz WAV, frame-batch sharded across 8 x MI355X": the file's
// frames dealt out over the workers, every worker reading its own part of the file (its hops + the
// history halo) through its own handle and pinned ring; rows to disjoint ranges of h_psd.
int glfer_hip_spectrogram_wav_workers(const glfer_hip_config *cfg, const int *devices, int nworkers, const char *path,
                                      float *h_psd, size_t max_frames, size_t *nframes_out, unsigned flags) {
  if (!cfg || !path || !nframes_out) return GLFER_E_ARG;
  int rc = check_devices(devices, nworkers);
  if (rc) return rc;
  Source src;
  rc = wav_source(path, (int)(cfg->n * (1.0 - cfg->overlap)), cfg->sample_format, cfg->mode, flags, max_frames, &src);
  if (rc) return rc;
  *nframes_out = src.frames;
  if (src.frames == 0) return GLFER_OK;
  if (!h_psd) return GLFER_E_ARG;
  return psd_workers_kept(cfg, devices, nworkers, src, h_psd);
}

// ---- the persistent set of workers (include/glfer_hip.h) ------------------------------------------------------------------------
int glfer_hip_workers_create(const glfer_hip_config *cfg, const int *devices, int nworkers, size_t hint_frames, glfer_hip_workers **out) {
  if (!cfg || !out) return GLFER_E_ARG;
  *out = nullptr;
  int rc = check_devices(devices, nworkers);
  if (rc) return rc;
  if (hint_frames == (size_t)-1) return GLFER_E_ARG;
  return make_workers(cfg, devices, nworkers, hint_frames, out);
}

void glfer_hip_workers_destroy(glfer_hip_workers *w) {
  if (!w) return;
  { std::lock_guard<std::mutex> wait_for_a_running_call(w->mu); }
  free_workers(w);
}

int glfer_hip_workers_spectrogram_wav(glfer_hip_workers *w, const char *path, float *h_psd, size_t max_frames, size_t *nframes_out,
                                      unsigned flags, glfer_hip_phases *phases) {
  if (!w || !path || !nframes_out) return GLFER_E_ARG;
  Source src;
  int rc = wav_source(path, (int)(w->cfg.n * (1.0 - w->cfg.overlap)), w->cfg.sample_format, w->cfg.mode, flags, max_frames, &src);
  if (rc) return rc;
  *nframes_out = src.frames;
  if (phases) memset(phases, 0, sizeof *phases);
  if (src.frames == 0) return GLFER_OK;
  if (!h_psd) return GLFER_E_ARG;
  std::lock_guard<std::mutex> one_call(w->mu);
  return psd_workers(&w->cfg, w->devices.data(), (int)w->devices.size(), src, h_psd, w->plans.data(), phases, &w->pools);
}

int glfer_hip_workers_spectrogram_host(glfer_hip_workers *w, const void *h_stream, size_t nsamples, float *h_psd, size_t *nframes_out,
                                       glfer_hip_phases *phases) {
  if (!w || !h_stream || !nframes_out) return GLFER_E_ARG;
  const int hop = (int)(w->cfg.n * (1.0 - w->cfg.overlap));
  const size_t frames = nsamples / (size_t)hop;
  *nframes_out = frames;
  if (phases) memset(phases, 0, sizeof *phases);
  if (frames == 0) return GLFER_OK;
  if (!h_psd) return GLFER_E_ARG;
  bool pinned_in = false;
  {
    DeviceGuard g0(w->devices[0]);                                 // (pointer attributes need a current device)
    pinned_in = g0.error() == hipSuccess && is_pinned_host(h_stream);
  }
  std::lock_guard<std::mutex> one_call(w->mu);
  return psd_workers(&w->cfg, w->devices.data(), (int)w->devices.size(),
                     array_source(h_stream, frames, (size_t)hop * sample_bytes(w->cfg.sample_format), pinned_in), h_psd, w->plans.data(), phases,
                     &w->pools);
}

int glfer_hip_spectrogram_wav_multi(const glfer_hip_config *cfg, unsigned device_mask, const char *path, float *h_psd,
                                    size_t max_frames, size_t *nframes_out, unsigned flags) {
  if (device_mask == 0) return GLFER_E_ARG;
  int devs[32];
  const int n = mask_to_devices(device_mask, devs);
  return glfer_hip_spectrogram_wav_workers(cfg, devs, n, path, h_psd, max_frames, nframes_out, flags);
}

int glfer_hip_waterfall_wav_workers(const glfer_hip_config *cfg, const int *devices, int nworkers, glfer_hip_display *disp,
                                    int avg_mode, int depth, int minbin, int maxbin, int max0, const char *path,
                                    size_t max_frames, unsigned char *h_rgb, short *h_lev, size_t *nframes_out, unsigned flags) {
  if (!cfg || !disp || !path || !nframes_out) return GLFER_E_ARG;
  if (disp->scale_type < GLFER_SCALE_LIN || disp->scale_type > GLFER_SCALE_LOG_MAX0) return GLFER_E_ARG;
  int rc = check_devices(devices, nworkers);
  if (rc) return rc;
  const AvgArgs av{avg_mode, depth, minbin, maxbin, max0};
  if (!avg_args_ok(av, (size_t)cfg->n / 2 + 1)) return GLFER_E_ARG;
  Source src;
  rc = wav_source(path, (int)(cfg->n * (1.0 - cfg->overlap)), cfg->sample_format, cfg->mode, flags, max_frames, &src);
  if (rc) return rc;
  *nframes_out = src.frames;
  if (src.frames == 0) return GLFER_OK;
  if (!h_rgb) return GLFER_E_ARG;
  return waterfall_workers(cfg, devices, nworkers, src, disp, av, h_rgb, h_lev);
}

int glfer_hip_waterfall_wav_multi(const glfer_hip_config *cfg, unsigned device_mask, glfer_hip_display *disp, int avg_mode,
                                  int depth, int minbin, int maxbin, int max0, const char *path, size_t max_frames,
                                  unsigned char *h_rgb, short *h_lev, size_t *nframes_out, unsigned flags) {
  if (device_mask == 0) return GLFER_E_ARG;
  int devs[32];
  const int n = mask_to_devices(device_mask, devs);
  return glfer_hip_waterfall_wav_workers(cfg, devs, n, disp, avg_mode, depth, minbin, maxbin, max0, path, max_frames, h_rgb, h_lev,
                                         nframes_out, flags);
}

}  // extern "C"
