// spectro16h.hip -- single-taper frames (the periodogram of fft_do/fft_psd, fft.c:190-226) as a
// REAL-input transform: the windowed frame y[0..N) is packed as z[n] = y[2n] + i*y[2n+1], one
// complex M = N/2 point Stockham FFT is run (stockham16.hpp, 16 points per lane, N/32 lanes per
// frame), and the N-point spectrum is recovered bin pair by bin pair,
//     E = Z[k] + conj(Z[M-k]),  O = Z[k] - conj(Z[M-k]),  P = -i * W_N^k * O,
//     X[k] = (E + P)/2,  X[M-k] = conj(E - P)/2,
// so |X[k]|^2 and |X[M-k]|^2 come from one E/P pair.  Against spectro16.hip's packed form (which
// spends a whole N-point complex transform on one real frame when there is no second taper to
// put in the imaginary part) this is half the butterflies and half the LDS exchange traffic.
// The 1/N of fft.c:212-216 and the two halvings above are folded into the window: w*sqrt(1/(4N)).
//
// Layout: frames per block = 256/(N/32) (N=4096: two frames, two wavefronts each).  The window
// is the same for every frame (in LDS up to N = 4096, re-read per frame above: see VAR); the next
// frame's samples are fetched -- one load per pair (y[2n], y[2n+1]) in every sample format, integer
// pairs staying raw until the frame is formed -- right after pass 0 has handed its data to LDS.
// Mirror step: only the upper half of Z (k >= M/2) goes through LDS, lane t keeps its own Z[k],
// k < M/2, in registers; the post twiddle W_N^(t + T*m) is the lane's W_N^t (two VGPRs) times the
// compile-time constant W_32^m.  MT = 1 runs several windows (the tapers of mtm_do, mtm.c:154-239)
// over the frame in turn and sums the spectra: the multitaper path for N >= 8192.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "stockham16.hpp"
#include "div_exact.hpp"

// Build variant per block size, chosen so that nothing spills (tools/hbench, -Rpass-analysis):
// window in LDS up to N = 4096 (165 VGPRs, 3 waves/SIMD, 53 KB LDS per block); re-read per frame
// above that, where the LDS copy would cost a resident block (N = 16384: 2 waves/SIMD, 194 VGPRs).
#ifndef GLFER16H_WAVES_PER_SIMD
#define GLFER16H_WAVES_PER_SIMD (GLFER_LOGN_OR(12) == 14 ? 2 : 3)
#endif
#ifndef GLFER16H_VAR
#define GLFER16H_VAR (GLFER_LOGN_OR(12) <= 12 ? 2 : 1)
#endif
#ifdef GLFER_LOGN
#define GLFER_LOGN_OR(d) GLFER_LOGN
#else
#define GLFER_LOGN_OR(d) d
#endif

// GLFER_H_ABL (tools/hbench timing ablations only; results are wrong): bit 0 = no PSD stores,
// bit 1 = no sample loads after a block's first frame
#ifndef GLFER_H_ABL
#define GLFER_H_ABL 0
#endif
#ifndef GLFER16H_PREFETCH_TOP
#define GLFER16H_PREFETCH_TOP 0    /* 1: next frame's samples requested at the top of the iteration, not after exchange 0's writes */
#endif
#ifndef GLFER16H_TW1_REGS
#define GLFER16H_TW1_REGS (GLFER_LOGN_OR(12) != 13)   /* the lane's 15 pass-1 twiddles in registers instead of 15 LDS reads per transform: the compiler still
                                     fits three wavefronts per SIMD (168 VGPRs, no spill at N = 4096); N = 512 +3.5 %, 1024 +-0, 2048 +1 %, 4096 +0.4...2 % (C2 +2.5 %),
                                     16384 +2...4 %; N = 8192 (window in registers there) spills and loses 7 %: off -- profiles/r03_h_tw1_regs.txt */
#endif
#ifndef GLFER16H_PREFETCH2
#define GLFER16H_PREFETCH2 0        /* 1: register-reuse forms (SHIFT 4 / 8, no mean removal): a frame's new pairs are requested TWO frames ahead, into landing registers */
#endif
#ifndef GLFER16H_SHIFT_BUILDS
#define GLFER16H_SHIFT_BUILDS 1    /* build the register-reuse forms for 75 % and 50 % overlap */
#endif
#ifndef GLFER16H_AVG_BUILDS
#define GLFER16H_AVG_BUILDS (GLFER_LOGN_OR(12) <= 12)   /* the average taken inside the kernel (AVG = 1): N = 512 .. 4096 */
#endif
#ifndef GLFER16H_AVG_STORE_AUX
#define GLFER16H_AVG_STORE_AUX 0   /* cache policy of the averaged rows' stores (2 = non-temporal) */
#endif
#ifndef GLFER16H_AVG_ABL
#define GLFER16H_AVG_ABL 0         /* timing ablations of the AVG forms (results wrong): 1 averaged rows not stored, 2 no window sum / quotient, 4 no band statistics */
#endif
#ifndef GLFER16H_AVG_TW1R
#define GLFER16H_AVG_TW1R 1        /* AVG forms: the lane's pass-1 twiddles in registers (0: read from LDS per transform, 32 VGPRs less) */
#endif
#ifndef GLFER16H_STORE_AUX
#define GLFER16H_STORE_AUX 0   /* default cache policy at every size.  Rounds 1-2 stored rows non-temporally from N = 4096 up (+3 % then,
                                  N = 8192 +25 %); with the register reuse and the contiguous frame ranges of round 2 in place the
                                  default policy is the faster one (N = 4096: +4...9 % at overlap 75 / 50 / 0 %, N = 8192: 0...5 %, N = 16384
                                  equal) and the non-temporal path is what wrote 1.09x the row bytes to HBM (default policy: 1.005x) --
                                  profiles/r03_store_policy.txt */
#endif

// Rows staged through LDS and stored 16 bytes per lane from a 64-byte boundary (round 3).  A PSD row is
// N/2+1 floats, so rows start 4 bytes further into a 64-byte granule with every frame: stored bin by
// bin, every wavefront store straddles a granule boundary (HBM writes 1.09x the row bytes, 33 store
// instructions per frame).  Staged: the row's floats go into the free half of the frame's exchange
// buffer at the offset they have inside their 64-byte granule in memory, one barrier, and a lane
// reads 16 aligned bytes and stores them; the row's first and last few floats (the parts of a
// 16-byte chunk it shares with its neighbours) go out as single floats.
#ifndef GLFER16H_OPAQUE_ROW
#define GLFER16H_OPAQUE_ROW 1    /* 0: A/B builds, the row offset left to the optimizer (tools/build_variant.sh) */
#endif
#ifndef GLFER16H_STAGE_ROWS
#define GLFER16H_STAGE_ROWS 0   /* measured, C2: 320 -> 264 M frames/s (HBM writes 1.09x -> see profiles/r03_c2_staged_rows.txt): one more barrier, 8 KB more LDS
                                   writes per frame and 24 spilled VGPRs cost more than the aligned stores save; kept for A/B builds */
#endif

namespace glfer {

template <int LOGN>
struct LaunchH {
  using C = Plan16<LOGN - 1>;
  static constexpr int M = C::N, TH = C::T;
#ifdef GLFER16H_FPB
  static constexpr int FPB = GLFER16H_FPB;                 // A/B builds: frames per workgroup
#else
  static constexpr int FPB = TH >= 256 ? 1 : 256 / TH;
#endif
  static constexpr int BLOCK = TH * FPB;
  static constexpr int PADM = M + M / 16;
  static constexpr int LDS_WORDS = FPB * PADM + 16 * 17;
};

// VAR (where the window lives): 0 = 32 VGPRs for the whole launch, 1 = re-read from the table
// at the top of every frame, 2 = in LDS ([16][T] pairs, shared by the block's frames)
// MT = 1: multitaper through the same transform -- p.htapers windows (each taper with its weight
// folded in) applied in turn to the frame held in registers, |X|^2 summed per bin in 17 more
// registers and stored after the last one.  Used where a frame's N-point exchange buffer would leave
// one workgroup per CU (N >= 8192): the real-input form needs half of it.  VAR must be 1.
// HIST = 1: history zeroed in every frame (history_mode ZERO_ALWAYS).  A template parameter because as
// a run-time test the compiler turns the zeroing into 32 unconditional selects per frame.
// SHIFT = K > 0: the hop is K of a lane's 16 sample registers (H/2 = K * N/32 points), so a frame's
// registers K..15 ARE its successor's registers 0..15-K: every frame slot of a workgroup walks
// CONSECUTIVE frames, keeps those 16-K pairs and loads only the K new ones (75 % overlap: 4 loads
// per lane and frame instead of 16).
//
// MEAN = 1: per-hop mean removal (fft.c:86-96, sub_mean = opt.autoscale: the reference's default) done HERE
// instead of by a pre-pass that writes a corrected copy of the stream.  The hop is KM = SHIFT (or 16:
// overlap 0) of a lane's 16 sample registers, so a frame spans NH = 16/KM hops, each a fixed group of
// registers, and a sample is corrected by the mean of the hop it arrived in: xs[m] - mu[m / KM].
// The mean of the NEWEST hop is summed from the next frame's prefetched samples at the end of an
// iteration (lane partial over its KM pairs in register order, butterfly over the wavefront, the
// frame's wavefronts combined through LDS across the iteration's last barrier -- no extra barrier);
// the older hops' means move down with the frames a slot walks.  Wherever and whenever a hop's
// mean is formed, it is formed from the same lanes' same registers in the same order: one value.
// Producer side of the fused launch (SpectroParams::nprod): hop means in the reference's own order (fft.c:88-92: a float
// accumulated sample after sample) for hops [prod_hop0, prod_hop0 + prod_nhops), the shape of submean_seq.hip -- a lane walks a
// hop, 64 hops side by side in a wavefront, the samples coming in coalesced and transposed through a wavefront-private LDS
// tile (here [64 hops][32 samples], row stride 33: 8.4 KB a wavefront out of the workgroup's exchange buffer) -- as waves of a
// launch whose other workgroups transform.  A wavefront takes hop groups g = its index, + the number of producer wavefronts,
// ...: in stream order, which is the order the consumer workgroups are dispatched in.  When a group's means are written (and
// fenced) one lane adds 1 to the group's chunk counter; a consumer polls the counters of the chunks its hops lie in.
template <int FMT>
__device__ __forceinline__ void produce_hop_means(const SpectroParams &p, float *tile, unsigned lane, long long wave, long long nwaves) {
  constexpr unsigned esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  const int H = p.H;
  const long long groups = (p.prod_nhops + 63) / 64;
  const unsigned half = lane >> 5, l32 = lane & 31u;
  // The lock-stepped form: this wavefront belongs to front `fr` and walks that front's groups only, never more than prod_look hops
  // past what the front's consumers have FINISHED (front_done x frames per workgroup) -- the consumers that are running need hops
  // inside that allowance (the launcher sizes it: resident ranges + a margin), so nobody waits for somebody who waits for him.
  long long g_first = wave, g_step = nwaves, g_end = groups, front_hop0 = 0;
  int fr = -1;
  if (p.prod_front_frames > 0) {
    fr = (int)(wave & 7);                                   // (the caller hands wave = wavefront-in-front * 8 + front)
    const long long nfronts_waves = nwaves >> 3;
    const long long f_lo = (long long)fr * p.prod_front_frames, f_hi = f_lo + p.prod_front_frames < p.nframes ? f_lo + p.prod_front_frames : p.nframes;
    if (f_lo >= p.nframes) return;
    front_hop0 = p.frame0 + f_lo - (p.frame0 - p.prod_hop0);            // the front's first frame's OLDEST hop (its history reaches prod_hop0's distance back)
    const long long g0 = (front_hop0 - p.prod_hop0) / 64;
    g_end = (p.frame0 + f_hi - p.prod_hop0 + 63) / 64;
    if (g_end > groups) g_end = groups;
    g_first = g0 + (wave >> 3);
    g_step = nfronts_waves;
  }
  for (long long g = g_first; g < g_end; g += g_step) {
    if (fr >= 0) {
      const long long ahead = p.prod_hop0 + g * 64 - front_hop0;        // hops from the front's start to this group
      for (int spin = 0; spin < (1 << 16); spin++) {
        const long long done = (long long)__hip_atomic_load(p.front_done + fr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * p.prod_block_frames;
        if (ahead <= done + p.prod_look) break;
        __builtin_amdgcn_s_sleep(64);
      }
    }
    const long long hop0 = p.prod_hop0 + g * 64;
    const long long left = p.prod_hop0 + p.prod_nhops - hop0;
    const int rows = (int)(left < 64 ? left : 64);
    const long long span = (long long)rows * H * esz;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.stream)) + hop0 * (long long)H * esz, 0,
        (unsigned)(span > 0x7fffffffLL ? 0x7fffffffLL : span), 0x00020000);
    // Two tiles in flight (round 5: with one -- 8 KB per ~2 us of HBM latency -- a producer wavefront streamed ~4 GB/s, and the lock-stepped
    // launch's few producers could not keep their front ahead of its consumers)
    float va[32], vb[32];
    auto fetch = [&](float (&v)[32], int k0) {       // instruction j: hops 2j (lanes 0-31) and 2j + 1 (lanes 32-63), sample k0 + (lane & 31)
      const unsigned base = k0 + (int)l32 < H ? ((unsigned)half * (unsigned)H + (unsigned)(k0 + (int)l32)) * esz : 0x80000000u;
#pragma unroll
      for (int j = 0; j < 32; j++) v[j] = buf_sample<FMT>(rs, base + (unsigned)(2 * j) * (unsigned)H * esz, 0u);
    };
    float s = 0.0f;
    auto step = [&](float (&v)[32], int k0) {        // tile k0 out of its registers, the tile after next into them, then the chain over its 32 samples
#pragma unroll
      for (int j = 0; j < 32; j++) tile[(2 * j + (int)half) * 33 + (int)l32] = v[j];
      if (k0 + 64 < H) fetch(v, k0 + 64);
      const int kn = H - k0 < 32 ? H - k0 : 32;
      const float *row = tile + lane * 33;
      if (kn == 32) {
#pragma unroll
        for (int k = 0; k < 32; k++) s += row[k];    // fft.c:89-91: one float add per sample, in order (adds only: nothing to contract)
      } else {
        for (int k = 0; k < kn; k++) s += row[k];
      }
    };
    fetch(va, 0);
    if (32 < H) fetch(vb, 32);
    for (int k0 = 0; k0 < H; k0 += 64) {
      step(va, k0);
      if (k0 + 32 < H) step(vb, k0 + 32);
    }
    // Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): the group's 64 means leave as `sc1` stores -- whole 128-byte
    // lines by one store instruction of one wavefront, written through, so NO agent release is needed (a `__threadfence()` here
    // is a `buffer_wbl2`: every group flushing its XCD's L2 full of the consumers' rows -- measured: the whole launch 2.3x slower) --,
    // the wavefront waits for them (vmcnt 0), then one lane adds 1 to the chunk's counter (an agent-scope atomic at the memory
    // side).  The consumer: a relaxed `sc1` poll by one wavefront, an agent acquire, a workgroup barrier, plain loads.
    if ((int)lane < rows) __hip_atomic_store(p.means_out + hop0 + lane, s / (float)H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // fft.c:92: float /= int
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      if (p.prod_front_frames > 0) __hip_atomic_store(p.means_ready + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // a flag per group (two fronts may both write a boundary group: the same bits)
      else __hip_atomic_fetch_add(p.means_ready + (g * 64) / p.prod_chunk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// MTAB = 1 (with MEAN = 1; round 4): the hop means are GIVEN (p.means, the reference's own summation order,
// submean_seq.hip) -- a form of its own, so that it carries none of the summing code: it fits the three
// wavefronts per SIMD of the plain periodogram where the summing form needs two, and the next frame's
// table entry is requested at the top of a frame, not where it is needed.  A sample belongs to ONE hop, so
// x - mu is formed once, IN PLACE, when a hop's pairs are first used (`absorb`: integer pairs become floats
// there) -- what fft.c:93-95 does to the caller's buffer -- and the frames that share the pair afterwards
// read the corrected value: no second copy of the frame in registers.
// AVG = 1 (round 5): update_avg_plain (avg.c:108-159) taken HERE, on the PSD values while they are in registers -- north_star's
// "|X|^2 + block-average fused in-register".  A slot walks consecutive frames (whatever the overlap) and keeps the PSD of its
// 17 bins for the last three frames (51 VGPRs); per frame and bin the window's sum is formed in double, oldest row first
// (float terms in a double: exact unless a bin spans more than ~2^26 within the window -- then it is the restarted sum of
// avg_cum_kernel, not the reference's running one, like every chunk start of the stand-alone kernels), divided by depth + 1
// with the same correctly rounded quotient (div_exact.hpp) and stored as 8-byte words; the band mean and the peak bin
// (update_avg_plain's return value and *peakbin) come from a reduction over the slot's lanes.  The slot recomputes the depth-1
// frames in front of its range (recompute, do not exchange): `lead` iterations without stores.  The PSD rows
// themselves are stored only if p.psd is given.  Two wavefronts per SIMD (the rings and the double sums do not fit 168 VGPRs).
template <int LOGN, int FMT, int WPS = GLFER16H_WAVES_PER_SIMD, int VAR = GLFER16H_VAR, int MT = 0, int HIST = 0, int SHIFT = 0, int MEAN = 0, int MTAB = 0, int AVG = 0>
__global__ __launch_bounds__(LaunchH<LOGN>::BLOCK, WPS) void spectro16h_kernel(SpectroParams p) {
  static_assert(AVG == 0 || (MT == 0 && HIST == 0 && ((MEAN == 0 && MTAB == 0) || (MEAN == 1 && MTAB == 1))),
                "the average inside the kernel: the periodogram, plain or with GIVEN hop means (the reference's default, sub_mean = opt.autoscale), history from the stream");
  constexpr bool CONSEC = SHIFT > 0 || AVG != 0;         // every frame slot walks consecutive frames
  static_assert(MTAB == 0 || (MEAN == 1 && MT == 0), "given means: the periodogram's mean form");
  static_assert(SHIFT == 0 || (MT == 0 && HIST == 0), "register reuse: periodogram, history from the stream");
  static_assert(MEAN == 0 || ((MT == 0 || SHIFT == 0) && HIST == 0 && GLFER16_BARRIER_AFTER_READS != 0),
                "in-kernel mean removal: the periodogram, or the multitaper form with hop = frame; history from the stream");
  constexpr int KM = SHIFT > 0 ? SHIFT : (MEAN > 1 ? MEAN : 16);   // MEAN: register pairs per hop (MEAN = 8, 4: the multitaper form at 50 / 75 % overlap, no register reuse)
  static_assert(MEAN <= 1 || (MT != 0 && SHIFT == 0 && LaunchH<LOGN>::FPB == 1 && (MEAN == 8 || MEAN == 4)), "MEAN = 8, 4: multitaper form, a slot walks consecutive frames");
  constexpr int NH = 16 / KM;                            //       hops per frame
  static_assert(MT == 0 || VAR == 1, "the multitaper form re-reads its window per taper");
  using L = LaunchH<LOGN>;
  using C = typename L::C;
  constexpr int N = 1 << LOGN, M = L::M, T = L::TH, FPB = L::FPB, PADM = L::PADM, NPASS = C::NPASS;
  constexpr int TW1 = 15;
  constexpr int NTWR = C::NTW - TW1;
  constexpr int NT = NTWR > 0 ? NTWR : 1;
  constexpr unsigned esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  // integer samples stay raw in their registers until the frame is formed (a conversion next to the
  // load would wait for it: no prefetch), and their power-of-two scale (wav_fmt.c:104-117) rides in
  // the window -- (x/32768)*w and x*(w/32768) are the same float
  constexpr float kSampleScale = FMT == GLFER_FMT_F32 ? 1.0f : (FMT == GLFER_FMT_S16 ? 1.0f / 32768.0f : 1.0f / 128.0f);
#ifndef GLFER16H_LDS_PAD
#define GLFER16H_LDS_PAD 0       /* experiment builds: extra 8-byte words of LDS per workgroup (3400 at N = 4096: two workgroups per CU instead of three) */
#endif
  __shared__ __attribute__((aligned(16))) v2f32 lds[L::LDS_WORDS + (VAR == 2 ? M : 0) + GLFER16H_LDS_PAD];
  constexpr int WPF = T > 64 ? T / 64 : 1;               // wavefronts per frame
  __shared__ float mred[MEAN && !MTAB ? FPB * WPF * NH : 1];      // MEAN: the frame's wavefronts' partial sums
  // AVG: the slot's wavefronts' band partials, by frame parity (written before a frame's barrier, read after it; the next
  // frame writes the other set: no second barrier), and the new row's psd[minbin] (the running maximum starts there, avg.c:111)
  __shared__ double a_sum[AVG ? 2 * FPB * WPF : 1], a_max[AVG ? 2 * FPB * WPF : 1], a_init[AVG ? 2 * FPB : 1];
  __shared__ int a_idx[AVG ? 2 * FPB * WPF : 1];

  const unsigned tid = threadIdx.x;
  // the fused launch (MTAB, p.nprod > 0): the first nprod workgroups produce the hop means, the others are the launch as it was
  unsigned grid_w = gridDim.x, block_w = blockIdx.x;
  if constexpr (MTAB != 0) {
    if (p.nprod > 0) {
      if (blockIdx.x < (unsigned)p.nprod) {
        static_assert(L::BLOCK % 64 == 0 && (L::BLOCK / 64) * 64 * 33 * 4 <= L::LDS_WORDS * 8, "a [64][33] float tile per wavefront out of the exchange buffer");
        const unsigned wv = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
        // (lock-stepped: producer workgroup j serves front j mod 8; its wavefronts are numbered (wavefront within the front) * 8 + front)
        const long long wave_id = p.prod_front_frames > 0 ? ((long long)(blockIdx.x >> 3) * (L::BLOCK / 64) + wv) * 8 + (blockIdx.x & 7u)
                                                          : (long long)blockIdx.x * (L::BLOCK / 64) + wv;
        produce_hop_means<FMT>(p, reinterpret_cast<float *>(lds) + wv * (64 * 33), tid & 63u, wave_id, (long long)p.nprod * (L::BLOCK / 64));
        return;
      }
      grid_w = gridDim.x - (unsigned)p.nprod;
      block_w = blockIdx.x - (unsigned)p.nprod;
    }
  }
  const unsigned t = tid % T;
  const unsigned fl = tid / T;
  v2f32 *xb = lds + fl * PADM;
  v2f32 *tw1 = lds + FPB * PADM;

  {
    const v2f32 *tw = reinterpret_cast<const v2f32 *>(p.htw);
    for (unsigned i = tid; i < 256; i += L::BLOCK) {
      const unsigned k = i >> 4, q = i & 15;
      tw1[k * 17 + q] = q ? tw[(q - 1) * T + k] : v2f32{1.0f, 0.0f};
    }
  }
  float twr[NT], twi[NT];
  {
    const v2f32 *tw = reinterpret_cast<const v2f32 *>(p.htw) + t;
#pragma unroll
    for (int e = 0; e < NTWR; e++) {
      const v2f32 w = tw[(TW1 + e) * T];
      twr[e] = w.x;
      twi[e] = w.y;
    }
    if constexpr (NTWR == 0) twr[0] = twi[0] = 0.0f;
  }
  const v2f32 rot = reinterpret_cast<const v2f32 *>(p.hrot)[t];        // (cos, sin)(2 pi t / N)
  // the window, in the lane's point order: wn[m] = (w[2n], w[2n+1]), n = t + T*m
  v2f32 wn[16];
  typedef float v4f32 __attribute__((ext_vector_type(4)));
  auto load_window = [&](int j = 0) {
    const v4f32 *ht = reinterpret_cast<const v4f32 *>(p.htaps) + (size_t)j * (N / 4) + t;
#pragma unroll
    for (int mh = 0; mh < 8; mh++) {
      const v4f32 q = ht[T * mh];
      wn[2 * mh] = v2f32{q.x, q.y} * kSampleScale;
      wn[2 * mh + 1] = v2f32{q.z, q.w} * kSampleScale;
    }
  };
  v2f32 *wl = lds + L::LDS_WORDS;                 // VAR 2: wl[m*T + t]
  if constexpr (VAR == 0) load_window();
  if constexpr (VAR == 2) {
    if (fl == 0) {
      load_window();
#pragma unroll
      for (int m = 0; m < 16; m++) wl[m * T + t] = wn[m];
    }
  }
  __syncthreads();
  const v2f32 *tw1row = tw1 + (t & 15) * 17;
  // (integer samples at 75 % overlap: the registers push the three-wavefront form over its 168 and the 2-4 spilled dwords cost
  // 4-6 %: the LDS row there -- profiles/r03_h_tw1_regs.txt)
  constexpr bool TW1R = GLFER16H_TW1_REGS != 0 && !(FMT != GLFER_FMT_F32 && SHIFT == 4 && MTAB == 0) && (AVG == 0 || GLFER16H_AVG_TW1R != 0);
  v2f32 tw1reg[16];
  if constexpr (TW1R) {
#pragma unroll
    for (int q = 0; q < 16; q++) tw1reg[q] = tw1row[q];
  }

  v2f32 px[16];
  // The launcher hands this kernel only frames that lie wholly inside the stream
  // ((frame0+f)*H >= R; the first ceil(R/H) frames of a stream go to spectro16.hip, which has the
  // zero-history gather), so every load is in range: one shared VGPR offset + immediates.
  // history_mode 1 zeroes the first R samples of every frame afterwards (fft.c:103-108).
  // A workgroup walks the frames start .. start + per*FPB - 1; `rel` is a frame's index in that
  // range, clamped to the launch's last frame (a slot past the end re-reads that frame; what it
  // computes is dropped by the output descriptor's range check).
  long long start = 0;                                   // first frame of the workgroup's range
  // Pair m of the frame at rotation ROT lives in px[(m + ROT) & 15] (ROT != 0 only with SHIFT 4 / 8,
  // where the frame loop is unrolled over the rotations instead of moving registers).
  constexpr bool PF2 = GLFER16H_PREFETCH2 != 0 && (SHIFT == 4 || SHIFT == 8) && MEAN == 0 && MT == 0 && HIST == 0;
  v2f32 pb[PF2 ? 2 : 1][PF2 ? SHIFT : 1];               // PF2: the new pairs of frames f+1 / f+2 land here (frame parity picks the set)
  auto load_pairs = [&](long long rel, auto fromc, auto rotc, auto destc) {
    constexpr int FROM = decltype(fromc)::value;         // pairs FROM..15 are loaded
    constexpr int ROT = decltype(rotc)::value;
    constexpr int DEST = decltype(destc)::value;         // 0: px (rotated); 1, 2: landing set DEST - 1
    // AVG: a slot starts `lead` frames in front of its range (rel >= -lead): indices and base shifted so that they stay unsigned
    const long long lead_l = AVG != 0 ? (long long)(p.avg_depth - 1) : 0;
    const long long last_rel = (long long)p.nframes - 1 - start + lead_l;
    const unsigned relc = (unsigned)(rel + lead_l < last_rel ? rel + lead_l : last_rel);
    const long long sblk = (p.frame0 + start - lead_l) * (long long)p.H - p.R;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.stream)) + sblk * (long long)esz, 0, 0x7fffffff, 0x00020000);
    const unsigned lrel = relc * (unsigned)p.H + 2u * t;
    if constexpr (HIST != 0) {
      // ZERO_ALWAYS: the history is never loaded (a piece cut for this mode carries none: glfer_hip.h,
      // "Cutting a stream", rule 2).  The descriptor starts at the range's own first hop; pairs that
      // lie in the history get an out-of-range offset and read 0 -- prefetch_x puts the format's
      // zero there.  (Integer formats: H is even on this kernel -- pairs are naturally aligned -- so R
      // is, and no pair straddles the history's end; f32: below.)
      const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char *>(reinterpret_cast<const char *>(p.stream)) + (sblk + p.R) * (long long)esz, 0, 0x7fffffff, 0x00020000);
      const int d = 2 * (int)t - p.R;
      const int hrel = (int)(relc * (unsigned)p.H) + d;
      static_for<FROM, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value, q = (m + ROT) & 15;
        const bool ok = d >= -2 * T * m;
        const unsigned off = ok ? (unsigned)(hrel + 2 * T * m) * (unsigned)esz : 0x80000000u;
        if constexpr (FMT == GLFER_FMT_F32) {
          // an odd hop (f32 only: 8-byte loads need no alignment) makes R odd, and ONE pair of the frame
          // straddles the history's end: its second sample is the hop's first.  That pair is fetched
          // one sample up (the hop's samples 0, 1) and its first sample moved into place.
          const bool strad = d + 2 * T * m == -1;
          const v2f32 v = __builtin_bit_cast(v2f32, __builtin_amdgcn_raw_buffer_load_b64(hrsrc, strad ? (unsigned)(hrel + 2 * T * m + 1) * 4u : off, 0u, 0));
          px[q] = strad ? v2f32{0.0f, v.x} : v;
        } else if constexpr (FMT == GLFER_FMT_S16) {
          px[q].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(hrsrc, off, 0u, 0));
        } else {
          px[q].x = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(hrsrc, off, 0u, 0));
        }
      });
      return;
    }
    // y[2n] and y[2n+1] are adjacent: ONE load per pair in every format (the launcher sends streams
    // whose pairs are not naturally aligned to spectro16.hip)
    static_for<FROM, 16>([&](auto mc) {
      constexpr int m = decltype(mc)::value, q = (m + ROT) & 15;
      v2f32 &dst = DEST == 0 ? px[q] : pb[DEST > 0 ? DEST - 1 : 0][PF2 ? m - FROM : 0];
      if constexpr (FMT == GLFER_FMT_F32) {
        dst = __builtin_bit_cast(v2f32, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, lrel * 4u, (unsigned)(2 * T * m) * 4u, 0));
      } else if constexpr (FMT == GLFER_FMT_S16) {
        dst.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, lrel * 2u, (unsigned)(2 * T * m) * 2u, 0));
      } else {
        dst.x = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(xrsrc, lrel, (unsigned)(2 * T * m), 0));
      }
    });
  };
  constexpr std::integral_constant<int, 0> kToPx{};
  // the next frame of this slot: with SHIFT its first 16-SHIFT pairs are already here
#ifndef GLFER16H_NO_UNROLL
#define GLFER16H_NO_UNROLL 0     /* experiment builds: 1 = no form unrolls its frame loop over the register rotations */
#endif
#ifndef GLFER16H_AVG_UNROLL
#define GLFER16H_AVG_UNROLL 0    /* experiment builds: 1 = the AVG forms unroll like the others */
#endif
  constexpr bool UNROLL = (SHIFT == 4 || SHIFT == 8) && (AVG == 0 || GLFER16H_AVG_UNROLL != 0) && GLFER16H_NO_UNROLL == 0;   // 16/SHIFT copies of the frame loop's body (AVG: one copy, the pairs are moved --
                                                                    // four copies of the averaging block spill 120 VGPRs at two wavefronts per SIMD)
  auto prefetch_next = [&](long long rel, auto rotc) {
    constexpr int ROT = decltype(rotc)::value;             // the rotation of the frame in flight
    if constexpr (UNROLL) {
      load_pairs(rel, std::integral_constant<int, 16 - SHIFT>{}, std::integral_constant<int, (ROT + SHIFT) & 15>{}, kToPx);
    } else if constexpr (SHIFT > 0) {
#pragma unroll
      for (int m = 0; m < 16 - SHIFT; m++) px[m] = px[m + SHIFT];
      load_pairs(rel, std::integral_constant<int, 16 - SHIFT>{}, rotc, kToPx);
    } else {
      load_pairs(rel, std::integral_constant<int, 0>{}, rotc, kToPx);
    }
  };
  auto prefetch_x = [&](long long rel) {
    load_pairs(rel, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, kToPx);
    if constexpr (HIST != 0) {   // history_mode 1: sample j = 2*(t + T*m) + e is kept iff j >= R.  Zeroed in
      const int d = 2 * (int)t - p.R;                  // place (this waits for the loads; a rare mode)
      static_for<0, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const bool k0 = d >= -2 * T * m, k1 = d + 1 >= -2 * T * m;
        if constexpr (FMT == GLFER_FMT_F32) {
          px[m].x = k0 ? px[m].x : 0.0f;
          px[m].y = k1 ? px[m].y : 0.0f;
        } else if constexpr (FMT == GLFER_FMT_S16) {   // raw 0 is sample 0.0
          const unsigned raw = __float_as_uint(px[m].x);
          px[m].x = __uint_as_float((k0 ? raw & 0xffffu : 0u) | (k1 ? raw & 0xffff0000u : 0u));
        } else {                                       // raw 128 is sample 0.0
          const unsigned raw = __float_as_uint(px[m].x);
          px[m].x = __uint_as_float((k0 ? raw & 0xffu : 0x80u) | (k1 ? raw & 0xff00u : 0x8000u));
        }
      });
    }
  };
  // the pair as floats (integer formats: unscaled, see kSampleScale)
  auto sample_pair = [&](auto mc, auto rotc) -> v2f32 {
    constexpr int m = (decltype(mc)::value + decltype(rotc)::value) & 15;
    v2f32 x;
    if constexpr (FMT == GLFER_FMT_F32 || MTAB != 0) {   // (MTAB: absorbed pairs are floats in every format)
      x = px[m];
    } else if constexpr (FMT == GLFER_FMT_S16) {
      const int raw = (int)__float_as_uint(px[m].x);
      x = v2f32{(float)(short)(raw & 0xffff), (float)(raw >> 16)};
    } else {
      const unsigned raw = __float_as_uint(px[m].x);
      x = v2f32{(float)(raw & 0xffu) - 128.0f, (float)((raw >> 8) & 0xffu) - 128.0f};
    }
    return x;
  };

  // Each workgroup walks a CONTIGUOUS range of frames: with overlapped frames the samples a frame
  // shares with its predecessor were read by the same workgroup one iteration earlier (L1 hits),
  // whatever the other workgroups are doing.  (Neighbouring ranges sit on the same XCD:
  // xcd_block_index.)  Without SHIFT the FPB slots take adjacent frames each iteration (measured:
  // 2 % better than a contiguous part per slot); with SHIFT each slot needs consecutive frames.
  const long long per = ((long long)p.nframes + (long long)grid_w * FPB - 1) / ((long long)grid_w * FPB);   // frames per slot
  // (xcd_block_index() over the consumer part of the grid: workgroup w runs on XCD w mod 8 whatever the producers in front)
  const unsigned lblock = (grid_w & 7u) || (MTAB != 0 && (p.nprod & 7)) ? block_w : (block_w & 7u) * (grid_w >> 3) + (block_w >> 3);
  start = (long long)lblock * per * FPB;
  if (start >= p.nframes) return;
  if constexpr (MTAB != 0) {
    if (p.nprod > 0) {
      // this workgroup's hops: the newest hop of frame F is hop F, a frame reaches NH - 1 hops back.  Wait for their chunks (the
      // producers walk the stream in order, as the consumers are dispatched); every wavefront polls for itself, so that the
      // acquire that follows is its own.  The poll is bounded: a launch must end whatever happens to its producers.
      const long long left = p.nframes - start, span = per * FPB;
      const long long h_lo = p.frame0 + start - (NH - 1), h_hi = p.frame0 + start + (left < span ? left : span);   // [h_lo, h_hi)
      const long long c_lo = (h_lo - p.prod_hop0) / p.prod_chunk, c_hi = (h_hi - 1 - p.prod_hop0) / p.prod_chunk;
      // ONE wavefront polls (every poll is a load that goes past L2: two thousand pollers on a handful of counters starve the
      // producers they are waiting for -- measured: 0.4 TB/s of means), the others wait at the barrier and then acquire for themselves
      if (p.prod_front_frames > 0) {
        if (tid < 64) {
          const long long g_lo = (h_lo - p.prod_hop0) / 64, g_hi = (h_hi - 1 - p.prod_hop0) / 64;
          for (long long g = g_lo; g <= g_hi; g++) {
            bool ok = false;
            for (int spin = 0; spin < (1 << 16) && !ok; spin++) {
              ok = __hip_atomic_load(p.means_ready + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
              if (!ok) __builtin_amdgcn_s_sleep(32);
            }
            // a producer that never came (a launch must end whatever happens): the hops' means are poisoned, the rows come out NaN
            if (!ok && tid == 0)
              for (int k = 0; k < 64 && g * 64 + k < p.prod_nhops; k++) p.means_out[p.prod_hop0 + g * 64 + k] = __builtin_nanf("");
          }
        }
      } else
      if (tid < 64) {
        for (long long c = c_lo; c <= c_hi; c++) {
          const long long in_chunk = p.prod_nhops - c * p.prod_chunk < p.prod_chunk ? p.prod_nhops - c * p.prod_chunk : p.prod_chunk;
          const unsigned need = (unsigned)((in_chunk + 63) / 64);
          for (int spin = 0; spin < (1 << 21); spin++) {
            if (__hip_atomic_load(p.means_ready + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) break;
            __builtin_amdgcn_s_sleep(127);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the invalidate has completed before the barrier releases anyone)
      __syncthreads();
    }
  }
  auto rel_of = [&](long long i) { return CONSEC ? (long long)fl * per + i : i * FPB + (long long)fl; };
  const int lead = AVG != 0 ? p.avg_depth - 1 : 0;       // AVG: frames recomputed in front of a slot's range (never stored)
  long long it = -(long long)lead;                         // frames done by every slot
  prefetch_x(rel_of(it));
  if constexpr (PF2) {                                     // frame 1's new pairs: landing set 1
    load_pairs(rel_of(1), std::integral_constant<int, 16 - SHIFT>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
  }

  // ---- MEAN: sums over the hop held in registers [q KM, (q+1) KM) of the frame at rotation ROT
  float mu[MEAN ? NH : 1];                               // the frame's hop means, oldest first (unscaled sample units)
  auto hop_partial = [&](auto qc, auto rotc) -> float {  // this lane's share, then the wavefront's (T < 64: the frame's lanes')
    constexpr int q = decltype(qc)::value;
    float sm = 0.0f;
    static_for<q * KM, (q + 1) * KM>([&](auto mc) {
      const v2f32 x = sample_pair(mc, rotc);
      sm += x.x;
      sm += x.y;
    });
#pragma unroll
    for (int o = 1; o < (T < 64 ? T : 64); o <<= 1) sm += __shfl_xor(sm, o);
    return sm;
  };
  // the frame's total from its wavefronts' partials (call between two frame_sync: the writer side is publish())
  auto publish = [&](float sm, int q) {
    if constexpr (WPF > 1) {
      if ((t & 63u) == 0) mred[(fl * WPF + (t >> 6)) * NH + q] = sm;
    }
  };
  auto collect = [&](float sm, int q) -> float {
    if constexpr (WPF > 1) {
      float tot = mred[(fl * WPF) * NH + q];
#pragma unroll
      for (int w = 1; w < WPF; w++) tot += mred[(fl * WPF + w) * NH + q];
      return tot / (float)p.H;                           // fft.c:91
    } else {
      return sm / (float)p.H;
    }
  };
  // p.means (GLFER_SUBMEAN_EXACT): the hop means are GIVEN -- taken in the reference's own order by
  // hop_means_seq_kernel -- as means[global hop index] in sample units; here they are used in the units
  // the samples are held in (integer formats: raw, the power-of-two scale rides in the window: exact).
  // The newest hop of frame F of the stream is hop F.
  const bool mean_table = MTAB != 0 || (MEAN != 0 && p.means != nullptr);
  auto table_mean = [&](long long rel, int back) -> float {      // the mean of the hop `back` hops before frame rel's newest
    const long long last_rel = (long long)p.nframes - 1 - start;
    const long long F = p.frame0 + start + (rel < last_rel ? rel : last_rel);
    return p.means[F - back] * (1.0f / kSampleScale);
  };
  // MTAB: pairs FROM..15 of the frame at rotation ROT as floats (integer formats: unscaled) minus `mean`, in place
  auto absorb = [&](auto fromc, auto toc, auto rotc, float mean) {
    static_for<decltype(fromc)::value, decltype(toc)::value>([&](auto mc) {
      constexpr int q = (decltype(mc)::value + decltype(rotc)::value) & 15;
      v2f32 x;
      if constexpr (FMT == GLFER_FMT_F32) {
        x = px[q];
      } else if constexpr (FMT == GLFER_FMT_S16) {
        const int raw = (int)__float_as_uint(px[q].x);
        x = v2f32{(float)(short)(raw & 0xffff), (float)(raw >> 16)};
      } else {
        const unsigned raw = __float_as_uint(px[q].x);
        x = v2f32{(float)(raw & 0xffu) - 128.0f, (float)((raw >> 8) & 0xffu) - 128.0f};
      }
      {
#pragma clang fp contract(off)
        px[q] = v2f32{x.x - mean, x.y - mean};             // fft.c:93-95, rounded on its own
      }
    });
  };
  float mu_new = 0.0f;                                     // MTAB: the mean of the newest hop of the frame about to be formed
  if constexpr (MTAB != 0) {
    // the first frame of the slot: its older hops now, its newest hop at the top of the frame like every frame's
    static_for<0, NH - 1>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      absorb(std::integral_constant<int, q * KM>{}, std::integral_constant<int, (q + 1) * KM>{}, std::integral_constant<int, 0>{},
             table_mean(rel_of(it), NH - 1 - q));                         // (it = 0, or -lead with AVG: the slot's first frame)
    });
    mu_new = table_mean(rel_of(it), 0);
  } else if constexpr (MEAN != 0) {
    if (mean_table) {
#pragma unroll
      for (int q = 0; q < NH; q++) mu[q] = table_mean(rel_of(0), NH - 1 - q);
    } else {
      float part[NH];
      static_for<0, NH>([&](auto qc) {
        part[decltype(qc)::value] = hop_partial(qc, std::integral_constant<int, 0>{});
        publish(part[decltype(qc)::value], decltype(qc)::value);
      });
      frame_sync<T>();
      static_for<0, NH>([&](auto qc) { mu[decltype(qc)::value] = collect(part[decltype(qc)::value], decltype(qc)::value); });
      frame_sync<T>();                                     // mred is free again
    }
  }

  constexpr int RL = C::radix(NPASS - 1), BL = 16 / RL;
  // register holding bin t + T*m after the last pass
  auto rho_of = [](int m) constexpr { return (m % BL) + BL * brev(m / BL, RL); };

  const int ntap = MT ? p.htapers : 1;
  // AVG: the PSD of this lane's 17 bins in the last three frames (ao1 = the previous frame), and the window's divisor
  struct NoDivisor {
    __device__ __forceinline__ explicit NoDivisor(double) {}
    __device__ __forceinline__ double operator()(double a) const { return a; }
  };
  using WindowDivisor = std::conditional_t<AVG != 0, Divisor, NoDivisor>;
  const WindowDivisor by_depth(AVG != 0 ? (double)(p.avg_depth + 1) : 1.0);          // avg.c:138-139,155 with the window full
  float ao1[AVG ? 17 : 1], ao2[AVG ? 17 : 1], ao3[AVG ? 17 : 1];
  bool holds_minbin = false;                             // AVG: one of this lane's bins is the band's first (psd[minbin] starts the running maximum)
  if constexpr (AVG != 0) {
#pragma unroll
    for (int i = 0; i < 17; i++) ao1[i] = ao2[i] = ao3[i] = 0.0f;
    const int mb = p.avg_minbin;
    holds_minbin = mb < M / 2 ? (mb % T) == (int)t : (mb == M / 2 ? t == 0 : ((M - mb) % T) == (int)t);
  }
  auto frame_body = [&](auto rotc) -> bool {
    const bool has_next = it + 1 < per;
    float mu_next = 0.0f;
    if constexpr (MTAB != 0) {
      mu_next = table_mean(rel_of(has_next ? it + 1 : it), 0);   // wanted at the end of the frame
      absorb(std::integral_constant<int, 16 - KM>{}, std::integral_constant<int, 16>{}, rotc, mu_new);   // this frame's newest hop
    }
    if constexpr (PF2) {
      // this frame's new pairs were requested two frames ago (frame 1's: before the loop) into the landing
      // set of its parity; their px slots -- the previous frame's oldest pairs -- are free since that frame's
      // samples were formed
      constexpr int ROT = decltype(rotc)::value, PAR = (ROT / SHIFT) & 1;
      if (it > 0) {
#pragma unroll
        for (int i = 0; i < SHIFT; i++) px[(16 - SHIFT + i + ROT) & 15] = pb[PAR][i];
      }
    }
    float acc[MT ? 17 : 1];                            // MT: bins k = t + T*m (m < 8), M - k (8 + m), M/2 (16)
    if constexpr (MT != 0) {
#pragma unroll
      for (int i = 0; i < 17; i++) acc[i] = 0.0f;
    }
   for (int j = 0; j < ntap; j++) {
    const bool last = j == ntap - 1;
    float zr[16], zi[16];
    if constexpr (VAR == 1) load_window(j);
    v2f32 xs[16];
    static_for<0, 16>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      xs[m] = sample_pair(mc, rotc);
      if constexpr (MEAN != 0 && MTAB == 0) {          // fft.c:93-95 (the subtraction is rounded on its own: no contraction into the window product)
#pragma clang fp contract(off)
        xs[m].x = xs[m].x - mu[m / KM];
        xs[m].y = xs[m].y - mu[m / KM];
      }
    });
    if constexpr (VAR == 2) {
      // 16 ds_read_b64 with immediate offsets: left to the compiler they become ds_read2_b64 (half
      // the rate) behind one address add each
      v2f32 wv[16];
      lds_read16_strided<T>(wl + t, wv);
#pragma unroll
      for (int m = 0; m < 16; m++) {
        // the products are rounded on their own in every copy of the loop body (left free, some
        // copies fuse them into the first butterfly's adds and a frame's bits depend on its slot)
#pragma clang fp contract(off)
        zr[m] = xs[m].x * wv[m].x;
        zi[m] = xs[m].y * wv[m].y;
      }
    } else {
#pragma unroll
      for (int m = 0; m < 16; m++) {
        zr[m] = xs[m].x * wn[m].x;
        zi[m] = xs[m].y * wn[m].y;
      }
    }
    if constexpr (GLFER16H_PREFETCH_TOP != 0 && !(GLFER_H_ABL & 2)) {
      if constexpr (PF2) {
        constexpr int PAR = (decltype(rotc)::value / SHIFT) & 1;
        if (it + 2 < per) load_pairs(rel_of(it + 2), std::integral_constant<int, 16 - SHIFT>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, PAR + 1>{});
      } else
      if (has_next && last) prefetch_next(rel_of(it + 1), rotc);   // px is free as soon as xs is formed
    }

    auto pass_hook = [&] {
      if constexpr (GLFER_H_ABL & 2) {                 // timing ablation: no sample loads after the first frame
#pragma unroll
        for (int m = 0; m < 16; m++) px[m] = px[m] * 0.999f;
      } else if constexpr (GLFER16H_PREFETCH_TOP == 0) {
        if constexpr (PF2) {                               // frame it + 2's new pairs into this frame's (emptied) landing set
          constexpr int PAR = (decltype(rotc)::value / SHIFT) & 1;
          if (it + 2 < per) load_pairs(rel_of(it + 2), std::integral_constant<int, 16 - SHIFT>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, PAR + 1>{});
        } else
        if (has_next && last) prefetch_next(rel_of(it + 1), rotc);   // the frame's last use of px is behind us
      }
    };
    if constexpr (TW1R) stockham16_passes<LOGN - 1, NT>(zr, zi, xb, t, tw1reg, twr, twi, pass_hook);
    else stockham16_passes<LOGN - 1, NT>(zr, zi, xb, t, tw1row, twr, twi, pass_hook);

    // ---- mirror step: Z[k], k >= M/2, through LDS (entry u = k - M/2)
    frame_sync<T>();
    static_for<8, 16>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int r = rho_of(m);
      xb[t + T * (m - 8)] = v2f32{zr[r], zi[r]};
    });
    frame_sync<T>();
    {
      // rows go out through a buffer descriptor over this block's frames: one VGPR offset per
      // direction (bins k upwards, bins M-k downwards) plus scalar offsets, no address arithmetic,
      // and frame slots past the last frame fall outside num_records (their stores are dropped)
      const unsigned ROWB = (unsigned)p.pitch * 4u;        // bytes from row to row (cfg.psd_pitch)
      // a descriptor over the workgroup's rows: rows past the launch's last frame fall outside
      // num_records and their stores are dropped
      const long long left = p.nframes - start, span = per * FPB;
      const bool store_psd = AVG == 0 || (it >= 0 && p.psd != nullptr);     // AVG: the rows themselves only if asked, never the lead frames'
      // (AVG: rows not asked for, and the lead frames', are dropped the same way -- an empty descriptor -- not by a branch per store:
      // a branch in front of every put cut the mirror step into seventeen basic blocks whose LDS reads could not be issued together)
      const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
          p.psd + (size_t)start * (size_t)p.pitch, 0, store_psd ? (unsigned)((left > span ? span : left) * (long long)ROWB) : 0u, 0x00020000);
      // row = the slot's part + the iteration's part.  The iteration's part is kept out of the optimizer's sight (a scalar it must
      // recompute every frame): left visible, the unrolled register-reuse loops kept `lane part + r * ROWB` of each of their copies in
      // registers of their own -- six VGPRs spilled in the C2 kernel (three wavefronts per SIMD: 168), and their reloads, counted
      // by vmcnt like the sample prefetches issued in front of them, made every frame wait for loads that were meant to have two
      // frames' time.  (The row offset stays in the VECTOR offset: the descriptor's range check, which drops the stores of frame
      // slots past the last frame, does not cover a scalar offset.)
      unsigned urow = (unsigned)it * ROWB * (unsigned)(CONSEC ? 1 : FPB);
      float pv[AVG ? 17 : 1];                                              // AVG: the frame's PSD at this lane's bins: t + T m, M - (t + T m), M/2
#if GLFER16H_OPAQUE_ROW
      asm volatile("" : "+s"(urow));
#endif
      const unsigned row = (unsigned)rel_of(0) * ROWB + urow;
      const unsigned vup = row + t * 4u, vdown = row + (unsigned)(M - 7 * T - (int)t) * 4u;
      // STAGE: the row's floats into the exchange buffer's free upper part (the mirror step uses entries
      // 0 .. M/2), at the offset s16 they have inside their 64-byte granule in memory
      constexpr bool STAGE = GLFER16H_STAGE_ROWS != 0 && T >= 64 && (PADM - M / 2 - 8) * 2 >= M + 1 + 16 + 4 && !(GLFER_H_ABL & 1);
      float *stg = reinterpret_cast<float *>(xb + (M / 2 + 8));
      const unsigned long long gfl = (unsigned long long)(start + rel_of(it)) * (unsigned long long)p.pitch +
                                     (unsigned long long)(reinterpret_cast<__SIZE_TYPE__>(p.psd) >> 2);
      const unsigned s16 = (unsigned)gfl & 15u;          // (uniform over the frame's wavefronts)
      float *sup = stg + (s16 + t), *sdown = stg + (s16 + (unsigned)(M - 7 * T - (int)t));   // the staged bins t and M - 7T - t
      auto put = [&](float v, unsigned voff, unsigned soff) {
        if constexpr (GLFER_H_ABL & 1) {               // timing ablation: arithmetic kept live, no store traffic
          if (v == 1.2345e-30f) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), orsrc, voff, soff, GLFER16H_STORE_AUX);
        } else if constexpr (STAGE) {
          (voff == vup ? sup : sdown)[soff >> 2] = v;  // (every call passes vup or vdown: folded at compile time)
        } else {
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), orsrc, voff, soff, GLFER16H_STORE_AUX);
        }
      };
      // the staged row out: after the barrier that follows the last put
      auto flush_row = [&] {
        if constexpr (STAGE) {
          frame_sync<T>();
          typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
          const unsigned head = (4u - (s16 & 3u)) & 3u;             // floats in front of the first whole 16-byte chunk
          const unsigned full = ((unsigned)(M + 1) - head) >> 2;   // whole chunks: 4 T or 4 T - 1 ... (M + 1 = 16 T + 1 floats)
          const unsigned tail = ((unsigned)(M + 1) - head) & 3u;
          const v4f32 *chunks = reinterpret_cast<const v4f32 *>(stg + (s16 + head));   // 16-byte aligned: s16 + head = 0 mod 4
          static_assert(M + 1 == 16 * T + 1, "a row is 4 T whole chunks, or 4 T - 1 and ragged ends");
          static_for<0, 4>([&](auto jc) {                          // chunk c = t + T j; the last pass may have one chunk too many
            constexpr int j = decltype(jc)::value;
            const unsigned c = t + (unsigned)(T * j);
            const v4f32 q = chunks[c];
            const unsigned off = row + (head + 4u * c) * 4u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, q), orsrc, (j < 3 || c < full) ? off : 0x80000000u, 0, GLFER16H_STORE_AUX);
          });
          if (t < 8u) {                                            // the frame's first wavefront: the row's ragged ends, float by float
            if (t < head) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(stg[s16 + t]), orsrc, row + t * 4u, 0, GLFER16H_STORE_AUX);
            const unsigned i = head + 4u * full + (t - 4u);
            if (t >= 4u && t - 4u < tail) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(stg[s16 + i]), orsrc, row + i * 4u, 0, GLFER16H_STORE_AUX);
          }
        }
      };
      static_for<0, 8>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int r = rho_of(m);
        const int k = (int)t + T * m;
        v2f32 b = xb[M / 2 - k];                       // Z[M-k]; entry M/2 (t = 0, m = 0) is never written
        const float ar = zr[r], ai = zi[r];
        if constexpr (m == 0) {
          if (t == 0) b = v2f32{ar, ai};               // k = 0 pairs with itself: bins 0 and N/2
        }
        const float er = ar + b.x, ei = ai - b.y, orr = ar - b.x, oi = ai + b.y;
        constexpr cplx64 u = unit_root(m, 32);         // W_N^(T*m) = exp(-i 2 pi m/32) = (u.c, -u.s)
        constexpr float cm = (float)u.c, sm = (float)u.s;
        const float c = m == 0 ? rot.x : __builtin_fmaf(rot.x, cm, -rot.y * sm);
        const float s = m == 0 ? rot.y : __builtin_fmaf(rot.y, cm, rot.x * sm);
        // P = -i * (c - i s) * O = (-s*Or + c*Oi) + i(-s*Oi - c*Or)
        const float pr = __builtin_fmaf(c, oi, -s * orr);
        const float pi = -__builtin_fmaf(c, orr, s * oi);
        const float x1r = er + pr, x1i = ei + pi, x2r = er - pr, x2i = ei - pi;
        if constexpr (MT != 0) {
          acc[m] = __builtin_fmaf(x1r, x1r, __builtin_fmaf(x1i, x1i, acc[m]));
          acc[8 + m] = __builtin_fmaf(x2r, x2r, __builtin_fmaf(x2i, x2i, acc[8 + m]));
          if (last) {
            put(acc[m], vup, (unsigned)(T * m) * 4u);
            put(acc[8 + m], vdown, (unsigned)(T * (7 - m)) * 4u);
          }
        } else {
          const float q1 = __builtin_fmaf(x1r, x1r, x1i * x1i), q2 = __builtin_fmaf(x2r, x2r, x2i * x2i);
          if constexpr (AVG != 0) {
            pv[m] = q1;
            pv[8 + m] = q2;
          }
          put(q1, vup, (unsigned)(T * m) * 4u);              // bin k
          put(q2, vdown, (unsigned)(T * (7 - m)) * 4u);     // bin M - k
        }
      });
      {                                                // k = M/2 pairs with itself: X = conj(Z) (lane 0's value is the bin)
        constexpr int r = rho_of(8);
        const float nyq = 4.0f * __builtin_fmaf(zr[r], zr[r], zi[r] * zi[r]);
        if constexpr (MT != 0) {
          acc[16] += nyq;
          if (t == 0 && last) put(acc[16], vup, (unsigned)(M / 2) * 4u);
        } else {
          if constexpr (AVG != 0) pv[16] = nyq;
          if (t == 0) put(nyq, vup, (unsigned)(M / 2) * 4u);
        }
      }
      if (MT == 0 || last) flush_row();
      if constexpr (AVG != 0) {
        // ---- update_avg_plain (avg.c:108-159) on the values just formed.  Bin of value i: see pv.  One value at a time -- sum, quotient,
        // store, band statistics -- so that nothing but the rings lives across the block, and WITHOUT A BRANCH per value (a first
        // form tested `frame to be stored?` / `return values wanted?` per value: seventeen basic blocks whose double-precision
        // chains could not overlap, 330 scalar instructions a frame): a lead frame's stores fall outside its empty descriptor, lane
        // != 0's copy of bin M/2 goes to an out-of-range offset, the band statistics are selects.
        // The window's sum is ((p[f-3] + p[f-2]) + p[f-1]) + p[f] in double, oldest row first -- the same additions in the same order
        // as avg_fused_kernel's ring form (aux_kernels.hip) makes for depth <= 4, so the two give the same doubles whether or not
        // the additions are exact.  A ring slot beyond the window is kept at +0.0 (adding +0.0 is exact): one copy of the block for
        // every window length -- a copy per length, or the frame loop's four copies, spill ~120 VGPRs at two wavefronts per SIMD.
        const int minbin = p.avg_minbin, maxbin = p.avg_maxbin, n_out = p.avg_nout;
        const bool out_now = it >= 0;                                                  // (the same in every lane of the workgroup)
        const int par = (int)(it & 1);
        typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
        const unsigned AROWB = (unsigned)n_out * 8u;
        const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
            p.avg + (size_t)start * (size_t)n_out, 0, out_now ? (unsigned)((left > span ? span : left) * (long long)AROWB) : 0u, 0x00020000);
        unsigned uarow = (unsigned)it * AROWB;                                         // (the iteration's part, a scalar kept out of the optimizer's sight: see urow)
#if GLFER16H_OPAQUE_ROW
        asm volatile("" : "+s"(uarow));
#endif
        const unsigned arow = (unsigned)rel_of(0) * AROWB + uarow;
        const unsigned aup = arow + t * 8u, adown = arow + (unsigned)(M - 7 * T - (int)t) * 8u;
        const unsigned amid = t == 0 ? arow : 0x80000000u;                              // bin M/2 is lane 0's
        const bool d2 = p.avg_depth >= 2, d3 = p.avg_depth >= 3, d4 = p.avg_depth >= 4;
        const double dv = (double)(p.avg_depth + 1), dy = by_depth.y;                  // avg.c:138-139,155: the divisor with the window full, and RN(1 / it)
        double sm = 0.0, mx = -1.0e300;
        float initv = 0.0f;
        int mi = 0x7fffffff;
        auto one = [&](auto ic, int b, bool mine, unsigned voff, unsigned soff) {
          constexpr int i = decltype(ic)::value;
#if GLFER16H_AVG_ABL & 2
          const double c = (double)(pv[i] + ao1[i]);
          ao1[i] = pv[i] + ao2[i];
          const bool in = (b >= minbin) & (b < maxbin) & mine;
          const double q = c;
#else
          const double c = (((double)ao3[i] + (double)ao2[i]) + (double)ao1[i]) + (double)pv[i];   // avgdata->cum[index], avg.c:116-127
          ao3[i] = d4 ? ao2[i] : 0.0f;
          ao2[i] = d3 ? ao1[i] : 0.0f;
          ao1[i] = d2 ? pv[i] : 0.0f;
          const bool in = (b >= minbin) & (b < maxbin) & mine;
          // c / (depth + 1), correctly rounded (div_exact.hpp; the divisor is 2 .. 5: never the slow path)
          const double q0 = c * dy;
          const double q = __builtin_fma(__builtin_fma(-dv, q0, c), dy, q0);
#endif
          const double val = in ? q : 1e-15;                                          // avg.c:150-155
#if GLFER16H_AVG_ABL & 1
          if (val == 1.2345e-300) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32, val), arsrc, voff, soff, GLFER16H_AVG_STORE_AUX);
#else
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32, val), arsrc, voff, soff, GLFER16H_AVG_STORE_AUX);
#endif
#if GLFER16H_AVG_ABL & 4
          initv += (float)val;
          return;
#endif
          // the band's sum and its maximum with the LOWEST bin among equals (avg.c:129-135 walks the bins upwards with a strict >):
          // a lane takes its own bins in ascending order, so a strict > keeps the lowest
          sm += in ? c : 0.0;
          const bool gt = in & (c > mx);
          mx = gt ? c : mx;
          mi = gt ? b : mi;
          initv = ((b == minbin) & mine) ? pv[i] : initv;                             // the running maximum starts at psd[minbin], avg.c:111
        };
        static_for<0, 8>([&](auto mc) {                                               // bins t + T m, upwards
          constexpr int m = decltype(mc)::value;
          one(std::integral_constant<int, m>{}, (int)t + T * m, true, aup, (unsigned)(T * m) * 8u);
        });
        one(std::integral_constant<int, 16>{}, M / 2, t == 0, amid, (unsigned)(M / 2) * 8u);
        static_for<0, 8>([&](auto mc) {                                               // bins M - (t + T m), m = 7 .. 0: upwards too
          constexpr int m = 7 - decltype(mc)::value;
          one(std::integral_constant<int, 8 + m>{}, M - ((int)t + T * m), true, adown, (unsigned)(T * (7 - m)) * 8u);
        });
        if (out_now) {
          for (int b = M + 1 + (int)t; b < n_out; b += T)                               // avgdata->avg is N wide (source.c:312): the columns past the last bin
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32, 1e-15), arsrc, arow + (unsigned)b * 8u, 0, GLFER16H_AVG_STORE_AUX);
          if (holds_minbin) a_init[par * FPB + (int)fl] = (double)initv;
          if constexpr (T >= 64) {
            wave_sum_max(sm, mx, mi);                                                  // DPP: no LDS round trips
          } else {
#pragma unroll
            for (int o = 1; o < T; o <<= 1) {
              const double os = __shfl_xor(sm, o), om = __shfl_xor(mx, o);
              const int oi = __shfl_xor(mi, o);
              sm += os;
              if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
            }
          }
          if constexpr (WPF > 1) {
            if ((t & 63u) == 0) {
              const int slot = (par * FPB + (int)fl) * WPF + (int)(t >> 6);
              a_sum[slot] = sm;
              a_max[slot] = mx;
              a_idx[slot] = mi;
            }
          }
          frame_sync<T>();
          if (t == 0) {
            if constexpr (WPF > 1) {
              const int base = (par * FPB + (int)fl) * WPF;
              sm = a_sum[base];
              mx = a_max[base];
              mi = a_idx[base];
#pragma unroll
              for (int w = 1; w < WPF; w++) {
                sm += a_sum[base + w];
                if (a_max[base + w] > mx || (a_max[base + w] == mx && a_idx[base + w] < mi)) { mx = a_max[base + w]; mi = a_idx[base + w]; }
              }
            }
            const long long frel = start + rel_of(it);
            if (frel < p.nframes) {
              const double init = a_init[par * FPB + (int)fl];
              double top = init;
              int peak = -1;
              if (mx > init) { top = mx; peak = mi; }
              double *o = p.avg_ret + (size_t)frel * 4;
              o[0] = (sm - top) / ((double)(maxbin - minbin - 1) * dv);               // avg.c:147 (effdepth = depth: the window is full)
              o[1] = (double)peak;
              o[2] = 0.0;
              o[3] = (double)p.avg_depth;
            }
          }
        }
      }
    }
    float next_part = 0.0f;
    if constexpr (MEAN != 0) {
      // the next frame's newest hop is in px by now (requested during this frame's passes): its sum
      // crosses the frame's wavefronts over the barrier that ends the iteration
      if (MTAB == 0 && has_next && last && !mean_table) {           // (the multitaper form: after the frame's last taper)
        constexpr int ROTN = (SHIFT == 4 || SHIFT == 8) ? ((decltype(rotc)::value + SHIFT) & 15) : 0;
        next_part = hop_partial(std::integral_constant<int, NH - 1>{}, std::integral_constant<int, ROTN>{});
        publish(next_part, NH - 1);
      }
    }
    if constexpr (GLFER16_BARRIER_AFTER_READS != 0) frame_sync<T>();     // mirror entries read: buffer free
    if constexpr (MEAN != 0) {
      if (has_next && last) {
        if constexpr (MTAB != 0) {
          mu_new = mu_next;
        } else {
          const float mn = mean_table ? table_mean(rel_of(it + 1), 0) : collect(next_part, NH - 1);
#pragma unroll
          for (int h = 0; h + 1 < NH; h++) mu[h] = mu[h + 1];
          mu[NH - 1] = mn;
        }
      }
    }
   }
    it++;
    return has_next;
  };
  if constexpr (SHIFT == 4 && UNROLL) {
    while (frame_body(std::integral_constant<int, 0>{}) && frame_body(std::integral_constant<int, 4>{}) &&
           frame_body(std::integral_constant<int, 8>{}) && frame_body(std::integral_constant<int, 12>{})) {}
  } else if constexpr (SHIFT == 8 && UNROLL) {
    while (frame_body(std::integral_constant<int, 0>{}) && frame_body(std::integral_constant<int, 8>{})) {}
  } else {
    while (frame_body(std::integral_constant<int, 0>{})) {}
  }
  if constexpr (MTAB != 0) {
    // the lock-stepped fused launch: this front's producers may move on (one count per finished consumer workgroup)
    if (p.nprod > 0 && p.prod_front_frames > 0 && tid == 0)
      __hip_atomic_fetch_add(p.front_done + (block_w & 7u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace glfer

#ifndef GLFER_NO_LAUNCHERS
using namespace glfer;

#ifndef GLFER_LOGN
#error "compile with -DGLFER_LOGN=<log2 of the block size>"
#endif
#define GLFER_CAT2(a, b) a##b
#define GLFER_CAT(a, b) GLFER_CAT2(a, b)

template <int FMT>
static hipError_t launch16h_fmt(const SpectroParams &p, hipStream_t st) {
  constexpr int L = GLFER_LOGN;
  using LC = LaunchH<L>;
  const long long work = ((long long)p.nframes + LC::FPB - 1) / LC::FPB;
  if (work == 0) return hipSuccess;
  const long long per_cu = (GLFER16H_WAVES_PER_SIMD * 256) / LC::BLOCK > 0 ? (GLFER16H_WAVES_PER_SIMD * 256) / LC::BLOCK : 1;
  const long long resident = 256LL * per_cu;
  unsigned grid = (unsigned)(work < 8 * resident ? work : 8 * resident);   // tools/hbench: 8x beats 4x by ~2 % with contiguous ranges
  if (grid >= 64) grid &= ~7u;                     // whole XCD slices: see xcd_block_index()
#if GLFER_LOGN >= 13
  if (p.htapers > 1) {
    if (p.mean_inkernel) {                          // hop = 16, 8 or 4 of a lane's register pairs (overlap 0, 50, 75 %): the hops' means before the frame's first taper
      const int km = (16 * p.H) % (1 << L) == 0 ? 16 * p.H / (1 << L) : 0;
      if (p.history_mode) return hipErrorInvalidValue;
      // (two wavefronts per SIMD: at three the hop means push the multitaper loop into spills, 152 B of scratch per lane)
      constexpr int W2 = GLFER16H_WAVES_PER_SIMD > 2 ? 2 : GLFER16H_WAVES_PER_SIMD;
      if (km == 16) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, W2, 1, 1, 0, 0, 1>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
      else if (km == 8) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, W2, 1, 1, 0, 0, 8>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
      else if (km == 4) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, W2, 1, 1, 0, 0, 4>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
      else return hipErrorInvalidValue;
      return hipGetLastError();
    }
    if (p.history_mode) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, GLFER16H_WAVES_PER_SIMD, 1, 1, 1>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
    else hipLaunchKernelGGL((spectro16h_kernel<L, FMT, GLFER16H_WAVES_PER_SIMD, 1, 1, 0>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
    return hipGetLastError();
  }
#endif
  if (p.htapers > 1) return hipErrorInvalidValue;      // the multitaper form is built for N >= 8192 only
  if (p.avg) {
#if GLFER16H_AVG_BUILDS
    // update_avg_plain inside the kernel: two wavefronts per SIMD, every slot walks consecutive frames and recomputes the
    // depth-1 frames in front of its range -- so a slot gets >= 32 frames where the launch is long enough (3 lead frames: < 10 %)
    if (p.history_mode || (p.mean_inkernel && !p.means) || p.nprod || p.avg_depth < 1 || p.avg_depth > 4 || p.avg_nout < (1 << (L - 1)) + 1) return hipErrorInvalidValue;
    const long long per_cu2 = (2 * 256) / LC::BLOCK > 0 ? (2 * 256) / LC::BLOCK : 1;
    static const long long mult = [] { const char *e = getenv("GLFER_AVG_GRID_MULT"); const long v = e && *e ? atol(e) : 4; return (long long)(v < 1 ? 1 : v); }();
    const long long cap = mult * 256LL * per_cu2, want = work / 32 > 0 ? work / 32 : 1;
    unsigned ga = (unsigned)(want < cap ? want : cap);
    if (ga >= 64) ga &= ~7u;
    const int k16 = (16 * p.H) % (1 << L) == 0 ? (16 * p.H) >> L : 0;
    if (p.mean_inkernel) {
      // the reference's default (sub_mean = opt.autoscale) with the means GIVEN (taken in its own order by hop_means_seq_kernel): the
      // table form's in-place correction in front of the same averaging block
      if (k16 == 2) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, 2, GLFER16H_VAR, 0, 0, 2, 1, 1, 1>), dim3(ga), dim3(LC::BLOCK), 0, st, p);
      else if (k16 == 4) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, 2, GLFER16H_VAR, 0, 0, 4, 1, 1, 1>), dim3(ga), dim3(LC::BLOCK), 0, st, p);
      else if (k16 == 8) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, 2, GLFER16H_VAR, 0, 0, 8, 1, 1, 1>), dim3(ga), dim3(LC::BLOCK), 0, st, p);
      else if (k16 == 16) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, 2, GLFER16H_VAR, 0, 0, 0, 1, 1, 1>), dim3(ga), dim3(LC::BLOCK), 0, st, p);
      else return hipErrorInvalidValue;
      return hipGetLastError();
    }
    if (k16 == 2) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, 2, GLFER16H_VAR, 0, 0, 2, 0, 0, 1>), dim3(ga), dim3(LC::BLOCK), 0, st, p);
    else if (k16 == 4) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, 2, GLFER16H_VAR, 0, 0, 4, 0, 0, 1>), dim3(ga), dim3(LC::BLOCK), 0, st, p);
    else if (k16 == 8) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, 2, GLFER16H_VAR, 0, 0, 8, 0, 0, 1>), dim3(ga), dim3(LC::BLOCK), 0, st, p);
    else hipLaunchKernelGGL((spectro16h_kernel<L, FMT, 2, GLFER16H_VAR, 0, 0, 0, 0, 0, 1>), dim3(ga), dim3(LC::BLOCK), 0, st, p);   // any other hop: every frame loaded whole
    return hipGetLastError();
#else
    return hipErrorInvalidValue;
#endif
  }
#ifdef GLFER16H_AVG_ONLY
  return hipErrorInvalidValue;                         // (experiment builds, tools/build_variant.sh: only the AVG forms are compiled)
#else
  if (p.history_mode) {
    if (p.mean_inkernel) return hipErrorInvalidValue;
    hipLaunchKernelGGL((spectro16h_kernel<L, FMT, GLFER16H_WAVES_PER_SIMD, GLFER16H_VAR, 0, 1>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
    return hipGetLastError();
  }
#if GLFER16H_SHIFT_BUILDS
  if (p.mean_inkernel) {
    // mean removal inside the kernel: the hop must be 2, 4, 8 or all 16 of a lane's sample registers
    // (overlap 87.5 / 75 / 50 / 0 %); a slot walks consecutive frames whatever the launch's length
    // (two wavefronts per SIMD: at three the hop means and their sums push the loop into spills, and the
    // periodogram runs as fast at two -- profiles/r02_register_reuse_ab.txt)
    constexpr int kMeanWps = GLFER16H_WAVES_PER_SIMD > 2 ? 2 : GLFER16H_WAVES_PER_SIMD;
    const int k16 = (16 * p.H) % (1 << L) == 0 ? (16 * p.H) >> L : 0;
    unsigned g = (unsigned)(work / 4 < 8 * resident ? (work / 4 ? work / 4 : 1) : 8 * resident);
    if (g >= 64) g &= ~7u;
    if (p.means) {                                   // the means are given: the table form, at the plain form's occupancy
      constexpr int W = GLFER16H_WAVES_PER_SIMD;
      if (p.nprod > 0 && p.prod_front_frames < 0) {
        // the lock-stepped fused launch (round 5): short consumer ranges (prod_block_frames asks for so many frames per workgroup) walked
        // front by front, the producers throttled to stay inside the Infinity Cache
        if ((p.nprod & 7) || p.prod_block_frames < 1) return hipErrorInvalidValue;
        SpectroParams q = p;
        long long gc = ((long long)p.nframes + p.prod_block_frames - 1) / p.prod_block_frames;
        gc = (gc + 7) / 8 * 8;
        const long long per = ((long long)p.nframes + gc * LC::FPB - 1) / (gc * LC::FPB);
        q.prod_block_frames = (int)(per * LC::FPB);
        q.prod_front_frames = (gc / 8) * per * LC::FPB;
        q.prod_look = 96 * q.prod_block_frames + p.prod_look;       // what a front's resident consumers span (3 workgroups x 32 CUs an XCD) + the margin asked for
        const unsigned gl = (unsigned)(gc + p.nprod);
        constexpr int WL = GLFER16H_WAVES_PER_SIMD;
        if (k16 == 16) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, WL, GLFER16H_VAR, 0, 0, 0, 1, 1>), dim3(gl), dim3(LC::BLOCK), 0, st, q);
        else if (k16 == 2) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, WL, GLFER16H_VAR, 0, 0, 2, 1, 1>), dim3(gl), dim3(LC::BLOCK), 0, st, q);
        else if (k16 == 4) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, WL, GLFER16H_VAR, 0, 0, 4, 1, 1>), dim3(gl), dim3(LC::BLOCK), 0, st, q);
        else if (k16 == 8) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, WL, GLFER16H_VAR, 0, 0, 8, 1, 1>), dim3(gl), dim3(LC::BLOCK), 0, st, q);
        else return hipErrorInvalidValue;
        return hipGetLastError();
      }
      if (p.nprod > 0) {                             // the fused launch: producers in front of the grid as it would have been
        if (p.nprod & 7) return hipErrorInvalidValue;
        grid += (unsigned)p.nprod;
        g += (unsigned)p.nprod;
      }
      // GLFER_MTAB_WPS=2 (experiment, profiles/r04_piecewise_means.txt): two wavefronts per SIMD, so that a hop-means launch of the
      // NEXT piece (side stream) finds registers and LDS beside this one
      static const bool two = [] { const char *e = getenv("GLFER_MTAB_WPS"); return e && *e == '2'; }();
      if (two && W > 2) {
        if (k16 == 16) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, 2, GLFER16H_VAR, 0, 0, 0, 1, 1>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
        else if (k16 == 2) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, 2, GLFER16H_VAR, 0, 0, 2, 1, 1>), dim3(g), dim3(LC::BLOCK), 0, st, p);
        else if (k16 == 4) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, 2, GLFER16H_VAR, 0, 0, 4, 1, 1>), dim3(g), dim3(LC::BLOCK), 0, st, p);
        else if (k16 == 8) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, 2, GLFER16H_VAR, 0, 0, 8, 1, 1>), dim3(g), dim3(LC::BLOCK), 0, st, p);
        else return hipErrorInvalidValue;
        return hipGetLastError();
      }
      if (k16 == 16) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, W, GLFER16H_VAR, 0, 0, 0, 1, 1>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
      else if (k16 == 2) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, W, GLFER16H_VAR, 0, 0, 2, 1, 1>), dim3(g), dim3(LC::BLOCK), 0, st, p);
      else if (k16 == 4) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, W, GLFER16H_VAR, 0, 0, 4, 1, 1>), dim3(g), dim3(LC::BLOCK), 0, st, p);
      else if (k16 == 8) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, W, GLFER16H_VAR, 0, 0, 8, 1, 1>), dim3(g), dim3(LC::BLOCK), 0, st, p);
      else return hipErrorInvalidValue;
      return hipGetLastError();
    }
    if (k16 == 16) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, kMeanWps, GLFER16H_VAR, 0, 0, 0, 1>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
    else if (k16 == 2) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, kMeanWps, GLFER16H_VAR, 0, 0, 2, 1>), dim3(g), dim3(LC::BLOCK), 0, st, p);
    else if (k16 == 4) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, kMeanWps, GLFER16H_VAR, 0, 0, 4, 1>), dim3(g), dim3(LC::BLOCK), 0, st, p);
    else if (k16 == 8) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, kMeanWps, GLFER16H_VAR, 0, 0, 8, 1>), dim3(g), dim3(LC::BLOCK), 0, st, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
#else
  if (p.mean_inkernel) return hipErrorInvalidValue;
#endif
#if GLFER16H_SHIFT_BUILDS
  // overlap 87.5 / 75 / 50 %: the hop is 2 / 4 / 8 of a lane's 16 sample registers -- a slot that
  // walks consecutive frames keeps the rest.  Worth it when a slot gets >= 4 frames and the
  // launch still fills the chip.
  const int shift = (16 * p.H) % (1 << L) == 0 ? (16 * p.H) >> L : 0;
  if ((shift == 2 || shift == 4 || shift == 8) && work >= 4 * resident) {
    unsigned g = (unsigned)(work / 4 < 8 * resident ? work / 4 : 8 * resident) & ~7u;
    if (shift == 2) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, GLFER16H_WAVES_PER_SIMD, GLFER16H_VAR, 0, 0, 2>), dim3(g), dim3(LC::BLOCK), 0, st, p);
    else if (shift == 4) hipLaunchKernelGGL((spectro16h_kernel<L, FMT, GLFER16H_WAVES_PER_SIMD, GLFER16H_VAR, 0, 0, 4>), dim3(g), dim3(LC::BLOCK), 0, st, p);
    else hipLaunchKernelGGL((spectro16h_kernel<L, FMT, GLFER16H_WAVES_PER_SIMD, GLFER16H_VAR, 0, 0, 8>), dim3(g), dim3(LC::BLOCK), 0, st, p);
    return hipGetLastError();
  }
#endif
  hipLaunchKernelGGL((spectro16h_kernel<L, FMT, GLFER16H_WAVES_PER_SIMD, GLFER16H_VAR, 0, 0>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
  return hipGetLastError();
#endif
}

// the real-input form of the single-taper path; needs p->htaps/htw/hrot (glfer_hip.cpp builds them)
extern "C" hipError_t GLFER_CAT(glfer_launch_spectro16h_n, GLFER_LOGN)(const SpectroParams *p, hipStream_t st) {
  if (!p->htaps || !p->htw || !p->hrot || p->nonlin || p->spec) return hipErrorInvalidValue;
  // the gather has no zero-history path: every frame must lie wholly inside the stream
  if (p->frame0 * (long long)p->H < (long long)p->R) return hipErrorInvalidValue;
  // integer samples are fetched in pairs (y[2j], y[2j+1]) with one load: pairs must be naturally aligned
  if (p->fmt != GLFER_FMT_F32) {
    const unsigned pair = p->fmt == GLFER_FMT_S16 ? 4u : 2u;
    if ((p->H & 1) || (reinterpret_cast<uintptr_t>(p->stream) & (pair - 1u))) return hipErrorInvalidValue;
  }
  switch (p->fmt) {
    case GLFER_FMT_F32: return launch16h_fmt<GLFER_FMT_F32>(*p, st);
    case GLFER_FMT_S16: return launch16h_fmt<GLFER_FMT_S16>(*p, st);
    case GLFER_FMT_U8: return launch16h_fmt<GLFER_FMT_U8>(*p, st);
  }
  return hipErrorInvalidValue;
}
#endif  // GLFER_NO_LAUNCHERS
