// spectro16xl.hip -- spectro16x.hip's scheme (odd taper count: the last taper of two frames shares
// one transform) with the taper tables RESIDENT IN LDS.
//
// In spectro16x.hip every round fetches its taper pair from the L2-resident table: 32 KB per round
// through the CU's 64 B/clk vector-memory path, 80 KB per frame at T = 5, five times the frame's
// own samples, plus 32 prefetch VGPRs.  DPSS tapers are symmetric (even order) or antisymmetric
// (odd order) about the frame centre, v_k[N-1-n] = (-1)^k v_k[n], so HALF of each table is enough:
// lane t's samples t + T*m, m >= 8, are lane T-1-t's samples at 15-m.  (2*NP+1) half tables are
// (2*NP+1)*N*2 bytes: 40 KB for N = 4096, T = 5, next to the 35 KB exchange buffer -- two blocks
// per CU, 8 wavefronts, which measures the same as three (tools/xbench: the kernel is bound by
// VALU issue, not by latency hiding).  Reading them is 16 ds_read_b64 per round (LDS reads run at
// 256 B/clk).  With 256 VGPRs per lane both frame groups' samples and both partial PSDs stay in
// registers: nothing is re-read from memory and nothing but the exchange goes through LDS.
//
// The host builds the half tables only when the tapers it computed are (anti)symmetric to 1e-7 of
// their peak (they are, to rounding; the check guards the identity, not the algorithm).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "odd_taper.hpp"

#ifndef GLFER16XL_TW1_REGS
#define GLFER16XL_TW1_REGS 1
#endif

namespace glfer {
hipError_t allow_dynamic_lds(const void *kernel, size_t bytes);   // plan.h / glfer_hip.cpp: once per device, kernel and size class
}

namespace glfer {

template <int LOGN>
struct LaunchXL {
  using C = Plan16<LOGN>;
  static constexpr int N = C::N, T = C::T;
  static_assert(T <= 256, "one block of 256 lanes holds whole frames");
  static constexpr int FPB = 256 / T;
  static constexpr int BLOCK = 256;
  static constexpr int PADN = N + N / 16;
  static constexpr int WPF = T >= 64 ? T / 64 : 1;
  // v2f32 words: exchange, pass-1 twiddles, power partials, then the taper half tables
  static constexpr int FIXED_WORDS = FPB * PADN + 16 * 17 + (FPB * WPF + 1) / 2 + 1;
  static constexpr size_t lds_bytes(int nfull) { return ((size_t)FIXED_WORDS + (size_t)nfull * 8 * T + 4 * T) * 8; }
};

// KM > 0: per-hop mean removal inside the kernel (load_frame16_mean, odd_taper.hpp)
template <int LOGN, int FMT, int KM = 0>
__global__ __launch_bounds__(256, 2) void spectro16xl_kernel(SpectroParams p) {
  using C = Plan16<LOGN>;
  using L = LaunchXL<LOGN>;
  constexpr int N = C::N, T = C::T, NPASS = C::NPASS, FPB = L::FPB, PADN = L::PADN, WPF = L::WPF;
  constexpr int TW1 = 15;
  constexpr int NTWR = C::NTW - TW1;
  constexpr int NT = NTWR > 0 ? NTWR : 1;
  constexpr unsigned esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  extern __shared__ v2f32 lds[];

  const unsigned tid = threadIdx.x;
  const unsigned t = tid % T;
  const unsigned fl = tid / T;
  const int NP = p.npairs - 1;                       // full (two-taper) rounds per frame
  v2f32 *xb = lds + fl * PADN;
  v2f32 *tw1 = lds + FPB * PADN;
  float *red = reinterpret_cast<float *>(lds + FPB * PADN + 16 * 17);
  v2f32 *tl2 = lds + L::FIXED_WORDS;                 // [NP][8][T] (taper 2p, taper 2p+1) at sample t + T*m, m < 8
  float *tl1 = reinterpret_cast<float *>(tl2 + NP * 8 * T);   // [8][T] the last (even-order) taper

  {
    const v2f32 *tw = reinterpret_cast<const v2f32 *>(p.tw);
    const unsigned k = tid >> 4, q = tid & 15;
    tw1[k * 17 + q] = q ? tw[(q - 1) * T + k] : v2f32{1.0f, 0.0f};
    const v2f32 *g2 = reinterpret_cast<const v2f32 *>(p.ltaps);
    for (int i = tid; i < NP * 8 * T; i += 256) tl2[i] = g2[i];
    const float *g1 = p.ltaps + (size_t)NP * 8 * T * 2;
    for (int i = tid; i < 8 * T; i += 256) tl1[i] = g1[i];
  }
  float twr[NT], twi[NT];
  {
    const v2f32 *tw = reinterpret_cast<const v2f32 *>(p.tw) + t;
#pragma unroll
    for (int e = 0; e < NTWR; e++) {
      const v2f32 w = tw[(TW1 + e) * T];
      twr[e] = w.x;
      twi[e] = w.y;
    }
    if constexpr (NTWR == 0) twr[0] = twi[0] = 0.0f;
  }
  __syncthreads();
  // the lane's pass-1 twiddles: a row of the LDS table, or 32 registers (N >= 1024, where the kernel is at two wavefronts
  // per SIMD anyway; below, the registers would cost the third wavefront) -- profiles/r03_tw1_regs_other_kernels.txt
  constexpr bool TW1R = (GLFER16XL_TW1_REGS) != 0 && LOGN >= 10;
  Tw1Source<TW1R> tw1row;
  tw1row.init(tw1 + (t & 15) * 17);
  const long long stride = (long long)gridDim.x * (2 * FPB);

  auto load_x = [&](float (&dst)[16], long long fblk) {
    if constexpr (KM > 0) load_frame16_mean<FMT, T, KM>(p, t, fl, fblk, dst);
    else load_frame16<FMT, T>(p, t, fl, fblk, dst);
  };

  long long fblk = (long long)xcd_block_index() * (2 * FPB);
  if (fblk >= p.nframes) return;
  float xA[16], xB[16];
  load_x(xA, fblk);
  if (fblk + FPB < p.nframes) load_x(xB, fblk + FPB);
  else {
#pragma unroll
    for (int m = 0; m < 16; m++) xB[m] = 0.0f;
  }

  constexpr int RL = C::radix(NPASS - 1), BL = 16 / RL;
  auto rho_of = [](int m) constexpr { return (m % BL) + BL * brev(m / BL, RL); };   // register of bin t + T*m

  // One frame group's NP full rounds; leaves the mirror-folded sums psd[k] = acc[k] + acc[N-k]
  // (k = t + T*m, m < 8, and k = N/2 on lane 0) and the exponent of the shared round's scale.
  auto full_rounds = [&](const float (&x)[16], float (&psd)[8], float &nyq, int &hx) {
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    for (int pair = 0; pair < NP; pair++) {
      const v2f32 *wlo = tl2 + pair * 8 * T + t, *whi = tl2 + pair * 8 * T + (T - 1 - t);
      float zr[16], zi[16];
      GLFER_STAMP(0);                                // round start
      static_for<0, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        if constexpr (m < 8) {
          const v2f32 w = wlo[m * T];
          zr[m] = x[m] * w.x;
          zi[m] = x[m] * w.y;
        } else {                                     // sample N-1-n: even taper as is, odd taper negated
          const v2f32 w = whi[(15 - m) * T];
          zr[m] = x[m] * w.x;
          zi[m] = -(x[m] * w.y);
        }
      });
      stockham16_passes<LOGN, NT>(zr, zi, xb, t, tw1row, twr, twi, [] {});
#pragma unroll
      for (int r = 0; r < 16; r++)
        acc[r] = __builtin_fmaf(zr[r], zr[r], __builtin_fmaf(zi[r], zi[r], acc[r]));
      GLFER_STAMP(15);                               // round end (accumulated)
    }
    float e = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) e += acc[r];
    constexpr int RW = T < 64 ? T : 64;
#pragma unroll
    for (int w = 1; w < RW; w <<= 1) e += __shfl_xor(e, w);
    float *fold = reinterpret_cast<float *>(xb);
    frame_sync<T>();
    static_for<8, 16>([&](auto mc) {                 // upper half (entry k - N/2) through LDS
      constexpr int m = decltype(mc)::value;
      fold[t + T * (m - 8)] = acc[rho_of(m)];
    });
    if constexpr (T >= 64) {
      if ((t & 63) == 0) red[fl * WPF + (t >> 6)] = e;
    }
    frame_sync<T>();
    if constexpr (T >= 64) {
      e = 0.0f;
#pragma unroll
      for (int w = 0; w < WPF; w++) e += red[fl * WPF + w];
    }
    hx = scale_exponent(e);
    static_for<0, 8>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const int k = T * m + (int)t;
      float other = fold[N / 2 - k];                 // acc[N-k]; entry N/2 (k = 0) is never written
      if constexpr (m == 0) {
        if (t == 0) other = acc[rho_of(0)];
      }
      psd[m] = acc[rho_of(m)] + other;
    });
    nyq = 2.0f * acc[rho_of(8)];
    if constexpr (GLFER16_BARRIER_AFTER_READS != 0) frame_sync<T>();     // fold buffer read: free for the next writes
  };

  while (true) {                                     // one iteration: frame groups A and B
    const bool hasB = fblk + FPB < p.nframes;        // block-uniform
    const long long nfblk = fblk + stride;
    const bool has_next = nfblk < p.nframes;
    float psdA[8], psdB[8], nyqA = 0.0f, nyqB = 0.0f;
    int hxA = kSilent, hxB = kSilent;
#pragma unroll
    for (int m = 0; m < 8; m++) psdA[m] = psdB[m] = 0.0f;
    // one copy of the rounds in the instruction stream: the group is chosen by register moves
#pragma clang loop unroll(disable)
    for (int which = 0; which < (hasB ? 2 : 1); which++) {
      float xw[16], psd[8], nyq;
      int hx;
#pragma unroll
      for (int m = 0; m < 16; m++) xw[m] = which ? xB[m] : xA[m];
      full_rounds(xw, psd, nyq, hx);
      if (which == 0) {
#pragma unroll
        for (int m = 0; m < 8; m++) psdA[m] = psd[m];
        nyqA = nyq;
        hxA = hx;
      } else {
#pragma unroll
        for (int m = 0; m < 8; m++) psdB[m] = psd[m];
        nyqB = nyq;
        hxB = hx;
      }
    }

    // ---- shared round: re = sA*(xA*v), im = sB*(xB*v), v the last taper (even order: symmetric)
    float zr[16], zi[16];
    GLFER_STAMP(0);                                  // shared round start
    {
      const float sA = scale_in(hxA), sB = scale_in(hxB);
      const float *vlo = tl1 + t, *vhi = tl1 + (T - 1 - t);
      static_for<0, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const float v = m < 8 ? vlo[m * T] : vhi[(15 - m) * T];
        zr[m] = (xA[m] * v) * sA;
        zi[m] = (xB[m] * v) * sB;
      });
    }
    // the next iteration's samples go out after the first exchange's writes
    stockham16_passes<LOGN, NT>(zr, zi, xb, t, tw1row, twr, twi, [&] {
      if (has_next) {
        load_x(xA, nfblk);
        if (nfblk + FPB < p.nframes) load_x(xB, nfblk + FPB);
      }
    });
    separate_and_store<LOGN, FPB>(p, zr, zi, xb, t, fl, fblk, hxA, hxB,
                                  [&](auto mc) { if constexpr (decltype(mc)::value == 8) return nyqA; else return psdA[decltype(mc)::value]; },
                                  [&](auto mc) { if constexpr (decltype(mc)::value == 8) return nyqB; else return psdB[decltype(mc)::value]; });
    GLFER_STAMP(15);                                 // shared round end (separated, stored)
    if (!has_next) break;
    fblk = nfblk;
    if (fblk + FPB >= p.nframes) {
#pragma unroll
      for (int m = 0; m < 16; m++) xB[m] = 0.0f;
    }
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
static hipError_t launch16xl_fmt(const SpectroParams &p, hipStream_t st) {
  constexpr int L = GLFER_LOGN;
  using LC = LaunchXL<L>;
  const size_t shmem = LC::lds_bytes(p.npairs - 1);
  if (shmem > 80 * 1024) return hipErrorInvalidValue;            // two blocks per CU or not at all
  const long long work = ((long long)p.nframes + 2 * LC::FPB - 1) / (2 * LC::FPB);
  if (work == 0) return hipSuccess;
  const long long resident = 256LL * 2;
  unsigned grid = (unsigned)(work < 4 * resident ? work : 4 * resident);
  if (grid >= 64) grid &= ~7u;                       // whole XCD slices: see xcd_block_index()
  auto kern = spectro16xl_kernel<L, FMT>;
  if (p.mean_inkernel) {
    if constexpr (Plan16<L>::T <= 64) {
      if (p.history_mode) return hipErrorInvalidValue;
      const int km = (16 * p.H) % (1 << L) == 0 ? (16 * p.H) >> L : 0;
      if (km == 16) kern = spectro16xl_kernel<L, FMT, 16>;
      else if (km == 8) kern = spectro16xl_kernel<L, FMT, 8>;
      else if (km == 4) kern = spectro16xl_kernel<L, FMT, 4>;
      else return hipErrorInvalidValue;
    } else {
      return hipErrorInvalidValue;
    }
  }
  hipError_t e = glfer::allow_dynamic_lds(reinterpret_cast<const void *>(kern), shmem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, st, p);
  return hipGetLastError();
}

// odd taper counts >= 3 whose half tables fit in LDS; needs p->ltaps (glfer_hip.cpp builds it)
extern "C" hipError_t GLFER_CAT(glfer_launch_spectro16xl_n, GLFER_LOGN)(const SpectroParams *p, hipStream_t st) {
  if (!p->ltaps || p->npairs < 2 || p->nonlin || p->spec) return hipErrorInvalidValue;
  if (p->frame0 * (long long)p->H < (long long)p->R) return hipErrorInvalidValue;   // no zero-history path here
  switch (p->fmt) {
    case GLFER_FMT_F32: return launch16xl_fmt<GLFER_FMT_F32>(*p, st);
    case GLFER_FMT_S16: return launch16xl_fmt<GLFER_FMT_S16>(*p, st);
    case GLFER_FMT_U8: return launch16xl_fmt<GLFER_FMT_U8>(*p, st);
  }
  return hipErrorInvalidValue;
}
#endif  // GLFER_NO_LAUNCHERS
