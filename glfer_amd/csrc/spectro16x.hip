// spectro16x.hip -- multitaper frames with an ODD number of tapers (mtm_do with kmax even,
// mtm.c:189-220: kmax+1 tapers; BASELINE configs 3 and 4 have 5 and 9).
//
// spectro16.hip packs tapers 2p, 2p+1 of ONE frame as re/im of a complex N-point transform, so an
// odd taper count leaves the imaginary half of the last transform empty.  Here the last taper of
// TWO frames shares one transform: z = sA*(xA*v) + i*sB*(xB*v).  The two real spectra are separated
// bin pair by bin pair, E = Z[k] + conj(Z[N-k]) = 2*sA*Y_A[k], O = Z[k] - conj(Z[N-k]) = 2i*sB*Y_B[k],
// through one mirror exchange in LDS, and |E|^2/sA^2, |O|^2/sB^2 are added to the two frames' sums.
// T tapers cost T/2 transforms per frame instead of ceil(T/2): 2.5 instead of 3 for T = 5.
//
// sA, sB are powers of two chosen per frame so that both halves enter the shared transform at the
// same magnitude (the frame's power summed over the even number of tapers already done, halved
// exponent).  Without them the rounding error of the louder frame (~1e-7 of ITS peak) would leak
// into the quieter one; with them each frame sees the error level of a transform of its own data
// (x2 at most), and a power-of-two factor changes no mantissa bit.
//
// Per block iteration: frame group A (NP full rounds, fold -> partial PSD parked in LDS), frame
// group B (NP full rounds, fold -> partial PSD in 9 VGPRs), one shared round.  The loop nest is
// explicit (not a round-kind state machine) so that register liveness is what it looks like:
// acc is dead outside a group's full rounds, psdB outside group B's fold .. the shared round.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "odd_taper.hpp"

#ifndef GLFER16X_WAVES_PER_SIMD
#define GLFER16X_WAVES_PER_SIMD 3
#endif

#ifndef GLFER16X_TW1_REGS
#define GLFER16X_TW1_REGS 0
#endif

namespace glfer {

template <int LOGN>
struct LaunchX {
  using C = Plan16<LOGN>;
  static constexpr int N = C::N, T = C::T;
  static constexpr int FPB = T >= 256 ? 1 : 256 / T;
  static constexpr int BLOCK = T * FPB;
  static constexpr int PADN = N + N / 16;
  static constexpr int WPF = T >= 64 ? T / 64 : 1;                    // wavefronts per frame
  static constexpr int PART = N / 2 + 2;                              // floats per parked partial PSD (even)
  static constexpr int LDS_WORDS = FPB * PADN + 16 * 17 + 2 * FPB * PART / 2 + (FPB * WPF + 1) / 2;
};

// KM > 0: per-hop mean removal inside the kernel (load_frame16_mean, odd_taper.hpp)
template <int LOGN, int FMT, int WPS = GLFER16X_WAVES_PER_SIMD, int KM = 0>
__global__ __launch_bounds__(LaunchX<LOGN>::BLOCK, WPS) void spectro16x_kernel(SpectroParams p) {
  using C = Plan16<LOGN>;
  using L = LaunchX<LOGN>;
  constexpr int N = C::N, T = C::T, NPASS = C::NPASS, FPB = L::FPB, PADN = L::PADN, WPF = L::WPF;
  constexpr int TW1 = 15;
  constexpr int NTWR = C::NTW - TW1;
  constexpr int NT = NTWR > 0 ? NTWR : 1;
  constexpr unsigned esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  typedef float v4f32 __attribute__((ext_vector_type(4)));
  __shared__ v2f32 lds[L::LDS_WORDS];

  const unsigned tid = threadIdx.x;
  const unsigned t = tid % T;
  const unsigned fl = tid / T;
  v2f32 *xb = lds + fl * PADN;
  v2f32 *tw1 = lds + FPB * PADN;
  float *part = reinterpret_cast<float *>(lds + FPB * PADN + 16 * 17) + fl * L::PART;   // folded sums: group A, then (+FPB*PART) group B
  float *red = reinterpret_cast<float *>(lds + FPB * PADN + 16 * 17 + 2 * FPB * L::PART / 2);

  {
    const v2f32 *tw = reinterpret_cast<const v2f32 *>(p.tw);
    if (tid < 256) {
      const unsigned k = tid >> 4, q = tid & 15;
      tw1[k * 17 + q] = q ? tw[(q - 1) * T + k] : v2f32{1.0f, 0.0f};
    }
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
  Tw1Source<(GLFER16X_TW1_REGS) != 0> tw1row;                   // the lane's pass-1 twiddles: a row of the LDS table, or registers
  tw1row.init(tw1 + (t & 15) * 17);

  const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.taps), 0, p.npairs * 2 * N * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.xtaps), 0, N * 4, 0x00020000);
  const unsigned toff = t * 16u;
  const int NP = p.npairs - 1;                       // full (two-taper) rounds per frame; the last pair is the shared one
  const long long stride = (long long)gridDim.x * (2 * FPB);

  float px[16];          // samples of the frame group in work (A, then B)
  v2f32 pt[16];          // next full round: taper pair; shared round: pt[0..7] = the odd taper, pt[8..15] = group A's samples
  auto load_x = [&](float (&dst)[16], long long fblk) {
    if constexpr (KM > 0) load_frame16_mean<FMT, T, KM>(p, t, fl, fblk, dst);
    else load_frame16<FMT, T>(p, t, fl, fblk, dst);
  };
  auto prefetch_taps = [&](int pair) {
    const unsigned tap_p = (unsigned)pair * (N * 8u);
    static_for<0, 8>([&](auto mc) {
      constexpr int mh = decltype(mc)::value;
      const v4f32 q = __builtin_bit_cast(v4f32, __builtin_amdgcn_raw_buffer_load_b128(trsrc, toff, tap_p + (unsigned)(T * mh) * 16u, 0));
      pt[2 * mh] = v2f32{q.x, q.y};
      pt[2 * mh + 1] = v2f32{q.z, q.w};
    });
  };
  // shared round: the odd taper ([m/4][T][4] floats: values at samples t + T*(4*(m/4) + j)) and a
  // second copy of group A's samples (px holds group B by then)
  auto prefetch_shared = [&](long long fblk) {
    static_for<0, 4>([&](auto mc) {
      constexpr int mq = decltype(mc)::value;
      const v4f32 q = __builtin_bit_cast(v4f32, __builtin_amdgcn_raw_buffer_load_b128(vrsrc, toff, (unsigned)(T * mq) * 16u, 0));
      pt[2 * mq] = v2f32{q.x, q.y};
      pt[2 * mq + 1] = v2f32{q.z, q.w};
    });
    float xa[16];
    load_x(xa, fblk);
#pragma unroll
    for (int m = 0; m < 8; m++) pt[8 + m] = v2f32{xa[2 * m], xa[2 * m + 1]};
  };

  long long fblk = (long long)xcd_block_index() * (2 * FPB);
  if (fblk >= p.nframes) return;
  load_x(px, fblk);
  prefetch_taps(0);

  constexpr int RL = C::radix(NPASS - 1), BL = 16 / RL;
  auto rho_of = [](int m) constexpr { return (m % BL) + BL * brev(m / BL, RL); };   // register of bin t + T*m

  while (true) {                                              // one iteration: frame groups A and B
    const bool hasB = fblk + FPB < p.nframes;                 // block-uniform
    const int ngroups = hasB ? 2 : 1;
    const long long nfblk = fblk + stride;
    const bool has_next = nfblk < p.nframes;
    int hxA = 0, hxB = 0;          // scale into the shared round = 2^-hx, chosen from the frame's power

    for (int which = 0; which < ngroups; which++) {
      float acc[16];
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] = 0.0f;
      for (int pair = 0; pair < NP; pair++) {
        // ---- full round: re = x*taper(2*pair), im = x*taper(2*pair+1)
        float zr[16], zi[16];
#pragma unroll
        for (int m = 0; m < 16; m++) {
          zr[m] = px[m] * pt[m].x;
          zi[m] = px[m] * pt[m].y;
        }
        // the next round's loads go out after the first exchange's writes
        stockham16_passes<LOGN, NT>(zr, zi, xb, t, tw1row, twr, twi, [&] {
          if (pair + 1 < NP) {
            prefetch_taps(pair + 1);
          } else if (which + 1 < ngroups) {
            prefetch_taps(0);
            load_x(px, fblk + FPB);
          } else {
            prefetch_shared(fblk);
          }
        });
#pragma unroll
        for (int r = 0; r < 16; r++)
          acc[r] = __builtin_fmaf(zr[r], zr[r], __builtin_fmaf(zi[r], zi[r], acc[r]));
      }
      // ---- the group's full rounds are done: mirror fold psd[k] = acc[k] + acc[N-k] through
      // LDS, and the frame's power (sum of acc over the frame) for the shared round's scale
      float e = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; r++) e += acc[r];
      constexpr int RW = T < 64 ? T : 64;
#pragma unroll
      for (int w = 1; w < RW; w <<= 1) e += __shfl_xor(e, w);
      float *fold = reinterpret_cast<float *>(xb);
      frame_sync<T>();
      // only the upper half (bins >= N/2, entry k - N/2) goes through LDS; the lane adds the
      // partner of each of its own lower bins
      static_for<8, 16>([&](auto mc) {
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
      if (which == 0) hxA = scale_exponent(e);
      else hxB = scale_exponent(e);
      float *dstp = part + which * (FPB * L::PART);
      static_for<0, 8>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const int k = T * m + (int)t;
        float other = fold[N / 2 - k];                 // acc[N-k]; entry N/2 (k = 0) is never written
        if constexpr (m == 0) {
          if (t == 0) other = acc[rho_of(0)];          // bin 0 pairs with itself
        }
        dstp[k] = acc[rho_of(m)] + other;
      });
      if (t == 0) dstp[N / 2] = 2.0f * acc[rho_of(8)];
      if constexpr (GLFER16_BARRIER_AFTER_READS != 0) frame_sync<T>();   // fold buffer read: free for the next writes
    }

    // ---- shared round: re = sA*(xA*v), im = sB*(xB*v), v the odd taper
    {
      float zr[16], zi[16];
      const float sA = scale_in(hxA), sB = scale_in(hxB);
#pragma unroll
      for (int m = 0; m < 16; m++) {
        const float v = (m & 1) ? pt[m / 2].y : pt[m / 2].x;
        const float xa = (m & 1) ? pt[8 + m / 2].y : pt[8 + m / 2].x;
        zr[m] = (xa * v) * sA;
        zi[m] = hasB ? (px[m] * v) * sB : 0.0f;
      }
      stockham16_passes<LOGN, NT>(zr, zi, xb, t, tw1row, twr, twi, [&] {
        if (has_next) {
          prefetch_taps(0);
          load_x(px, nfblk);
        }
      });
      const float *partB = part + FPB * L::PART;
      separate_and_store<LOGN, FPB>(p, zr, zi, xb, t, fl, fblk, hxA, hxB,
                                    [&](auto mc) { return part[decltype(mc)::value == 8 ? N / 2 : T * decltype(mc)::value + (int)t]; },
                                    [&](auto mc) { return partB[decltype(mc)::value == 8 ? N / 2 : T * decltype(mc)::value + (int)t]; });
    }
    if (!has_next) break;
    fblk = nfblk;
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
static hipError_t launch16x_fmt(const SpectroParams &p, hipStream_t st) {
  constexpr int L = GLFER_LOGN;
  using LC = LaunchX<L>;
  const long long work = ((long long)p.nframes + 2 * LC::FPB - 1) / (2 * LC::FPB);
  if (work == 0) return hipSuccess;
  const long long per_cu = (GLFER16X_WAVES_PER_SIMD * 256) / LC::BLOCK > 0 ? (GLFER16X_WAVES_PER_SIMD * 256) / LC::BLOCK : 1;
  const long long resident = 256LL * per_cu;
  unsigned grid = (unsigned)(work < 4 * resident ? work : 4 * resident);
  if (grid >= 64) grid &= ~7u;                     // whole XCD slices: see xcd_block_index()
  if (p.mean_inkernel) {
    if constexpr (Plan16<L>::T <= 64) {
      if (p.history_mode) return hipErrorInvalidValue;
      const int km = (16 * p.H) % (1 << L) == 0 ? (16 * p.H) >> L : 0;
      if (km == 16) hipLaunchKernelGGL((spectro16x_kernel<L, FMT, GLFER16X_WAVES_PER_SIMD, 16>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
      else if (km == 8) hipLaunchKernelGGL((spectro16x_kernel<L, FMT, GLFER16X_WAVES_PER_SIMD, 8>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
      else if (km == 4) hipLaunchKernelGGL((spectro16x_kernel<L, FMT, GLFER16X_WAVES_PER_SIMD, 4>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
      else return hipErrorInvalidValue;
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  hipLaunchKernelGGL((spectro16x_kernel<L, FMT>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
  return hipGetLastError();
}

// odd taper counts >= 3; needs p->xtaps (the last taper alone, glfer_hip.cpp builds it)
extern "C" hipError_t GLFER_CAT(glfer_launch_spectro16x_n, GLFER_LOGN)(const SpectroParams *p, hipStream_t st) {
  if (!p->xtaps || p->npairs < 2 || p->nonlin || p->spec) return hipErrorInvalidValue;
  // the gather has no zero-history path: every frame must lie wholly inside the stream
  if (p->frame0 * (long long)p->H < (long long)p->R) return hipErrorInvalidValue;
  switch (p->fmt) {
    case GLFER_FMT_F32: return launch16x_fmt<GLFER_FMT_F32>(*p, st);
    case GLFER_FMT_S16: return launch16x_fmt<GLFER_FMT_S16>(*p, st);
    case GLFER_FMT_U8: return launch16x_fmt<GLFER_FMT_U8>(*p, st);
  }
  return hipErrorInvalidValue;
}
#endif  // GLFER_NO_LAUNCHERS
