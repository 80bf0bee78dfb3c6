// odd_taper.hpp -- what the three kernels for odd taper counts (spectro16x / xl / y) have in common:
// the in-stream gather of a frame, the power-of-two scale of a frame entering the shared transform,
// and the separation of the shared transform into the two frames' PSD rows.
#pragma once
#include "stockham16.hpp"

#ifndef GLFER_PSD_STORE_AUX
#define GLFER_PSD_STORE_AUX 0      /* default cache policy: the non-temporal path wrote 1.07-1.26x the row bytes to HBM and is ~1 % slower (profiles/r03_store_policy.txt) */
#endif

namespace glfer {

// The 16 samples t + T*m of frame (fblk + fl) of the launch.  Only frames that lie wholly inside
// the stream reach these kernels (the launcher sends a stream's first ceil(R/H) frames to
// spectro16.hip, which has the zero-history gather), so every load is in range: one shared VGPR
// offset + immediates.  history_mode 1 (fft.c:103-108 with glfer.first_buffer stuck at TRUE) zeroes
// the first R samples of every frame afterwards.  A frame slot past the last frame re-reads the
// last one (its results are dropped).
// HIST: -1 = test p.history_mode at run time (the compiler turns that into 16 unconditional
// selects per frame); 0 / 1 = decided at compile time by the caller.
template <int FMT, int T, int HIST = -1>
__device__ __forceinline__ void load_frame16(const SpectroParams &p, unsigned t, unsigned fl, long long fblk, float (&dst)[16]) {
  constexpr unsigned esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  const long long f = fblk + fl;
  const unsigned flc = f < p.nframes ? fl : (unsigned)(p.nframes - 1 - fblk);
  const long long sblk = (p.frame0 + fblk) * (long long)p.H - p.R;
  if (HIST == 1 || (HIST < 0 && p.history_mode)) {
    // ZERO_ALWAYS: a frame is R zeros + its own hop (fft.c:99-108), and a piece cut for that mode
    // carries NO history (glfer_hip.h, "Cutting a stream", rule 2) -- so the history is never
    // loaded: the descriptor starts at the block's own first hop and the history registers get an
    // out-of-range offset (they read 0 by the range check; the select is for u8, whose raw 0 is
    // not sample 0.0).
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.stream)) + (sblk + p.R) * (long long)esz, 0, 0x7fffffff, 0x00020000);
    const int d = (int)t - p.R;                       // sample t + T*m of the frame is kept iff d + T*m >= 0
    const int lrel = (int)(flc * (unsigned)p.H) + d;
    static_for<0, 16>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const bool ok = d >= -T * m;
      const float x = buf_sample<FMT>(hrsrc, ok ? (unsigned)(lrel + T * m) * esz : 0x80000000u, 0u);
      dst[m] = ok ? x : 0.0f;
    });
    return;
  }
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char *>(reinterpret_cast<const char *>(p.stream)) + sblk * (long long)esz, 0, 0x7fffffff, 0x00020000);
  const unsigned lrel = flc * (unsigned)p.H + t;
  static_for<0, 16>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    dst[m] = buf_sample<FMT>(xrsrc, lrel * esz, (unsigned)(T * m) * esz);
  });
}

// Sum over the wavefront by DPP (row_shr 1,2,4,8 leave a row's sum in its lane 15, row_bcast 15 / 31
// carry it to lane 63), broadcast from lane 63: a tenth of the latency of six ds_bpermute rounds.
__device__ __forceinline__ float wave_total_f32(float v) {
  auto dpp = [](float x, auto ctrl, auto rowmask) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(rowmask)::value, 0xf, false));
  };
  v += dpp(v, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});
  v += dpp(v, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
  v += dpp(v, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});
  v += dpp(v, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
  v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
  v += dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// The same with per-hop mean removal (fft.c:86-96) for a hop of KM of the lane's 16 registers
// (history from the stream; T <= 64: a frame's lanes sit in one wavefront): the hop sums are taken
// from the registers -- lane partial in register order, a butterfly over the frame's T lanes -- and
// x - mu replaces x.  The same lanes, registers and order whichever frame a hop is seen in.
template <int FMT, int T, int KM>
__device__ __forceinline__ void load_frame16_mean(const SpectroParams &p, unsigned t, unsigned fl, long long fblk, float (&dst)[16]) {
  static_assert(T <= 64 && (KM == 16 || KM == 8 || KM == 4), "a frame within one wavefront; hop = 4, 8 or 16 registers");
  load_frame16<FMT, T, 0>(p, t, fl, fblk, dst);
  constexpr int NH = 16 / KM;
  float mu[NH];
  if (p.means) {
    // given means (cfg.sub_mean = 1: the reference's own summation order, submean_seq.hip), indexed by GLOBAL hop = the frame
    // whose newest hop it is; the samples are floats in sample units here in every format
    const long long f = fblk + fl;
    const long long F = p.frame0 + (f < p.nframes ? f : (long long)p.nframes - 1);
#pragma unroll
    for (int q = 0; q < NH; q++) mu[q] = p.means[F - (NH - 1) + q];
  } else {
#pragma unroll
    for (int q = 0; q < NH; q++) {
      float sm = 0.0f;
#pragma unroll
      for (int m = 0; m < KM; m++) sm += dst[q * KM + m];
#pragma unroll
      for (int o = 1; o < T; o <<= 1) sm += __shfl_xor(sm, o);
      mu[q] = sm / (float)p.H;                         // fft.c:91
    }
  }
#pragma unroll
  for (int m = 0; m < 16; m++) dst[m] = dst[m] - mu[m / KM];
}

// A frame enters the shared transform scaled by 2^-hx, hx = half the binary exponent of its power
// summed over the tapers already done, so that both halves of the transform have the same
// magnitude and each frame sees the rounding of a transform of its own size.  A frame whose power
// is exactly 0 (digital silence) must stay exactly 0, as in the reference, whatever its partner's
// rounding leaves in the shared transform: scale 0 (kSilent).
constexpr int kSilent = 0x7fff;
__device__ __forceinline__ int scale_exponent(float power) {
  int ex = __builtin_amdgcn_frexp_expf(power);       // 0 for 0, inf, nan
  ex = ex > 120 ? 120 : (ex < -120 ? -120 : ex);
  return power == 0.0f ? kSilent : ex >> 1;
}
__device__ __forceinline__ float scale_in(int hx) { return hx == kSilent ? 0.0f : __builtin_amdgcn_ldexpf(1.0f, -hx); }
__device__ __forceinline__ float scale_out(int hx) { return hx == kSilent ? 0.0f : __builtin_amdgcn_ldexpf(1.0f, 2 * hx); }

// After the shared transform Z = FFT(sA*yA + i*sB*yB) (register b + B*brev(q',R) of lane t holds bin
// t + T*(b + B*q')): the mirror pairs (k, N-k) give E = Z[k] + conj Z[N-k] = 2 sA Y_A[k] and
// O = Z[k] - conj Z[N-k] = 2i sB Y_B[k]; |E|^2/sA^2 and |O|^2/sB^2 (the 1/4 is folded into the
// taper) are added to the frames' folded sums and stored.  Z[k], k >= N/2, goes through LDS
// (entry k - N/2 of xb); the lane keeps its own Z[k], k < N/2.  xb is free again on return.  sumA(mc) / sumB(mc): the folded sum of bin t + T*m,
// m = 0..7 as a compile-time constant, m = 8 standing for bin N/2 (lane 0 only).
// Rows go out through buffer descriptors: one shared VGPR offset plus SGPR/immediate offsets, and
// frame slots past the last frame fall outside num_records, so their stores are dropped.
template <int LOGN, int FPB, class SumA, class SumB>
__device__ __forceinline__ void separate_and_store(const SpectroParams &p, const float (&zr)[16], const float (&zi)[16],
                                                   v2f32 *xb, unsigned t, unsigned fl, long long fblk, int hxA, int hxB,
                                                   SumA &&sumA, SumB &&sumB) {
  using C = Plan16<LOGN>;
  constexpr int N = C::N, T = C::T, RL = C::radix(C::NPASS - 1), BL = 16 / RL;
  auto rho_of = [](int m) constexpr { return (m % BL) + BL * brev(m / BL, RL); };
  if constexpr (GLFER16_BARRIER_AFTER_READS == 0) frame_sync<T>();      // else the passes left xb free
  static_for<8, 16>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    constexpr int r = rho_of(m);
    xb[t + T * (m - 8)] = v2f32{zr[r], zi[r]};
  });
  frame_sync<T>();
  const unsigned ROWB = (unsigned)p.pitch * 4u;            // bytes from row to row (cfg.psd_pitch)
  const long long leftA = p.nframes - fblk, leftB = p.nframes - (fblk + FPB);
  const unsigned recA = (unsigned)((leftA > FPB ? FPB : leftA) * (long long)ROWB);
  const unsigned recB = leftB > 0 ? (unsigned)((leftB > FPB ? FPB : leftB) * (long long)ROWB) : 0u;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(p.psd + (size_t)fblk * (size_t)p.pitch, 0, recA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(p.psd + (size_t)(fblk + (leftB > 0 ? FPB : 0)) * (size_t)p.pitch, 0, recB, 0x00020000);
  const unsigned voff = fl * ROWB + t * 4u;
  const float uA = scale_out(hxA), uB = scale_out(hxB);
  static_for<0, 8>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    constexpr int r = rho_of(m);
    const int k = (int)t + T * m;
    v2f32 b = xb[N / 2 - k];                          // Z[N-k]; entry N/2 (k = 0) is never written
    const float ar = zr[r], ai = zi[r];
    if constexpr (m == 0) {
      if (t == 0) b = v2f32{ar, ai};                  // k = 0 pairs with itself
    }
    const float er = ar + b.x, ei = ai - b.y, orr = ar - b.x, oi = ai + b.y;
    const float pa = __builtin_fmaf(er, er, ei * ei), pb = __builtin_fmaf(orr, orr, oi * oi);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_fmaf(pa, uA, sumA(mc))), ra, voff, (unsigned)(T * m) * 4u, GLFER_PSD_STORE_AUX);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_fmaf(pb, uB, sumB(mc))), rb, voff, (unsigned)(T * m) * 4u, GLFER_PSD_STORE_AUX);
  });
  if (t == 0) {                                       // k = N/2 pairs with itself: E = 2 Re Z, O = 2i Im Z
    constexpr int r = rho_of(8);
    constexpr std::integral_constant<int, 8> nyq{};
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_fmaf(4.0f * zr[r] * zr[r], uA, sumA(nyq))), ra, voff, (unsigned)(N / 2) * 4u, GLFER_PSD_STORE_AUX);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_fmaf(4.0f * zi[r] * zi[r], uB, sumB(nyq))), rb, voff, (unsigned)(N / 2) * 4u, GLFER_PSD_STORE_AUX);
  }
  frame_sync<T>();                                    // mirror entries read: xb is free again
}

}  // namespace glfer
