// fft_inreg.hpp -- compile-time-unrolled complex FFTs on per-lane register arrays.
//
// Part of the MI355X spectral engine (replaces the butterfly nest of the
// reference's fft_radix2.c:104-175).  Everything here is resolved at compile
// time: register indices, twiddle constants and the trivial-twiddle special
// cases, so a radix-R transform is a straight line of v_fma/v_add/v_sub on
// VGPRs with no address arithmetic.
//
// Butterfly form: radix-2 decimation in time with the 6-FMA butterfly
//     X  = a + w*b      (4 fma)         X' = 2a - X   (2 fma)
// which costs 3*R*log2(R) VALU ops per R-point transform -- the same count as
// a radix-4 kernel without FMA fusion -- and 4 ops where w is 1 or -i.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#ifndef GLFER_BFLY_GROUP
#define GLFER_BFLY_GROUP 4
#endif

namespace glfer {

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n / 2); }

// bit reversal of k in log2(R) bits
constexpr int brev(int k, int R) {
  int r = 0;
  for (int b = 1; b < R; b <<= 1) {
    r = (r << 1) | (k & 1);
    k >>= 1;
  }
  return r;
}

// ---- compile-time unit roots, exact octant reduction + Taylor core ----
constexpr double kPi = 3.141592653589793238462643383279502884;

constexpr double sin_core(double x) {  // |x| <= pi/4
  double x2 = x * x, term = x, sum = x;
  for (int i = 1; i < 14; i++) {
    term *= -x2 / double((2 * i) * (2 * i + 1));
    sum += term;
  }
  return sum;
}
constexpr double cos_core(double x) {
  double x2 = x * x, term = 1.0, sum = 1.0;
  for (int i = 1; i < 14; i++) {
    term *= -x2 / double((2 * i - 1) * (2 * i));
    sum += term;
  }
  return sum;
}
struct cplx64 { double c, s; };
// exp(+i*2*pi*j/R), R a power of two
constexpr cplx64 unit_root(int j, int R) {
  j %= R;
  if (j < 0) j += R;
  if (j == 0) return {1.0, 0.0};
  if (2 * j == R) return {-1.0, 0.0};
  if (4 * j == R) return {0.0, 1.0};
  if (4 * j == 3 * R) return {0.0, -1.0};
  if (2 * j > R) { cplx64 u = unit_root(R - j, R); return {u.c, -u.s}; }
  if (4 * j > R) { cplx64 u = unit_root(R / 2 - j, R); return {-u.c, u.s}; }
  if (8 * j > R) { cplx64 u = unit_root(R / 4 - j, R); return {u.s, u.c}; }
  double a = 2.0 * kPi * double(j) / double(R);
  return {cos_core(a), sin_core(a)};
}

// One DIT combine stage entry: a at index PA, b at PB, forward twiddle exp(-i 2 pi K / R).
template <int R, int K, int PA, int PB, int LEN>
__device__ __forceinline__ void bfly(float (&re)[LEN], float (&im)[LEN]) {
  const float ar = re[PA], ai = im[PA], br = re[PB], bi = im[PB];
  if constexpr (K == 0) {
    re[PA] = ar + br; im[PA] = ai + bi;
    re[PB] = ar - br; im[PB] = ai - bi;
  } else if constexpr (4 * K == R) {          // w = -i : t = (bi, -br)
    re[PA] = ar + bi; im[PA] = ai - br;
    re[PB] = ar - bi; im[PB] = ai + br;
  } else {
    constexpr cplx64 u = unit_root(K, R);
    constexpr float wr = float(u.c), wi = float(-u.s);   // w = exp(-i theta)
    const float xr = __builtin_fmaf(wr, br, __builtin_fmaf(-wi, bi, ar));
    const float xi = __builtin_fmaf(wr, bi, __builtin_fmaf(wi, br, ai));
    re[PA] = xr; im[PA] = xi;
    re[PB] = __builtin_fmaf(2.0f, ar, -xr);
    im[PB] = __builtin_fmaf(2.0f, ai, -xi);
  }
}

// R-point forward DFT over the elements OFF + S*i (i = 0..R-1, natural order).
// Result X[k] is left at index OFF + S*brev(k, R).
template <int R, int S, int OFF, int LEN>
__device__ __forceinline__ void dit(float (&re)[LEN], float (&im)[LEN]) {
  if constexpr (R >= 2) {
    dit<R / 2, 2 * S, OFF, LEN>(re, im);
    dit<R / 2, 2 * S, OFF + S, LEN>(re, im);
    static_for<0, R / 2>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int pa = OFF + 2 * S * brev(k, R / 2);
      bfly<R, k, pa, pa + S, LEN>(re, im);
      // keep at most GLFER_BFLY_GROUP butterflies in one scheduling region: left alone, the
      // scheduler interleaves all R/2 independent butterflies and their temporaries spill
      if constexpr (GLFER_BFLY_GROUP > 0 && (k % GLFER_BFLY_GROUP) == GLFER_BFLY_GROUP - 1)
        __builtin_amdgcn_sched_barrier(0);
    });
  }
}

// ---- the inter-pass twiddles folded into the first butterfly stage ----
// A Stockham pass multiplies its inputs by twiddles T_m and then runs dit<R>.  The first stage's butterflies have w = 1
// (pairs m, m + 8S', four operations each) on inputs that cost four operations each to twiddle: 12 per pair.  Folded,
//     a' = T_a a (4; none when T_a = 1)      X = a' + T_b b (4 fma)      X' = 2a' - X (2 fma)
// the pair costs 10 (6 with T_a = 1): 16 operations fewer per lane and twiddled pass, and one rounding fewer on X.
// tw(integral_constant<int, m>) returns register m's twiddle as a v2f32, or NoTwiddle.
struct NoTwiddle {};

template <int R, int S, int OFF, int LEN, class TW>
__device__ __forceinline__ void dit_tw(float (&re)[LEN], float (&im)[LEN], const TW &tw) {
  if constexpr (R == 2) {
    constexpr int PA = OFF, PB = OFF + S;
    const auto wa = tw(std::integral_constant<int, PA>{});
    const auto wb = tw(std::integral_constant<int, PB>{});
    float ar = re[PA], ai = im[PA];
    if constexpr (!std::is_same_v<std::decay_t<decltype(wa)>, NoTwiddle>) {
      const float r0 = ar, i0 = ai;
      ar = __builtin_fmaf(r0, wa.x, -i0 * wa.y);
      ai = __builtin_fmaf(r0, wa.y, i0 * wa.x);
    }
    const float br = re[PB], bi = im[PB];
    if constexpr (!std::is_same_v<std::decay_t<decltype(wb)>, NoTwiddle>) {
      const float xr = __builtin_fmaf(br, wb.x, __builtin_fmaf(-bi, wb.y, ar));
      const float xi = __builtin_fmaf(br, wb.y, __builtin_fmaf(bi, wb.x, ai));
      re[PA] = xr; im[PA] = xi;
      re[PB] = __builtin_fmaf(2.0f, ar, -xr);
      im[PB] = __builtin_fmaf(2.0f, ai, -xi);
    } else {
      re[PA] = ar + br; im[PA] = ai + bi;
      re[PB] = ar - br; im[PB] = ai - bi;
    }
  } else if constexpr (R > 2) {
    dit_tw<R / 2, 2 * S, OFF, LEN>(re, im, tw);
    dit_tw<R / 2, 2 * S, OFF + S, LEN>(re, im, tw);
    static_for<0, R / 2>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int pa = OFF + 2 * S * brev(k, R / 2);
      bfly<R, k, pa, pa + S, LEN>(re, im);
      if constexpr (GLFER_BFLY_GROUP > 0 && (k % GLFER_BFLY_GROUP) == GLFER_BFLY_GROUP - 1)
        __builtin_amdgcn_sched_barrier(0);
    });
  }
}
template <int R, int S, int OFF, int LEN, class TW>
__device__ __forceinline__ void dit_head_tw(float (&re)[LEN], float (&im)[LEN], const TW &tw) {
  static_assert(R >= 4, "radix");
  dit_tw<R / 2, 2 * S, OFF, LEN>(re, im, tw);
  dit_tw<R / 2, 2 * S, OFF + S, LEN>(re, im, tw);
}

// dit<R,...> whose LAST stage hands every finished output to emit(k, reg) -- X[k] sits in register
// reg -- as soon as its butterfly is done (two outputs per butterfly: k and k + R/2), so that the
// consumer (an LDS store) is issued between butterflies instead of in one burst after the transform.
// dit_emit in two halves, for callers that put something (a barrier) in front of the last stage: head = all stages
// but the last, tail = the last stage with its emits.  head + tail is dit_emit, operation for operation.
template <int R, int S, int OFF, int LEN>
__device__ __forceinline__ void dit_head(float (&re)[LEN], float (&im)[LEN]) {
  static_assert(R >= 2, "radix");
  dit<R / 2, 2 * S, OFF, LEN>(re, im);
  dit<R / 2, 2 * S, OFF + S, LEN>(re, im);
}
template <int R, int S, int OFF, int LEN, class F>
__device__ __forceinline__ void dit_tail(float (&re)[LEN], float (&im)[LEN], F &&emit) {
  static_for<0, R / 2>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int pa = OFF + 2 * S * brev(k, R / 2);
    bfly<R, k, pa, pa + S, LEN>(re, im);
    emit(std::integral_constant<int, k>{}, std::integral_constant<int, pa>{});
    emit(std::integral_constant<int, k + R / 2>{}, std::integral_constant<int, pa + S>{});
    if constexpr (GLFER_BFLY_GROUP > 0 && (k % GLFER_BFLY_GROUP) == GLFER_BFLY_GROUP - 1)
      __builtin_amdgcn_sched_barrier(0);
  });
}

template <int R, int S, int OFF, int LEN, class F>
__device__ __forceinline__ void dit_emit(float (&re)[LEN], float (&im)[LEN], F &&emit) {
  static_assert(R >= 2, "radix");
  dit<R / 2, 2 * S, OFF, LEN>(re, im);
  dit<R / 2, 2 * S, OFF + S, LEN>(re, im);
  static_for<0, R / 2>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int pa = OFF + 2 * S * brev(k, R / 2);
    bfly<R, k, pa, pa + S, LEN>(re, im);
    emit(std::integral_constant<int, k>{}, std::integral_constant<int, pa>{});
    emit(std::integral_constant<int, k + R / 2>{}, std::integral_constant<int, pa + S>{});
    if constexpr (GLFER_BFLY_GROUP > 0 && (k % GLFER_BFLY_GROUP) == GLFER_BFLY_GROUP - 1)
      __builtin_amdgcn_sched_barrier(0);
  });
}

}  // namespace glfer
