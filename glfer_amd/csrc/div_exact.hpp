// div_exact.hpp -- a / d for many a and one d, correctly rounded (shared by aux_kernels.hip's update_avg kernels and the
// average taken inside the periodogram kernel, spectro16h.hip: both must produce the SAME doubles, avg.c:155).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace glfer {

#ifndef GLFER_AVG_ABL
#define GLFER_AVG_ABL 0    /* timing ablations (results wrong): 1 no row stores, 2 no wavefront reduction, 4 no barrier, 8 no quotients */
#endif

// With y = RN(1/d) (one true division), q0 = RN(a y), r = a - d q0 (exact in an fma), q = RN(q0 + r y) is RN(a/d) (Markstein) as
// long as nothing over- or underflows on the way -- so only for 1e-100 < |d| < 1e100 (the operands here are sums of float32
// bins: below 1e40, and a quotient of 1e-45/1e100 is still a normal double); any other divisor (0, inf, NaN, denormal) takes the
// division itself.
struct Divisor {
  double d, y;
  bool fast;
  __device__ __forceinline__ explicit Divisor(double dd) : d(dd), y(1.0 / dd) {
    const double a = dd < 0 ? -dd : dd;
    fast = __builtin_amdgcn_readfirstlane((a > 1e-100 && a < 1e100) ? 1 : 0) != 0;   // d is the same in every lane
  }
  __device__ __forceinline__ double operator()(double a) const {
#if GLFER_AVG_ABL & 8
    return a + y;
#endif
    if (!fast) return a / d;
    const double q0 = a * y;
    const double r = __builtin_fma(-d, q0, a);
    return __builtin_fma(r, y, q0);
  }
};

// ---- double-precision wavefront reductions by DPP: row_shr 1,2,4,8 leave a row's result in its lane
// 15, row_bcast 15 / 31 carry it to lane 63, readlane broadcasts it.  A lane without a source takes
// `ZERO ? 0 : itself` -- the identity of a sum / of a maximum or minimum.  (Six ds_bpermute rounds
// per value, as __shfl_xor does it, cost an LDS round trip each.)
template <int CTRL, int ROWMASK, bool ZERO>
__device__ __forceinline__ double dpp_f64(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const int lo = (int)(unsigned)b, hi = (int)(unsigned)(b >> 32);
  const unsigned rl = (unsigned)__builtin_amdgcn_update_dpp(ZERO ? 0 : lo, lo, CTRL, ROWMASK, 0xf, false);
  const unsigned rh = (unsigned)__builtin_amdgcn_update_dpp(ZERO ? 0 : hi, hi, CTRL, ROWMASK, 0xf, false);
  return __longlong_as_double((long long)(((unsigned long long)rh << 32) | rl));
}
__device__ __forceinline__ double lane63_f64(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// over the wavefront: the sum of s, the maximum of mx with the LOWEST index mi among equals, the minimum of mn
// (INDEX = false: the index is not wanted -- a column that is only mapped has no peak bin to return)
template <bool INDEX = true>
__device__ __forceinline__ void wave_sum_max_min(double &s, double &mx, int &mi, double &mn) {
  auto step = [&](auto ctrl, auto rowmask) {
    constexpr int CT = decltype(ctrl)::value, RM = decltype(rowmask)::value;
    const double os = dpp_f64<CT, RM, true>(s), om = dpp_f64<CT, RM, false>(mx), on = dpp_f64<CT, RM, false>(mn);
    s += os;
    if constexpr (INDEX) {
      const int oi = __builtin_amdgcn_update_dpp(mi, mi, CT, RM, 0xf, false);
      if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
    } else {
      mx = om > mx ? om : mx;
    }
    mn = on < mn ? on : mn;
  };
  step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
  step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
  s = lane63_f64(s);
  mx = lane63_f64(mx);
  mn = lane63_f64(mn);
  if constexpr (INDEX) mi = __builtin_amdgcn_readlane(mi, 63);
}


// the sum of s and the maximum of mx with the LOWEST index mi among equals, over the wavefront (no minimum: the plain average's
// return values need none)
__device__ __forceinline__ void wave_sum_max(double &s, double &mx, int &mi) {
  auto step = [&](auto ctrl, auto rowmask) {
    constexpr int CT = decltype(ctrl)::value, RM = decltype(rowmask)::value;
    const double os = dpp_f64<CT, RM, true>(s), om = dpp_f64<CT, RM, false>(mx);
    const int oi = __builtin_amdgcn_update_dpp(mi, mi, CT, RM, 0xf, false);
    s += os;
    if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
  };
  step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
  step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
  s = lane63_f64(s);
  mx = lane63_f64(mx);
  mi = __builtin_amdgcn_readlane(mi, 63);
}

}  // namespace glfer
