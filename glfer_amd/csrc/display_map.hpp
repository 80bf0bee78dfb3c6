// display_map.hpp -- one bin of one waterfall column (g_main.c:1186-1226): shared by map_kernel
// (display.hip) and the average-and-map kernel (aux_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef GLFER_MAP_GUARD
#define GLFER_MAP_GUARD 2.5e-4f    /* the float logarithm is within 8e-5 of 10 log10 x (1 ulp of v_log_f32 at |log2 x| <= 128, the product, the constant) */
#endif

namespace glfer {

// double -> int32 the way the reference's implicit double->short / double->unsigned char
// conversions behave on x86-64 (cvttsd2si: out of range or NaN gives INT_MIN; low bits kept)
__device__ __forceinline__ int x86_d2i(double d) {
  if (!(d > -2147483649.0 && d < 2147483648.0)) return (int)0x80000000u;
  return (int)d;
}

// One bin of one column: the dB short and the 0..255 colour index (g_main.c:1186-1226).
//   levbuf = (short)(10 log10 x)  [double log10, truncated]       colour = (uchar)((f - thr255) / one_m_thr)
// Both truncate a double: the value only matters next to an integer.  So the logarithm is taken in
// float (v_log_f32: error < 1e-4 over |y| <= 400) and a result further than GLFER_MAP_GUARD from an
// integer is truncated as it is; a result inside the guard band (0.05 % of the bins; a wavefront
// takes the branch when ANY of its lanes does, 3 % of the time -- with the 2e-3 band and a double
// log10 behind it, round 1's form, that was 23 % and a third of the kernel) is decided by ONE comparison with the
// point where the reference's own 10.0*log10(x) crosses that integer (log_thr: host_tables.cpp
// log_thresholds(), built with the host libm the reference itself would run on), not by a double
// log10 on the device.  The quotient is a product with the reciprocal, recomputed the reference's
// way only within 1e-9 of an integer (a few double ulp of at most 255).  Same integers as the
// all-double form, at a fraction of the instructions.
constexpr int kLogThrK = 400;                              // host_tables.h
struct RowScale {                                          // per column: display_min and 1/(display_max - display_min)
  float display_min, span, inv_span;
  bool fast;                                               // the reciprocal form of x / span is exact (see fdiv)
};
// a / span, correctly rounded, for many a and one span: y = RN(1/span), q0 = RN(a y),
// r = a - span q0 (exact in an fma), RN(q0 + r y) = RN(a / span) (Markstein) while nothing over- or
// underflows: |span| and |a| within 2^-60 .. 2^59; anything else (0 included) takes the division.
__device__ __forceinline__ float fdiv(float a, const RowScale &rs) {
  const unsigned e = (__float_as_uint(a) >> 23) & 0xffu;   // biased exponent
  if (rs.fast && e - 67u < 120u) {
    const float q0 = a * rs.inv_span;
    const float r = __builtin_fmaf(-rs.span, q0, a);
    return __builtin_fmaf(r, rs.inv_span, q0);
  }
  return a / rs.span;
}
__device__ __forceinline__ RowScale row_scale(float display_max, float display_min) {
  RowScale rs;
  rs.display_min = display_min;
  rs.span = display_max - display_min;
  rs.inv_span = 1.0f / rs.span;
  const unsigned e = (__float_as_uint(rs.span) >> 23) & 0xffu;
  rs.fast = e - 67u < 120u;
  return rs;
}

// levbuf's whole-dB short of one bin; sf = (float)s, what the linear scale maps
template <typename SRC>
__device__ __forceinline__ short db_short(SRC s, int scale_log, const double *__restrict__ log_thr, float &sf) {
#pragma clang fp contract(off)                             // the reference's float/double expression order is the contract
  sf = (float)s;                                           // the linear scale maps (float)s, and logs that
  const double sd = scale_log ? (double)s : (double)sf;
  const float y = __builtin_amdgcn_logf(sf) * 3.010299956639812f;         // 10 log10 = log2 * 10 log10(2)
  const float yr = __builtin_rintf(y);
  int li;
  if (sf > 1e-37f && sf < 3e38f) {                         // normal floats (a double source rounded to float moves y by 3e-7)
    if (__builtin_fabsf(y - yr) > GLFER_MAP_GUARD) {
      li = (int)y;
    } else {                                               // 10 log10(sd) is within the guard band of k: k, or the integer before it
      const int k = (int)yr;
      const double t = log_thr[kLogThrK + k];
      li = k > 0 ? (sd >= t ? k : k - 1) : (k < 0 ? (sd <= t ? k : k + 1) : 0);
    }
  } else {
    li = x86_d2i(10.0 * log10(sd));
  }
  return (short)li;
}

// the 0..255 colour index of a level (g_main.c:1204-1217); above: the `f > 255` branch was the one taken
__device__ __forceinline__ unsigned colour_index(float sig_level, const RowScale &rs, double thr255, double one_m_thr,
                                                 double inv_one_m_thr, bool &above) {
#pragma clang fp contract(off)
  const float f = 255.0f * fdiv(sig_level - rs.display_min, rs);
  above = false;
  if ((double)f < thr255) return 0;
  if (f > 255.0f) {
    above = true;
    return 255;
  }
  const double num = (double)f - thr255, q = num * inv_one_m_thr;
  const int qi = (__builtin_fabs(q - __builtin_rint(q)) > 1e-9) ? (int)q : x86_d2i(num / one_m_thr);
  return (unsigned)qi & 0xffu;                             // (unsigned char) of the conversion
}

template <typename SRC>
__device__ __forceinline__ void map_bin(SRC s, int scale_log, const RowScale &rs, double thr255,
                                        double one_m_thr, double inv_one_m_thr, const double *__restrict__ log_thr,
                                        short &l, unsigned &v) {
  float sf;
  l = db_short<SRC>(s, scale_log, log_thr, sf);
  bool above;
  v = colour_index(scale_log ? (float)l : sf, rs, thr255, one_m_thr, inv_one_m_thr, above);
}

// On the logarithmic scales the colour index is a function of the column's levels and of the bin's
// whole-dB short alone (sig_level = (float)levbuf, g_main.c:1188-1192), and it is constant outside
// [display_min, display_max]: 0 below (f <= 0, thr255 >= 0) and 255 above (f > 255).  So a column
// needs the quotient, the threshold and the conversion only for the 256 shorts from
// l0 = floor(display_min) - 1 on -- one per thread, by the very expression of the reference -- and a
// bin is its logarithm and one LDS read.  The table stands for the column when its last entry took
// the `f > 255` branch (the quotient is monotonic in the level: so does every level above it) and
// the levels are ordered and finite; any other column (a span of 254 dB and more, max <= min, NaN)
// maps bin by bin.
struct DbTable {
  int l0;
  bool low_ok;
};
__device__ __forceinline__ DbTable db_table(const RowScale &rs, double thr255) {
  DbTable t;
  const float fl = __builtin_floorf(rs.display_min);
  t.low_ok = rs.span > 0.0f && thr255 >= 0.0 && __builtin_fabsf(fl) < 30000.0f;   // a NaN fails each
  t.l0 = t.low_ok ? (int)fl - 1 : 0;
  return t;
}
__device__ __forceinline__ unsigned db_table_slot(const DbTable &t, short l) {
  const int i = (int)l - t.l0;
  return (unsigned)(i < 0 ? 0 : (i > 255 ? 255 : i));
}

}  // namespace glfer
