// stats_kernels.hip -- per-bin epilogues over spectra the estimator kernels have produced, and the
// frame-preparation kernel.  Built with -ffp-contract=off: the float/double expression order of
// the reference IS the contract here (every statement below is the reference's statement with the
// same operand types).
//   lmp_kernel      lmp.c:132-160   detection statistic over the ring of the last nl periodograms
//   ftest_kernel    mtm.c:203-210, 222-233   harmonic F statistic from the tapered spectra and mu
//   prepare_kernel  fft.c:98-156    what prepare_audio leaves in inbuf_fft, for a batch of frames
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <type_traits>
#include <algorithm>
#include "spectro_params.h"

#ifndef GLFER_LMP_RING
#define GLFER_LMP_RING 1    /* 0: every frame through lmp_kernel (one thread per frame and bin, nl row reads each) */
#endif

namespace glfer {
hipError_t allow_dynamic_lds(const void *kernel, size_t bytes);   // plan.h / glfer_hip.cpp

// a / d for many a and one d, correctly rounded: with y = RN(1/d) (one true division), q0 = RN(a y),
// r = a - d q0 (exact in an fma), RN(q0 + r y) = RN(a/d) (Markstein) -- the divisors here are the
// small integers nl and nl - 1, the dividends sums of a few float32 bins: nothing over- or underflows.
struct SmallDivisor {
  double d, y;
  __device__ __forceinline__ SmallDivisor(double dd, double recip) : d(dd), y(recip) {}   // recip = RN(1/dd), from the host: IEEE division either side
  __device__ __forceinline__ double operator()(double a) const {
    const double q0 = a * y;
    return __builtin_fma(__builtin_fma(-d, q0, a), y, q0);
  }
};

// One thread per (frame, bin).  rows: periodograms of frames [row0, row0 + nrows) (global frame
// indices), [nrows][bins]; out: frames [first, first + nframes).  The reference keeps the last nl
// periodograms in a ring written round-robin (slot = frame mod nl, zero before its first write)
// and sums over the SLOTS in slot order; slot j of frame f holds frame f - ((f - j) mod nl).
// first_mod = first mod nl; c_neg = -sqrt(nl / 2.0) and c_den = 2.0 * sqrt(2.0 * nl) are the
// expression's constants (lmp.c:156), evaluated once on the host (IEEE sqrt: the same doubles).
// (Round 2's first form took f % nl in 64 bits and (jl - j + nl) % nl per term, two f64 square
// roots of constants and three f64 divisions per bin: 7x the time of the periodograms under it.)
// NL > 0: nl is that constant -- the ring's rows are loaded once into registers and the two sums are
// straight-line code, one copy per rotation of the ring (f mod nl is the same for the whole
// workgroup); NL = 0: any nl, by loops.
template <int NL>
__global__ __launch_bounds__(256) void lmp_kernel(const float *__restrict__ rows, long long row0, long long first,
                                                  long long nframes, int bins, int nl_arg, int first_mod, double c_neg,
                                                  double c_den, double recip_nl, double recip_nl1, float *__restrict__ out) {
  const int nl = NL > 0 ? NL : nl_arg;
  const unsigned fi = blockIdx.y;                                  // < 65535
  const int i = blockIdx.x * 256 + threadIdx.x;
  if ((long long)fi >= nframes || i >= bins) return;
  const long long f = first + fi;
  float *o = out + (size_t)fi * bins;
  if (i == 0) {                                                    // lmp.c:160
    o[0] = 1e-3;
    return;
  }
  const int jl = (int)(((unsigned)first_mod + fi % (unsigned)nl) % (unsigned)nl);   // f mod nl
  const float *col = rows + (size_t)(f - row0) * bins + i;         // this frame's row; frame f - d is d rows up
  // (the two reciprocals were a true f64 division each, per bin: two fifths of the kernel's instructions)
  const SmallDivisor by_nl((double)nl, recip_nl), by_nl1((double)(nl - 1), recip_nl1);
  double my = 0.0, sy = 0.0;
  if constexpr (NL > 0) {
    float v[NL];                                                   // v[d]: bin i of frame f - d (zero before the first frame)
#pragma unroll
    for (int d = 0; d < NL; d++) v[d] = (long long)d <= f ? *(col - (size_t)d * bins) : 0.0f;
    // slot j holds frame f - ((jl - j) mod nl); the sums run over the slots in slot order
    auto sums = [&](auto rc) {
      constexpr int R = decltype(rc)::value;
#pragma unroll
      for (int j = 0; j < NL; j++) my += v[(R - j + NL) % NL];     // lmp.c:134-140
      my = by_nl(my);
#pragma unroll
      for (int j = 0; j < NL; j++) {                               // lmp.c:143-149
        const double t = v[(R - j + NL) % NL] - my;
        sy += t * t;
      }
    };
    if (jl == 0) sums(std::integral_constant<int, 0>{});
    if constexpr (NL > 1) { if (jl == 1) sums(std::integral_constant<int, 1 % (NL > 0 ? NL : 1)>{}); }
    if constexpr (NL > 2) { if (jl == 2) sums(std::integral_constant<int, 2 % (NL > 0 ? NL : 1)>{}); }
    if constexpr (NL > 3) { if (jl == 3) sums(std::integral_constant<int, 3 % (NL > 0 ? NL : 1)>{}); }
    if constexpr (NL > 4) { if (jl == 4) sums(std::integral_constant<int, 4 % (NL > 0 ? NL : 1)>{}); }
    if constexpr (NL > 5) { if (jl == 5) sums(std::integral_constant<int, 5 % (NL > 0 ? NL : 1)>{}); }
    if constexpr (NL > 6) { if (jl == 6) sums(std::integral_constant<int, 6 % (NL > 0 ? NL : 1)>{}); }
    if constexpr (NL > 7) { if (jl == 7) sums(std::integral_constant<int, 7 % (NL > 0 ? NL : 1)>{}); }
    static_assert(NL <= 8, "rotations are spelled out up to 8");
  } else {
    int d = jl;                                                    // slot j holds frame f - ((jl - j) mod nl)
    for (int j = 0; j < nl; j++) {                                 // lmp.c:134-140
      const float v = (long long)d <= f ? *(col - (size_t)d * bins) : 0.0f;
      my += v;
      d = d == 0 ? nl - 1 : d - 1;
    }
    my = by_nl(my);
    d = jl;
    for (int j = 0; j < nl; j++) {                                 // lmp.c:143-149
      const float v = (long long)d <= f ? *(col - (size_t)d * bins) : 0.0f;
      sy += (v - my) * (v - my);
      d = d == 0 ? nl - 1 : d - 1;
    }
  }
  sy = by_nl1(sy);
  double v_hat = my * my - sy;                                     // lmp.c:153-159
  if (v_hat < 0.0) v_hat = 0.0;
  v_hat = 0.5 * (my - sqrt(v_hat));
  float r = c_neg + (nl * my) / (c_den * v_hat);
  if (r <= 1.0e-3) r = 1e-3;
  o[i] = r;
}

// The same statistic, a thread walking G consecutive frames of its bin with the ring in registers,
// indexed by SLOT as the reference's is (slot = frame mod nl): a group starts at a frame that is a
// multiple of NL, so the slot a frame goes into is known at compile time and the sums run over the
// slots in slot order with no rotation at all.  One row read per frame (plus NL - 1 per group)
// instead of NL: lmp_kernel's blocks of consecutive frames land on different XCDs, each with its own
// L2, and the re-reads came from HBM -- 40 KB per frame where 16 are needed.
template <int NL, int G>
__global__ __launch_bounds__(256) void lmp_ring_kernel(const float *__restrict__ rows, long long row0, long long first,
                                                       long long nframes, int bins, double c_neg, double c_den,
                                                       double recip_nl, double recip_nl1, float *__restrict__ out) {
  static_assert(G % NL == 0, "a group is whole turns of the ring");
  const int i = blockIdx.x * 256 + threadIdx.x;
  const long long f0 = first + (long long)blockIdx.y * G;          // a multiple of NL (the launcher sees to `first`)
  const long long end = first + nframes;
  if (i >= bins || f0 >= end) return;
  const float *col = rows + (size_t)(f0 - row0) * bins + i;        // row of frame f0; frame f0 + k is k rows down
  float w[NL], r[G];
#pragma unroll
  for (int j = 0; j < NL; j++)     // frame f0 - NL + j sits in slot j; slot 0 is f0's own before it is ever summed (and its old row may lie before `rows`)
    w[j] = (j > 0 && f0 - NL + j >= 0) ? *(col - (size_t)(NL - j) * bins) : 0.0f;
#pragma unroll
  for (int k = 0; k < G; k++) r[k] = f0 + k < end ? col[(size_t)k * bins] : 0.0f;
  const SmallDivisor by_nl((double)NL, recip_nl), by_nl1((double)(NL - 1), recip_nl1);
#pragma unroll
  for (int k = 0; k < G; k++) {
    if (f0 + k >= end) return;
    w[k % NL] = r[k];
    float *o = out + (size_t)(f0 + k - first) * bins;
    if (i == 0) {                                                  // lmp.c:160
      o[0] = 1e-3;
      continue;
    }
    double my = 0.0, sy = 0.0;
#pragma unroll
    for (int j = 0; j < NL; j++) my += w[j];                       // lmp.c:134-140
    my = by_nl(my);
#pragma unroll
    for (int j = 0; j < NL; j++) {                                 // lmp.c:143-149
      const double t = w[j] - my;
      sy += t * t;
    }
    sy = by_nl1(sy);
    double v_hat = my * my - sy;                                   // lmp.c:153-159
    if (v_hat < 0.0) v_hat = 0.0;
    v_hat = 0.5 * (my - sqrt(v_hat));
    float q = c_neg + (NL * my) / (c_den * v_hat);
    if (q <= 1.0e-3) q = 1e-3;
    o[i] = q;
  }
}

// The same for ANY ring size up to 64 (round 4; lmp_av is a free entry of glfer's options dialog): the thread's ring in LDS
// ([slot][thread]: a column of its own, no barrier), the slot of a frame taken as frame mod nl at run time, the sums over the slots in
// slot order by loops.  One row read per frame (plus nl - 1 per group of G) where lmp_kernel<0> reads 2 nl: lmp_av = 16 at N = 4096
// ran at 16.5 M frames/s, a tenth of the rate of the periodograms under it.
__global__ __launch_bounds__(256) void lmp_ring_any_kernel(const float *__restrict__ rows, long long row0, long long first, long long nframes,
                                                           int bins, int nl, int G, double c_neg, double c_den, double recip_nl,
                                                           double recip_nl1, float *__restrict__ out) {
  extern __shared__ float ring[];                                  // [nl][256]
  const int i = blockIdx.x * 256 + threadIdx.x;
  const long long f0 = first + (long long)blockIdx.y * G;
  const long long last = first + nframes, end = f0 + G < last ? f0 + G : last;
  if (f0 >= last) return;
  const bool live = i < bins;
  float *w = ring + threadIdx.x;
  const float *col = rows + (live ? (size_t)i : 0);
  // what the ring holds when frame f0 arrives: frames f0 - nl + 1 .. f0 - 1 in their slots (zero before the stream), f0's own slot still to come
  for (int d = 1; d < nl; d++) {
    const long long f = f0 - d;
    const int slot = (int)(((f % nl) + nl) % nl);
    w[slot * 256] = (f >= 0 && live) ? col[(size_t)(f - row0) * bins] : 0.0f;
  }
  const SmallDivisor by_nl((double)nl, recip_nl), by_nl1((double)(nl - 1), recip_nl1);
  int slot = (int)(f0 % nl);
  for (long long f = f0; f < end; f++) {
    w[slot * 256] = live ? col[(size_t)(f - row0) * bins] : 0.0f;
    slot = slot + 1 == nl ? 0 : slot + 1;
    if (!live) continue;
    float *o = out + (size_t)(f - first) * bins;
    if (i == 0) {                                                  // lmp.c:160
      o[0] = 1e-3;
      continue;
    }
    double my = 0.0, sy = 0.0;
    for (int j = 0; j < nl; j++) my += w[j * 256];                 // lmp.c:134-140
    my = by_nl(my);
    for (int j = 0; j < nl; j++) {                                 // lmp.c:143-149
      const double t = w[j * 256] - my;
      sy += t * t;
    }
    sy = by_nl1(sy);
    double v_hat = my * my - sy;                                   // lmp.c:153-159
    if (v_hat < 0.0) v_hat = 0.0;
    v_hat = 0.5 * (my - sqrt(v_hat));
    float q = c_neg + (nl * my) / (c_den * v_hat);
    if (q <= 1.0e-3) q = 1e-3;
    o[i] = q;
  }
}

// One thread per (frame, bin).  spec: [ntap + 1][nframes][n] halfcomplex spectra (fft_radix2.c
// layout) of the frame under taper j, the last one under hn (mu); mu_live = 0: mu is all zeros (the
// reference build without FFTW never writes it, mtm.c:173).
__global__ __launch_bounds__(256) void ftest_kernel(const float *__restrict__ spec, long long nframes, int n, int ntap,
                                                    const double *__restrict__ U0, float sum_U0_sqr, int mu_live,
                                                    float *__restrict__ ftest) {
  const long long fi = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int bins = n / 2 + 1;
  if (fi >= nframes || i >= bins) return;
  const int k = ntap - 1;                                          // params->kmax
  const size_t fstride = (size_t)nframes * n;
  const float *mu = spec + (size_t)ntap * fstride + (size_t)fi * n;
  const float mur = mu_live ? mu[i] : 0.0f;                        // mu[i]
  const float mui = mu_live ? mu[(n - i) % n] : 0.0f;              // mu[n_fft - i]  (i = 0: unused)
  float ft = 0.0;                                                  // mtm.c:179-186
  double tmpr, tmpi, num_ftest;
  if (i < (n + 1) / 2) {
    for (int j = 0; j < ntap; j++) {                               // mtm.c:203-210
      const float *ob = spec + (size_t)j * fstride + (size_t)fi * n;
      if (i == 0) {
        tmpr = ob[0] - mur * U0[j];
        ft += tmpr * tmpr;
      } else {
        tmpr = ob[i] - mur * U0[j];
        tmpi = ob[n - i] - mui * U0[j];
        ft += tmpr * tmpr + tmpi * tmpi;
      }
    }
  }
  if (i == 0) num_ftest = k * (mur * mur) * sum_U0_sqr;            // mtm.c:222-223
  else num_ftest = k * (mur * mur + mui * mui) * sum_U0_sqr;       // mtm.c:225, 230 (the Nyquist bin counts mu[n/2] twice)
  ftest[(size_t)fi * bins + i] = num_ftest / ft;
}

// prepare_audio's output (fft.c:98-156) for frames [frame0, frame0 + nframes): history / zero
// history, RA9MB, window (NULL = rectangular: no multiply, fft.c:132,139), limiter.
template <int FMT>
__global__ __launch_bounds__(256) void prepare_kernel(SpectroParams p, int n, const float *__restrict__ window,
                                                      float *__restrict__ out) {
  const long long fi = blockIdx.x;
  if (fi >= p.nframes) return;
  const long long s0 = (p.frame0 + fi) * (long long)p.H - p.R;
  constexpr int esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  const char *base = reinterpret_cast<const char *>(p.stream);
  for (int i = threadIdx.x; i < n; i += 256) {
    const long long s = s0 + i;
    float x = 0.0f;
    if (s >= 0 && !(p.history_mode && i < p.R)) {
      if constexpr (FMT == GLFER_FMT_F32) x = *reinterpret_cast<const float *>(base + s * esz);
      else if constexpr (FMT == GLFER_FMT_S16) x = (float)*reinterpret_cast<const short *>(base + s * esz) / 32768;
      else x = ((float)*reinterpret_cast<const unsigned char *>(base + s) - 128) / 128;
    }
    float y;
    if (p.a > 0.0) {                                               // fft.c:127-136
      y = x / (p.a + x * x);
      if (window) y *= window[i];
    } else if (window) {                                           // fft.c:139-146
      y = window[i] * x;
    } else {                                                       // fft.c:147-148
      y = x;
    }
    if (p.limiter) {                                               // fft.c:151-156
      const float ftmp = log(fabs(y));
      y = (y > 0 ? exp(ftmp * 0.1) : -exp(ftmp * 0.1));
    }
    out[(size_t)fi * n + i] = y;
  }
}

}  // namespace glfer

using namespace glfer;

extern "C" hipError_t glfer_launch_lmp(const float *rows, long long row0, long long first, size_t nframes, int bins,
                                       int nl, float *out, hipStream_t st) {
  if (nframes == 0) return hipSuccess;
  if (nl < 1 || bins < 1 || nframes > 65535u * 65535ull) return hipErrorInvalidValue;
  const double c_neg = -sqrt(nl / 2.0), c_den = 2.0 * sqrt(2.0 * nl), recip_nl = 1.0 / (double)nl, recip_nl1 = 1.0 / (double)(nl - 1);
  // the ring sizes with a register form: the frames up to the first multiple of nl one by one (at most
  // nl - 1 of them, below), the rest in groups through lmp_ring_kernel
  if ((nl == 2 || nl == 3 || nl == 4 || nl == 8) && GLFER_LMP_RING) {
    const size_t head = std::min<size_t>(nframes, (size_t)((nl - first % nl) % nl));
    const size_t body = nframes - head;
    if (body) {
      const long long f0 = first + (long long)head;
      float *o = out + head * (size_t)bins;
#define GLFER_LMP_RING_LAUNCH(NLC, GC)                                                                               \
  do {                                                                                                               \
    const size_t groups = (body + GC - 1) / GC;                                                                      \
    for (size_t g0 = 0; g0 < groups; g0 += 65535) {                                                                  \
      const size_t ng = std::min<size_t>(65535, groups - g0);                                                        \
      const long long fg = f0 + (long long)(g0 * GC);                                                                \
      hipLaunchKernelGGL((lmp_ring_kernel<NLC, GC>), dim3((unsigned)((bins + 255) / 256), (unsigned)ng), dim3(256), 0, st, rows, row0, \
                         fg, (long long)std::min<size_t>(ng * GC, body - g0 * GC), bins, c_neg, c_den, recip_nl, recip_nl1,            \
                         o + g0 * GC * (size_t)bins);                                                                \
    }                                                                                                                \
  } while (0)
      if (nl == 2) GLFER_LMP_RING_LAUNCH(2, 16);
      else if (nl == 3) GLFER_LMP_RING_LAUNCH(3, 15);
      else if (nl == 4) GLFER_LMP_RING_LAUNCH(4, 16);
      else GLFER_LMP_RING_LAUNCH(8, 16);
#undef GLFER_LMP_RING_LAUNCH
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
    }
    nframes = head;                                               // what is left for the frame-by-frame kernel
    if (nframes == 0) return hipSuccess;
  }
  if (nl > 1 && nl <= 64 && nframes >= 64 && GLFER_LMP_RING) {     // any other ring size: the ring in LDS (lmp_ring_any_kernel)
    size_t G = 64;
    while ((nframes + G - 1) / G > 65535) G *= 2;
    const size_t lds = (size_t)nl * 256 * sizeof(float);
    hipError_t e = glfer::allow_dynamic_lds((const void *)lmp_ring_any_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lmp_ring_any_kernel, dim3((unsigned)((bins + 255) / 256), (unsigned)((nframes + G - 1) / G)), dim3(256), lds, st, rows, row0,
                       first, (long long)nframes, bins, nl, (int)G, c_neg, c_den, recip_nl, recip_nl1, out);
    return hipGetLastError();
  }
  // blockIdx.y carries the frame: at most 65535 per launch
  for (size_t done = 0; done < nframes; done += 65535) {
    const size_t nf = nframes - done < 65535 ? nframes - done : 65535;
    const long long f0 = first + (long long)done;
    const dim3 grid((unsigned)((bins + 255) / 256), (unsigned)nf);
#define GLFER_LMP(NLC)                                                                                       \
  hipLaunchKernelGGL(lmp_kernel<NLC>, grid, dim3(256), 0, st, rows, row0, f0, (long long)nf, bins, nl, (int)(f0 % nl), \
                     c_neg, c_den, recip_nl, recip_nl1, out + done * (size_t)bins)
    switch (nl) {
      case 2: GLFER_LMP(2); break;
      case 3: GLFER_LMP(3); break;
      case 4: GLFER_LMP(4); break;                                   // the reference's default (glfer.c:252)
      case 8: GLFER_LMP(8); break;
      default: GLFER_LMP(0); break;
    }
#undef GLFER_LMP
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

extern "C" hipError_t glfer_launch_ftest(const float *spec, size_t nframes, int n, int ntap, const double *U0,
                                         float sum_U0_sqr, int mu_live, float *ftest, hipStream_t st) {
  if (nframes == 0) return hipSuccess;
  if (nframes > 65535 || ntap < 1) return hipErrorInvalidValue;    // the caller chunks the frames
  hipLaunchKernelGGL(ftest_kernel, dim3((unsigned)((n / 2 + 1 + 255) / 256), (unsigned)nframes), dim3(256), 0, st, spec,
                     (long long)nframes, n, ntap, U0, sum_U0_sqr, mu_live, ftest);
  return hipGetLastError();
}

extern "C" hipError_t glfer_launch_prepare(const SpectroParams *p, int n, const float *window, float *out,
                                           hipStream_t st) {
  if (p->nframes <= 0) return hipSuccess;
  const dim3 grid((unsigned)p->nframes), block(256);
  switch (p->fmt) {
    case GLFER_FMT_F32: hipLaunchKernelGGL(prepare_kernel<GLFER_FMT_F32>, grid, block, 0, st, *p, n, window, out); break;
    case GLFER_FMT_S16: hipLaunchKernelGGL(prepare_kernel<GLFER_FMT_S16>, grid, block, 0, st, *p, n, window, out); break;
    case GLFER_FMT_U8: hipLaunchKernelGGL(prepare_kernel<GLFER_FMT_U8>, grid, block, 0, st, *p, n, window, out); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
