// spectro_small.hip -- block sizes below the 16-points-per-lane kernels' range (N = 8 .. 128).
// The reference takes any power of two the user types (g_options.c:386-387, fft_radix2.c:27-44);
// nobody runs a waterfall at these sizes for speed, so this is the plain form: N/2 lanes per frame,
// the frame in LDS, radix-2 decimation in time with a twiddle table, everything spectro16.hip does
// (taper pairs packed as re/im, zero history, RA9MB / limiter, halfcomplex spectrum output) in the
// same arithmetic: tapers carry sqrt(1/(2N(1+sig))), psd[k] = acc[k] + acc[N-k].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "stockham16.hpp"

namespace glfer {

template <int FMT>
__global__ __launch_bounds__(256) void spectro_small_kernel(SpectroParams p, int logn, const float *__restrict__ staps) {
  const int N = 1 << logn, LPF = N / 2, FPB = 256 / LPF;        // lanes per frame, frames per block
  __shared__ v2f32 zbuf[256 * 2];                                 // FPB frames of N points
  __shared__ float accb[256 * 2];
  __shared__ v2f32 twid[64];                                      // exp(-2 pi i k / N), k < N/2
  constexpr int esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  const int tid = threadIdx.x, fl = tid / LPF, l = tid % LPF;
  if (tid < N / 2) {
    float sn, cs;
    sincospif(-2.0f * (float)tid / (float)N, &sn, &cs);
    twid[tid] = v2f32{cs, sn};
  }
  __syncthreads();
  v2f32 *z = zbuf + fl * N;
  float *acc = accb + fl * N;
  const char *base = reinterpret_cast<const char *>(p.stream);
  for (long long f0 = (long long)blockIdx.x * FPB; f0 < p.nframes; f0 += (long long)gridDim.x * FPB) {
    const long long f = f0 + fl;
    const bool live = f < p.nframes;
    const long long s0 = (p.frame0 + (live ? f : f0)) * (long long)p.H - p.R;
    float x[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const int i = l + e * LPF;
      const long long s = s0 + i;
      float v = 0.0f;
      if (s >= 0 && !(p.history_mode && i < p.R)) {               // zero history: fft.c:103-108
        if constexpr (FMT == GLFER_FMT_F32) v = *reinterpret_cast<const float *>(base + s * esz);
        else if constexpr (FMT == GLFER_FMT_S16) v = (float)*reinterpret_cast<const short *>(base + s * esz) / 32768.0f;
        else v = ((float)*reinterpret_cast<const unsigned char *>(base + s) - 128.0f) / 128.0f;
      }
      x[e] = v;
      acc[i] = 0.0f;
    }
    for (int pr = 0; pr < p.npairs; pr++) {
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int i = l + e * LPF;
        const float ta = staps[((size_t)pr * N + i) * 2], tb = staps[((size_t)pr * N + i) * 2 + 1];
        float re, im;
        if (p.nonlin) {                                           // fft.c:127-156, as spectro16.hip
          float xv = x[e];
          if (p.a > 0.0f) xv = xv / (p.a + xv * xv);
          float y = xv * ta;
          if (p.limiter) y = limiter_value(y);
          re = y * p.post_scale;
          im = 0.0f;
        } else {
          re = x[e] * ta;
          im = x[e] * tb;
        }
        z[__brev((unsigned)i) >> (32 - logn)] = v2f32{re, im};
      }
      __syncthreads();
      for (int st = 0; st < logn; st++) {                         // radix-2 decimation in time
        const int half = 1 << st, j = l & (half - 1), i0 = ((l >> st) << (st + 1)) + j, i1 = i0 + half;
        const v2f32 w = twid[j << (logn - 1 - st)];
        const v2f32 a = z[i0], b = z[i1];
        const float tr = __builtin_fmaf(b.x, w.x, -b.y * w.y), ti = __builtin_fmaf(b.x, w.y, b.y * w.x);
        __syncthreads();
        z[i0] = v2f32{a.x + tr, a.y + ti};
        z[i1] = v2f32{a.x - tr, a.y - ti};
        __syncthreads();
      }
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int k = l + e * LPF;
        const v2f32 v = z[k];
        acc[k] = __builtin_fmaf(v.x, v.x, __builtin_fmaf(v.y, v.y, acc[k]));
        if (live && p.spec) {                                     // halfcomplex layout of fft_radix2.c:75-177
          float *o = p.spec + (size_t)f * N;
          const float inv = 1.0f / p.spec_unscale;
          if (k <= N / 2) o[k] = v.x * inv;
          if (k > 0 && k < N / 2) o[N - k] = v.y * inv;
        }
      }
      __syncthreads();
    }
    if (live) {
      float *o = p.psd + (size_t)f * (size_t)p.pitch;
      o[l] = acc[l] + acc[(N - l) & (N - 1)];
      if (l == 0) o[N / 2] = 2.0f * acc[N / 2];
    }
    __syncthreads();
  }
}

}  // namespace glfer

using namespace glfer;

extern "C" hipError_t glfer_launch_spectro_small(const SpectroParams *p, int n, const float *staps, hipStream_t st) {
  if (n < 8 || n > 128 || (n & (n - 1)) || !staps) return hipErrorInvalidValue;
  if (p->nframes <= 0) return hipSuccess;
  int logn = 0;
  while ((1 << logn) < n) logn++;
  const int fpb = 256 / (n / 2);
  const long long work = ((long long)p->nframes + fpb - 1) / fpb;
  const unsigned grid = (unsigned)(work < 8192 ? work : 8192);
  switch (p->fmt) {
    case GLFER_FMT_F32: hipLaunchKernelGGL(spectro_small_kernel<GLFER_FMT_F32>, dim3(grid), dim3(256), 0, st, *p, logn, staps); break;
    case GLFER_FMT_S16: hipLaunchKernelGGL(spectro_small_kernel<GLFER_FMT_S16>, dim3(grid), dim3(256), 0, st, *p, logn, staps); break;
    case GLFER_FMT_U8: hipLaunchKernelGGL(spectro_small_kernel<GLFER_FMT_U8>, dim3(grid), dim3(256), 0, st, *p, logn, staps); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
