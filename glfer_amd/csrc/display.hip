// display.hip -- the waterfall column mapping that follows the estimator in the reference's draw
// routine (main_window_draw, g_main.c:1099-1236), batched over frames:
//   K7a levels_kernel : the autoscale recurrence on (sig, floor) -> display_max/min per frame
//                       (g_main.c:1111-1139).  The recurrence rounds to float at every frame, so
//                       it is inherently sequential: one wave walks the frames, its 64 lanes
//                       only stage the inputs and outputs through LDS.
//   K7b map_kernel    : PSD (or averaged PSD) -> dB short (levbuf) -> 0..255 -> palette RGB
//                       (g_main.c:1186-1236).  One block per frame, HBM-bound byte/short stores;
//                       the 768-byte palette sits in LDS.
// Built with -ffp-contract=off: the reference's float/double expression order is the contract.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace glfer {

// double -> int32 the way the reference's implicit double->short / double->unsigned char
// conversions behave on x86-64 (cvttsd2si: out of range or NaN gives INT_MIN; low bits kept)
__device__ __forceinline__ int x86_d2i(double d) {
  if (!(d > -2147483649.0 && d < 2147483648.0)) return (int)0x80000000u;
  return (int)d;
}

struct LevelsParams {
  int scale_log, autoscale, first_buffer;
  float overlap;
  float max_lvl0, min_lvl0;      // state carried in (autoscale) or the fixed levels (not autoscale)
};

// levels: [nframes][4] = {display_max, display_min, display_max_lvl, display_min_lvl}
__global__ __launch_bounds__(64) void levels_kernel(const float *__restrict__ stats, long long nframes,
                                                    LevelsParams p, float *__restrict__ levels) {
  __shared__ float sx[2][64];
  __shared__ float sy[2][64];
  const int lane = threadIdx.x;
  float lvl = (lane == 0) ? p.max_lvl0 : p.min_lvl0;       // lanes 0 / 1 carry the two chains
  bool first = p.first_buffer != 0;
  for (long long base = 0; base < nframes; base += 64) {
    const long long f = base + lane;
    if (f < nframes) {
      sx[0][lane] = stats[f * 4 + 0];
      sx[1][lane] = stats[f * 4 + 1];
    }
    __syncthreads();
    if (lane < 2) {
      const int cnt = (int)((nframes - base < 64) ? (nframes - base) : 64);
      for (int j = 0; j < cnt; j++) {
        float x = sx[lane][j];
        if (p.autoscale) {
          if (first) {                                     // g_main.c:1112-1120
            if (p.overlap > 0.0) x /= p.overlap;
            lvl = x;
            first = false;
          } else {                                         // g_main.c:1122-1123
            lvl = (float)((1.0 - 0.99) * (double)x + 0.99 * (double)lvl);
          }
        }
        sy[lane][j] = lvl;
      }
    }
    __syncthreads();
    if (f < nframes) {
      const float mx = sy[0][lane], mn = sy[1][lane];
      float *o = levels + f * 4;
      o[0] = p.scale_log ? (float)(10.0 * log10((double)mx)) : mx;   // g_main.c:1132-1139
      o[1] = p.scale_log ? (float)(10.0 * log10((double)mn)) : mn;
      o[2] = mx;
      o[3] = mn;
    }
    __syncthreads();
  }
}

// autoscale off: the levels are the same for every frame (g_main.c:1125-1139, evaluated on the host)
__global__ __launch_bounds__(256) void levels_fixed_kernel(long long nframes, float dmax, float dmin,
                                                           float max_lvl, float min_lvl,
                                                           float *__restrict__ levels) {
  for (long long f = blockIdx.x * 256ll + threadIdx.x; f < nframes; f += (long long)gridDim.x * 256) {
    float *o = levels + f * 4;
    o[0] = dmax; o[1] = dmin; o[2] = max_lvl; o[3] = min_lvl;
  }
}

template <typename SRC>
__global__ __launch_bounds__(256) void map_kernel(const SRC *__restrict__ src, int n, int scale_log,
                                                  double thr255, double one_m_thr,
                                                  const float *__restrict__ levels,
                                                  const unsigned char *__restrict__ colortab,
                                                  unsigned char *__restrict__ rgb, short *__restrict__ lev) {
  __shared__ unsigned char tab[768];
  for (int i = threadIdx.x; i < 768; i += 256) tab[i] = colortab[i];
  __syncthreads();
  const size_t fr = blockIdx.x;
  const SRC *row = src + fr * (size_t)n;
  const float display_max = levels[fr * 4 + 0];
  const float display_min = levels[fr * 4 + 1];
  const float span = display_max - display_min;
  unsigned char *orow = rgb + fr * (size_t)n * 3;
  short *lrow = lev ? lev + fr * (size_t)n : nullptr;
  for (int i = threadIdx.x; i < n; i += 256) {
    const SRC s = row[n - i - 1];
    float sig_level;
    short l;
    if (scale_log) {
      l = (short)x86_d2i(10.0 * log10((double)s));         // levbuf[..] = 10.0*log10(x)
      sig_level = (float)l;                                // sig_level = (that short)
    } else {
      sig_level = (float)s;
      l = (short)x86_d2i(10.0 * log10((double)sig_level));
    }
    const float f = 255.0f * ((sig_level - display_min) / span);
    unsigned char v;
    if ((double)f < thr255)
      v = 0;
    else if (f > 255.0f)
      v = 255;
    else
      v = (unsigned char)x86_d2i(((double)f - thr255) / one_m_thr);
    orow[3 * i] = tab[3 * v];
    orow[3 * i + 1] = tab[3 * v + 1];
    orow[3 * i + 2] = tab[3 * v + 2];
    if (lrow) lrow[i] = l;
  }
}

}  // namespace glfer

using namespace glfer;

extern "C" hipError_t glfer_launch_levels(const float *stats, size_t nframes, int scale_log, int autoscale,
                                          int first_buffer, float overlap, float max_lvl0, float min_lvl0,
                                          float *levels, hipStream_t st) {
  if (nframes == 0) return hipSuccess;
  LevelsParams p{scale_log, autoscale, first_buffer, overlap, max_lvl0, min_lvl0};
  hipLaunchKernelGGL(levels_kernel, dim3(1), dim3(64), 0, st, stats, (long long)nframes, p, levels);
  return hipGetLastError();
}

extern "C" hipError_t glfer_launch_levels_fixed(size_t nframes, float dmax, float dmin, float max_lvl,
                                                float min_lvl, float *levels, hipStream_t st) {
  if (nframes == 0) return hipSuccess;
  const unsigned grid = (unsigned)((nframes + 255) / 256 < 4096 ? (nframes + 255) / 256 : 4096);
  hipLaunchKernelGGL(levels_fixed_kernel, dim3(grid), dim3(256), 0, st, (long long)nframes, dmax, dmin,
                     max_lvl, min_lvl, levels);
  return hipGetLastError();
}

extern "C" hipError_t glfer_launch_map(const float *psd, const double *avg, size_t nframes, int n,
                                       int scale_log, double thr255, double one_m_thr, const float *levels,
                                       const unsigned char *colortab, unsigned char *rgb, short *lev,
                                       hipStream_t st) {
  if (nframes == 0) return hipSuccess;
  if (avg)
    hipLaunchKernelGGL(map_kernel<double>, dim3((unsigned)nframes), dim3(256), 0, st, avg, n, scale_log,
                       thr255, one_m_thr, levels, colortab, rgb, lev);
  else
    hipLaunchKernelGGL(map_kernel<float>, dim3((unsigned)nframes), dim3(256), 0, st, psd, n, scale_log,
                       thr255, one_m_thr, levels, colortab, rgb, lev);
  return hipGetLastError();
}
