// spectro_kernels.hip -- fused frame -> taper(s) -> FFT -> |X|^2 -> taper-sum kernels (gfx950).
//
// Replaces, per audio frame, the reference chain
//   prepare_audio (fft.c:66-165) -> fft_real_radix2_transform (fft_radix2.c:75-177)
//   -> fft_psd (fft.c:203-226) [-> the taper loop of mtm_do, mtm.c:189-220]
// with ONE kernel that reads the overlapped sample stream once and writes N/2+1 PSD bins.
//
// Decomposition (N = 64*W, W = N/64 lanes per frame, 64 complex points per lane):
//   * two real tapered copies of the frame are packed as re/im of one complex
//     N-point transform (taper 2p -> re, taper 2p+1 -> im);
//   * pass 1: each lane owns the stride-W subsequence n = W*r + t and does a
//     64-point DFT in registers; multiply by W_N^(t*k1);
//   * one exchange through LDS (wave-local, no workgroup barrier for W <= 64);
//   * pass 2: each lane does 64/W DFTs of length W in registers;
//   * acc[k] += |Z_k|^2.  Because the weights are folded into the tapers,
//     sum_j w_j |Y_j[k]|^2 = sum_pairs (|Z_k|^2 + |Z_{N-k}|^2)/2, so the real/imag
//     separation is never done per taper: one mirror-add through LDS per FRAME.
// No MFMA: there is no dense contraction on this path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft_inreg.hpp"
#include "spectro_params.h"

#ifndef GLFER_WAVES_PER_SIMD
#define GLFER_WAVES_PER_SIMD 2
#endif
namespace glfer {

// sample formats: wav_fmt.c:104-117
template <int FMT>
__device__ __forceinline__ float load_sample(const void *base, long long idx) {
  if constexpr (FMT == GLFER_FMT_F32) {
    return reinterpret_cast<const float *>(base)[idx];
  } else if constexpr (FMT == GLFER_FMT_S16) {
    return (float)reinterpret_cast<const short *>(base)[idx] / 32768.0f;
  } else {
    return ((float)reinterpret_cast<const unsigned char *>(base)[idx] - 128.0f) / 128.0f;
  }
}

// ---------------------------------------------------------------------------
// Two-pass kernel, N = 64*W, W in {1,2,4,...,64}.  Block = 256 threads = 4 waves;
// a wave never talks to another wave, so there is no __syncthreads().
template <int W, int FMT, bool GEN, int WPS = GLFER_WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WPS) void spectro2_kernel(SpectroParams p) {
  constexpr int N = 64 * W;
  constexpr int G = 64 / W;            // pass-2 transforms per lane
  constexpr int FPB = 256 / W;         // frames per block
  constexpr int LDW = W + 1;           // padded row length (dwords)
  constexpr int REGION = 64 * LDW + (W < 32 ? 32 / 2 : 0);  // per-frame LDS dwords (+skew)
  __shared__ float lds[FPB * REGION];

  const int tid = threadIdx.x;
  const int t = tid % W;               // lane within the frame
  const int fl = tid / W;              // frame within the block
  const long long f = (long long)blockIdx.x * FPB + fl;
  const bool live = f < p.nframes;
  float *xch = lds + fl * REGION;

  // frame-relative sample j = W*r + t  <->  stream index s0 + j
  const long long s0 = (p.frame0 + f) * (long long)p.H - p.R;

  float acc[64];
#pragma unroll
  for (int r = 0; r < 64; r++) acc[r] = 0.0f;

  for (int pair = 0; pair < p.npairs; pair++) {
    float zr[64], zi[64];
    const float *ta = p.taps + (size_t)(2 * pair) * N;
    const float *tb = ta + N;
#pragma unroll
    for (int r = 0; r < 64; r++) {
      const int j = W * r + t;
      const long long s = s0 + j;
      const bool ok = live && (p.history_mode ? (j >= p.R) : (s >= 0));
      float x = ok ? load_sample<FMT>(p.stream, s) : 0.0f;
      if (GEN && p.nonlin) {
        // fft.c:127-156: RA9MB x/(a+x^2), window, then sign(y)*|y|^0.1; unit-power
        // scaling is applied afterwards (post_scale) because the limiter is not linear.
        if (p.a > 0.0f) x = x / (p.a + x * x);
        float y = x * ta[j];
        if (p.limiter) {
          float m = __expf(0.1f * __logf(fabsf(y)));
          y = (y > 0.0f) ? m : -m;
        }
        zr[r] = y * p.post_scale;
        zi[r] = 0.0f;
      } else {
        zr[r] = x * ta[j];
        zi[r] = x * tb[j];
      }
    }

    // ---- pass 1: 64-point DFT over r; X[k1] lands at register brev(k1,64)
    dit<64, 1, 0>(zr, zi);

    // ---- twiddle W_N^(t*k1) and scatter to LDS: row k1, column t
    if constexpr (W > 1) {
      static_for<1, 64>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value;
        constexpr int q = brev(k1, 64);
        const float2 w = p.tw[k1 * W + t];
        const float a = zr[q], b = zi[q];
        zr[q] = __builtin_fmaf(a, w.x, -b * w.y);
        zi[q] = __builtin_fmaf(a, w.y, b * w.x);
      });
      // real parts, then imaginary parts, through the same wave-private region
      static_for<0, 64>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value;
        xch[k1 * LDW + t] = zr[brev(k1, 64)];
      });
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // lane a now owns k1 = a + W*g; register g*W + n2
      static_for<0, 64>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        constexpr int g = r / W, n2 = r % W;
        zr[r] = xch[(t + W * g) * LDW + n2];
      });
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      static_for<0, 64>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value;
        xch[k1 * LDW + t] = zi[brev(k1, 64)];
      });
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      static_for<0, 64>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        constexpr int g = r / W, n2 = r % W;
        zi[r] = xch[(t + W * g) * LDW + n2];
      });
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();

      // ---- pass 2: G transforms of length W over n2; X[k2] at g*W + brev(k2,W)
      static_for<0, G>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        dit<W, 1, g * W>(zr, zi);
      });
    }

    if constexpr (GEN) {
      // halfcomplex spectrum of the (single, real) tapered frame: fft_radix2.c layout
      // data[k] = Re X_k (k<=N/2), data[N-k] = Im X_k (0<k<N/2).  Debug/compat output.
      if (live && p.spec) {
        float *o = p.spec + (size_t)f * N;
        const float inv = 1.0f / p.spec_unscale;
        static_for<0, 64>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          constexpr int g = (W > 1) ? r / W : 0;
          constexpr int k2 = (W > 1) ? brev(r % W, W) : 0;
          const int k = (W > 1) ? (t + W * g) + 64 * k2 : brev(r, 64);
          if (k <= N / 2) o[k] = zr[r] * inv;
          if (k > 0 && k < N / 2) o[N - k] = zi[r] * inv;
        });
      }
    }

#pragma unroll
    for (int r = 0; r < 64; r++)
      acc[r] = __builtin_fmaf(zr[r], zr[r], __builtin_fmaf(zi[r], zi[r], acc[r]));
  }

  // ---- mirror fold: psd[k] = acc[k] + acc[(N-k) mod N]  (scales are in the tapers)
  static_for<0, 64>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    constexpr int g = (W > 1) ? r / W : 0;
    constexpr int k2 = (W > 1) ? brev(r % W, W) : 0;
    const int k = (W > 1) ? (t + W * g) + 64 * k2 : brev(r, 64);
    xch[k] = acc[r];
  });
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (live) {
    float *o = p.psd + (size_t)f * (N / 2 + 1);
#pragma unroll
    for (int r = 0; r < 32; r++) {
      const int k = W * r + t;
      o[k] = xch[k] + xch[(N - k) & (N - 1)];
    }
    if (t == 0) o[N / 2] = 2.0f * xch[N / 2];
  }
}

// ---------------------------------------------------------------------------
// K0: per-hop mean removal (fft.c:86-96).  One block per hop; writes a float copy
// of the stream (the reference mutates the caller's hop buffer in place).
template <int FMT>
__global__ __launch_bounds__(256) void submean_kernel(const void *in, float *out, int H,
                                                      long long nhops) {
  __shared__ float part[256];
  const long long hop = blockIdx.x;
  if (hop >= nhops) return;
  const long long base = hop * (long long)H;
  float s = 0.0f;
  for (int i = threadIdx.x; i < H; i += 256) s += load_sample<FMT>(in, base + i);
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  const float mean = part[0] / (float)H;
  for (int i = threadIdx.x; i < H; i += 256) out[base + i] = load_sample<FMT>(in, base + i) - mean;
}

}  // namespace glfer

// ---------------------------------------------------------------------------
// host-side launchers (called from glfer_hip.cpp).  One translation unit per W
// (-DGLFER_W=...) so the lanes-per-frame variants compile in parallel.
using namespace glfer;

#ifndef GLFER_NO_LAUNCHERS
#ifndef GLFER_W
#error "compile with -DGLFER_W=<lanes per frame>"
#endif
#define GLFER_CAT2(a, b) a##b
#define GLFER_CAT(a, b) GLFER_CAT2(a, b)

template <int FMT>
static hipError_t launch2_fmt(const SpectroParams &p, hipStream_t st) {
  constexpr int W = GLFER_W;
  constexpr int FPB = 256 / W;
  const unsigned grid = (unsigned)((p.nframes + FPB - 1) / FPB);
  if (grid == 0) return hipSuccess;
  if (p.nonlin || p.spec)
    hipLaunchKernelGGL((spectro2_kernel<W, FMT, true>), dim3(grid), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((spectro2_kernel<W, FMT, false>), dim3(grid), dim3(256), 0, st, p);
  return hipGetLastError();
}

extern "C" hipError_t GLFER_CAT(glfer_launch_spectro2_w, GLFER_W)(const SpectroParams *p, hipStream_t st) {
  switch (p->fmt) {
    case GLFER_FMT_F32: return launch2_fmt<GLFER_FMT_F32>(*p, st);
    case GLFER_FMT_S16: return launch2_fmt<GLFER_FMT_S16>(*p, st);
    case GLFER_FMT_U8: return launch2_fmt<GLFER_FMT_U8>(*p, st);
  }
  return hipErrorInvalidValue;
}

#if GLFER_W == 64
extern "C" hipError_t glfer_launch_submean(const void *in, float *out, int H, long long nhops,
                                           int fmt, hipStream_t st) {
  if (nhops <= 0) return hipSuccess;
  switch (fmt) {
    case GLFER_FMT_F32: hipLaunchKernelGGL((submean_kernel<GLFER_FMT_F32>), dim3((unsigned)nhops), dim3(256), 0, st, in, out, H, nhops); break;
    case GLFER_FMT_S16: hipLaunchKernelGGL((submean_kernel<GLFER_FMT_S16>), dim3((unsigned)nhops), dim3(256), 0, st, in, out, H, nhops); break;
    case GLFER_FMT_U8: hipLaunchKernelGGL((submean_kernel<GLFER_FMT_U8>), dim3((unsigned)nhops), dim3(256), 0, st, in, out, H, nhops); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
#endif
#endif  // GLFER_NO_LAUNCHERS
