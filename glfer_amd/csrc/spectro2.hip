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

// Launder a pointer through an empty asm so the optimiser cannot fold it back into
// "base + large constant": each 16-element chunk then addresses its loads as
// chunk_base + immediate (<= 4 KiB) instead of materialising (and hoisting, and spilling)
// one 64-bit address per element.
// The result is typed as a global (address space 1) pointer so the loads stay global_load.
typedef float v2f32 __attribute__((ext_vector_type(2)));
template <class T>
using gptr = const __attribute__((address_space(1))) T *;
template <class T>
__device__ __forceinline__ gptr<T> opaque(const T *p) {
  gptr<T> g = (gptr<T>)p;
  asm volatile("" : "+v"(g));
  return g;
}

// ---------------------------------------------------------------------------
// Two-pass kernel, N = 64*W, W in {4,...,64}.  Block = 256 threads = 4 waves; a frame never
// spans two waves, so there is no __syncthreads() anywhere.
//
// Register plan for 2 waves/SIMD (<= 256 VGPRs): z = 128, acc = 64, ~48 for data in flight.
// The tap loads land directly in the z registers (no staging), the samples come in
// double-buffered chunks of 16 and the twiddles in chunks of 16, fenced with
// sched_barrier so the scheduler cannot hoist every load to the top and spill.
template <int W, int FMT, bool GEN, bool FAST>
__device__ __forceinline__ void spectro2_body(const SpectroParams &p, float *lds) {
  constexpr int N = 64 * W;
  constexpr int G = 64 / W;            // pass-2 transforms per lane
  constexpr int FPB = 256 / W;         // frames per block
  constexpr int LDW = W + 1;           // padded row length (dwords)
  constexpr int REGION = 64 * LDW + (W < 32 ? 16 : 0);  // per-frame LDS dwords (+bank skew)

  const unsigned tid = threadIdx.x;
  const unsigned t = tid % W;          // lane within the frame
  const unsigned fl = tid / W;         // frame within the block
  const long long fblk = (long long)blockIdx.x * FPB;
  const long long f = fblk + fl;
  const bool live = f < p.nframes;
  const unsigned flc = live ? fl : (unsigned)(p.nframes - 1 - fblk);   // clamp: loads stay in range
  float *xch = lds + fl * REGION;

  // frame-relative sample j = W*r + t  <->  stream index s0 + j.  The block's first frame
  // starts at sblk (wave-uniform): with sblk >= 0 and history kept, every load is in range
  // and no per-element predicate is needed.
  const long long sblk = (p.frame0 + fblk) * (long long)p.H - p.R;
  const unsigned loff = flc * (unsigned)p.H + t;      // lane's offset from sblk, in samples
  // FAST: wave-uniform base pointer + one 32-bit lane offset + immediates (no 64-bit
  // per-element address arithmetic for the scheduler to hoist and spill)
  // (per-lane pointer + compile-time element offsets: the offsets fold into the loads'
  // immediate fields; an index like base[loff + W*r] in unsigned arithmetic would not)
  const float *xl = reinterpret_cast<const float *>(p.stream) + (FAST ? sblk : 0) + loff;
  const v2f32 *twl = reinterpret_cast<const v2f32 *>(p.tw) + t;

  float acc[64];
#pragma unroll
  for (int r = 0; r < 64; r++) acc[r] = 0.0f;

  for (int pair = 0; pair < p.npairs; pair++) {
    float zr[64], zi[64];
    const float *ta = p.taps + (size_t)(2 * pair) * N + t;
    const float *tb = ta + N;

    // ---- taps straight into the z registers
    static_for<0, 4>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      gptr<float> tac = opaque(ta + 16 * c * W);
      gptr<float> tbc = opaque(tb + 16 * c * W);
      static_for<0, 16>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        zr[16 * c + i] = tac[W * i];
        zi[16 * c + i] = tbc[W * i];
      });
    });
    // ---- samples, in chunks of 16
    auto load_x = [&](auto rc, gptr<float> xc) -> float {
      constexpr int r = decltype(rc)::value;
      if constexpr (FAST) {
        if constexpr (FMT == GLFER_FMT_F32) return xc[W * (r % 16)];
        else return load_sample<FMT>(p.stream, sblk + (loff + W * r));
      } else {
        const int j = W * r + (int)t;
        const long long s = sblk + (long long)(loff + W * r);
        const bool ok = p.history_mode ? (j >= p.R) : (s >= 0);
        return ok ? load_sample<FMT>(p.stream, s) : 0.0f;
      }
    };
    static_for<0, 4>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      float xv[16];
      gptr<float> xc = opaque(xl + 16 * c * W);
      static_for<0, 16>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        xv[i] = load_x(std::integral_constant<int, 16 * c + i>{}, xc);
      });
      static_for<0, 16>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int r = 16 * c + i;
        float x = xv[i];
        if (GEN && p.nonlin) {
          // fft.c:127-156: RA9MB x/(a+x^2), window, then sign(y)*|y|^0.1; the unit-power
          // scale is applied afterwards (post_scale) because the limiter is not linear.
          if (p.a > 0.0f) x = x / (p.a + x * x);
          float y = x * zr[r];
          if (p.limiter) {
            const float m = __expf(0.1f * __logf(fabsf(y)));
            y = (y > 0.0f) ? m : -m;
          }
          zr[r] = y * p.post_scale;
          zi[r] = 0.0f;
        } else {
          zr[r] *= x;
          zi[r] *= x;
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });

    // ---- pass 1: 64-point DFT over r; X[k1] lands at register brev(k1,64)
    dit<64, 1, 0, 64>(zr, zi);

    if constexpr (W > 1) {
      // ---- twiddle W_N^(t*k1), in chunks of 16 rows of the [64][W] table
      static_for<0, 4>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        v2f32 w[16];
        gptr<v2f32> twc = opaque(twl + 16 * c * W);
        static_for<0, 16>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          w[i] = twc[i * W];
        });
        static_for<0, 16>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          constexpr int k1 = 16 * c + i;
          if constexpr (k1 > 0) {
            constexpr int q = brev(k1, 64);
            const float a = zr[q], b = zi[q];
            zr[q] = __builtin_fmaf(a, w[i].x, -b * w[i].y);
            zi[q] = __builtin_fmaf(a, w[i].y, b * w[i].x);
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      });

      // ---- exchange through the wave-private LDS region: row k1, column t; real parts,
      // then imaginary parts.  Afterwards lane a owns k1 = a + W*g in registers g*W + n2.
      static_for<0, 64>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value;
        xch[k1 * LDW + t] = zr[brev(k1, 64)];
      });
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
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
        dit<W, 1, g * W, 64>(zr, zi);
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
          const int k = (W > 1) ? (int)(t + W * g) + 64 * k2 : brev(r, 64);
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
    const int k = (W > 1) ? (int)(t + W * g) + 64 * k2 : brev(r, 64);
    xch[k] = acc[r];
  });
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (live) {
    float *o = p.psd + (size_t)f * (N / 2 + 1);
#pragma unroll
    for (int r = 0; r < 32; r++) {
      const int k = W * r + (int)t;
      o[k] = xch[k] + xch[(N - k) & (N - 1)];
    }
    if (t == 0) o[N / 2] = 2.0f * xch[N / 2];
  }
}

template <int W, int FMT, bool GEN, int WPS = GLFER_WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WPS) void spectro2_kernel(SpectroParams p) {
  constexpr int FPB = 256 / W;
  constexpr int REGION = 64 * (W + 1) + (W < 32 ? 16 : 0);
  __shared__ float lds[FPB * REGION];
  // blocks whose first frame starts at or after sample 0 (all but the first few) and keep
  // their history need no per-element bounds predicate: wave-uniform choice of body
  const long long sblk = (p.frame0 + (long long)blockIdx.x * FPB) * (long long)p.H - p.R;
  if (sblk >= 0 && p.history_mode == 0) spectro2_body<W, FMT, GEN, true>(p, lds);
  else spectro2_body<W, FMT, GEN, false>(p, lds);
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


#endif  // GLFER_NO_LAUNCHERS
