// spectro16y.hip -- multitaper, odd taper count, N = 4096: the two frames that share their last
// taper's transform (spectro16x.hip) are also carried through the full rounds TOGETHER, interleaved
// in every wavefront (stockham16_passes2): while frame A's exchange writes drain through the LDS
// pipe the wavefront does frame B's butterflies, and every workgroup barrier serves both frames.
//
// tools/stampbench on the one-frame-at-a-time kernels: a round is ~36 % butterflies, ~37 % waiting
// to issue the 2x16 ds_write_b64 of the two exchanges (all four wavefronts of a frame burst into
// the LDS pipe at the same moment, 80 B/clk) and ~28 % barriers; more resident blocks do not help
// (2 and 3 per CU measure the same).  Here the overlap is built into the instruction stream.
//
// Both frames use the same taper pair in the same round, so one set of 32 prefetch VGPRs serves
// both; per lane: 2x32 transform registers, 2x16 accumulators, 2x16 samples, 32 taper values,
// 30 twiddles -- 256 VGPRs, 2 blocks (8 wavefronts) per CU, two 35 KB exchange buffers per block.
// Per frame the arithmetic is exactly spectro16x.hip's (same pairing, same scales), so the rows are
// bit-identical to it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "odd_taper.hpp"

#ifndef GLFER16Y_DPP_SUM
#define GLFER16Y_DPP_SUM 1
#endif
#ifndef GLFER16Y_SKEW
#define GLFER16Y_SKEW 1           /* stockham16_passes2s: frame B's exchange reads and the buffer-release barrier under frame A's arithmetic (+1.5 %,
                                     profiles/r03_y_skew.txt); 2: the shared round's barrier in front of its last stage too (stockham16_passes1s) */
#endif
#ifndef GLFER16Y_TW1_REGS
#define GLFER16Y_TW1_REGS 1       /* the lane's 15 pass-1 twiddles in registers instead of 15 LDS reads per transform (where they fit without a spill:
                                     not with history zeroed per frame, not the 75 % mean form): +1.5...2 %, profiles/r03_y_tw1_regs.txt */
#endif

namespace glfer {

hipError_t allow_dynamic_lds(const void *kernel, size_t bytes);   // plan.h / glfer_hip.cpp: once per device, kernel and size class

struct LaunchY {
  static constexpr int N = 4096, T = 256, PADN = N + N / 16;
  static constexpr int LDS_WORDS = 2 * PADN + 16 * 17 + 4;       // two exchange buffers, pass-1 twiddles, power partials
};

// ABL (tools/xbench timing ablations only; results are wrong): 1 = skip the shared round
// KM > 0: per-hop mean removal (fft.c:86-96, the reference's default) inside the kernel, for a hop of
// KM of a lane's 16 sample registers (KM = 16, 8, 4: overlap 0, 50, 75 %).  A frame spans NH = 16/KM
// hops, each a fixed group of registers; frames A and A+1 share all but one.  The sums are taken from
// the NEXT iteration's samples once they are in registers (lane partial in register order, a
// butterfly over the wavefront, the four wavefronts through LDS across the barriers that end the
// shared round), and x - mu is formed once, before the frames' first round.  A hop's mean comes from
// the same lanes' same registers in the same order whichever frame it is seen in.
template <int FMT, int ABL = 0, int HIST = 0, int KM = 0>
__global__ __launch_bounds__(256, 2) void spectro16y_kernel(SpectroParams p) {
  static_assert(KM == 0 || (HIST == 0 && (KM == 16 || KM == 8 || KM == 4)), "in-kernel mean removal: history from the stream");
  constexpr int NH = KM > 0 ? 16 / KM : 1;
  __shared__ float mred[KM > 0 ? 4 * (NH + 1) : 1];
  using C = Plan16<12>;
  using L = LaunchY;
  constexpr int N = L::N, T = L::T, PADN = L::PADN, NPASS = C::NPASS;
  constexpr int TW1 = 15;
  constexpr int NT = C::NTW - TW1;
  constexpr unsigned esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  typedef float v4f32 __attribute__((ext_vector_type(4)));
  extern __shared__ v2f32 lds[];                    // L::LDS_WORDS entries (72 KB: above the static limit)

  const unsigned t = threadIdx.x;
  v2f32 *xbA = lds, *xbB = lds + PADN;
  v2f32 *tw1 = lds + 2 * PADN;
  float *red = reinterpret_cast<float *>(lds + 2 * PADN + 16 * 17);   // [2][4]

  {
    const v2f32 *tw = reinterpret_cast<const v2f32 *>(p.tw);
    const unsigned k = t >> 4, q = t & 15;
    tw1[k * 17 + q] = q ? tw[(q - 1) * T + k] : v2f32{1.0f, 0.0f};
  }
  float twr[NT], twi[NT];
  {
    const v2f32 *tw = reinterpret_cast<const v2f32 *>(p.tw) + t;
#pragma unroll
    for (int e = 0; e < NT; e++) {
      const v2f32 w = tw[(TW1 + e) * T];
      twr[e] = w.x;
      twi[e] = w.y;
    }
  }
  __syncthreads();
  const v2f32 *tw1row = tw1 + (t & 15) * 17;
  constexpr bool TW1R = GLFER16Y_TW1_REGS != 0 && HIST == 0 && KM != 4;
  v2f32 tw1reg[16];
  if constexpr (TW1R) {
#pragma unroll
    for (int q = 0; q < 16; q++) tw1reg[q] = tw1row[q];
  }

  const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.taps), 0, p.npairs * 2 * N * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.xtaps), 0, N * 4, 0x00020000);
  const unsigned toff = t * 16u;
  const int NP = p.npairs - 1;                       // full (two-taper) rounds per frame
  const long long stride = (long long)gridDim.x * 2;

  // (loads are unconditional -- a frame index past the end re-reads the last frame -- so that the
  // compiler has nothing to turn into selects; what such a slot computes is dropped)
  auto load_x = [&](float (&dst)[16], long long f) {
    load_frame16<FMT, T, HIST>(p, t, 0u, f < p.nframes ? f : (long long)p.nframes - 1, dst);
  };
  v2f32 pt[16];          // full round: the next taper pair; shared round: pt[0..7] = the last taper
  auto prefetch_taps = [&](int pair) {
    const unsigned tap_p = (unsigned)pair * (N * 8u);
    static_for<0, 8>([&](auto mc) {
      constexpr int mh = decltype(mc)::value;
      const v4f32 q = __builtin_bit_cast(v4f32, __builtin_amdgcn_raw_buffer_load_b128(trsrc, toff, tap_p + (unsigned)(T * mh) * 16u, 0));
      pt[2 * mh] = v2f32{q.x, q.y};
      pt[2 * mh + 1] = v2f32{q.z, q.w};
    });
  };
  auto prefetch_last = [&] {                         // [m/4][T][4] floats: samples t + T*(4*(m/4) + j)
    static_for<0, 4>([&](auto mc) {
      constexpr int mq = decltype(mc)::value;
      const v4f32 q = __builtin_bit_cast(v4f32, __builtin_amdgcn_raw_buffer_load_b128(vrsrc, toff, (unsigned)(T * mq) * 16u, 0));
      pt[2 * mq] = v2f32{q.x, q.y};
      pt[2 * mq + 1] = v2f32{q.z, q.w};
    });
  };

  long long fA = (long long)xcd_block_index() * 2;
  if (fA >= p.nframes) return;
  float xA[16], xB[16];
  load_x(xA, fA);
  if (fA + 1 < p.nframes) load_x(xB, fA + 1);
  else {
#pragma unroll
    for (int m = 0; m < 16; m++) xB[m] = 0.0f;
  }
  prefetch_taps(0);

  // ---- KM: the hop sums of frames A (NH hops) and B (its newest hop), wavefront-reduced, into mred
  auto publish_hop_sums = [&] {
    static_for<0, NH + 1>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      float sm = 0.0f;
      static_for<0, KM>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        sm += q < NH ? xA[q * KM + m] : xB[(NH - 1) * KM + m];
      });
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) sm += __shfl_xor(sm, o);
      if ((t & 63u) == 0) mred[(t >> 6) * (NH + 1) + q] = sm;
    });
  };
  // (after a barrier) x - mu for both frames: frame B's hops are frame A's moved down by one
  auto subtract_hop_means = [&] {
    float mu[NH + 1];
#pragma unroll
    for (int q = 0; q <= NH; q++)
      mu[q] = ((mred[q] + mred[(NH + 1) + q]) + (mred[2 * (NH + 1) + q] + mred[3 * (NH + 1) + q])) / (float)p.H;   // fft.c:91
#pragma unroll
    for (int m = 0; m < 16; m++) {
      xA[m] = xA[m] - mu[m / (KM ? KM : 1)];
      xB[m] = xB[m] - mu[m / (KM ? KM : 1) + 1];
    }
  };
  // p.means (GLFER_SUBMEAN_EXACT): the hop means are given -- taken in the reference's own order by
  // hop_means_seq_kernel, means[global hop index]; the newest hop of frame F of the stream is hop F
  auto subtract_table_means = [&](long long fa) {      // fa: frame A's index in the launch
    float mu[NH + 1];
    const long long F = p.frame0 + fa;
#pragma unroll
    for (int q = 0; q < NH; q++) mu[q] = p.means[F - (NH - 1) + q];
    mu[NH] = p.means[fa + 1 < p.nframes ? F + 1 : F];  // frame B's newest hop (no frame B: its registers are zeroed afterwards)
#pragma unroll
    for (int m = 0; m < 16; m++) {
      xA[m] = xA[m] - mu[m / (KM ? KM : 1)];
      xB[m] = xB[m] - mu[m / (KM ? KM : 1) + 1];
    }
  };
  if constexpr (KM > 0) {
    if (p.means) {
      subtract_table_means(fA);
    } else {
      publish_hop_sums();
      __syncthreads();
      subtract_hop_means();
      __syncthreads();                               // mred is free again
    }
    if (fA + 1 >= p.nframes) {
#pragma unroll
      for (int m = 0; m < 16; m++) xB[m] = 0.0f;
    }
  }

  constexpr int RL = C::radix(NPASS - 1), BL = 16 / RL;
  auto rho_of = [](int m) constexpr { return (m % BL) + BL * brev(m / BL, RL); };   // register of bin t + T*m

  while (true) {                                     // one iteration: frames A = fA and B = fA + 1
    const long long nfA = fA + stride;
    const bool has_next = nfA < p.nframes;
    float psdA[8], psdB[8], nyqA, nyqB;
    int hxA, hxB;
    {
      float accA[16], accB[16];
#pragma unroll
      for (int r = 0; r < 16; r++) accA[r] = accB[r] = 0.0f;     // (dead when NP >= 1: pair 0 overwrites)
      for (int pair = 0; pair < NP; pair++) {
        // ---- one full round of both frames: re = x*taper(2*pair), im = x*taper(2*pair+1)
        float zrA[16], ziA[16], zrB[16], ziB[16];
        GLFER_STAMP(0);                              // dual round start
#pragma unroll
        for (int m = 0; m < 16; m++) {
          zrA[m] = xA[m] * pt[m].x;
          ziA[m] = xA[m] * pt[m].y;
          zrB[m] = xB[m] * pt[m].x;
          ziB[m] = xB[m] * pt[m].y;
        }
        auto dual = [&](const auto &tw1sel) {
          auto hook = [&] {
            if (pair + 1 < NP) prefetch_taps(pair + 1);
            else prefetch_last();
          };
          if constexpr (GLFER16Y_SKEW != 0) stockham16_passes2s<12, NT>(zrA, ziA, xbA, zrB, ziB, xbB, t, tw1sel, twr, twi, hook);
          else stockham16_passes2<12, NT>(zrA, ziA, xbA, zrB, ziB, xbB, t, tw1sel, twr, twi, hook);
        };
        if constexpr (TW1R) dual(tw1reg);
        else dual(tw1row);
        if (pair == 0) {                             // the first pair starts the sums (no zeroing pass)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            accA[r] = __builtin_fmaf(zrA[r], zrA[r], ziA[r] * ziA[r]);
            accB[r] = __builtin_fmaf(zrB[r], zrB[r], ziB[r] * ziB[r]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; r++) {
            accA[r] = __builtin_fmaf(zrA[r], zrA[r], __builtin_fmaf(ziA[r], ziA[r], accA[r]));
            accB[r] = __builtin_fmaf(zrB[r], zrB[r], __builtin_fmaf(ziB[r], ziB[r], accB[r]));
          }
        }
        GLFER_STAMP(15);                             // dual round end
      }
      // ---- mirror fold psd[k] = acc[k] + acc[N-k] of both frames (upper half through LDS, entry
      // k - N/2) and the frames' powers for the shared round's scales
      float eA = 0.0f, eB = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        eA += accA[r];
        eB += accB[r];
      }
#if GLFER16Y_DPP_SUM
      eA = wave_total_f32(eA);
      eB = wave_total_f32(eB);
#else
#pragma unroll
      for (int w = 1; w < 64; w <<= 1) {
        eA += __shfl_xor(eA, w);
        eB += __shfl_xor(eB, w);
      }
#endif
      float *foldA = reinterpret_cast<float *>(xbA), *foldB = reinterpret_cast<float *>(xbB);
      static_for<8, 16>([&](auto mc) {               // the buffers are free: barrier after the last reads
        constexpr int m = decltype(mc)::value;
        foldA[t + T * (m - 8)] = accA[rho_of(m)];
        foldB[t + T * (m - 8)] = accB[rho_of(m)];
      });
      if ((t & 63) == 0) {
        red[t >> 6] = eA;
        red[4 + (t >> 6)] = eB;
      }
      __syncthreads();
      eA = (red[0] + red[1]) + (red[2] + red[3]);
      eB = (red[4] + red[5]) + (red[6] + red[7]);
      hxA = scale_exponent(eA);
      hxB = scale_exponent(eB);
      static_for<0, 8>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const int k = T * m + (int)t;
        float oa = foldA[N / 2 - k], ob = foldB[N / 2 - k];   // acc[N-k]; entry N/2 (k = 0) is never written
        if constexpr (m == 0) {
          if (t == 0) {
            oa = accA[rho_of(0)];
            ob = accB[rho_of(0)];
          }
        }
        psdA[m] = accA[rho_of(m)] + oa;
        psdB[m] = accB[rho_of(m)] + ob;
      });
      nyqA = 2.0f * accA[rho_of(8)];
      nyqB = 2.0f * accB[rho_of(8)];
      __syncthreads();                               // fold buffers read: free for the next writes
    }

    if constexpr (ABL == 1) {
      if (has_next) {
        prefetch_taps(0);
        load_x(xA, nfA);
        load_x(xB, nfA + 1);
      }
      if (t == 0) p.psd[(size_t)fA * (size_t)p.pitch] = psdA[0] + psdB[0] + nyqA + nyqB + (float)(hxA + hxB);
      if (!has_next) break;
      fA = nfA;
      continue;
    }
    // ---- shared round: re = sA*(xA*v), im = sB*(xB*v), v the last taper
    float zr[16], zi[16];
    GLFER_STAMP(0);                                  // shared round start
    {
      const float sA = scale_in(hxA), sB = scale_in(hxB);
#pragma unroll
      for (int m = 0; m < 16; m++) {
        const float v = (m & 1) ? pt[m / 2].y : pt[m / 2].x;
        zr[m] = (xA[m] * v) * sA;
        zi[m] = (xB[m] * v) * sB;                    // no frame B: xB is all zeros (set below)
      }
    }
    // the next iteration's samples and first taper pair go out after the first exchange's writes
    auto shared_round = [&](const auto &tw1sel) {
      auto hook = [&] {
        if (has_next) {
          prefetch_taps(0);
          load_x(xA, nfA);
          load_x(xB, nfA + 1);
        }
      };
      if constexpr (GLFER16Y_SKEW >= 2) stockham16_passes1s<12, NT>(zr, zi, xbA, t, tw1sel, twr, twi, hook);
      else stockham16_passes<12, NT>(zr, zi, xbA, t, tw1sel, twr, twi, hook);
    };
    if constexpr (TW1R) shared_round(tw1reg);
    else shared_round(tw1row);
    if constexpr (KM > 0) {
      if (has_next && !p.means) publish_hop_sums();  // the next iteration's samples are in registers (requested during the passes above)
    }
    separate_and_store<12, 1>(p, zr, zi, xbA, t, 0u, fA, hxA, hxB,
                              [&](auto mc) { if constexpr (decltype(mc)::value == 8) return nyqA; else return psdA[decltype(mc)::value]; },
                              [&](auto mc) { if constexpr (decltype(mc)::value == 8) return nyqB; else return psdB[decltype(mc)::value]; });
    GLFER_STAMP(15);                                 // shared round end (separated, stored)
    if constexpr (KM > 0) {
      if (has_next) {                                // (separate_and_store's barriers lie between the sums and this)
        if (p.means) subtract_table_means(nfA);
        else subtract_hop_means();
      }
    }
    if (!has_next) break;
    fA = nfA;
    if (fA + 1 >= p.nframes) {
#pragma unroll
      for (int m = 0; m < 16; m++) xB[m] = 0.0f;
    }
  }
}

}  // namespace glfer

#ifndef GLFER_NO_LAUNCHERS
using namespace glfer;

template <int FMT>
static hipError_t launch16y_fmt(const SpectroParams &p, hipStream_t st) {
  const long long work = ((long long)p.nframes + 1) / 2;
  if (work == 0) return hipSuccess;
  const long long resident = 256LL * 2;
  unsigned grid = (unsigned)(work < 16 * resident ? work : 16 * resident);   // tools/xbench: 16x beats 4x by ~2 %
  if (grid >= 64) grid &= ~7u;                       // whole XCD slices: see xcd_block_index()
  constexpr size_t shmem = (size_t)LaunchY::LDS_WORDS * 8;
  {
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(p.history_mode ? spectro16y_kernel<FMT, 0, 1> : spectro16y_kernel<FMT, 0, 0>), shmem);
    if (e != hipSuccess) return e;
  }
  if (p.mean_inkernel) {
    if (p.history_mode) return hipErrorInvalidValue;
    const int km = p.H % 256 == 0 ? p.H / 256 : 0;
#define GLFER_Y_MEAN(K)                                                                                              \
  do {                                                                                                               \
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(spectro16y_kernel<FMT, 0, 0, K>), shmem);       \
    if (e != hipSuccess) return e;                                                                                   \
    hipLaunchKernelGGL((spectro16y_kernel<FMT, 0, 0, K>), dim3(grid), dim3(256), shmem, st, p);                     \
    return hipGetLastError();                                                                                        \
  } while (0)
    if (km == 16) GLFER_Y_MEAN(16);
    if (km == 8) GLFER_Y_MEAN(8);
    if (km == 4) GLFER_Y_MEAN(4);
#undef GLFER_Y_MEAN
    return hipErrorInvalidValue;
  }
  if (p.history_mode) hipLaunchKernelGGL((spectro16y_kernel<FMT, 0, 1>), dim3(grid), dim3(256), shmem, st, p);
  else hipLaunchKernelGGL((spectro16y_kernel<FMT, 0, 0>), dim3(grid), dim3(256), shmem, st, p);
  return hipGetLastError();
}

// N = 4096, odd taper counts >= 3; needs p->xtaps (glfer_hip.cpp builds it)
extern "C" hipError_t glfer_launch_spectro16y_n12(const SpectroParams *p, hipStream_t st) {
  if (!p->xtaps || p->npairs < 2 || p->nonlin || p->spec) return hipErrorInvalidValue;
  if (p->frame0 * (long long)p->H < (long long)p->R) return hipErrorInvalidValue;   // no zero-history path here
  switch (p->fmt) {
    case GLFER_FMT_F32: return launch16y_fmt<GLFER_FMT_F32>(*p, st);
    case GLFER_FMT_S16: return launch16y_fmt<GLFER_FMT_S16>(*p, st);
    case GLFER_FMT_U8: return launch16y_fmt<GLFER_FMT_U8>(*p, st);
  }
  return hipErrorInvalidValue;
}
#endif  // GLFER_NO_LAUNCHERS
