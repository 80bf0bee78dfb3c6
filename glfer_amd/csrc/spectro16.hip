// spectro16.hip -- fused frame -> taper(s) -> FFT -> |X|^2 -> taper-sum kernel (gfx950): the
// general form.  It serves every configuration; the specialised forms (spectro16h: one taper as a
// real-input transform; spectro16x / xl / y: odd taper counts) take over where they apply, and
// this kernel keeps what only it can do: the zero-history frames at the start of a stream, the
// RA9MB / limiter path, the halfcomplex spectrum output, even taper counts, N = 256 and N >= 8192.
//
// Replaces, per audio frame, the reference chain
//   prepare_audio (fft.c:66-165) -> fft_real_radix2_transform (fft_radix2.c:75-177)
//   -> fft_psd (fft.c:203-226) [-> the taper loop of mtm_do, mtm.c:189-220]
// with ONE kernel that reads the overlapped sample stream and writes N/2+1 PSD bins.
//
// Layout: N/16 lanes per frame, 16 complex points per lane (two real tapered copies of the
// frame packed as re/im of one complex N-point transform).  Stockham autosort passes of
// radix 16 (last pass radix N/256 or N/4096), each pass a straight-line in-register DFT,
// with the frame exchanged through LDS between passes (stockham16.hpp: layouts that are
// bank-conflict free for writes and reads).  All inter-pass twiddles of a lane are fixed for
// the whole launch and live in registers (N <= 4096: at most 30 complex).  168 VGPRs, three
// 256-lane blocks per CU at N = 4096.
//
// Because taper weights and 1/N are folded into the tapers,
//   sum_j w_j |Y_j[k]|^2 = sum_pairs (|Z_k|^2 + |Z_{N-k}|^2)/2,
// so acc[k] += |Z_k|^2 per pair and one mirror-add through LDS per FRAME finishes the PSD.
// No MFMA: there is no dense contraction on this path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "stockham16.hpp"

#ifndef GLFER16_STAGGER
#define GLFER16_STAGGER 8     /* start delay, in s_sleep units, per (blockIdx/256)%4 */
#endif
#ifndef GLFER16_WAVES_PER_SIMD
/* 3 runs with a few dozen spilled registers; at N = 1024 and 2048 two spill-free wavefronts per SIMD are faster
   (+3..9 %, tools/packed_sizes.py), at 256 / 512 / 4096 three are (or equal) */
#if defined(GLFER_LOGN) && (GLFER_LOGN == 10 || GLFER_LOGN == 11)
#define GLFER16_WAVES_PER_SIMD 2
#else
#define GLFER16_WAVES_PER_SIMD 3
#endif
#endif

#ifndef GLFER16_TW1_REGS
#define GLFER16_TW1_REGS 1
#endif

namespace glfer {

// One kernel, persistent blocks.  The work of a block is a flat sequence of ROUNDS
// (frame group, taper pair); the 48 loads of round r+1 (16 samples, 2x16 taper values per
// lane) are issued right after round r has handed its data to LDS, so they fly under the
// remaining two passes instead of stalling the next round.
// (The timing ablations and the alternative exchange layout of the first round of work are kept
// with tools/experiments/kbench.hip; their results are in profiles/r01_kbench_ablation.txt.)
// KM > 0: per-hop mean removal (fft.c:86-96, the reference's default) inside the kernel for frames
// that lie wholly inside the stream, the hop being KM of a lane's 16 sample registers (16, 8, 4:
// overlap 0, 50, 75 %): when a frame's samples arrive, the sums of its 16/KM hops are taken from
// the registers (lane partial in register order, a butterfly over the frame's lanes of a
// wavefront, the frame's wavefronts through LDS and one barrier) and x - mu replaces x for all the
// frame's rounds.  The same lanes, registers and order whichever frame a hop is seen in.
// FT: the harmonic F statistic of mtm_do (mtm.c:165-174, 203-233) instead of the PSD.  Every round
// transforms the frame under ONE taper (im part zero, so Z[k] = X[k] for k <= N/2): first hn -- its
// spectrum mu stays in 32 registers -- then tapers 0..ntap-1, each adding |X_j - mu U0_j|^2 to a
// per-bin float sum with the reference's own double/float statement types; the quotient is formed
// and stored after the last round.  No spectrum goes through HBM (round 2's first form wrote
// ntap+1 spectra per frame and read them back: 7.4 M frames/s at N = 4096).
// FT = 2 (round 5): the same statistic from HALF the transforms -- the six real sequences of N = 4096, 5 tapers (hn and the tapers)
// go through three packed transforms, two sequences each (re / im), and every pair is separated through the mirror bins in LDS:
// with Z = FFT(a + i b), X_a[k] = (Z[k] + conj Z[N-k]) / 2 and X_b[k] = (Z[k] - conj Z[N-k]) / (2 i) (the halves ride in the tables).
// The accumulation per taper, its order and its statement types are those of FT = 1.
template <int LOGN, int FMT, bool GEN, int WPS = GLFER16_WAVES_PER_SIMD, int STG = GLFER16_STAGGER, int KM = 0, int FT = 0>
__global__ __launch_bounds__(Launch16<LOGN>::BLOCK, WPS) void spectro16_kernel(SpectroParams p) {
  static_assert(KM == 0 || (!GEN && (KM == 16 || KM == 8 || KM == 4)), "in-kernel mean removal: the plain path");
  static_assert(FT == 0 || !GEN, "F statistic: the plain path");
  constexpr int NH = KM > 0 ? 16 / KM : 1;
  using C = Plan16<LOGN>;
  using L = Launch16<LOGN>;
  constexpr int N = C::N, T = C::T, NPASS = C::NPASS, FPB = L::FPB, PADN = L::PADN;
  constexpr int TW1 = 15;                       // pass-1 (Ls=16) twiddles: shared LDS table
  constexpr int NTWR = C::NTW - TW1;            // later passes: per lane, in registers
  constexpr unsigned esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  __shared__ v2f32 lds[L::LDS_WORDS];
  constexpr int WPF = T > 64 ? T / 64 : 1;        // wavefronts per frame
  __shared__ float mred[KM > 0 && WPF > 1 ? FPB * WPF * NH : 1];

  const unsigned tid = threadIdx.x;
  const unsigned t = tid % T;
  const unsigned fl = tid / T;
  v2f32 *xb = lds + fl * PADN;
  v2f32 *tw1 = lds + FPB * PADN;                // [k][q] = W_256^(k*q), k,q < 16

  // ---- twiddles: pass 1's 16x16 table to LDS (same for every lane with equal t%16), the
  // later passes' per-lane values to registers; both fixed for the launch
  {
    const v2f32 *tw = reinterpret_cast<const v2f32 *>(p.tw);
    if (tid < 256) {
      const unsigned k = tid >> 4, q = tid & 15;
      // rows padded to 17 entries: the 16 distinct rows a wave reads land in different banks
      tw1[k * 17 + q] = q ? tw[(q - 1) * T + k] : v2f32{1.0f, 0.0f};     // slot q-1, lane k (k < 16 <= T)
    }
  }
  constexpr int NT = NTWR > 0 ? NTWR : 1;
  float twr[NT], twi[NT];
  {
    const v2f32 *tw = reinterpret_cast<const v2f32 *>(p.tw) + t;
#pragma unroll
    for (int e = 0; e < NTWR; e++) {
      const v2f32 w = tw[(TW1 + e) * T];
      twr[e] = w.x;
      twi[e] = w.y;
    }
  }
  __syncthreads();
  // the lane's pass-1 twiddles: a row of the LDS table, or 32 registers where the form has them to spare (two wavefronts
  // per SIMD: N = 1024 / 2048, and N = 256 outside the generic-window form) -- profiles/r03_tw1_regs_other_kernels.txt
  constexpr bool TW1R = (GLFER16_TW1_REGS) != 0 && (LOGN == 10 || LOGN == 11 || (LOGN == 8 && !GEN));
  Tw1Source<TW1R> tw1row;
  tw1row.init(tw1 + (t & 15) * 17);

  const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.taps), 0, p.npairs * 2 * N * 4, 0x00020000);
  const unsigned toff = t * 16u;
  const long long stride = (long long)gridDim.x * FPB;

  // ---- registers filled ahead of use: the frame's samples (once per frame) and the NEXT
  // round's taper pair (tables are stored interleaved, taps[pair][j] = (taper 2p, taper 2p+1)[j],
  // so one 8-byte load brings both)
  float px[16];
  v2f32 pt[16];
  auto prefetch_x = [&](long long fblk) {
    // Stream index of frame-relative sample j is sblk + flc*H + j with sblk wave-uniform.  The
    // descriptor starts at sample max(sblk,0); samples before the stream (first frames only)
    // get a negative offset -- a huge unsigned one -- and read 0 by the range check: the
    // zero history of fft.c:103-108 without a branch.
    const long long f = fblk + fl;
    const unsigned flc = f < p.nframes ? fl : (unsigned)(p.nframes - 1 - fblk);   // clamp: loads stay in range
    const long long sblk = (p.frame0 + fblk) * (long long)p.H - p.R;
    const long long sbase = sblk > 0 ? sblk : 0;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.stream)) + sbase * (long long)esz, 0, 0x7fffffff, 0x00020000);
    const int lrel = (int)(sblk - sbase) + (int)(flc * (unsigned)p.H + t);   // lane's first sample, relative to sbase
    if (sblk >= 0 && p.history_mode == 0) {        // wave-uniform: no per-element predicate needed
      static_for<0, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        px[m] = buf_sample<FMT>(xrsrc, (unsigned)lrel * esz, (unsigned)(T * m) * esz);
      });
    } else {
      // first frames of the stream, or history zeroed in every frame (fft.c:103-108 with
      // glfer.first_buffer stuck): per-element offset, forced out of range where zero is due
      static_for<0, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const int j = T * m + (int)t;
        const int rel = lrel + T * m;
        const bool ok = p.history_mode ? (j >= p.R) : (rel >= 0);
        const float x = buf_sample<FMT>(xrsrc, ok ? (unsigned)rel * esz : 0x80000000u, 0u);
        px[m] = ok ? x : 0.0f;             // raw 0 is not sample 0.0 for u8 ((0-128)/128)
      });
    }
  };
  auto prefetch_taps = [&](int pair) {
#ifdef GLFER16_ABL_TAPS                            /* timing ablation: the first pair's table for every round (results wrong) */
    if (pair != 0) return;
#endif
    // table layout [pair][m/2][lane][4] = (taper 2p, taper 2p+1) at samples t+T*m and t+T*(m+1):
    // one 16-byte load per lane brings two complex points' worth of tapers (8 loads per round)
    const unsigned tap_p = (unsigned)pair * (N * 8u);              // byte offset of this pair's table (uniform)
    static_for<0, 8>([&](auto mc) {
      constexpr int mh = decltype(mc)::value;
      typedef float v4f32 __attribute__((ext_vector_type(4)));
      const v4f32 q = __builtin_bit_cast(v4f32, __builtin_amdgcn_raw_buffer_load_b128(trsrc, toff, tap_p + (unsigned)(T * mh) * 16u, 0));
      pt[2 * mh] = v2f32{q.x, q.y};
      pt[2 * mh + 1] = v2f32{q.z, q.w};
    });
  };

  if constexpr (STG > 0) {
    // de-phase co-resident blocks so that their VALU, LDS and load phases interleave
    const unsigned ph = (blockIdx.x >> 8) & 3;
    for (unsigned i = 0; i < ph; i++) __builtin_amdgcn_s_sleep(STG);
  }
  long long fblk = (long long)xcd_block_index() * FPB;
  if (fblk >= p.nframes) return;
  int pair = 0;
  prefetch_x(fblk);
  prefetch_taps(0);

  float acc[16];
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.0f;
  float ftmur[FT ? 16 : 1], ftmui[FT ? 16 : 1], ftsum[FT ? 16 : 1];   // FT: mu (re, im) and the per-bin sum, by register

  while (true) {
    if constexpr (KM > 0) {
      if (pair == 0 && p.means) {
        // GLFER_SUBMEAN_EXACT: the hop means are given (taken in the reference's own order by hop_means_seq_kernel,
        // means[global hop index]; the newest hop of frame F of the stream is hop F; px holds sample units here)
        const long long fc = fblk + fl < p.nframes ? fblk + fl : (long long)p.nframes - 1;   // (the clamp of prefetch_x)
        const long long F = p.frame0 + fc;
#pragma unroll
        for (int m = 0; m < 16; m++) px[m] = px[m] - p.means[F - (NH - 1) + m / KM];          // fft.c:93-95
      } else
      if (pair == 0) {                               // a new frame's samples: x <- x - (mean of the hop x arrived in)
        float sm[NH];
#pragma unroll
        for (int q = 0; q < NH; q++) {
          sm[q] = 0.0f;
#pragma unroll
          for (int m = 0; m < KM; m++) sm[q] += px[q * KM + m];
#pragma unroll
          for (int o = 1; o < (T < 64 ? T : 64); o <<= 1) sm[q] += __shfl_xor(sm[q], o);
        }
        if constexpr (WPF > 1) {
          if ((t & 63u) == 0) {
#pragma unroll
            for (int q = 0; q < NH; q++) mred[(fl * WPF + (t >> 6)) * NH + q] = sm[q];
          }
          frame_sync<T>();
#pragma unroll
          for (int q = 0; q < NH; q++) {
            float tot = mred[(fl * WPF) * NH + q];
#pragma unroll
            for (int w = 1; w < WPF; w++) tot += mred[(fl * WPF + w) * NH + q];
            sm[q] = tot;
          }
          // (the next frame's partials are written a whole frame of barriers later)
        }
#pragma unroll
        for (int m = 0; m < 16; m++) px[m] = px[m] - sm[m / KM] / (float)p.H;   // fft.c:91, 93-95
      }
    }
    // ---- form the packed complex frame: re = x*taper(2*pair), im = x*taper(2*pair+1)
    float zr[16], zi[16];
#pragma unroll
    for (int m = 0; m < 16; m++) {
      float x = px[m];
      if (GEN && p.nonlin) {
        // fft.c:127-156: RA9MB x/(a+x^2), window, then sign(y)*|y|^0.1; the unit-power
        // scale is applied afterwards (post_scale) because the limiter is not linear.
        if (p.a > 0.0f) x = x / (p.a + x * x);
        float y = x * pt[m].x;
        if (p.limiter) y = limiter_value(y);
        zr[m] = y * p.post_scale;
        zi[m] = 0.0f;
      } else {
        zr[m] = x * pt[m].x;
        zi[m] = x * pt[m].y;
      }
    }

    // ---- which round comes next (wave-uniform)
    int npair = pair + 1;
    long long nfblk = fblk;
    if (npair == p.npairs) {
      npair = 0;
      nfblk += stride;
    }
    const bool has_next = nfblk < p.nframes;

    // ---- Stockham passes (stockham16.hpp); the next round's loads go out after the first
    // exchange's writes: the prefetch registers are free by then
    stockham16_passes<LOGN, NT>(zr, zi, xb, t, tw1row, twr, twi, [&] {
      if (has_next) {
        prefetch_taps(npair);
        if (npair == 0) prefetch_x(nfblk);
      }
    });

    // After the last pass register rho = b + B*brev(q',R) holds bin t + T*(b + B*q').
    const long long f = fblk + fl;
    const bool live = f < p.nframes;
    if constexpr (FT == 2) {
      constexpr int R = C::radix(NPASS - 1), B = 16 / R;
      // Z to LDS by bin, one barrier, and every lane reads the mirror bins of its bins k <= N/2 (registers with b + B q' <= 8)
      frame_sync<T>();
      static_for<0, 16>([&](auto rc) {
        constexpr int rho = decltype(rc)::value;
        constexpr int b = rho % B, qp = brev(rho / B, R);
        xb[(int)t + T * (b + B * qp)] = v2f32{zr[rho], zi[rho]};
      });
      frame_sync<T>();
      // one sequence's spectrum at the lane's bin of register rho: mu (the hn sequence) is kept, a taper's goes into the sum (mtm.c:203-210)
      auto take = [&](auto rc, int seq, float xr, float xi, int k) {
        constexpr int rho = decltype(rc)::value;
        if (seq >= p.ft_nseq) return;                              // (an odd number of sequences: the last table's second half is zero)
        const int jt = p.ft_mu_live ? seq - 1 : seq;               // taper of this sequence (-1: hn)
        if (jt < 0) {
          ftmur[rho] = xr;
          ftmui[rho] = xi;
          ftsum[rho] = 0.0f;
          return;
        }
        if (!p.ft_mu_live && seq == 0) {
          ftmur[rho] = 0.0f;
          ftmui[rho] = 0.0f;
          ftsum[rho] = 0.0f;
        }
        const double U0j = p.ft_U0[jt];
        {
#pragma clang fp contract(off)
          const double tmpr = (double)xr - (double)ftmur[rho] * U0j;
          const double tmpi = (double)xi - (double)ftmui[rho] * U0j;
          const double both = tmpr * tmpr + tmpi * tmpi, one = tmpr * tmpr;
          ftsum[rho] = (float)((double)ftsum[rho] + (k == 0 ? one : both));
        }
      };
      static_for<0, 16>([&](auto rc) {
        constexpr int rho = decltype(rc)::value;
        constexpr int b = rho % B, qp = brev(rho / B, R);
        if constexpr (b + B * qp <= 8) {
          const int k = (int)t + T * (b + B * qp);
          const v2f32 zm = xb[(N - k) & (N - 1)];                  // Z[N-k] (k = 0: Z[0] itself; k = N/2: itself)
          float ar, ai, br, bi;
          {
#pragma clang fp contract(off)
            ar = zr[rho] + zm.x;                                   // X_a = Z[k] + conj Z[N-k]       (the 1/2 is in the tables)
            ai = zi[rho] - zm.y;
            br = zi[rho] + zm.y;                                   // X_b = (Z[k] - conj Z[N-k]) / i
            bi = zm.x - zr[rho];
          }
          take(rc, 2 * pair, ar, ai, k);
          take(rc, 2 * pair + 1, br, bi, k);
        }
      });
      if constexpr (GLFER16_BARRIER_AFTER_READS != 0) frame_sync<T>();   // mirror bins read: the buffer is free for the next round's writes
    }
    if constexpr (FT == 1) {
      constexpr int R = C::radix(NPASS - 1), B = 16 / R;
      const int jt = p.ft_mu_live ? pair - 1 : pair;             // taper of this round (-1: the hn round)
      if (jt < 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) { ftmur[r] = zr[r]; ftmui[r] = zi[r]; ftsum[r] = 0.0f; }
      } else {
        if (!p.ft_mu_live && pair == 0) {
#pragma unroll
          for (int r = 0; r < 16; r++) { ftmur[r] = 0.0f; ftmui[r] = 0.0f; ftsum[r] = 0.0f; }
        }
        const double U0j = p.ft_U0[jt];
        static_for<0, 16>([&](auto rc) {
          constexpr int rho = decltype(rc)::value;
          constexpr int b = rho % B, qp = brev(rho / B, R);
          const int k = (int)t + T * (b + B * qp);
          {
#pragma clang fp contract(off)
            // mtm.c:203-210: tmpr = ob[i] - mu[i]*U0[j] (double), ft (float) += tmpr*tmpr + tmpi*tmpi; bin 0 has no imaginary part
            const double tmpr = (double)zr[rho] - (double)ftmur[rho] * U0j;
            const double tmpi = (double)zi[rho] - (double)ftmui[rho] * U0j;
            const double both = tmpr * tmpr + tmpi * tmpi, one = tmpr * tmpr;
            ftsum[rho] = (float)((double)ftsum[rho] + (k == 0 ? one : both));
          }
        });
      }
    }
    if constexpr (FT != 0) {
      constexpr int R = C::radix(NPASS - 1), B = 16 / R;
      if (npair == 0 && live) {                                  // the frame's last round: the quotient (mtm.c:222-233)
        float *o = p.ftest + (size_t)f * (N / 2 + 1);
        const int kk = (FT == 2 ? p.ft_nseq : p.npairs) - (p.ft_mu_live ? 2 : 1);        // params->kmax = ntap - 1
        static_for<0, 16>([&](auto rc) {
          constexpr int rho = decltype(rc)::value;
          constexpr int b = rho % B, qp = brev(rho / B, R);
          const int k = (int)t + T * (b + B * qp);
          if (k <= N / 2) {
#pragma clang fp contract(off)
            const float mur = ftmur[rho];
            const float mui = k == N / 2 ? mur : ftmui[rho];     // mu[n_fft - i] at i = N/2 is mu[N/2] again
            const float ft = k < (N + 1) / 2 ? ftsum[rho] : 0.0f;
            double num;
            if (k == 0) num = kk * (mur * mur) * p.ft_sum_U0_sqr;
            else num = kk * (mur * mur + mui * mui) * p.ft_sum_U0_sqr;
            o[k] = (float)(num / ft);
          }
        });
      }
    }
    if constexpr (GEN) {
      // halfcomplex spectrum of the (single, real) tapered frame in fft_radix2.c's layout:
      // data[k] = Re X_k (k<=N/2), data[N-k] = Im X_k (0<k<N/2).  Compat/debug output.
      if (live && p.spec) {
        float *o = p.spec + (size_t)f * N;
        const float inv = 1.0f / p.spec_unscale;
        constexpr int R = C::radix(NPASS - 1), B = 16 / R;
        static_for<0, 16>([&](auto rc) {
          constexpr int rho = decltype(rc)::value;
          constexpr int b = rho % B, qp = brev(rho / B, R);
          const int k = (int)t + T * (b + B * qp);
          if (k <= N / 2) o[k] = zr[rho] * inv;
          if (k > 0 && k < N / 2) o[N - k] = zi[rho] * inv;
        });
      }
    }
#pragma unroll
    for (int r = 0; r < 16; r++)
      acc[r] = __builtin_fmaf(zr[r], zr[r], __builtin_fmaf(zi[r], zi[r], acc[r]));

    if (FT == 0 && npair == 0) {
      // ---- last pair of this frame: mirror fold through LDS, psd[k] = acc[k] + acc[(N-k) mod N]
      float *fold = reinterpret_cast<float *>(xb);
      frame_sync<T>();
      {
        constexpr int R = C::radix(NPASS - 1), B = 16 / R;
        static_for<0, 16>([&](auto rc) {
          constexpr int rho = decltype(rc)::value;
          constexpr int b = rho % B, qp = brev(rho / B, R);
          fold[(int)t + T * (b + B * qp)] = acc[rho];
          acc[rho] = 0.0f;
        });
      }
      frame_sync<T>();
      if (live) {
        float *o = p.psd + (size_t)f * (size_t)p.pitch;
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const int k = T * m + (int)t;
          o[k] = fold[k] + fold[(N - k) & (N - 1)];
        }
        if (t == 0) o[N / 2] = 2.0f * fold[N / 2];
      }
      if constexpr (GLFER16_BARRIER_AFTER_READS != 0) frame_sync<T>();   // fold buffer read: free for the next writes
    }
    if (!has_next) break;
    fblk = nfblk;
    pair = npair;
  }
}

// ---------------------------------------------------------------------------
// K0: per-hop mean removal (fft.c:86-96).  One block per hop; writes a float copy of the
// stream (the reference mutates the caller's hop buffer in place).
template <int FMT>
__global__ __launch_bounds__(256) void submean_kernel(const void *in, float *out, int H, long long nhops, const float *means) {
  __shared__ float part[256];
  const long long hop = blockIdx.x;
  if (hop >= nhops) return;
  constexpr int esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  const char *src = reinterpret_cast<const char *>(in) + hop * (long long)H * esz;   // wave-uniform
  float *dst = out + hop * (long long)H;
  float s = 0.0f;
  for (int i = threadIdx.x; i < H; i += 256) s += cvt_sample<FMT>(src, (unsigned)i);
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  const float mean = means ? means[hop] : part[0] / (float)H;   // means: taken in the reference's order (submean_seq.hip)
  for (int i = threadIdx.x; i < H; i += 256) dst[i] = cvt_sample<FMT>(src, (unsigned)i) - mean;
}

// The same, with a hop's samples held in registers between the sum and the subtraction (one read of
// the stream, no second pass), the loads range-checked by a buffer descriptor and issued together,
// and the sum by DPP instead of an LDS tree with eight barriers -- the per-hop form above ran at
// 0.7 TB/s, and mean removal is the reference's DEFAULT (opt.autoscale = 1, glfer.c:275 ->
// fft.c:186).  GROUP lanes share a hop: 64 (a wavefront per hop, four hops per workgroup, no barrier)
// or 256 (a workgroup per hop, one barrier); lane l holds samples l + GROUP j, j < EPL.
// (The mean is accumulated in another order than the reference's sequential float loop,
// fft.c:88-92 -- as in the form above: a rounding-level difference in a value that is subtracted.)
__device__ __forceinline__ float wave_sum_f32(float v) {
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

template <int FMT, int GROUP, int EPL>
__global__ __launch_bounds__(256) void submean_reg_kernel(const void *in, float *out, int H, long long nhops, const float *means) {
  constexpr unsigned esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  __shared__ float part[4];
  const unsigned l = GROUP == 64 ? (threadIdx.x & 63u) : threadIdx.x;
  const long long hop = GROUP == 64 ? (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))
                                    : (long long)blockIdx.x;
  if (hop >= nhops) return;                        // (GROUP 64: wavefront-uniform; GROUP 256: the whole workgroup)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char *>(reinterpret_cast<const char *>(in)) + hop * (long long)H * esz, 0, (unsigned)H * esz, 0x00020000);
  float x[EPL];
  float s = 0.0f;
#pragma unroll
  for (int j = 0; j < EPL; j++) {
    const float v = buf_sample<FMT>(rs, l * esz, (unsigned)(GROUP * j) * esz);
    x[j] = (int)(l + GROUP * j) < H ? v : 0.0f;    // (past the hop's end the descriptor returns raw 0: a sample of -1 in the u8 format)
  }
#pragma unroll
  for (int j = 0; j < EPL; j++) s += x[j];
  s = wave_sum_f32(s);
  if constexpr (GROUP == 256) {
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    s = (part[0] + part[1]) + (part[2] + part[3]);
  }
  const float mean = means ? means[hop] : s / (float)H;           // means: taken in the reference's order (submean_seq.hip)
  const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(out + hop * (long long)H, 0, (unsigned)H * 4u, 0x00020000);
#pragma unroll
  for (int j = 0; j < EPL; j++)                    // (stores past the hop's end are dropped by the descriptor)
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x[j] - mean), ws, l * 4u, (unsigned)(GROUP * j) * 4u, 0);
}

// The file source's trailing partial block (wav_fmt.c:102-119): `fresh` new samples over the stale
// tail of the reader's buffer, which holds the PREVIOUS block as prepare_audio left it -- mean
// removed in place (fft.c:93-95); prev = that corrected hop (NULL: the buffer was never filled,
// calloc zeros, wav_fmt.c:99).  Then this block's own mean removal.  One block.
template <int FMT>
__global__ __launch_bounds__(256) void submean_tail_kernel(const void *raw_last, const float *prev, float *out, int H,
                                                           int fresh, int exact) {
  __shared__ float part[256];
  float s = 0.0f;
  for (int i = threadIdx.x; i < H; i += 256) {
    const float v = i < fresh ? cvt_sample<FMT>(raw_last, (unsigned)i) : (prev ? prev[i] : 0.0f);
    out[i] = v;
    s += v;
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  float mean = part[0] / (float)H;
  if (exact) {                             // GLFER_SUBMEAN_EXACT: the reference's own order (fft.c:88-92), one thread; once per file
    __shared__ float seq;
    if (threadIdx.x == 0) {
      float a = 0.0f;
      for (int i = 0; i < H; i++) a += out[i];
      seq = a / (float)H;
    }
    __syncthreads();
    mean = seq;
  }
  for (int i = threadIdx.x; i < H; i += 256) out[i] -= mean;       // each thread revisits its own elements
}

}  // namespace glfer

// ---------------------------------------------------------------------------
// host-side launchers (called from glfer_hip.cpp).  One translation unit per LOGN
// (-DGLFER_LOGN=...) so the size variants compile in parallel.
#ifndef GLFER_NO_LAUNCHERS
using namespace glfer;

#ifndef GLFER_LOGN
#error "compile with -DGLFER_LOGN=<log2 of the block size>"
#endif
#define GLFER_CAT2(a, b) a##b
#define GLFER_CAT(a, b) GLFER_CAT2(a, b)

template <int FMT>
static hipError_t launch16_fmt(const SpectroParams &p, hipStream_t st) {
  constexpr int L = GLFER_LOGN;
  using LC = Launch16<L>;
  // persistent blocks: enough to fill every CU at the kernel's occupancy, never more than the work
  const long long work = ((long long)p.nframes + LC::FPB - 1) / LC::FPB;
  if (work == 0) return hipSuccess;
  // the general path (limiter, RA9MB, spectrum output) at two wavefronts per SIMD whatever the size: at three it spills 113-123 VGPRs
  // (N = 4096) and runs at 30-36 M frames/s against 50 (N = 512: 328 against 425) -- tools/packed_rate.py, round 4
  const int wps = (p.nonlin || p.spec) && !p.ftest && !p.mean_inkernel ? 2 : GLFER16_WAVES_PER_SIMD;
  const long long resident = 256LL * ((wps * 256) / LC::BLOCK > 0 ? (wps * 256) / LC::BLOCK : 1);
  unsigned grid = (unsigned)(work < 4 * resident ? work : 4 * resident);
  if (grid >= 64) grid &= ~7u;                     // whole XCD slices: see xcd_block_index()
  if (p.ftest) {
    // the F statistic: one taper per round (two spill-free wavefronts per SIMD: mu and the sums are 48 more registers)
    if (p.nonlin || p.spec || p.mean_inkernel || !p.ft_U0) return hipErrorInvalidValue;
    // (the paired form keeps mu and the sums for the bins k <= N/2 only -- 27 registers, not 48: three wavefronts per SIMD)
    if (p.ft_nseq > 0) hipLaunchKernelGGL((spectro16_kernel<L, FMT, false, 3, GLFER16_STAGGER, 0, 2>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
    else hipLaunchKernelGGL((spectro16_kernel<L, FMT, false, 2, GLFER16_STAGGER, 0, 1>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
    return hipGetLastError();
  }
  if (p.mean_inkernel) {
    // frames inside the stream only, history from the stream, a hop of 4, 8 or 16 sixteenths of the block
    if (p.nonlin || p.spec || p.history_mode || p.frame0 * (long long)p.H < (long long)p.R) return hipErrorInvalidValue;
    const int km = (16 * p.H) % (1 << L) == 0 ? (16 * p.H) >> L : 0;
    if (km == 16) hipLaunchKernelGGL((spectro16_kernel<L, FMT, false, GLFER16_WAVES_PER_SIMD, GLFER16_STAGGER, 16>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
    else if (km == 8) hipLaunchKernelGGL((spectro16_kernel<L, FMT, false, GLFER16_WAVES_PER_SIMD, GLFER16_STAGGER, 8>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
    else if (km == 4) hipLaunchKernelGGL((spectro16_kernel<L, FMT, false, GLFER16_WAVES_PER_SIMD, GLFER16_STAGGER, 4>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if (p.nonlin || p.spec)
    hipLaunchKernelGGL((spectro16_kernel<L, FMT, true, 2>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
  else
    hipLaunchKernelGGL((spectro16_kernel<L, FMT, false>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
  return hipGetLastError();
}

extern "C" hipError_t GLFER_CAT(glfer_launch_spectro16_n, GLFER_LOGN)(const SpectroParams *p, hipStream_t st) {
  switch (p->fmt) {
    case GLFER_FMT_F32: return launch16_fmt<GLFER_FMT_F32>(*p, st);
    case GLFER_FMT_S16: return launch16_fmt<GLFER_FMT_S16>(*p, st);
    case GLFER_FMT_U8: return launch16_fmt<GLFER_FMT_U8>(*p, st);
  }
  return hipErrorInvalidValue;
}

#if GLFER_LOGN == 12
extern "C" hipError_t glfer_launch_submean(const void *in, float *out, int H, long long nhops, int fmt,
                                           hipStream_t st, const float *means) {
  if (nhops <= 0) return hipSuccess;
  if (fmt != GLFER_FMT_F32 && fmt != GLFER_FMT_S16 && fmt != GLFER_FMT_U8) return hipErrorInvalidValue;
  // hops up to 16384 samples: held in registers (a wavefront per hop up to 1024 samples, a workgroup above)
#define GLFER_SUBMEAN_REG(G, E)                                                                                     \
  do {                                                                                                              \
    const unsigned grid = G == 64 ? (unsigned)((nhops + 3) / 4) : (unsigned)nhops;                                  \
    if (fmt == GLFER_FMT_F32) hipLaunchKernelGGL((submean_reg_kernel<GLFER_FMT_F32, G, E>), dim3(grid), dim3(256), 0, st, in, out, H, nhops, means); \
    else if (fmt == GLFER_FMT_S16) hipLaunchKernelGGL((submean_reg_kernel<GLFER_FMT_S16, G, E>), dim3(grid), dim3(256), 0, st, in, out, H, nhops, means); \
    else hipLaunchKernelGGL((submean_reg_kernel<GLFER_FMT_U8, G, E>), dim3(grid), dim3(256), 0, st, in, out, H, nhops, means); \
    return hipGetLastError();                                                                                       \
  } while (0)
  if (H <= 64 * 2) GLFER_SUBMEAN_REG(64, 2);
  if (H <= 64 * 4) GLFER_SUBMEAN_REG(64, 4);
  if (H <= 64 * 8) GLFER_SUBMEAN_REG(64, 8);
  if (H <= 64 * 16) GLFER_SUBMEAN_REG(64, 16);
  if (H <= 256 * 8) GLFER_SUBMEAN_REG(256, 8);
  if (H <= 256 * 16) GLFER_SUBMEAN_REG(256, 16);
  if (H <= 256 * 32) GLFER_SUBMEAN_REG(256, 32);
  if (H <= 256 * 64) GLFER_SUBMEAN_REG(256, 64);
#undef GLFER_SUBMEAN_REG
  switch (fmt) {
    case GLFER_FMT_F32: hipLaunchKernelGGL((submean_kernel<GLFER_FMT_F32>), dim3((unsigned)nhops), dim3(256), 0, st, in, out, H, nhops, means); break;
    case GLFER_FMT_S16: hipLaunchKernelGGL((submean_kernel<GLFER_FMT_S16>), dim3((unsigned)nhops), dim3(256), 0, st, in, out, H, nhops, means); break;
    case GLFER_FMT_U8: hipLaunchKernelGGL((submean_kernel<GLFER_FMT_U8>), dim3((unsigned)nhops), dim3(256), 0, st, in, out, H, nhops, means); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

extern "C" hipError_t glfer_launch_submean_tail_ex(const void *raw_last, const float *prev, float *out, int H, int fresh, int exact,
                                                int fmt, hipStream_t st) {
  switch (fmt) {
    case GLFER_FMT_F32: hipLaunchKernelGGL((submean_tail_kernel<GLFER_FMT_F32>), dim3(1), dim3(256), 0, st, raw_last, prev, out, H, fresh, exact); break;
    case GLFER_FMT_S16: hipLaunchKernelGGL((submean_tail_kernel<GLFER_FMT_S16>), dim3(1), dim3(256), 0, st, raw_last, prev, out, H, fresh, exact); break;
    case GLFER_FMT_U8: hipLaunchKernelGGL((submean_tail_kernel<GLFER_FMT_U8>), dim3(1), dim3(256), 0, st, raw_last, prev, out, H, fresh, exact); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
#endif
#endif  // GLFER_NO_LAUNCHERS
