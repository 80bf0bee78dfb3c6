// spectro16w.hip -- real-input frames (the periodogram of fft_do/fft_psd, fft.c:190-226, and the
// taper loop of mtm_do, mtm.c:189-220) with WAVEFRONT-PRIVATE transforms: no workgroup barrier
// inside a transform.
//
// The windowed frame y[0..N) is packed as z[n] = y[2n] + i*y[2n+1] (M = N/2 complex points) and
// the M-point DFT is split decimation-in-time over the W = M/1024 wavefronts of the frame:
//     Z[k1 + 1024*k2] = sum_w  W_M^(w*k1) * A_w[k1] * W_W^(w*k2),     A_w = DFT_1024{ z[W*j + w] }.
// Wavefront w runs A_w entirely by itself -- 16 points per lane, Stockham radix 16,16,4
// (stockham16.hpp), both exchanges through its own strip of LDS, ordered by the wavefront's own
// LDS queue -- and leaves A_w in that strip.  ONE workgroup barrier later a lane picks up
// A_0..A_{W-1} at bin k1 and at bin 1024-k1, applies the combine twiddles, runs the two radix-W
// butterflies in registers and has Z[k1 + 1024*k2] together with its mirror partners
// Z[M - k1 - 1024*k2], which is exactly what the real-input split needs:
//     E = Z[k] + conj(Z[M-k]),  O = Z[k] - conj(Z[M-k]),  P = -i * W_N^k * O,
//     X[k] = (E + P)/2,  X[M-k] = conj(E - P)/2,
// so the cross-wavefront pass and the mirror step are one exchange.  Per transform: two
// wavefront-private exchanges and one shared one with one barrier (two strips sets, SETS = 2) or
// two (SETS = 1), against three shared exchanges plus the mirror with eight barriers in
// spectro16h.hip.  1/N (fft.c:212-216), the halvings above and the taper weights 1/(1+sig_j)
// (mtm.c:214-219) are folded into the window tables.
//
// MT = 1: the tapers in turn on the frame held in registers, |X|^2 summed per bin in registers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "stockham16.hpp"

#ifndef GLFER_LOGN
#error "compile with -DGLFER_LOGN=<log2 of the block size>"
#endif
// strip sets: 2 = the next transform's exchanges go to the other set, so nothing waits for the
// slowest reader of the shared pass; 1 where two sets would cost a resident workgroup
#ifndef GLFER16W_SETS
#define GLFER16W_SETS (GLFER_LOGN >= 13 && GLFER_LOGN <= 14 ? 2 : 1)
#endif
#ifndef GLFER16W_WAVES_PER_SIMD
#define GLFER16W_WAVES_PER_SIMD (GLFER_LOGN == 12 ? 3 : (GLFER_LOGN == 15 ? 4 : 2))
#endif
#ifndef GLFER16W_WPREFETCH
#define GLFER16W_WPREFETCH 1      /* VAR 1: the next transform's table is requested under this transform's passes */
#endif
#ifndef GLFER16W_SETPRIO
#define GLFER16W_SETPRIO 0        /* 1: the later-dispatched half of a >= 8-wavefront workgroup runs at s_setprio 1 */
#endif
#ifndef GLFER16W_DEPHASE
#define GLFER16W_DEPHASE 0        /* 1: workgroups of >= 8 wavefronts run S before C in their upper half (measured: -4 %, profiles/r02_spectro16w_variants.txt) */
#endif
#ifndef GLFER16W_STORE_AUX
#define GLFER16W_STORE_AUX 0   /* default cache policy (round 3: equal speed, and no write excess; profiles/r03_store_policy.txt) */
#endif

// GLFER16W_STAMPS (diagnostic builds only, tools/build_variant.sh): lane 0 of every wavefront of one
// workgroup records the shader clock at the phase boundaries of its first 64 transforms into the
// buffer passed in p.spec; the launcher prints them.  No stamp executes in the product build.
#ifdef GLFER16W_STAMPS
#define W_STAMP(T, id)                                                                                       \
  do {                                                                                                       \
    if (blockIdx.x == 8 && t == 0 && (T) < 64u) {                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      reinterpret_cast<unsigned long long *>(p.spec)[(wv * 64u + (T)) * 8u + (id)] = __builtin_amdgcn_s_memtime(); \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
    }                                                                                                        \
  } while (0)
#else
#define W_STAMP(T, id)
#endif

#ifndef GLFER16W_TW1_REGS
#define GLFER16W_TW1_REGS 1
#endif

namespace glfer {

template <int LOGN>
struct LaunchW {
  static constexpr int N = 1 << LOGN, M = N / 2;
  static constexpr int W = M / 1024;                       // wavefronts per frame
  static_assert(W >= 1 && W <= 16, "N = 2048 .. 32768");
  static constexpr int FPB = W >= 4 ? 1 : 4 / W;           // frames per workgroup (>= 256 threads)
  static constexpr int BLOCK = 64 * W * FPB;
  static constexpr int LF = 64 * W;                        // lanes per frame
  static constexpr int IPL = LF <= 512 ? 512 / LF : 1;     // bin pairs (k1, 1024-k1) per lane (W = 16: lanes 512.. idle)
  static constexpr int STRIP = 1024 + 64;                  // entries per wavefront strip (both exchange layouts)
};

// VAR 1: window / taper read from its table per transform; 2: the window in LDS (periodogram)
// GEN 1: the general form -- samples gathered one by one, range-checked (frames that reach back
// before sample 0 read zeros, fft.c:103-108; integer pairs need no alignment), RA9MB / limiter
// (fft.c:127-156) and the halfcomplex spectrum output of fft_do.  Used where no other kernel takes
// those cases: N = 32768 (the packed form stops at N = 16384).
// HIST = 1: history zeroed in every frame (a template parameter: as a run-time test the zeroing
// becomes 32 unconditional selects per frame).
// KM = 16, 8, 4 (multitaper form; overlap 0, 50, 75 %: a hop is KM of a lane's 16 sample registers):
// per-hop mean removal (fft.c:86-96, the reference's default) inside the kernel -- the frame's samples
// are in registers for all its tapers anyway; before the first taper the lanes' sums per hop go
// through the wavefronts (butterfly) and the frame's W wavefronts (LDS, one more workgroup barrier
// per FRAME, not per transform), and x - mu[hop] is what every taper multiplies.  A hop seen again in
// the next frame sits KM registers lower in the same lanes: the same sums in the same order.  Integer samples: the mean is taken on the raw values, the scale
// (a power of two) sits in the taper tables as before.
template <int LOGN, int FMT, int MT, int VAR, int SETS, int WPS, int GEN = 0, int HIST = 0, int KM = 0>
__global__ __launch_bounds__(LaunchW<LOGN>::BLOCK, WPS) void spectro16w_kernel(SpectroParams p) {
  static_assert(KM == 0 || ((KM == 16 || KM == 8 || KM == 4) && MT != 0 && GEN == 0 && HIST == 0),
                "in-kernel mean removal: the multitaper form, hop = 16, 8 or 4 of a lane's 16 sample registers");
  constexpr int NH = KM ? 16 / KM : 1;                     // hops per frame
  using L = LaunchW<LOGN>;
  using C = Plan16<10>;
  constexpr int M = L::M, W = L::W, FPB = L::FPB, LF = L::LF, IPL = L::IPL, STRIP = L::STRIP;
  constexpr int TW1 = 15, NT = C::NTW - TW1;               // pass 2's twiddles per lane (12)
  constexpr unsigned esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  constexpr float kSampleScale = FMT == GLFER_FMT_F32 ? 1.0f : (FMT == GLFER_FMT_S16 ? 1.0f / 32768.0f : 1.0f / 128.0f);
  static_assert(MT == 0 || VAR == 1, "the multitaper form reads a table per taper");
  // MT: bin 512 of every sub-transform and taper is set aside (side[frame parity][taper][w]) and the
  // k1 = 512 item of ALL tapers is worked off once per frame, one taper per lane (red[taper][k2] holds
  // the lanes' |X|^2 for the sum over tapers): one lane's extra item per taper would lengthen one
  // wavefront of eight by 45 % and every barrier with it
  constexpr int NTMAX = 32;                                // mtm_k <= 31 (glfer_hip_plan_create)
  constexpr int SIDE = MT ? 2 * NTMAX * W : 0, RED = MT ? NTMAX * W / 2 : 0;   // v2f32 entries per frame slot
  __shared__ v2f32 lds[FPB * W * STRIP * SETS + 16 * 17 + (VAR == 2 ? M : 0) + FPB * (SIDE + RED)];
  __shared__ float mred[KM ? FPB * W * NH : 1];            // KM: the wavefronts' sums of the frame's hops

  const unsigned tid = threadIdx.x;
  const unsigned t = tid & 63u;
  const unsigned wv = tid >> 6;
  const unsigned w = wv % W;                               // which sub-transform of the frame
  const unsigned fl = wv / W;                              // frame slot in the workgroup
  const unsigned u = w * 64u + t;                          // lane index within the frame
  v2f32 *tw1 = lds + FPB * W * STRIP * SETS;
  v2f32 *wl = tw1 + 16 * 17;                               // VAR 2: window pairs, [w][m][t]
  v2f32 *side = wl + (VAR == 2 ? M : 0) + fl * (SIDE + RED);
  float *red = reinterpret_cast<float *>(side + SIDE);

  {                                                        // pass 1's 16 x 16 table (rows padded to 17)
    const v2f32 *tw = reinterpret_cast<const v2f32 *>(p.wtw);
    if (tid < 256) {
      const unsigned k = tid >> 4, q = tid & 15;
      tw1[k * 17 + q] = q ? tw[(q - 1) * 64 + k] : v2f32{1.0f, 0.0f};
    }
  }
  float twr[NT], twi[NT];
  {
    const v2f32 *tw = reinterpret_cast<const v2f32 *>(p.wtw) + t;
#pragma unroll
    for (int e = 0; e < NT; e++) {
      const v2f32 x = tw[(TW1 + e) * 64];
      twr[e] = x.x;
      twi[e] = x.y;
    }
  }
  // combine twiddles of this lane's bins: slot [i][0] = (cos, sin)(2 pi k1 / N) (the split's post
  // twiddle), [i][w'] = (cos, sin)(2 pi w' k1 / M), k1 = u + LF*i
  // (W = 2: only [i][0] is kept; W_M^k1 = (W_N^k1)^2 is squared out of it per item)
  constexpr int CTW = W == 2 ? 1 : W;
  v2f32 ct[IPL][CTW];
  {
    const v2f32 *cw = reinterpret_cast<const v2f32 *>(p.wcomb) + u;
#pragma unroll
    for (int i = 0; i < IPL; i++)
#pragma unroll
      for (int ww = 0; ww < CTW; ww++) ct[i][ww] = cw[(i * W + ww) * LF];
  }
  typedef float v4f32 __attribute__((ext_vector_type(4)));
  v4f32 wq[8];                                             // window / taper pairs of the lane, two registers per load
  auto load_window = [&](int j) {                          // table [taper][w][m/2][t][4]
    const v4f32 *ht = reinterpret_cast<const v4f32 *>(p.wtaps) + ((size_t)j * W + w) * (8 * 64) + t;
#pragma unroll
    for (int mh = 0; mh < 8; mh++) wq[mh] = ht[64 * mh];
  };
  auto window_pair = [&](auto mc) -> v2f32 {
    constexpr int m = decltype(mc)::value;
    const v4f32 q = wq[m / 2];
    return (m & 1) ? v2f32{q.z, q.w} : v2f32{q.x, q.y};
  };
  if constexpr (VAR == 2) {
    if (fl == 0) {
      load_window(0);
      static_for<0, 16>([&](auto mc) { wl[(w * 16 + decltype(mc)::value) * 64 + t] = window_pair(mc) * kSampleScale; });
    }
  } else if constexpr (GLFER16W_WPREFETCH != 0) {
    load_window(0);
  }
  __syncthreads();
  // the lane's pass-1 twiddles: a row of the LDS table, or 32 registers in the two-wavefronts-per-SIMD forms that hold them
  // without spilling (profiles/r03_tw1_regs_other_kernels.txt: +3.9 % on N = 16384, 9 tapers)
  constexpr bool TW1R = (GLFER16W_TW1_REGS) != 0 && WPS == 2 && LOGN <= 14 && !(LOGN >= 13 && KM != 0) && !(LOGN == 14 && HIST != 0);
  Tw1Source<TW1R> tw1row;
  tw1row.init(tw1 + (t & 15) * 17);

  // ---- samples: z index n = W*(t + 64 m) + w, the pair (y[2n], y[2n+1]) with one load.  The
  // launcher hands this kernel only frames that lie wholly inside the stream.
  v2f32 px[16];
  auto prefetch_x = [&](long long fblk) {
    const long long f = fblk + fl;
    const unsigned flc = f < p.nframes ? fl : (unsigned)(p.nframes - 1 - fblk);
    const long long sblk = (p.frame0 + fblk) * (long long)p.H - p.R;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.stream)) + sblk * (long long)esz, 0, 0x7fffffff, 0x00020000);
    const unsigned lrel = flc * (unsigned)p.H + 2u * (W * t + w);
    if constexpr (GEN != 0) {
      // one element at a time, through a descriptor that starts at sample max(sblk, 0): samples before
      // the stream get an out-of-range offset and read 0 (raw value 0 is not sample 0.0 for u8, hence
      // the select); px keeps FLOAT samples here, whatever the format
      const long long sbase = sblk > 0 ? sblk : 0;
      const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char *>(reinterpret_cast<const char *>(p.stream)) + sbase * (long long)esz, 0, 0x7fffffff, 0x00020000);
      const int rel0 = (int)(sblk - sbase) + (int)lrel;
      static_for<0, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int off = 2 * W * 64 * m;
        const int jfr = 2 * (int)(W * t + w) + off;       // index in the frame
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const int rel = rel0 + off + e;
          const bool ok = p.history_mode ? (jfr + e >= p.R) : (rel >= 0);
          const float v = buf_sample<FMT>(grsrc, ok ? (unsigned)rel * esz : 0x80000000u, 0u);
          if (e == 0) px[m].x = ok ? v : 0.0f;
          else px[m].y = ok ? v : 0.0f;
        }
      });
      return;
    }
    if constexpr (HIST != 0) {
      // ZERO_ALWAYS: the history is never loaded (a piece cut for this mode carries none: glfer_hip.h,
      // "Cutting a stream", rule 2): descriptor at the block's own first hop, history pairs out of
      // range (they read 0; the masks below put the format's zero there).
      const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char *>(reinterpret_cast<const char *>(p.stream)) + (sblk + p.R) * (long long)esz, 0, 0x7fffffff, 0x00020000);
      const int d = 2 * (int)(W * t + w) - p.R;
      const int hrel = (int)(flc * (unsigned)p.H) + d;
      static_for<0, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int off = 2 * W * 64 * m;
        const bool ok = d + off >= 0;
        const unsigned vo = ok ? (unsigned)(hrel + off) * (unsigned)esz : 0x80000000u;
        if constexpr (FMT == GLFER_FMT_F32) {
          // odd hop (f32 only) => odd R: the one pair that straddles the history's end is fetched one
          // sample up and its first sample moved into place (spectro16h.hip)
          const bool strad = d + off == -1;
          const v2f32 v = __builtin_bit_cast(v2f32, __builtin_amdgcn_raw_buffer_load_b64(hrsrc, strad ? (unsigned)(hrel + off + 1) * 4u : vo, 0u, 0));
          px[m] = strad ? v2f32{0.0f, v.x} : v;
        } else if constexpr (FMT == GLFER_FMT_S16) {
          px[m].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(hrsrc, vo, 0u, 0));
        } else {
          px[m].x = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(hrsrc, vo, 0u, 0));
        }
      });
    } else
    static_for<0, 16>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr unsigned off = 2u * W * 64u * m;           // samples
      if constexpr (FMT == GLFER_FMT_F32) {
        px[m] = __builtin_bit_cast(v2f32, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, lrel * 4u, off * 4u, 0));
      } else if constexpr (FMT == GLFER_FMT_S16) {
        px[m].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, lrel * 2u, off * 2u, 0));
      } else {
        px[m].x = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(xrsrc, lrel, off, 0));
      }
    });
    if constexpr (HIST != 0) {   // sample j = 2n + e is kept iff j >= R (fft.c:103-108); a rare mode
      const int d = 2 * (int)(W * t + w) - p.R;
      static_for<0, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int off = 2 * W * 64 * m;
        const bool k0 = d + off >= 0, k1 = d + off + 1 >= 0;
        if constexpr (FMT == GLFER_FMT_F32) {
          px[m].x = k0 ? px[m].x : 0.0f;
          px[m].y = k1 ? px[m].y : 0.0f;
        } else if constexpr (FMT == GLFER_FMT_S16) {
          const unsigned raw = __float_as_uint(px[m].x);
          px[m].x = __uint_as_float((k0 ? raw & 0xffffu : 0u) | (k1 ? raw & 0xffff0000u : 0u));
        } else {
          const unsigned raw = __float_as_uint(px[m].x);
          px[m].x = __uint_as_float((k0 ? raw & 0xffu : 0x80u) | (k1 ? raw & 0xff00u : 0x8000u));
        }
      });
    }
  };
  // MT: an integer frame is converted ONCE, in place, when its first taper starts (phase_S) -- the same floats as converting
  // at every use, a twentieth of C4's arithmetic less for 16-bit and 8-bit streams
  constexpr bool kConvertOnce = MT != 0 && FMT != GLFER_FMT_F32 && GEN == 0;
  auto raw_pair = [&](auto mc) -> v2f32 {
    constexpr int m = decltype(mc)::value;
    if constexpr (FMT == GLFER_FMT_S16) {
      const int raw = (int)__float_as_uint(px[m].x);
      return v2f32{(float)(short)(raw & 0xffff), (float)(raw >> 16)};
    } else {
      const unsigned raw = __float_as_uint(px[m].x);
      return v2f32{(float)(raw & 0xffu) - 128.0f, (float)((raw >> 8) & 0xffu) - 128.0f};
    }
  };
  auto sample_pair = [&](auto mc) -> v2f32 {
    constexpr int m = decltype(mc)::value;
    if constexpr (FMT == GLFER_FMT_F32 || GEN != 0 || kConvertOnce) {
      return px[m];
    } else if constexpr (FMT == GLFER_FMT_S16) {
      const int raw = (int)__float_as_uint(px[m].x);
      return v2f32{(float)(short)(raw & 0xffff), (float)(raw >> 16)};
    } else {
      const unsigned raw = __float_as_uint(px[m].x);
      return v2f32{(float)(raw & 0xffu) - 128.0f, (float)((raw >> 8) & 0xffu) - 128.0f};
    }
  };

  // contiguous frame ranges per workgroup, neighbouring ranges on one XCD (spectro16h.hip)
  const long long groups = (p.nframes + FPB - 1) / FPB, per = (groups + gridDim.x - 1) / gridDim.x;
  long long fblk = (long long)xcd_block_index() * per * FPB;
  const long long fend = (fblk + per * FPB < p.nframes) ? fblk + per * FPB : (long long)p.nframes;
  if (fblk >= fend) return;
  prefetch_x(fblk);

  // after the last pass register rho_of(m) holds bin t + 64 m of the sub-transform
  auto rho_of = [](int m) constexpr { return (m % 4) + 4 * brev(m / 4, 4); };
  auto shared_sync = [&] {
    if constexpr (W > 1) __syncthreads();
    else frame_sync<64>();                                 // W = 1: the frame lives in one wavefront
  };
  const int ntap = MT ? p.wtapers : 1;

  // The k1 = 512 item: 1024 - 512 is 512 again, so one radix-W transform gives all of
  // Z[512 + 1024 k2] = sum_w a_w W_2W^w W_W^(w k2), and bin 512 + 1024 k2 pairs with bin
  // 512 + 1024 (W-1-k2).  v[k2] = |X[512 + 1024 k2]|^2.
  auto special512 = [](float (&ar)[W], float (&ai)[W], float (&v)[W], auto &&spec_out) {
    static_for<1, W>([&](auto wc) {
      constexpr int ww = decltype(wc)::value;
      constexpr cplx64 a = unit_root(ww, 2 * W);           // (cos, sin)(2 pi 512 ww / M)
      constexpr float c = (float)a.c, s = (float)a.s;
      const float xr = ar[ww], xi = ai[ww];
      ar[ww] = __builtin_fmaf(xr, c, xi * s);
      ai[ww] = __builtin_fmaf(xi, c, -xr * s);
    });
    if constexpr (W > 1) dit<W, 1, 0, W>(ar, ai);
    static_for<0, (W > 1 ? W / 2 : 1)>([&](auto kc) {
      constexpr int k2 = decltype(kc)::value;
      constexpr int ia = brev(k2, W), ib = brev(W - 1 - k2, W);
      const float er = ar[ia] + ar[ib], ei = ai[ia] - ai[ib], orr = ar[ia] - ar[ib], oi = ai[ia] + ai[ib];
      constexpr cplx64 uu = unit_root(1 + 2 * k2, 4 * W);  // (cos, sin)(2 pi (512 + 1024 k2) / N)
      constexpr float c = (float)uu.c, s = (float)uu.s;
      const float pr = __builtin_fmaf(c, oi, -s * orr);
      const float pi = -__builtin_fmaf(c, orr, s * oi);
      const float x1r = er + pr, x1i = ei + pi, x2r = er - pr, x2i = ei - pi;
      v[k2] = __builtin_fmaf(x1r, x1r, x1i * x1i);
      v[W - 1 - k2] = __builtin_fmaf(x2r, x2r, x2i * x2i);
      spec_out(k2, x1r, x1i);
      if (W > 1) spec_out(W - 1 - k2, x2r, -x2i);
    });
  };

  // ---- the work of a workgroup is a flat sequence of transforms T = (frame group, taper); each is a
  // sub-transform phase S(T) -- window, the wavefront's own 1024-point transform, A_w into the strips
  // of set T & 1 -- and, one barrier later, a shared phase C(T) -- combine, split, |X|^2.
  //   lockstep:   S(0) | C(0) S(1) | C(1) S(2) | ...          ( | = workgroup barrier )
  // With eight or more wavefronts in the workgroup (two per SIMD, wavefront i and i + W/2 on the same
  // one) the upper half runs each round in the OTHER order, S(T+1) before C(T): after a barrier one
  // wavefront of every SIMD is in the LDS-store-heavy phase and the other in the arithmetic-only
  // one, instead of both queueing for the LDS store path and then both for the VALU.  Legal with
  // two strip sets: S(T+1) touches set (T+1) & 1 only, C(T) reads set T & 1, which nobody writes
  // before the next barrier.
  constexpr bool kStagger = GLFER16W_DEPHASE != 0 && SETS == 2 && W * FPB >= 8;
  const bool late = kStagger && w >= W / 2;
  const unsigned ROWB = (unsigned)p.pitch * 4u;            // bytes from row to row (cfg.psd_pitch)
  constexpr int KMAX = LF * (IPL - 1) + 1024 * (W - 1);    // largest k1 + 1024 k2 offset of a lane's items
  float acc[MT ? IPL * 2 * W : 1];                         // MT: [item][k2][bin k | bin M-k]

  // S: frame group sf, taper sj.  px holds the group's samples, wq taper sj's table (VAR 1).
  long long sf = fblk;
  int sj = 0;
  unsigned sit = 0;                                        // transforms started: picks the strip set
  float mu[NH];                                            // KM: the means of the hops of the frame in px (raw sample units)
#pragma unroll
  for (int q = 0; q < NH; q++) mu[q] = 0.0f;
  auto phase_S = [&] {
    const bool last = sj == ntap - 1;
    const long long nf = sf + FPB;
    const bool has_next = nf < fend;
    if constexpr (kConvertOnce) {
      if (sj == 0) static_for<0, 16>([&](auto mc) { px[decltype(mc)::value] = raw_pair(mc); });
    }
    if constexpr (KM != 0) {
      if (sj == 0 && p.means) {
        // given means (cfg.sub_mean = 1: the reference's own summation order, submean_seq.hip), indexed by GLOBAL hop = the frame
        // whose newest hop it is; in the units the samples are held in (integer formats: raw, the power of two rides in the tapers)
        const long long fr = sf + fl < p.nframes ? sf + fl : (long long)p.nframes - 1;
        const long long F = p.frame0 + fr;
#pragma unroll
        for (int q = 0; q < NH; q++) mu[q] = p.means[F - (NH - 1) + q] * (1.0f / kSampleScale);
      } else if (sj == 0) {                                // a new frame in px: its hops' means, once (every wavefront of the workgroup is here)
        float part[NH];
#pragma unroll
        for (int q = 0; q < NH; q++) part[q] = 0.0f;
        static_for<0, 16>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          const v2f32 x = sample_pair(mc);
          part[m / KM] += x.x;
          part[m / KM] += x.y;
        });
#pragma unroll
        for (int q = 0; q < NH; q++) {
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) part[q] += __shfl_xor(part[q], o);
          if (t == 0) mred[(fl * W + w) * NH + q] = part[q];
        }
        shared_sync();
#pragma unroll
        for (int q = 0; q < NH; q++) {
          float sm = 0.0f;
#pragma unroll
          for (int ww = 0; ww < W; ww++) sm += mred[(fl * W + ww) * NH + q];
          mu[q] = sm / (float)p.H;                         // fft.c:91
        }
      }
    }
    v2f32 *xb = lds + ((SETS == 2 ? (sit & 1u) : 0u) * FPB + fl) * (W * STRIP) + w * STRIP;
    W_STAMP(sit, 0);                                       // S begins
    float zr[16], zi[16];
    if constexpr (VAR == 1 && GLFER16W_WPREFETCH == 0) load_window(sj);
    v2f32 wv[VAR == 2 ? 16 : 1];
    if constexpr (VAR == 2) lds_read16_strided<64>(wl + w * (16 * 64) + t, wv);   // plain ds_read_b64, immediate offsets
    static_for<0, 16>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      v2f32 x = sample_pair(mc);
      if constexpr (KM != 0) x = v2f32{x.x - mu[m / (KM ? KM : 1)], x.y - mu[m / (KM ? KM : 1)]};   // fft.c:93-95
      const v2f32 ww = VAR == 2 ? wv[VAR == 2 ? m : 0] : window_pair(mc) * (GEN ? 1.0f : kSampleScale);
      if (GEN != 0 && p.nonlin) {
        // fft.c:127-156: RA9MB x/(a+x^2), window, then sign(y)|y|^0.1; the unit-power scale comes
        // afterwards (post_scale) because the limiter is not linear (the table holds the plain window)
        if (p.a > 0.0f) x = v2f32{x.x / (p.a + x.x * x.x), x.y / (p.a + x.y * x.y)};
        v2f32 y = v2f32{x.x * ww.x, x.y * ww.y};
        if (p.limiter) y = v2f32{limiter_value(y.x), limiter_value(y.y)};
        zr[m] = y.x * p.post_scale;
        zi[m] = y.y * p.post_scale;
      } else {
        zr[m] = x.x * ww.x;
        zi[m] = x.y * ww.y;
      }
    });
    // the next transform's table (VAR 1) and the next frame's samples are requested once pass 0 has
    // handed its data to LDS
    stockham16_passes<10, NT>(zr, zi, xb, t, tw1row, twr, twi, [&] {
      if constexpr (VAR == 1 && GLFER16W_WPREFETCH != 0) {
        if (!last) load_window(sj + 1);
        else if (has_next) load_window(0);
      }
      if (has_next && last) prefetch_x(nf);                // the frame's last use of px is behind us
      W_STAMP(sit, 1);                                     // pass 0 done, exchange 0 written
    });
    W_STAMP(sit, 2);                                       // passes done
    // A_w into the strip, bin k at entry k (the wavefront's last exchange reads have landed)
    static_for<0, 16>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int r = rho_of(m);
      xb[t + 64 * m] = v2f32{zr[r], zi[r]};
    });
    if constexpr (MT != 0) {
      if (t == 0) side[(((unsigned)(sf / FPB) & 1u) * NTMAX + sj) * W + w] = v2f32{zr[rho_of(8)], zi[rho_of(8)]};   // bin 512 = 0 + 64*8
    }
    W_STAMP(sit, 3);                                       // A_w written
    sit++;
    if (last) { sj = 0; sf = nf; } else { sj++; }
  };

  // C: frame group cf, taper cj
  long long cf = fblk;
  int cj = 0;
  unsigned cit = 0;
  auto phase_C = [&] {
    const bool last = cj == ntap - 1;
    if constexpr (MT != 0) {
      if (cj == 0) {                                       // a new frame: the sums start over
#pragma unroll
        for (int i = 0; i < IPL * 2 * W; i++) acc[i] = 0.0f;
      }
    }
    const unsigned cpar = (unsigned)(cf / FPB) & 1u;
    const v2f32 *strips = lds + ((SETS == 2 ? (cit & 1u) : 0u) * FPB + fl) * (W * STRIP);
    W_STAMP(cit, 4);                                       // C begins
    // rows go out through a buffer descriptor over this workgroup's frames; frame slots past the
    // last frame fall outside num_records (their stores are dropped)
    const long long left = p.nframes - cf;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.psd + (size_t)cf * (size_t)p.pitch, 0, (unsigned)((left > FPB ? FPB : left) * (long long)ROWB), 0x00020000);
    // one VGPR offset per direction (bins k upwards from u, bins M-k from the lane's lowest one),
    // everything else a compile-time scalar offset.  (MT stores once per frame, after the last
    // taper: there the offsets are computed at the store, so that no scalar registers stay reserved
    // for them across the taper loop.)
    const unsigned vup = fl * ROWB + u * 4u, vdown = fl * ROWB + ((unsigned)M - u - (unsigned)KMAX) * 4u;
    auto put = [&](float v, unsigned voff, unsigned soff) {
      if constexpr (MT != 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), orsrc, voff + soff, 0, GLFER16W_STORE_AUX);
      else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), orsrc, voff, soff, GLFER16W_STORE_AUX);
    };
    // one item = the bins k1 + 1024 k2 and their mirrors
    auto item = [&](unsigned k1, auto offc, const v2f32 (&tw)[W], float *sum) {
      constexpr int OFF = decltype(offc)::value;           // k1 - u
      float ar[W], ai[W], br[W], bi[W];
      const unsigned k1m = (1024u - k1) & 1023u;           // k1 = 0 pairs with itself
#pragma unroll
      for (int ww = 0; ww < W; ww++) {
        const v2f32 a = strips[ww * STRIP + k1], b = strips[ww * STRIP + k1m];
        if (ww == 0) {
          ar[0] = a.x; ai[0] = a.y; br[0] = b.x; bi[0] = b.y;
        } else {                                           // a * W_M^(ww k1),  b * conj(that)
          const float c = tw[ww].x, sn = tw[ww].y;
          ar[ww] = __builtin_fmaf(a.x, c, a.y * sn);
          ai[ww] = __builtin_fmaf(a.y, c, -a.x * sn);
          br[ww] = __builtin_fmaf(b.x, c, -b.y * sn);
          bi[ww] = __builtin_fmaf(b.y, c, b.x * sn);
        }
      }
      if constexpr (W > 1) {
        dit<W, 1, 0, W>(ar, ai);                           // Z[k1 + 1024 k2] at index brev(k2)
        dit<W, 1, 0, W>(br, bi);                           // Z[1024 - k1 + 1024 (k2 - 1)] at index brev(k2)
      }
      static_for<0, W>([&](auto kc) {
        constexpr int k2 = decltype(kc)::value;
        constexpr int ia = brev(k2, W), ib = brev((W - k2) % W, W);
        const float er = ar[ia] + br[ib], ei = ai[ia] - bi[ib], orr = ar[ia] - br[ib], oi = ai[ia] + bi[ib];
        constexpr cplx64 uu = unit_root(k2, 2 * W);        // (cos, sin)(2 pi 1024 k2 / N)
        constexpr float cm = (float)uu.c, sm = (float)uu.s;
        const float c = k2 == 0 ? tw[0].x : __builtin_fmaf(tw[0].x, cm, -tw[0].y * sm);
        const float sn = k2 == 0 ? tw[0].y : __builtin_fmaf(tw[0].y, cm, tw[0].x * sm);
        const float pr = __builtin_fmaf(c, oi, -sn * orr);             // P = -i (c - i s) O
        const float pi = -__builtin_fmaf(c, orr, sn * oi);
        const float x1r = er + pr, x1i = ei + pi, x2r = er - pr, x2i = ei - pi;
        float v1, v2;
        if constexpr (MT != 0) {
          v1 = sum[2 * k2] = __builtin_fmaf(x1r, x1r, __builtin_fmaf(x1i, x1i, sum[2 * k2]));
          v2 = sum[2 * k2 + 1] = __builtin_fmaf(x2r, x2r, __builtin_fmaf(x2i, x2i, sum[2 * k2 + 1]));
        } else {
          v1 = __builtin_fmaf(x1r, x1r, x1i * x1i);
          v2 = __builtin_fmaf(x2r, x2r, x2i * x2i);
        }
        if (last) {
          put(v1, vup, (unsigned)(OFF + 1024 * k2) * 4u);                       // bin k
          put(v2, vdown, (unsigned)(KMAX - OFF - 1024 * k2) * 4u);              // bin M - k
        }
        if constexpr (GEN != 0 && MT == 0) {
          // fft_do's halfcomplex spectrum (fft_radix2.c:75-177): data[k] = Re X_k, data[N-k] = Im X_k.
          // x1 = 2 s X[k], x2 = 2 s conj(X[M-k]) with s the factor folded into the window.
          if (p.spec && cf + fl < p.nframes) {
            float *o = p.spec + (size_t)(cf + fl) * (2 * M);
            const float inv = 0.5f / p.spec_unscale;
            const unsigned k = k1 + 1024u * k2, km = (unsigned)M - k;
            o[k] = x1r * inv;
            if (k > 0) o[2 * M - k] = x1i * inv;
            o[km] = x2r * inv;
            if (km < (unsigned)M) o[2 * M - km] = -x2i * inv;
          }
        }
      });
    };
    if (LF <= 512 || u < 512u) {
      static_for<0, IPL>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        v2f32 tw[W];
        tw[0] = ct[i][0];
        if constexpr (W == 2) {                            // W_M^k1 = (W_N^k1)^2
          tw[1] = v2f32{__builtin_fmaf(tw[0].x, tw[0].x, -tw[0].y * tw[0].y), 2.0f * tw[0].x * tw[0].y};
        } else {
#pragma unroll
          for (int ww = 1; ww < W; ww++) tw[ww] = ct[i][ww];
        }
        item(u + LF * i, std::integral_constant<int, LF * i>{}, tw, MT ? acc + i * 2 * W : acc);
      });
    }
    if constexpr (MT == 0) {
      if (u == 0) {                                        // k1 = 512, this frame
        float ar[W], ai[W], v[W];
#pragma unroll
        for (int ww = 0; ww < W; ww++) {
          const v2f32 a = strips[ww * STRIP + 512];
          ar[ww] = a.x;
          ai[ww] = a.y;
        }
        special512(ar, ai, v, [&](int k2, float xr, float xi) {
          if constexpr (GEN != 0) {
            if (p.spec && cf + fl < p.nframes) {
              float *o = p.spec + (size_t)(cf + fl) * (2 * M);
              const float inv = 0.5f / p.spec_unscale;
              const unsigned k = 512u + 1024u * k2;
              o[k] = xr * inv;
              o[2 * M - k] = xi * inv;
            }
          }
        });
#pragma unroll
        for (int k2 = 0; k2 < W; k2++) put(v[k2], fl * ROWB, (512u + 1024u * k2) * 4u);
      }
    } else if (last) {
      // every taper's bin 512 is in `side` (written before the barrier that precedes this phase):
      // lane l of the frame's first wavefront takes taper l, then lanes k2 < W add the tapers up in
      // taper order
      if (w == 0) {
        if ((int)t < ntap) {
          float ar[W], ai[W], v[W];
#pragma unroll
          for (int ww = 0; ww < W; ww++) {
            const v2f32 a = side[(cpar * NTMAX + t) * W + ww];
            ar[ww] = a.x;
            ai[ww] = a.y;
          }
          special512(ar, ai, v, [](int, float, float) {});
#pragma unroll
          for (int k2 = 0; k2 < W; k2++) red[t * W + k2] = v[k2];
        }
        frame_sync<64>();
        if (t < (unsigned)W) {
          float sum = 0.0f;
          for (int l = 0; l < ntap; l++) sum += red[l * W + t];
          put(sum, fl * ROWB + (512u + 1024u * t) * 4u, 0u);
        }
        frame_sync<64>();                                  // red is rewritten only a frame later, by this wavefront
      }
    }
    W_STAMP(cit, 5);                                       // C done
    cit++;
    if (last) { cj = 0; cf += FPB; } else { cj++; }
  };

  if constexpr (GLFER16W_SETPRIO != 0 && W * FPB >= 8) {
    if (wv >= (unsigned)(W * FPB) / 2) __builtin_amdgcn_s_setprio(1);
  }
  phase_S();
  shared_sync();
  while (true) {
    const bool more = sf < fend;                           // a transform is left to start
    // two half-steps: C then S, or S then C for the late half of a de-phased workgroup.  One copy of
    // each phase's code: the order is a wave-uniform branch inside a two-trip loop.
    if constexpr (kStagger) {
#pragma unroll 1
      for (int half = 0; half < 2; half++) {
        if ((half == 0) != late) phase_C();
        else if (more) phase_S();
      }
    } else {
      phase_C();
      if constexpr (SETS == 1) shared_sync();              // every reader is done: the strips may be rewritten
      if (more) phase_S();
    }
    if (!more) break;
    W_STAMP(cit, 6);                                       // at the barrier
    shared_sync();
    W_STAMP(cit, 7);                                       // through the barrier
  }
}

}  // namespace glfer

#ifndef GLFER_NO_LAUNCHERS
using namespace glfer;

#define GLFER_CAT2(a, b) a##b
#define GLFER_CAT(a, b) GLFER_CAT2(a, b)

template <int FMT>
static hipError_t launch16w_fmt(const SpectroParams &p, hipStream_t st) {
  constexpr int L = GLFER_LOGN;
  using LC = LaunchW<L>;
  const long long work = ((long long)p.nframes + LC::FPB - 1) / LC::FPB;
  if (work == 0) return hipSuccess;
  constexpr int WPS = GLFER16W_WAVES_PER_SIMD;
  const long long per_cu = (WPS * 256) / LC::BLOCK > 0 ? (WPS * 256) / LC::BLOCK : 1;
  const long long resident = 256LL * per_cu;
  unsigned grid = (unsigned)(work < 8 * resident ? work : 8 * resident);
  if (grid >= 64) grid &= ~7u;                     // whole XCD slices: see xcd_block_index()
#if GLFER_LOGN >= 15
  // N = 32768: the only kernel for this size, so also its general form (zero history, unaligned
  // integer pairs, RA9MB / limiter, spectrum output)
  const bool general = p.nonlin || p.spec || p.frame0 * (long long)p.H < (long long)p.R ||
                       (p.fmt != GLFER_FMT_F32 && ((p.H & 1) || (reinterpret_cast<uintptr_t>(p.stream) & (p.fmt == GLFER_FMT_S16 ? 3u : 1u))));
  if (general) {
    if (p.wtapers > 1) hipLaunchKernelGGL((spectro16w_kernel<L, FMT, 1, 1, GLFER16W_SETS, WPS, 1>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
    else hipLaunchKernelGGL((spectro16w_kernel<L, FMT, 0, 1, GLFER16W_SETS, WPS, 1>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
    return hipGetLastError();
  }
#endif
  if (p.wtapers > 1) {                             // the multitaper form keeps its sums in registers: two waves per SIMD
    constexpr int WPS_MT = (WPS > 2 && L < 15) ? 2 : WPS;
    if (p.mean_inkernel) {
      const int km = (16 * p.H) % (1 << L) == 0 ? 16 * p.H / (1 << L) : 0;
      if (p.history_mode) return hipErrorInvalidValue;
      if (km == 16) hipLaunchKernelGGL((spectro16w_kernel<L, FMT, 1, 1, GLFER16W_SETS, WPS_MT, 0, 0, 16>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
      else if (km == 8) hipLaunchKernelGGL((spectro16w_kernel<L, FMT, 1, 1, GLFER16W_SETS, WPS_MT, 0, 0, 8>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
      else if (km == 4) hipLaunchKernelGGL((spectro16w_kernel<L, FMT, 1, 1, GLFER16W_SETS, WPS_MT, 0, 0, 4>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
      else return hipErrorInvalidValue;
    } else if (p.history_mode) hipLaunchKernelGGL((spectro16w_kernel<L, FMT, 1, 1, GLFER16W_SETS, WPS_MT, 0, 1>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
    else hipLaunchKernelGGL((spectro16w_kernel<L, FMT, 1, 1, GLFER16W_SETS, WPS_MT, 0, 0>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
  } else {
    if (p.mean_inkernel) return hipErrorInvalidValue;
    constexpr int VAR = (L <= 12) ? 2 : 1;         // the window in LDS where it costs no resident workgroup
    if (p.history_mode) hipLaunchKernelGGL((spectro16w_kernel<L, FMT, 0, VAR, GLFER16W_SETS, WPS, 0, 1>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
    else hipLaunchKernelGGL((spectro16w_kernel<L, FMT, 0, VAR, GLFER16W_SETS, WPS, 0, 0>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
  }
  return hipGetLastError();
}

#ifdef GLFER16W_STAMPS
#include <cstdio>
#include <vector>
static hipError_t launch16w_stamped(const SpectroParams &p0, hipStream_t st) {
  static unsigned long long *d_st = nullptr;
  const size_t n = (size_t)16 * 64 * 8;
  if (!d_st && hipMalloc((void **)&d_st, n * 8) != hipSuccess) return hipErrorOutOfMemory;
  (void)hipMemsetAsync(d_st, 0, n * 8, st);
  SpectroParams p = p0;
  p.spec = reinterpret_cast<float *>(d_st);
  hipError_t e = p.fmt == GLFER_FMT_F32 ? launch16w_fmt<GLFER_FMT_F32>(p, st) : hipErrorInvalidValue;
  if (e != hipSuccess) return e;
  std::vector<unsigned long long> h(n);
  if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(h.data(), d_st, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return hipErrorUnknown;
  static int dumps = 0;
  if (dumps++ == 2) {                               // the third launch: caches and clocks are warm
    const int waves = LaunchW<GLFER_LOGN>::BLOCK / 64;
    const unsigned long long t0 = h[(0 * 64 + 20) * 8 + 0];
    fprintf(stderr, "# spectro16w stamps, workgroup 8, transforms 20..25; ticks since wave 0 began S(20)\n");
    fprintf(stderr, "# wave T   S-begin  pass0+x0   passes     A_w-out  | C-begin   C-done   | at-barrier  through\n");
    for (int T = 20; T < 26; T++)
      for (int wvi = 0; wvi < waves; wvi++) {
        const unsigned long long *r = &h[((size_t)wvi * 64 + T) * 8];
        fprintf(stderr, "  %2d  %2d", wvi, T);
        for (int i = 0; i < 8; i++) fprintf(stderr, " %9lld", r[i] ? (long long)(r[i] - t0) : -1LL);
        fprintf(stderr, "\n");
      }
  }
  return hipSuccess;
}
#endif

extern "C" hipError_t GLFER_CAT(glfer_launch_spectro16w_n, GLFER_LOGN)(const SpectroParams *p, hipStream_t st) {
  if (!p->wtaps || !p->wtw || !p->wcomb) return hipErrorInvalidValue;
#ifdef GLFER16W_STAMPS
  if (p->fmt == GLFER_FMT_F32) return launch16w_stamped(*p, st);
#endif
#if GLFER_LOGN < 15
  if (p->nonlin || p->spec) return hipErrorInvalidValue;
  // the gather has no zero-history path: every frame must lie wholly inside the stream
  if (p->frame0 * (long long)p->H < (long long)p->R) return hipErrorInvalidValue;
  if (p->fmt != GLFER_FMT_F32) {                   // integer pairs (y[2n], y[2n+1]) come with one load: naturally aligned
    const unsigned pair = p->fmt == GLFER_FMT_S16 ? 4u : 2u;
    if ((p->H & 1) || (reinterpret_cast<uintptr_t>(p->stream) & (pair - 1u))) return hipErrorInvalidValue;
  }
#endif
  switch (p->fmt) {
    case GLFER_FMT_F32: return launch16w_fmt<GLFER_FMT_F32>(*p, st);
    case GLFER_FMT_S16: return launch16w_fmt<GLFER_FMT_S16>(*p, st);
    case GLFER_FMT_U8: return launch16w_fmt<GLFER_FMT_U8>(*p, st);
  }
  return hipErrorInvalidValue;
}
#endif  // GLFER_NO_LAUNCHERS
