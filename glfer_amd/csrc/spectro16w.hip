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
#ifndef GLFER16W_STORE_AUX
#define GLFER16W_STORE_AUX (GLFER_LOGN >= 12 ? 2 : 0)   /* non-temporal rows from N = 4096 up (spectro16h.hip measured it) */
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
template <int LOGN, int FMT, int MT, int VAR, int SETS, int WPS>
__global__ __launch_bounds__(LaunchW<LOGN>::BLOCK, WPS) void spectro16w_kernel(SpectroParams p) {
  using L = LaunchW<LOGN>;
  using C = Plan16<10>;
  constexpr int N = L::N, M = L::M, W = L::W, FPB = L::FPB, LF = L::LF, IPL = L::IPL, STRIP = L::STRIP;
  constexpr int TW1 = 15, NT = C::NTW - TW1;               // pass 2's twiddles per lane (12)
  constexpr unsigned esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  constexpr float kSampleScale = FMT == GLFER_FMT_F32 ? 1.0f : (FMT == GLFER_FMT_S16 ? 1.0f / 32768.0f : 1.0f / 128.0f);
  static_assert(MT == 0 || VAR == 1, "the multitaper form reads a table per taper");
  __shared__ v2f32 lds[FPB * W * STRIP * SETS + 16 * 17 + (VAR == 2 ? M : 0)];

  const unsigned tid = threadIdx.x;
  const unsigned t = tid & 63u;
  const unsigned wv = tid >> 6;
  const unsigned w = wv % W;                               // which sub-transform of the frame
  const unsigned fl = wv / W;                              // frame slot in the workgroup
  const unsigned u = w * 64u + t;                          // lane index within the frame
  v2f32 *tw1 = lds + FPB * W * STRIP * SETS;
  v2f32 *wl = tw1 + 16 * 17;                               // VAR 2: window pairs, [w][m][t]

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
  v2f32 ct[IPL][W];
  {
    const v2f32 *cw = reinterpret_cast<const v2f32 *>(p.wcomb) + u;
#pragma unroll
    for (int i = 0; i < IPL; i++)
#pragma unroll
      for (int ww = 0; ww < W; ww++) ct[i][ww] = cw[(i * W + ww) * LF];
  }
  typedef float v4f32 __attribute__((ext_vector_type(4)));
  v2f32 wn[16];
  auto load_window = [&](int j) {                          // table [taper][w][m/2][t][4]
    const v4f32 *ht = reinterpret_cast<const v4f32 *>(p.wtaps) + ((size_t)j * W + w) * (8 * 64) + t;
#pragma unroll
    for (int mh = 0; mh < 8; mh++) {
      const v4f32 q = ht[64 * mh];
      wn[2 * mh] = v2f32{q.x, q.y} * kSampleScale;
      wn[2 * mh + 1] = v2f32{q.z, q.w} * kSampleScale;
    }
  };
  if constexpr (VAR == 2) {
    if (fl == 0) {
      load_window(0);
#pragma unroll
      for (int m = 0; m < 16; m++) wl[(w * 16 + m) * 64 + t] = wn[m];
    }
  }
  __syncthreads();
  const v2f32 *tw1row = tw1 + (t & 15) * 17;

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
    if (p.history_mode) {        // sample j = 2n + e is kept iff j >= R (fft.c:103-108); a rare mode
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
  auto sample_pair = [&](auto mc) -> v2f32 {
    constexpr int m = decltype(mc)::value;
    if constexpr (FMT == GLFER_FMT_F32) {
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
  unsigned it = 0;                                         // transforms done: picks the strip set

  while (true) {
    const long long nfblk = fblk + FPB;
    const bool has_next = nfblk < fend;
    float acc[MT ? IPL * 2 * W : 1], accs[MT ? 2 * W : 1];  // MT: [item][k2][bin k | bin M-k]; accs: the k1 = 512 item
    if constexpr (MT != 0) {
#pragma unroll
      for (int i = 0; i < IPL * 2 * W; i++) acc[i] = 0.0f;
#pragma unroll
      for (int i = 0; i < 2 * W; i++) accs[i] = 0.0f;
    }
    // rows go out through a buffer descriptor over this workgroup's frames; frame slots past the
    // last frame fall outside num_records (their stores are dropped)
    constexpr unsigned ROWB = (unsigned)(M + 1) * 4u;
    const long long left = p.nframes - fblk;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.psd + (size_t)fblk * (M + 1), 0, (unsigned)((left > FPB ? FPB : left) * (long long)ROWB), 0x00020000);
    auto put = [&](float v, unsigned bin) {
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), orsrc, fl * ROWB + bin * 4u, 0, GLFER16W_STORE_AUX);
    };

    for (int j = 0; j < ntap; j++) {
      const bool last = j == ntap - 1;
      v2f32 *strips = lds + ((SETS == 2 ? (it & 1u) : 0u) * FPB + fl) * (W * STRIP);   // this frame's W strips
      v2f32 *xb = strips + w * STRIP;
      it++;
      float zr[16], zi[16];
      if constexpr (VAR == 1) load_window(j);
      static_for<0, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const v2f32 x = sample_pair(mc);
        const v2f32 ww = VAR == 2 ? wl[(w * 16 + m) * 64 + t] : wn[m];
        zr[m] = x.x * ww.x;
        zi[m] = x.y * ww.y;
      });

      // ---- the wavefront's own 1024-point transform
      stockham16_passes<10, NT>(zr, zi, xb, t, tw1row, twr, twi, [&] {
        if (has_next && last) prefetch_x(nfblk);           // the frame's last use of px is behind us
      });
      // A_w into the strip, bin k at entry k (the wavefront's last exchange reads have landed)
      static_for<0, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int r = rho_of(m);
        xb[t + 64 * m] = v2f32{zr[r], zi[r]};
      });
      shared_sync();

      // ---- shared pass + real-input split, one item = the bins k1 + 1024 k2 and their mirrors
      auto item = [&](unsigned k1, const v2f32 (&tw)[W], float *sum, bool store) {
        float ar[W], ai[W], br[W], bi[W];
        const unsigned k1m = (1024u - k1) & 1023u;         // k1 = 0 pairs with itself
#pragma unroll
        for (int ww = 0; ww < W; ww++) {
          const v2f32 a = strips[ww * STRIP + k1], b = strips[ww * STRIP + k1m];
          if (ww == 0) {
            ar[0] = a.x; ai[0] = a.y; br[0] = b.x; bi[0] = b.y;
          } else {                                         // a * W_M^(ww k1),  b * conj(that)
            const float c = tw[ww].x, s = tw[ww].y;
            ar[ww] = __builtin_fmaf(a.x, c, a.y * s);
            ai[ww] = __builtin_fmaf(a.y, c, -a.x * s);
            br[ww] = __builtin_fmaf(b.x, c, -b.y * s);
            bi[ww] = __builtin_fmaf(b.y, c, b.x * s);
          }
        }
        if constexpr (W > 1) {
          dit<W, 1, 0, W>(ar, ai);                         // Z[k1 + 1024 k2] at index brev(k2)
          dit<W, 1, 0, W>(br, bi);                         // Z[1024 - k1 + 1024 (k2 - 1)] at index brev(k2)
        }
        static_for<0, W>([&](auto kc) {
          constexpr int k2 = decltype(kc)::value;
          constexpr int ia = brev(k2, W), ib = brev((W - k2) % W, W);
          const float er = ar[ia] + br[ib], ei = ai[ia] - bi[ib], orr = ar[ia] - br[ib], oi = ai[ia] + bi[ib];
          constexpr cplx64 uu = unit_root(k2, 2 * W);      // (cos, sin)(2 pi 1024 k2 / N)
          constexpr float cm = (float)uu.c, sm = (float)uu.s;
          const float c = k2 == 0 ? tw[0].x : __builtin_fmaf(tw[0].x, cm, -tw[0].y * sm);
          const float s = k2 == 0 ? tw[0].y : __builtin_fmaf(tw[0].y, cm, tw[0].x * sm);
          const float pr = __builtin_fmaf(c, oi, -s * orr);            // P = -i (c - i s) O
          const float pi = -__builtin_fmaf(c, orr, s * oi);
          const float x1r = er + pr, x1i = ei + pi, x2r = er - pr, x2i = ei - pi;
          float v1, v2;
          if constexpr (MT != 0) {
            v1 = sum[2 * k2] = __builtin_fmaf(x1r, x1r, __builtin_fmaf(x1i, x1i, sum[2 * k2]));
            v2 = sum[2 * k2 + 1] = __builtin_fmaf(x2r, x2r, __builtin_fmaf(x2i, x2i, sum[2 * k2 + 1]));
          } else {
            v1 = __builtin_fmaf(x1r, x1r, x1i * x1i);
            v2 = __builtin_fmaf(x2r, x2r, x2i * x2i);
          }
          if (store) {
            put(v1, k1 + 1024u * k2);                      // bin k
            put(v2, (unsigned)M - k1 - 1024u * k2);        // bin M - k
          }
        });
      };
      if (LF <= 512 || u < 512u) {
#pragma unroll
        for (int i = 0; i < IPL; i++) item(u + LF * i, ct[i], MT ? acc + i * 2 * W : acc, last);
      }
      if (u == 0) {                                        // k1 = 512: 1024 - k1 is k1 again
        v2f32 tw[W];
        static_for<0, W>([&](auto wc) {
          constexpr int ww = decltype(wc)::value;
          constexpr cplx64 a = unit_root(ww == 0 ? 1 : ww, ww == 0 ? 4 * W : 2 * W);   // [0]: 2 pi 512/N; [ww]: 2 pi 512 ww/M
          tw[ww] = v2f32{(float)a.c, (float)a.s};
        });
        item(512u, tw, accs, last);
      }
      if constexpr (SETS == 1) shared_sync();              // every reader is done: the strips may be rewritten
    }
    if (!has_next) break;
    fblk = nfblk;
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
  if (p.wtapers > 1) {
    hipLaunchKernelGGL((spectro16w_kernel<L, FMT, 1, 1, GLFER16W_SETS, WPS>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
  } else {
    constexpr int VAR = (L <= 12) ? 2 : 1;         // the window in LDS where it costs no resident workgroup
    hipLaunchKernelGGL((spectro16w_kernel<L, FMT, 0, VAR, GLFER16W_SETS, WPS>), dim3(grid), dim3(LC::BLOCK), 0, st, p);
  }
  return hipGetLastError();
}

extern "C" hipError_t GLFER_CAT(glfer_launch_spectro16w_n, GLFER_LOGN)(const SpectroParams *p, hipStream_t st) {
  if (!p->wtaps || !p->wtw || !p->wcomb || p->nonlin || p->spec) return hipErrorInvalidValue;
  // the gather has no zero-history path: every frame must lie wholly inside the stream
  if (p->frame0 * (long long)p->H < (long long)p->R) return hipErrorInvalidValue;
  if (p->fmt != GLFER_FMT_F32) {                   // integer pairs (y[2n], y[2n+1]) come with one load: naturally aligned
    const unsigned pair = p->fmt == GLFER_FMT_S16 ? 4u : 2u;
    if ((p->H & 1) || (reinterpret_cast<uintptr_t>(p->stream) & (pair - 1u))) return hipErrorInvalidValue;
  }
  switch (p->fmt) {
    case GLFER_FMT_F32: return launch16w_fmt<GLFER_FMT_F32>(*p, st);
    case GLFER_FMT_S16: return launch16w_fmt<GLFER_FMT_S16>(*p, st);
    case GLFER_FMT_U8: return launch16w_fmt<GLFER_FMT_U8>(*p, st);
  }
  return hipErrorInvalidValue;
}
#endif  // GLFER_NO_LAUNCHERS
