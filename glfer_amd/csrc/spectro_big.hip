// spectro_big.hip -- block sizes whose sub-transforms no longer fit one workgroup's LDS (N = 65536:
// W = N/2048 = 32 wavefront-private 1024-point transforms per frame and taper, 256 KB of results;
// N = 32768 too, where the one-kernel form of spectro16w.hip spills at the 128 VGPRs its 16 wavefronts leave it).
// The same decimation-in-time split as spectro16w.hip, cut into two kernels around a scratch buffer
// in HBM:
//   subfft_kernel  : one wavefront per (frame, taper, w): gather z[W j + w] = y[2n] + i y[2n+1],
//                    window, the 1024-point transform (stockham16.hpp), times W_M^(w k1), out to
//                    scratch[frame][taper][w][k1]
//   combine_kernel : one lane per bin pair (k1, 1024 - k1): the radix-W butterflies, the real-input
//                    split X[k] = (E + P)/2, |X|^2 summed over the tapers, PSD row out
// General by construction: range-checked gather (zero history, any alignment), RA9MB / limiter,
// periodogram and multitaper.  Very-slow-CW sizes; a correct path, not a tuned one.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "stockham16.hpp"

namespace glfer {

template <int FMT>
__global__ __launch_bounds__(256) void subfft_kernel(SpectroParams p, int W, int ntap, long long f0, int nf,
                                                     v2f32 *__restrict__ scratch) {
  using C = Plan16<10>;
  constexpr int TW1 = 15, NT = C::NTW - TW1, STRIP = 1024 + 64;
  constexpr unsigned esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  __shared__ v2f32 lds[4 * STRIP + 16 * 17];
  const unsigned tid = threadIdx.x, t = tid & 63u, wv = tid >> 6;
  v2f32 *tw1 = lds + 4 * STRIP;
  {
    const v2f32 *tw = reinterpret_cast<const v2f32 *>(p.wtw);
    const unsigned k = tid >> 4, q = tid & 15;
    tw1[k * 17 + q] = q ? tw[(q - 1) * 64 + k] : v2f32{1.0f, 0.0f};
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
  __syncthreads();
  const v2f32 *tw1row = tw1 + (t & 15) * 17;
  v2f32 *xb = lds + wv * STRIP;
  const int M = 1024 * W;
  // work item = (frame, taper, w), four consecutive w per workgroup
  const long long items = (long long)nf * ntap * W;
  for (long long it0 = (long long)blockIdx.x * 4; it0 < items; it0 += (long long)gridDim.x * 4) {
    const long long it = it0 + wv;                          // W is a multiple of 4: the four wavefronts share frame and taper
    const bool live = it < items;
    const long long itc = live ? it : items - 1;
    const int w = (int)(itc % W), j = (int)((itc / W) % ntap);
    const long long fr = itc / ((long long)W * ntap);
    const long long f = f0 + fr;
    // ---- gather: sample index in the frame 2(W(t+64m)+w)+e, in the stream s0 + that
    const long long s0 = (p.frame0 + f) * (long long)p.H - p.R;
    const long long sbase = s0 > 0 ? s0 : 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.stream)) + sbase * (long long)esz, 0, 0x7fffffff, 0x00020000);
    const int rel0 = (int)(s0 - sbase) + 2 * (W * (int)t + w);
    typedef float v4f32 __attribute__((ext_vector_type(4)));
    const v4f32 *ht = reinterpret_cast<const v4f32 *>(p.wtaps) + ((size_t)j * W + w) * (8 * 64) + t;
    float zr[16], zi[16];
    const bool inside = s0 >= 0 && p.history_mode == 0;          // workgroup-uniform: no per-sample range test needed
    // The sub-transforms' inputs lie 2W samples apart: gathered lane by lane, a wavefront's load touches 64 cache
    // lines for 8 bytes each (N = 65536: 14 TB/s of L2 traffic for 0.9 TB/s of samples -- what bounded this
    // kernel).  The workgroup's four sub-transforms w0..w0+3 (same frame, same taper) need 32 contiguous bytes
    // per position W(t + 64 m): the workgroup fetches those runs with 16-byte loads and deals them into the four
    // wavefronts' exchange strips (free until the first pass writes them), one barrier, and every wavefront reads
    // its 16 pairs from its own strip.
    __syncthreads();                                             // the strips are free: last iteration's exchanges are over
    if (inside) {
      const int w0 = w - (int)wv;                                // this workgroup's first sub-transform (wave-uniform, a multiple of 4)
      const int relb = (int)(s0 - sbase) + 2 * w0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int pos = (int)tid + 256 * i;
        const unsigned boff = (unsigned)(relb + 2 * W * pos) * esz;      // 8 consecutive samples = the pairs of w0..w0+3
        float sv[8];
        if constexpr (FMT == GLFER_FMT_F32) {
          const auto a = __builtin_amdgcn_raw_buffer_load_b128(rs, boff, 0u, 0), b = __builtin_amdgcn_raw_buffer_load_b128(rs, boff + 16u, 0u, 0);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            sv[e] = __uint_as_float(a[e]);
            sv[4 + e] = __uint_as_float(b[e]);
          }
        } else if constexpr (FMT == GLFER_FMT_S16) {
          const auto a = __builtin_amdgcn_raw_buffer_load_b128(rs, boff, 0u, 0);
          const unsigned q[4] = {a[0], a[1], a[2], a[3]};
#pragma unroll
          for (int e = 0; e < 4; e++) {
            sv[2 * e] = (float)(short)(q[e] & 0xffffu) / 32768.0f;
            sv[2 * e + 1] = (float)(short)(q[e] >> 16) / 32768.0f;
          }
        } else {
          const auto a = __builtin_amdgcn_raw_buffer_load_b64(rs, boff, 0u, 0);
          const unsigned q[2] = {a[0], a[1]};
#pragma unroll
          for (int e = 0; e < 8; e++) sv[e] = ((float)((q[e >> 2] >> (8 * (e & 3))) & 0xffu) - 128.0f) / 128.0f;
        }
#pragma unroll
        for (int ww = 0; ww < 4; ww++) lds[ww * STRIP + pos] = v2f32{sv[2 * ww], sv[2 * ww + 1]};
      }
    }
    __syncthreads();
    static_for<0, 16>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const int off = 2 * W * 64 * m;
      float xv[2];
      if (inside) {
        const v2f32 pr = xb[t + 64 * m];
        xv[0] = pr.x;
        xv[1] = pr.y;
      } else {
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const int rel = rel0 + off + e, jfr = 2 * (W * (int)t + w) + off + e;
          const bool ok = p.history_mode ? (jfr >= p.R) : (rel >= 0);
          const float v = buf_sample<FMT>(rs, ok ? (unsigned)rel * esz : 0x80000000u, 0u);
          xv[e] = ok ? v : 0.0f;
        }
      }
      const v4f32 q = ht[64 * (m / 2)];
      const float w0 = (m & 1) ? q.z : q.x, w1 = (m & 1) ? q.w : q.y;
      if (p.nonlin) {                                       // fft.c:127-156 (the table holds the plain window)
        if (p.a > 0.0f) {
          xv[0] = xv[0] / (p.a + xv[0] * xv[0]);
          xv[1] = xv[1] / (p.a + xv[1] * xv[1]);
        }
        float y0 = xv[0] * w0, y1 = xv[1] * w1;
        if (p.limiter) {
          y0 = limiter_value(y0);
          y1 = limiter_value(y1);
        }
        zr[m] = y0 * p.post_scale;
        zi[m] = y1 * p.post_scale;
      } else {
        zr[m] = xv[0] * w0;
        zi[m] = xv[1] * w1;
      }
    });
    stockham16_passes<10, NT>(zr, zi, xb, t, tw1row, twr, twi, [] {});
    // ---- A_w[k1] * W_M^(w k1) to the scratch; register rho_of(m) holds bin t + 64 m
    if (live) {
      v2f32 *o = scratch + (((size_t)fr * ntap + j) * W + w) * 1024;
      // W_M^(w k1), k1 = t + 64 m: the lane's exp(-2 pi i w t / M) (w t < M: no reduction) times the
      // table's exp(-2 pi i 64 w m / M), which is the same for the whole wavefront (16 sincospif per
      // lane and sub-transform cost as much as the transform itself)
      float sa, ca;
      sincospif(-2.0f * (float)(w * (int)t) / (float)M, &sa, &ca);
      const v2f32 *bt = reinterpret_cast<const v2f32 *>(p.bigtw) + (size_t)__builtin_amdgcn_readfirstlane(w) * 16;
      static_for<0, 16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int r = (m % 4) + 4 * brev(m / 4, 4);
        const int k1 = (int)t + 64 * m;
        const v2f32 b = bt[m];
        const float cs = __builtin_fmaf(ca, b.x, -sa * b.y), sn = __builtin_fmaf(sa, b.x, ca * b.y);
        o[k1] = v2f32{__builtin_fmaf(zr[r], cs, -zi[r] * sn), __builtin_fmaf(zr[r], sn, zi[r] * cs)};
      });
    }
  }
}

// one lane per item k1 = 0..512 of a frame: bins k1 + 1024 k2 and their mirrors M - k
template <int W>
__global__ __launch_bounds__(64) void combine_kernel(SpectroParams p, int ntap, long long f0, int nf,
                                                     const v2f32 *__restrict__ scratch) {
  const int M = 1024 * W;
  const long long fr = blockIdx.y;
  const int k1 = blockIdx.x * 64 + threadIdx.x;
  if (fr >= nf || k1 > 512) return;
  const int k1m = (1024 - k1) & 1023;
  float acc[2 * W];
#pragma unroll
  for (int i = 0; i < 2 * W; i++) acc[i] = 0.0f;
  float c0, s0;                                            // (cos, sin)(2 pi k1 / N), N = 2M
  {
    float sn, cs;
    sincospif((float)k1 / (float)M, &sn, &cs);
    c0 = cs;
    s0 = sn;
  }
  for (int j = 0; j < ntap; j++) {
    const v2f32 *S = scratch + (((size_t)fr * ntap + j) * W) * 1024;
    float ar[W], ai[W], br[W], bi[W];
#pragma unroll
    for (int w = 0; w < W; w++) {
      const v2f32 a = S[(size_t)w * 1024 + k1], b = S[(size_t)w * 1024 + k1m];
      ar[w] = a.x; ai[w] = a.y; br[w] = b.x; bi[w] = b.y;
    }
    dit<W, 1, 0, W>(ar, ai);                               // Z[k1 + 1024 k2] at index brev(k2)
    dit<W, 1, 0, W>(br, bi);                               // Z[1024 - k1 + 1024 k2] at index brev(k2)
    static_for<0, W>([&](auto kc) {
      constexpr int k2 = decltype(kc)::value;
      constexpr int ia = brev(k2, W), ib = brev(W - 1 - k2, W), ic = brev((W - k2) % W, W);
      // mirror partner of k = k1 + 1024 k2: M - k = (1024 - k1) + 1024 (W-1-k2); for k1 = 0 it is 1024 (W - k2)
      const float qr = k1 == 0 ? ar[ic] : br[ib], qi = k1 == 0 ? ai[ic] : bi[ib];
      const float er = ar[ia] + qr, ei = ai[ia] - qi, orr = ar[ia] - qr, oi = ai[ia] + qi;
      constexpr cplx64 uu = unit_root(k2, 2 * W);          // (cos, sin)(2 pi 1024 k2 / N)
      constexpr float cm = (float)uu.c, sm = (float)uu.s;
      const float c = __builtin_fmaf(c0, cm, -s0 * sm), s = __builtin_fmaf(s0, cm, c0 * sm);
      const float pr = __builtin_fmaf(c, oi, -s * orr), pi = -__builtin_fmaf(c, orr, s * oi);
      const float x1r = er + pr, x1i = ei + pi, x2r = er - pr, x2i = ei - pi;
      acc[2 * k2] = __builtin_fmaf(x1r, x1r, __builtin_fmaf(x1i, x1i, acc[2 * k2]));
      acc[2 * k2 + 1] = __builtin_fmaf(x2r, x2r, __builtin_fmaf(x2i, x2i, acc[2 * k2 + 1]));
    });
  }
  float *o = p.psd + (size_t)(f0 + fr) * (size_t)p.pitch;
#pragma unroll
  for (int k2 = 0; k2 < W; k2++) {
    o[k1 + 1024 * k2] = acc[2 * k2];
    o[M - k1 - 1024 * k2] = acc[2 * k2 + 1];               // (k1 = 0 and 512 write some bins twice, with equal values)
  }
}

// ---- N >= 131072 (W = N/2048 >= 64 sub-transforms per frame and taper): the W-point transforms over w no longer fit
// a lane's registers, so the combine is cut once more, decimation in time, W = 32 W1:
//     Z[k1 + 1024 (k2a + 32 k2b)] = sum_{w1 < W1} W_W1^(w1 k2b) * { W_W^(w1 k2a) * sum_{w2 < 32} c[W1 w2 + w1][k1] W_32^(w2 k2a) }
// with c[w][k1] = W_M^(w k1) A_w[k1] what subfft_kernel leaves in the scratch.
//   combine1_kernel : one lane per (frame, taper, w1, k1): the radix-32 transform over w2 in registers, times
//                     W_W^(w1 k2a), out to a second scratch D[frame][taper][w1][k2a][k1]
//   combine2_kernel : one lane per (frame, k2a, bin pair k1 / 1024 - k1): the radix-W1 transforms over w1 of the lane's
//                     column and of its mirror partner's, the real-input split, |X|^2 summed over the tapers, rows out
// Very-slow-CW sizes (a 131072-point block is 16 s of audio at 8 kHz): a correct path, not a tuned one.
__global__ __launch_bounds__(64) void combine1_kernel(int W, int W1, const v2f32 *__restrict__ S, v2f32 *__restrict__ D) {
  const int k1 = blockIdx.x * 64 + threadIdx.x;                  // 0 .. 1023
  const int w1 = blockIdx.y;
  const size_t ft = blockIdx.z;                                  // frame * ntap + taper
  float ar[32], ai[32];
#pragma unroll
  for (int w2 = 0; w2 < 32; w2++) {
    const v2f32 a = S[(ft * W + (size_t)(W1 * w2 + w1)) * 1024 + k1];
    ar[w2] = a.x;
    ai[w2] = a.y;
  }
  dit<32, 1, 0, 32>(ar, ai);                                     // the sum over w2 for k2a sits at index brev(k2a, 32)
  v2f32 *o = D + ((ft * W1 + w1) * 32) * 1024 + k1;
  static_for<0, 32>([&](auto kc) {
    constexpr int k2a = decltype(kc)::value;
    constexpr int r = brev(k2a, 32);
    float sn, cs;
    sincospif(-2.0f * (float)(w1 * k2a) / (float)W, &sn, &cs);   // W_W^(w1 k2a); w1 k2a < W: no reduction needed
    o[(size_t)k2a * 1024] = v2f32{__builtin_fmaf(ar[r], cs, -ai[r] * sn), __builtin_fmaf(ar[r], sn, ai[r] * cs)};
  });
}

template <int W1>
__global__ __launch_bounds__(64) void combine2_kernel(SpectroParams p, int ntap, long long f0, int nf,
                                                      const v2f32 *__restrict__ D) {
  constexpr int W = 32 * W1;
  const long long M = 1024LL * W;
  const int k1 = blockIdx.x * 64 + threadIdx.x;
  const int k2a = blockIdx.y;
  const long long fr = blockIdx.z;
  if (fr >= nf || k1 > 512) return;
  // the mirror partner of k = k1 + 1024 k2 is M - k = (1024 - k1) + 1024 (W - 1 - k2): column (31 - k2a) of item
  // 1024 - k1, index W1 - 1 - k2b; for k1 = 0 it is 1024 (W - k2): column (32 - k2a) mod 32 of item 0, index
  // W1 - 1 - k2b (k2a > 0) or (W1 - k2b) mod W1 (k2a = 0)
  const int k1m = (1024 - k1) & 1023;
  const int k2am = k1 == 0 ? ((32 - k2a) & 31) : 31 - k2a;
  const bool selfcol = k1 == 0 && k2a == 0;
  float acc[2 * W1];
#pragma unroll
  for (int i = 0; i < 2 * W1; i++) acc[i] = 0.0f;
  float c0, s0;                                                  // (cos, sin)(2 pi (k1 + 1024 k2a) / N), N = 2M
  {
    float sn, cs;
    sincospif((float)(k1 + 1024 * k2a) / (float)M, &sn, &cs);
    c0 = cs;
    s0 = sn;
  }
  for (int j = 0; j < ntap; j++) {
    const v2f32 *Dj = D + (((size_t)fr * ntap + j) * W1) * 32 * 1024;
    float ar[W1], ai[W1], br[W1], bi[W1];
#pragma unroll
    for (int w1 = 0; w1 < W1; w1++) {
      const v2f32 a = Dj[((size_t)w1 * 32 + k2a) * 1024 + k1], b = Dj[((size_t)w1 * 32 + k2am) * 1024 + k1m];
      ar[w1] = a.x; ai[w1] = a.y; br[w1] = b.x; bi[w1] = b.y;
    }
    dit<W1, 1, 0, W1>(ar, ai);                                   // Z[k1 + 1024 (k2a + 32 k2b)] at index brev(k2b, W1)
    dit<W1, 1, 0, W1>(br, bi);                                   // the partner column, likewise
    static_for<0, W1>([&](auto kc) {
      constexpr int k2b = decltype(kc)::value;
      constexpr int ia = brev(k2b, W1), ib = brev(W1 - 1 - k2b, W1), ic = brev((W1 - k2b) % W1, W1);
      const float qr = selfcol ? ar[ic] : br[ib], qi = selfcol ? ai[ic] : bi[ib];
      const float er = ar[ia] + qr, ei = ai[ia] - qi, orr = ar[ia] - qr, oi = ai[ia] + qi;
      constexpr cplx64 uu = unit_root(k2b, 2 * W1);              // (cos, sin)(2 pi 1024 * 32 k2b / N)
      constexpr float cm = (float)uu.c, sm = (float)uu.s;
      const float c = __builtin_fmaf(c0, cm, -s0 * sm), s = __builtin_fmaf(s0, cm, c0 * sm);
      const float pr = __builtin_fmaf(c, oi, -s * orr), pi = -__builtin_fmaf(c, orr, s * oi);
      const float x1r = er + pr, x1i = ei + pi, x2r = er - pr, x2i = ei - pi;
      acc[2 * k2b] = __builtin_fmaf(x1r, x1r, __builtin_fmaf(x1i, x1i, acc[2 * k2b]));
      acc[2 * k2b + 1] = __builtin_fmaf(x2r, x2r, __builtin_fmaf(x2i, x2i, acc[2 * k2b + 1]));
    });
  }
  float *o = p.psd + (size_t)(f0 + fr) * (size_t)p.pitch;
#pragma unroll
  for (int k2b = 0; k2b < W1; k2b++) {
    const long long k = k1 + 1024LL * (k2a + 32 * k2b);
    o[k] = acc[2 * k2b];
    // the mirrored bin M - k: for 0 < k1 < 512 nobody else computes it.  For k1 = 0 and k1 = 512 it is another lane's
    // bin k (another k2a), computed there in another operation order -- one writer per bin, or a row's last bits would
    // depend on which lane stores last; only bin M (the mirror of bin 0) has no lane of its own.
    if ((k1 > 0 && k1 < 512) || k == 0) o[M - k] = acc[2 * k2b + 1];
  }
}

}  // namespace glfer

using namespace glfer;

namespace glfer {
hipError_t scratch_malloc(void **p, size_t bytes, hipStream_t st);   // plan.h / glfer_hip.cpp: stream-ordered; large requests from blocks the library keeps
void scratch_free(void *p, hipStream_t st);
}

// frames [p->frame0 .. +nframes) in groups that keep the scratch under ~512 MiB; p->psd is row 0 of the launch
extern "C" hipError_t glfer_launch_spectro_big(const SpectroParams *p, int n, hipStream_t st) {
  if (n < 32768 || n > (1 << 20) || (n & (n - 1)) || !p->wtaps || !p->wtw || !p->bigtw || p->spec) return hipErrorInvalidValue;
  if (p->nframes <= 0) return hipSuccess;
  const int W = n / 2048;
  const bool two_level = W > 32;                                 // N >= 131072: combine1 + combine2 around a second scratch
  const int W1 = two_level ? W / 32 : 1;
  const int ntap = p->wtapers > 0 ? p->wtapers : 1;
  const size_t per_frame = (size_t)ntap * W * 1024 * sizeof(v2f32);
  long long group = (long long)(((size_t)512 << 20) / per_frame);
  if (group < 1) group = 1;
  if (group > p->nframes) group = p->nframes;
  if (two_level && group > 65535 / ntap) group = 65535 / ntap;   // (grid.z of combine1)
  v2f32 *scratch = nullptr, *scratch2 = nullptr;
  hipError_t e = glfer::scratch_malloc((void **)&scratch, (size_t)group * per_frame, st);
  if (e != hipSuccess) return e;
  if (two_level) {
    e = glfer::scratch_malloc((void **)&scratch2, (size_t)group * per_frame, st);
    if (e != hipSuccess) {
      glfer::scratch_free(scratch, st);
      return e;
    }
  }
  for (long long f0 = 0; f0 < p->nframes && e == hipSuccess; f0 += group) {
    const int nf = (int)((p->nframes - f0 < group) ? p->nframes - f0 : group);
    const long long blocks = ((long long)nf * ntap * W + 3) / 4;
    const unsigned grid = (unsigned)(blocks < 4096 ? blocks : 4096);
    switch (p->fmt) {
      case GLFER_FMT_F32: hipLaunchKernelGGL(subfft_kernel<GLFER_FMT_F32>, dim3(grid), dim3(256), 0, st, *p, W, ntap, f0, nf, scratch); break;
      case GLFER_FMT_S16: hipLaunchKernelGGL(subfft_kernel<GLFER_FMT_S16>, dim3(grid), dim3(256), 0, st, *p, W, ntap, f0, nf, scratch); break;
      case GLFER_FMT_U8: hipLaunchKernelGGL(subfft_kernel<GLFER_FMT_U8>, dim3(grid), dim3(256), 0, st, *p, W, ntap, f0, nf, scratch); break;
      default: e = hipErrorInvalidValue;
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess && two_level) {
      hipLaunchKernelGGL(combine1_kernel, dim3(16, (unsigned)W1, (unsigned)(nf * ntap)), dim3(64), 0, st, W, W1, scratch, scratch2);
      e = hipGetLastError();
      if (e == hipSuccess) {
        const dim3 g2(9, 32, (unsigned)nf);
        switch (W1) {
          case 2: hipLaunchKernelGGL(combine2_kernel<2>, g2, dim3(64), 0, st, *p, ntap, f0, nf, scratch2); break;
          case 4: hipLaunchKernelGGL(combine2_kernel<4>, g2, dim3(64), 0, st, *p, ntap, f0, nf, scratch2); break;
          case 8: hipLaunchKernelGGL(combine2_kernel<8>, g2, dim3(64), 0, st, *p, ntap, f0, nf, scratch2); break;
          case 16: hipLaunchKernelGGL(combine2_kernel<16>, g2, dim3(64), 0, st, *p, ntap, f0, nf, scratch2); break;
          default: e = hipErrorInvalidValue;
        }
        if (e == hipSuccess) e = hipGetLastError();
      }
    } else if (e == hipSuccess) {
      if (W == 32) hipLaunchKernelGGL(combine_kernel<32>, dim3(9, (unsigned)nf), dim3(64), 0, st, *p, ntap, f0, nf, scratch);
      else hipLaunchKernelGGL(combine_kernel<16>, dim3(9, (unsigned)nf), dim3(64), 0, st, *p, ntap, f0, nf, scratch);
      e = hipGetLastError();
    }
  }
  glfer::scratch_free(scratch, st);
  if (scratch2) glfer::scratch_free(scratch2, st);
  return e;
}
