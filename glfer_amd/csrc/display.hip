// display.hip -- the waterfall column mapping that follows the estimator in the reference's draw
// routine (main_window_draw, g_main.c:1099-1236), batched over frames:
//   K7a levels_kernel : the autoscale recurrence on (sig, floor) -> display_max/min per frame
//                       (g_main.c:1111-1139).  The recurrence rounds to float at every frame:
//                       sequential, but contractive -- chunks walk in parallel after a warm-up and a
//                       fix-up kernel verifies every seam bit for bit (see below).
//   K7b map_kernel    : PSD (or averaged PSD) -> dB short (levbuf) -> 0..255 -> palette RGB
//                       (g_main.c:1186-1236).  One block per frame, HBM-bound byte/short stores;
//                       the 768-byte palette sits in LDS.
// Built with -ffp-contract=off: the reference's float/double expression order is the contract.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "display_map.hpp"

namespace glfer {

struct LevelsParams {
  int scale_log, autoscale, first_buffer;
  float overlap;
  float max_lvl0, min_lvl0;      // state carried in (autoscale) or the fixed levels (not autoscale)
};

// levels: [nframes][4] = {display_max, display_min, display_max_lvl, display_min_lvl}
//
// The recurrence lvl = (float)(0.01 x + 0.99 lvl) rounds to float at every frame: sequential.  It
// is also a contraction (0.99 per frame), so two walks over the same frames from different states
// approach each other and, once within an ulp, merge exactly (each frame merges a 1-ulp gap with
// probability ~1 %).  The frames are therefore cut into chunks of LEV_CHUNK walked in parallel, each
// chunk warming up over the frames before it from an approximate state (from the true state when
// that reaches back to frame 0).  That the warm-up has merged with the true walk is not left to
// chance: levels_fixup_kernel compares every chunk's warm-up end state with its predecessor's final
// state, bit for bit, and re-walks the chunk from the true state where they differ.
// 131 072 columns: 10 M columns/s as one walk, ~40x that in chunks.
// A block walks warm-up + CHUNK frames, ~60 ns a frame, and a chunk that fails the check costs CHUNK
// frames in the one fix-up block: the walk is ~(warm-up + CHUNK + nframes x P(fail)) frames long whatever
// CHUNK is, so short chunks -- as long as the fix-up does not visit them one by one (it compares 64
// at a time).  256 against 1024: display with autoscale +3 % (two same-box runs).
#ifndef GLFER_LEV_CHUNK
#define GLFER_LEV_CHUNK 256
#endif
constexpr int LEV_CHUNK = GLFER_LEV_CHUNK;
// A better starting state shortens the warm-up: without the rounding to float the recurrence has the closed form
// lvl[f] = sum_d 0.01 * 0.99^d * x[f - d] (0.99^2048 = 1e-9: 2048 terms are all of it), a sum every lane can take a
// share of.  That value is a few float ulp from the true state (whose roundings it ignores), so the walk from it is
// within an ulp of the true walk at once and merges with it like any 1-ulp gap, ~1 % per frame: LEV_WARM_SEEDED frames
// of warm-up (round 1: 4096 from an arbitrary state), the same verification (and re-walk) by levels_fixup_kernel.
// Measured (131 072 columns, display with autoscale): 512 frames 169, 1024 frames 190, 2048 frames 178 M rows/s --
// shorter, and the re-walks of the chunks that have not merged cost more than the warm-up saves.
#ifndef GLFER_LEV_WARM_SEEDED
#define GLFER_LEV_WARM_SEEDED 1024
#endif
constexpr int LEV_WARM_SEEDED = GLFER_LEV_WARM_SEEDED, LEV_SEED = 2048;

// Lanes 0 / 1 of a 64-lane block carry the max / min chains over frames [from, to); the other lanes
// stage loads and stores through LDS.  Frames >= out_from get their levels written.  The chain is the
// critical path (four dependent conversions/FMAs per frame), so nothing else may sit on it: a batch's
// 64 inputs are in the chain lane's registers before it starts, and the next batch's statistics are
// already on their way from memory.
__device__ __forceinline__ void levels_walk(const float *__restrict__ stats, long long from, long long to,
                                            long long out_from, const LevelsParams &p, float overlap, float &lvl,
                                            bool &first, float *__restrict__ levels, float (&sx)[2][64],
                                            float (&sy)[2][64]) {
  const int lane = threadIdx.x;
  float r0 = 0.0f, r1 = 0.0f;
  if (from + lane < to) {
    r0 = stats[(from + lane) * 4 + 0];
    r1 = stats[(from + lane) * 4 + 1];
  }
  for (long long base = from; base < to; base += 64) {
    const long long f = base + lane;
    sx[0][lane] = r0;
    sx[1][lane] = r1;
    __syncthreads();
    if (f + 64 < to) {                                     // the next batch, in flight under this one's chain
      r0 = stats[(f + 64) * 4 + 0];
      r1 = stats[(f + 64) * 4 + 1];
    }
    if (lane < 2) {
      const int cnt = (int)((to - base < 64) ? (to - base) : 64);
      float x[64];
#pragma unroll
      for (int j = 0; j < 64; j++) x[j] = sx[lane][j];
      if (cnt == 64 && !first) {
        // a whole batch in steady state -- nearly every batch: nothing but the recurrence on the
        // chain (one conversion, one multiply, one add, one conversion per frame); the products
        // (1.0 - 0.99) * x do not depend on it
        double xs[64];
#pragma unroll
        for (int j = 0; j < 64; j++) xs[j] = (1.0 - 0.99) * (double)x[j];
#pragma unroll
        for (int j = 0; j < 64; j++) {
          lvl = (float)(xs[j] + 0.99 * (double)lvl);       // g_main.c:1122-1123
          x[j] = lvl;
        }
      } else {
#pragma unroll 1
        for (int j = 0; j < 64; j++) {
          if (j < cnt) {
            if (first) {                                   // g_main.c:1112-1120
              float x0 = x[j];
              if (overlap > 0.0) x0 /= overlap;
              lvl = x0;
              first = false;
            } else {                                       // g_main.c:1122-1123
              lvl = (float)((1.0 - 0.99) * (double)x[j] + 0.99 * (double)lvl);
            }
          }
          x[j] = lvl;
        }
      }
#pragma unroll
      for (int j = 0; j < 64; j++) sy[lane][j] = x[j];
    }
    __syncthreads();
    if (f < to && f >= out_from) {
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

// chunk state: [chunk][4] = {warm-up end max, min, final max, min}
__global__ __launch_bounds__(64) void levels_kernel(const float *__restrict__ stats, long long nframes,
                                                    LevelsParams p, float *__restrict__ levels,
                                                    float *__restrict__ chunk_state) {
  __shared__ float sx[2][64];
  __shared__ float sy[2][64];
  const int lane = threadIdx.x;
  const long long c = blockIdx.x, begin = c * LEV_CHUNK;
  const long long end = begin + LEV_CHUNK < nframes ? begin + LEV_CHUNK : nframes;
  const long long ws = begin - LEV_WARM_SEEDED;
  float lvl = (lane == 0) ? p.max_lvl0 : p.min_lvl0;       // the state carried into the call
  bool first;
  if (ws <= 0) {                                           // the warm-up would start at frame 0: walk from the true state
    first = p.first_buffer != 0;
    levels_walk(stats, 0, begin, begin, p, p.overlap, lvl, first, levels, sx, sy);
  } else {
    // warm-up from the closed form of the recurrence at frame ws - 1: lane l sums the terms d = l + 64 i.
    // Where the 2048 terms reach frame 0 the sum ends there with the start of the true walk: the
    // state carried into the call (0.99^ws of it is left), or -- first buffer -- frame 0 taken whole
    // (g_main.c:1112-1120: lvl = x0 [/ overlap], of which 0.99^(ws-1) is left).  (Walking those
    // chunks from frame 0 instead made the first of them, up to 3072 frames, the longest block by 3x.)
    double w = 0.01 * pow(0.99, (double)lane), s0 = 0.0, s1 = 0.0;
    const double r64 = pow(0.99, 64.0);
    const bool whole0 = p.first_buffer != 0;
#pragma unroll 4
    for (int i = 0; i < LEV_SEED / 64; i++) {
      const long long f = ws - 1 - (lane + 64 * i);
      if (f > 0 || (f == 0 && !whole0)) {
        s0 += w * (double)stats[f * 4 + 0];
        s1 += w * (double)stats[f * 4 + 1];
      } else if (f == 0) {
        float a = stats[0], b = stats[1];
        if (p.overlap > 0.0) {
          a /= p.overlap;
          b /= p.overlap;
        }
        s0 += 100.0 * w * (double)a;
        s1 += 100.0 * w * (double)b;
      }
      w *= r64;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      s0 += __shfl_xor(s0, o);
      s1 += __shfl_xor(s1, o);
    }
    if (!whole0 && ws <= LEV_SEED) {
      const double left = pow(0.99, (double)ws);
      s0 += left * (double)p.max_lvl0;
      s1 += left * (double)p.min_lvl0;
    }
    lvl = lane == 0 ? (float)s0 : (float)s1;
    first = false;
    levels_walk(stats, ws, begin, begin, p, p.overlap, lvl, first, levels, sx, sy);
  }
  if (lane < 2) chunk_state[c * 4 + lane] = lvl;
  levels_walk(stats, begin, end, begin, p, p.overlap, lvl, first, levels, sx, sy);
  if (lane < 2) chunk_state[c * 4 + 2 + lane] = lvl;
}

// One block (one wavefront): where a chunk's warm-up did not end in its predecessor's final state
// (bit for bit), walk the chunk again from that state.  The comparisons of 64 chunks are taken at
// once, a lane each (one after the other they cost a memory round trip per chunk: 26 us for the 128
// chunks of 131 072 columns, a fifth of the whole level tracking); the chunks that differ -- a few
// per thousand -- are then walked in order.  A re-walk that changes its chunk's final state makes
// the successor's comparison stale: the successor is walked again too (always correct, and rarer still).
__global__ __launch_bounds__(64) void levels_fixup_kernel(const float *__restrict__ stats, long long nframes,
                                                          LevelsParams p, float *__restrict__ levels,
                                                          float *__restrict__ chunk_state, int nchunks) {
  __shared__ float sx[2][64];
  __shared__ float sy[2][64];
  const int lane = threadIdx.x;
  bool carry = false;                                      // the chunk before this batch changed its final state
  for (int c0 = 1; c0 < nchunks; c0 += 64) {
    const int c = c0 + lane;
    bool differs = false;
    if (c < nchunks)
      differs = __float_as_uint(chunk_state[c * 4 + 0]) != __float_as_uint(chunk_state[(c - 1) * 4 + 2]) ||
                __float_as_uint(chunk_state[c * 4 + 1]) != __float_as_uint(chunk_state[(c - 1) * 4 + 3]);
    unsigned long long todo = __ballot(differs);
    if (carry) todo |= 1ull;
    carry = false;
    while (todo) {                                         // wavefront-uniform
      const int b = __builtin_ctzll(todo);
      todo &= todo - 1;
      const int cc = c0 + b;
      if (cc >= nchunks) break;
      const long long begin = (long long)cc * LEV_CHUNK;
      const long long end = begin + LEV_CHUNK < nframes ? begin + LEV_CHUNK : nframes;
      float lvl = lane < 2 ? chunk_state[(cc - 1) * 4 + 2 + lane] : 0.0f;
      const float was = lane < 2 ? chunk_state[cc * 4 + 2 + lane] : 0.0f;
      bool first = false;
      levels_walk(stats, begin, end, begin, p, p.overlap, lvl, first, levels, sx, sy);
      if (lane < 2) chunk_state[cc * 4 + 2 + lane] = lvl;
      const bool moved = __ballot(lane < 2 && __float_as_uint(lvl) != __float_as_uint(was)) != 0;
      if (moved) {
        if (b < 63) todo |= 1ull << (b + 1);
        else carry = true;
      }
    }
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

// One block per column.  A thread maps FOUR consecutive pixels: the 12 RGB bytes and the 4 shorts go
// out as one 12-byte and one 8-byte store (byte-aligned: rows of 3n bytes start anywhere; gfx950 runs
// with unaligned global access enabled and the compiler emits dwordx3/dwordx2 for it), instead of
// twelve byte stores and four short stores; the palette sits in LDS as one dword per colour.
template <typename SRC>
__global__ __launch_bounds__(256) void map_kernel(const SRC *__restrict__ src, int n, int src_pitch, int scale_log,
                                                  double thr255, double one_m_thr,
                                                  const float *__restrict__ levels,
                                                  const unsigned char *__restrict__ colortab,
                                                  const double *__restrict__ log_thr,
                                                  unsigned char *__restrict__ rgb, short *__restrict__ lev) {
  __shared__ unsigned tab[256];      // palette: colour index -> RGB dword
  __shared__ unsigned ctab[256];     // logarithmic scales: dB short l0 + i -> RGB dword (display_map.hpp, DbTable)
  __shared__ int table_ok;
  const int c0 = threadIdx.x;
  auto rgb_of = [&](unsigned v) {
    return (unsigned)colortab[3 * v] | ((unsigned)colortab[3 * v + 1] << 8) | ((unsigned)colortab[3 * v + 2] << 16);
  };
  tab[c0] = rgb_of(c0);
  const size_t fr = blockIdx.x;
  const SRC *row = src + fr * (size_t)src_pitch;
  const float display_max = levels[fr * 4 + 0];
  const float display_min = levels[fr * 4 + 1];
  const RowScale rs = row_scale(display_max, display_min);
  const double inv = 1.0 / one_m_thr;
  const DbTable dt = db_table(rs, thr255);
  if (scale_log) {
    bool above;
    ctab[c0] = rgb_of(colour_index((float)(short)(dt.l0 + c0), rs, thr255, one_m_thr, inv, above));
    if (c0 == 255) table_ok = dt.low_ok && above;
  } else if (c0 == 255) {
    table_ok = 0;
  }
  __syncthreads();
  unsigned char *orow = rgb + fr * (size_t)n * 3;
  short *lrow = lev ? lev + fr * (size_t)n : nullptr;
  const int n4 = n & ~3;
  auto columns = [&](auto by_table) {
    constexpr bool TABLE = decltype(by_table)::value;
    auto pixel = [&](SRC q, short &l) -> unsigned {
      if constexpr (TABLE) {
        float sf;
        l = db_short<SRC>(q, 1, log_thr, sf);
        return ctab[db_table_slot(dt, l)];
      } else {
        unsigned v;
        map_bin<SRC>(q, scale_log, rs, thr255, one_m_thr, inv, log_thr, l, v);
        return tab[v];
      }
    };
    for (int i = 4 * (int)threadIdx.x; i < n4; i += 4 * 256) {         // pixels i..i+3 <- bins n-1-i .. n-4-i
      SRC q[4];
      __builtin_memcpy(q, row + (n - 4 - i), sizeof q);
      short l[4];
      unsigned c[4];
#pragma unroll
      for (int u = 0; u < 4; u++) c[u] = pixel(q[3 - u], l[u]);
      const unsigned w[3] = {c[0] | (c[1] << 24), (c[1] >> 8) | (c[2] << 16), (c[2] >> 16) | (c[3] << 8)};
      __builtin_memcpy(orow + 3 * (size_t)i, w, 12);
      if (lrow) __builtin_memcpy(lrow + i, l, 8);
    }
    for (int i = n4 + (int)threadIdx.x; i < n; i += 256) {              // the last n mod 4 pixels
      short l;
      const unsigned c = pixel(row[n - i - 1], l);
      orow[3 * i] = (unsigned char)c;
      orow[3 * i + 1] = (unsigned char)(c >> 8);
      orow[3 * i + 2] = (unsigned char)(c >> 16);
      if (lrow) lrow[i] = l;
    }
  };
  if (table_ok) columns(std::true_type{});
  else columns(std::false_type{});
}

}  // namespace glfer

using namespace glfer;

extern "C" size_t glfer_levels_scratch_floats(size_t nframes) { return 4 * ((nframes + LEV_CHUNK - 1) / LEV_CHUNK); }

// chunk_state: device scratch of glfer_levels_scratch_floats(nframes) floats
extern "C" hipError_t glfer_launch_levels(const float *stats, size_t nframes, int scale_log, int autoscale,
                                          int first_buffer, float overlap, float max_lvl0, float min_lvl0,
                                          float *levels, float *chunk_state, hipStream_t st) {
  if (nframes == 0) return hipSuccess;
  LevelsParams p{scale_log, autoscale, first_buffer, overlap, max_lvl0, min_lvl0};
  const int nchunks = (int)((nframes + LEV_CHUNK - 1) / LEV_CHUNK);
  hipLaunchKernelGGL(levels_kernel, dim3(nchunks), dim3(64), 0, st, stats, (long long)nframes, p, levels, chunk_state);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || nchunks == 1) return e;
  hipLaunchKernelGGL(levels_fixup_kernel, dim3(1), dim3(64), 0, st, stats, (long long)nframes, p, levels, chunk_state, nchunks);
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

// psd_pitch: floats from one PSD row to the next (cfg.psd_pitch; the averaged rows are dense)
extern "C" hipError_t glfer_launch_map(const float *psd, const double *avg, size_t nframes, int n, int psd_pitch,
                                       int scale_log, double thr255, double one_m_thr, const float *levels,
                                       const unsigned char *colortab, const double *log_thr, unsigned char *rgb,
                                       short *lev, hipStream_t st) {
  if (nframes == 0) return hipSuccess;
  if (avg)
    hipLaunchKernelGGL(map_kernel<double>, dim3((unsigned)nframes), dim3(256), 0, st, avg, n, n, scale_log,
                       thr255, one_m_thr, levels, colortab, log_thr, rgb, lev);
  else
    hipLaunchKernelGGL(map_kernel<float>, dim3((unsigned)nframes), dim3(256), 0, st, psd, n, psd_pitch, scale_log,
                       thr255, one_m_thr, levels, colortab, log_thr, rgb, lev);
  return hipGetLastError();
}
