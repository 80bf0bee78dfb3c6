// aux_kernels.hip -- the per-column statistics that follow the estimator in the reference:
//   K6  compute_floor            (fft.c:240-294)   -> floor_kernel
//   K5  update_avg_{plain,sumextreme,sumavg} (avg.c:108-298) -> avg_cum_kernel + avg_norm_kernel
// HBM-bound streaming/reduction kernels over PSD rows; no LDS tiling beyond the row itself.
#include <hip/hip_runtime.h>
#include "display_map.hpp"
#include "div_exact.hpp"
#include <stdint.h>
#include <type_traits>

namespace glfer {
hipError_t allow_dynamic_lds(const void *kernel, size_t bytes);   // plan.h / glfer_hip.cpp: once per device, kernel and size class
}

namespace glfer {

// order-preserving map float -> uint32 (total order, handles -0 and negatives)
__device__ __forceinline__ uint32_t fkey(float v) {
  uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// ---------------------------------------------------------------------------
// K6.  One 256-thread block per PSD row.  sig = max bin; peak = first arg max (strict >
// scan from 0.0, fft.c:284-291); floor = (sum of the `m` smallest bins)/0.05/bins where the
// reference sorts all bins (qsort, fft.c:265) and adds sorted[(int)(bins*0.95) ..].  Here the
// m-th smallest is found by a 4x8-bit radix select on the row held in LDS; no sort.
__global__ __launch_bounds__(256) void floor_kernel(const float *__restrict__ psd, int bins, int pitch, int m,
                                                    float *__restrict__ stats) {
  extern __shared__ float row[];                 // bins floats, then 256 uint32 + scratch
  uint32_t *hist = reinterpret_cast<uint32_t *>(row + bins);
  __shared__ float red_v[256];
  __shared__ int red_i[256];
  __shared__ double red_d[256];
  __shared__ uint32_t sel_prefix, sel_need, wtot[4];

  const int tid = threadIdx.x;
  const float *src = psd + (size_t)blockIdx.x * pitch;
  float best = 0.0f;
  int besti = 0;
  for (int i = tid; i < bins; i += 256) {
    const float v = src[i];
    row[i] = v;
    if (v > best) { best = v; besti = i; }       // strided scan keeps the lowest index per thread
  }
  red_v[tid] = best;
  red_i[tid] = besti;
  if (tid == 0) { sel_prefix = 0; sel_need = (uint32_t)m; }
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) {
      const float ov = red_v[tid + w];
      const int oi = red_i[tid + w];
      if (ov > red_v[tid] || (ov == red_v[tid] && oi < red_i[tid])) { red_v[tid] = ov; red_i[tid] = oi; }
    }
    __syncthreads();
  }
  const float peak = red_v[0];
  const int peak_i = (peak > 0.0f) ? red_i[0] : 0;

  // radix select: key of the m-th smallest element
  for (int shift = 24; shift >= 0; shift -= 8) {
    hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = sel_prefix;
    const uint32_t mask_hi = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = tid; i < bins; i += 256) {
      const uint32_t k = fkey(row[i]);
      if ((k & mask_hi) == prefix) atomicAdd(&hist[(k >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    // the digit d whose bucket holds the need-th smallest: inclusive scan of the 256 counts (wave
    // shuffles + 4 wave totals), the one thread with excl < need <= incl owns it
    const uint32_t need = sel_need, h = hist[tid];
    uint32_t incl = h;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
      if ((tid & 63) >= o) incl += up;
    }
    if ((tid & 63) == 63) wtot[tid >> 6] = incl;
    __syncthreads();
    for (int w = 0; w < (tid >> 6); w++) incl += wtot[w];
    if (incl - h < need && need <= incl) {
      sel_prefix = prefix | ((uint32_t)tid << shift);
      sel_need = need - (incl - h);
    }
    __syncthreads();
  }
  const uint32_t kth = sel_prefix;               // key of the m-th smallest
  const uint32_t ties = sel_need;                // how many copies of it belong to the m smallest

  double s = 0.0;
  float kth_val = 0.0f;
  for (int i = tid; i < bins; i += 256) {
    const float v = row[i];
    const uint32_t k = fkey(v);
    if (k < kth) s += (double)v;
    else if (k == kth) kth_val = v;
  }
  red_d[tid] = s;
  red_v[tid] = kth_val;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) {
      red_d[tid] += red_d[tid + w];
      if (red_v[tid + w] != 0.0f) red_v[tid] = red_v[tid + w];
    }
    __syncthreads();
  }
  if (tid == 0) {
    float fl = (float)(red_d[0] + (double)ties * (double)red_v[0]);
    fl = (float)(fl / 0.05);                     // fft.c:274 (float / double)
    fl = fl / (float)bins;                       // fft.c:276
    float *o = stats + (size_t)blockIdx.x * 4;
    o[0] = peak;                                 // *sig_pwr_p = tmp_buf[0] = largest bin (fft.c:279)
    o[1] = fl;
    o[2] = (peak > 0.0f) ? peak : 0.0f;          // *peak_pwr_p
    o[3] = (float)peak_i;
  }
}

// ---------------------------------------------------------------------------
// K6, rows of up to 64*EPL bins: ONE WAVEFRONT per row, the row in registers (lane l holds bins
// l + 64 j), no LDS and no barriers.  The m-th smallest key is found bit by bit: with the prefix P
// decided so far, T = P | bit; if fewer than m keys are below T the answer has that bit set.  A
// step is one compare per key (a ballot) and the population counts added on the scalar unit; bits
// above the first one in which the row's smallest and largest key differ are skipped.  (The histogram form above
// serialises on LDS atomics -- a PSD row's keys share their leading digits, so a whole wavefront
// adds to one bucket -- and needs ~34 workgroup barriers per row.)
__device__ __forceinline__ float fkey_inv(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
// sum over the wavefront, returned in an SGPR: row_shr 1,2,4,8 leave each row's total in its lane 15,
// row_bcast 15 / 31 carry the totals up to lane 63
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {      // lanes shifted in from outside read 0
  auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { return ~wave_max_u32(~v); }

// c += (k0 < T) + (k1 < T) + (k2 < T) + (k3 < T): four compares into four SGPR pairs, then four
// add-with-carry-in -- 2 VALU per key, and no wait states between a compare and its use (the compiler's
// own form goes through VCC: v_cmp, s_nop 1, v_cndmask, v_add)
__device__ __forceinline__ void count_below4(uint32_t &c, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t T) {
  asm("v_cmp_gt_u32_e64 s[20:21], %5, %1\n\t"
      "v_cmp_gt_u32_e64 s[22:23], %5, %2\n\t"
      "v_cmp_gt_u32_e64 s[24:25], %5, %3\n\t"
      "v_cmp_gt_u32_e64 s[26:27], %5, %4\n\t"
      "v_addc_co_u32_e64 %0, s[28:29], 0, %0, s[20:21]\n\t"
      "v_addc_co_u32_e64 %0, s[28:29], 0, %0, s[22:23]\n\t"
      "v_addc_co_u32_e64 %0, s[28:29], 0, %0, s[24:25]\n\t"
      "v_addc_co_u32_e64 %0, s[28:29], 0, %0, s[26:27]"
      : "+v"(c)
      : "v"(k0), "v"(k1), "v"(k2), "v"(k3), "v"(T)
      : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29");
}

// inclusive prefix sum over the wavefront (the same DPP walk, every lane keeps its value)
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return v;
}

// Number of keys below T, over the wavefront.  Up to GLFER_FLOOR_BALLOT_MAX keys per lane: one compare
// per key into an SGPR pair (a ballot), the population counts added on the scalar unit -- no
// per-lane counters, no DPP sum (+4..10 % at 129 and 513 bins).  More keys per lane: per-lane
// counters (compare + add-with-carry) and one DPP sum -- there the scalar chain of the ballot form
// is the longer one (-6 % at 2049 bins, -22 % at 8193).
#ifndef GLFER_FLOOR_BALLOT_MAX
#define GLFER_FLOOR_BALLOT_MAX 9
#endif
template <int NK>
__device__ __forceinline__ uint32_t count_below(const uint32_t (&key)[NK], uint32_t T) {
  uint32_t c = 0, Tv;
  asm("v_mov_b32 %0, %1" : "=v"(Tv) : "s"(T));              // compares against a VGPR: an SGPR operand halves the VALU rate (tools/pkbench3)
  if constexpr (NK <= GLFER_FLOOR_BALLOT_MAX) {
#pragma unroll
    for (int j = 0; j < NK; j++) c += (uint32_t)__builtin_popcountll(__ballot(key[j] < Tv));
    return c;
  } else {
    constexpr int G = (NK / 4) * 4;
#pragma unroll
    for (int j = 0; j < G; j += 4) count_below4(c, key[j], key[j + 1], key[j + 2], key[j + 3], Tv);
#pragma unroll
    for (int j = G; j < NK; j++) c += key[j] < Tv ? 1u : 0u;
    return wave_sum_u32(c);
  }
}

// P = the m-th smallest of the wavefront's `total` real keys (the arrays are padded with 0xFFFFFFFF),
// below = number of keys < P.  kmin / kmax: smallest and largest real key.
template <int NK>
__device__ __forceinline__ void select_mth(const uint32_t (&key)[NK], uint32_t total, uint32_t m, uint32_t kmin, uint32_t kmax,
                                           uint32_t &P, uint32_t &below) {
  const uint32_t diff = kmin ^ kmax;
  P = kmin;
  below = 0;
  if (!diff) return;
  const int hb = 31 - __builtin_clz(diff);       // first bit in which the keys differ
  P = (hb == 31) ? 0u : (kmin >> (hb + 1)) << (hb + 1);
  uint32_t upper = total;                        // number of keys below the end of the bucket [P, P + 2^(b+1)) under search
  for (int b = hb; b >= 0; b--) {
    const uint32_t T = P | (1u << b);
    const uint32_t c = count_below<NK>(key, T);
    if (c < m) { P = T; below = c; } else upper = c;
    if (upper - below == 1) {                    // one key left in the bucket: it is the smallest key >= P
      uint32_t cand = 0xFFFFFFFFu;
#pragma unroll
      for (int j = 0; j < NK; j++) cand = (key[j] >= P && key[j] < cand) ? key[j] : cand;
      P = wave_min_u32(cand);
      return;
    }
  }
}

template <int NK>
__device__ __forceinline__ double sum_below(const uint32_t (&key)[NK], uint32_t P) {
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < NK; j++) s += (double)(key[j] < P ? fkey_inv(key[j]) : 0.0f);
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
  return s;
}

// Rows of 577 bins and more first shrink the problem.  The lane's keys are dealt into G groups
// (j mod G) and the two smallest of every group are tracked (v_med3 + v_min per key): the largest
// of all lanes' and groups' second-smallest keys is a pivot with at least 2*64*G keys at or below
// it (64*G for the groups' smallest) -- the m smallest are all among them when m <= 128 G -- and on
// a typical row a fifth of the keys are.  Those are compacted through a wavefront-private LDS strip
// (a DPP prefix sum gives each lane its offset) into 8 G keys per lane (48 for G = 4), and the bitwise
// search runs on them: 8..48 compares per step instead of 33..129.  Rows on which the pivot does not cut
// enough (many ties) take the search over the whole row.  G = 1 up to 2049 bins (m = 103), 2 for
// 4097 (m = 205), 4 for 8193 (m = 410).
constexpr int floor_groups(int epl) { return epl <= 33 ? 1 : (epl <= 65 ? 2 : 4); }

// EXACT: bins == 64 (EPL-1) + 1, the N/2+1 bins of a power-of-two block -- every group of 64 bins but
// the last is whole and needs no range test.
//
// The load loop is what the kernel's time is made of (the selection after it runs on ~100-400
// compacted keys): per bin one buffer load (range-checked by the descriptor: no exec masking, no
// 64-bit address arithmetic), three integer ops for the key, one v_max_f32 for the peak and
// v_med3 + v_min for the lane's two smallest keys.  The peak's INDEX is found afterwards, by
// ballots over the keys in ascending group order (33 compares at worst instead of 66 selects).
template <int EPL, bool EXACT>
__global__ __launch_bounds__(256) void floor_wave_kernel(const float *__restrict__ psd, long long nframes, int bins, int pitch, int m,
                                                         float *__restrict__ stats) {
  constexpr bool kCompact = EPL >= 17;           // (at 9 keys per lane the detour costs what it saves)
  constexpr int G = floor_groups(EPL), CAP = G == 4 ? 48 : 8 * G;   // (8193 bins: a quarter of the rows pass 32 keys per lane)
  __shared__ uint32_t strip[kCompact ? 4 : 1][kCompact ? 64 * CAP : 1];
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (r >= nframes) return;                      // wavefront-uniform
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(psd + (size_t)r * pitch), 0, bins * 4, 0x00020000);   // reads past the row's end return 0
  uint32_t key[EPL];                             // the row as keys only: fkey_inv() gives the value back
  float best = 0.0f;                             // strict > scan from 0.0 (fft.c:284-291): NaNs and bins <= 0 never win
  uint32_t kmin = 0xFFFFFFFFu, s1[G], s2[G];     // s1 <= s2: the two smallest keys of the lane's group g (keys j = g mod G)
#pragma unroll
  for (int g = 0; g < G; g++) s1[g] = s2[g] = 0xFFFFFFFFu;
  const unsigned voff = (unsigned)lane * 4u;
#pragma unroll
  for (int j = 0; j < EPL; j++) {
    const uint32_t b = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (unsigned)j * 256u, 0);
    best = fmaxf(best, __uint_as_float(b));      // (a read past the end is 0.0: no effect)
    uint32_t k = b ^ ((uint32_t)((int)b >> 31) | 0x80000000u);      // fkey()
    if (!EXACT || j == EPL - 1) k = (lane + 64 * j < bins) ? k : 0xFFFFFFFFu;   // padding sorts last: never among the m <= bins smallest
    key[j] = k;
    if constexpr (kCompact) {
      asm("v_med3_u32 %0, %1, %2, %0" : "+v"(s2[j % G]) : "v"(k), "v"(s1[j % G]));   // second smallest of {k, s1 <= s2} (no builtin; not matched from min/max)
      s1[j % G] = k < s1[j % G] ? k : s1[j % G];
    } else {
      kmin = k < kmin ? k : kmin;
    }
  }
  uint32_t g1 = 0u, g2 = 0u;                     // the largest of the lane's groups' smallest / second smallest
  if constexpr (kCompact) {
#pragma unroll
    for (int g = 0; g < G; g++) {
      kmin = s1[g] < kmin ? s1[g] : kmin;
      g1 = s1[g] > g1 ? s1[g] : g1;
      g2 = s2[g] > g2 ? s2[g] : g2;
    }
  }
  // largest bin and its first index: bins that can win are > 0, so the float order is the order of
  // the bit patterns, and the first index is in the first group of 64 that holds the peak's key
  const uint32_t peak_bits = wave_max_u32(__float_as_uint(best));
  const float peak = __uint_as_float(peak_bits);
  int peak_i = 0;
  if (peak > 0.0f) {
    const uint32_t pk = peak_bits | 0x80000000u;
    bool found = false;
#pragma unroll
    for (int j = 0; j < EPL; j++) {
      if (!found) {
        const unsigned long long hit = __ballot(key[j] == pk);
        if (hit) {
          peak_i = 64 * j + __builtin_ctzll(hit);
          found = true;
        }
      }
    }
  }

  // key of the m-th smallest bin, the number of keys below it, the sum of those
  kmin = wave_min_u32(kmin);
  uint32_t P = 0, below = 0;
  double s = 0.0;
  bool done = false;
#ifdef GLFER_FLOOR_ABL                              /* timing ablation: loads and the peak only */
  P = kmin; s = (double)kmax; done = true;
#endif
  if constexpr (kCompact) {
    // 64 G (128 G) keys are at or below the largest smallest (second smallest) key of a group
    const uint32_t pivot = m <= 64 * G ? wave_max_u32(g1) : (m <= 128 * G ? wave_max_u32(g2) : 0xFFFFFFFFu);
    if (pivot != 0xFFFFFFFFu) {
      uint32_t n = 0;
#pragma unroll
      for (int j = 0; j < EPL; j++) n += key[j] <= pivot ? 1u : 0u;
      const uint32_t incl = wave_scan_u32(n);
      const uint32_t C = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      if (C <= 64u * CAP) {                // wavefront-uniform
        uint32_t *mine = strip[threadIdx.x >> 6];
        uint32_t off = incl - n;
#pragma unroll
        for (int j = 0; j < EPL; j++) {
          const bool is = key[j] <= pivot;
          if (is) mine[off] = key[j];
          off += is ? 1u : 0u;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t cand[CAP];
#pragma unroll
        for (int q = 0; q < CAP; q++) cand[q] = (uint32_t)(lane + 64 * q) < C ? mine[lane + 64 * q] : 0xFFFFFFFFu;
        select_mth<CAP>(cand, C, (uint32_t)m, kmin, pivot, P, below);
        s = sum_below<CAP>(cand, P);
        done = true;
      }
    }
  }
  if (!done) {
    uint32_t kmax = 0u;                          // largest real key (the padding is 0xFFFFFFFF)
#pragma unroll
    for (int j = 0; j < EPL; j++) kmax = (key[j] != 0xFFFFFFFFu && key[j] > kmax) ? key[j] : kmax;
    kmax = wave_max_u32(kmax);
    if (kmax < kmin) kmax = kmin;                // a row of NaN patterns only
    select_mth<EPL>(key, (uint32_t)bins, (uint32_t)m, kmin, kmax, P, below);
    s = sum_below<EPL>(key, P);
  }
  const uint32_t ties = (uint32_t)m - below;     // copies of the m-th smallest that belong to the m smallest
  if (lane == 0) {
    float fl = (float)(s + (double)ties * (double)fkey_inv(P));
    fl = (float)(fl / 0.05);                     // fft.c:274 (float / double)
    fl = fl / (float)bins;                       // fft.c:276
    float *o = stats + (size_t)r * 4;
    o[0] = peak;
    o[1] = fl;
    o[2] = (peak > 0.0f) ? peak : 0.0f;
    o[3] = (float)peak_i;
  }
}

// ---------------------------------------------------------------------------
// K5a.  The sliding sum of avg.c:116-127 per bin, as the same double recurrence
//   f <  depth : cum += psd[f]                      (shift register still filling)
//   f >= depth : cum += psd[f] - psd[f-depth]       (avgarray[index][0] is the row f-depth)
// One thread per in-band bin, frames in order; rows are read coalesced across bins and eight rows
// are kept in flight to cover latency.  cum is written into the avg rows.
// The recurrence is sequential in f, and one thread per bin is only ~2000 threads, so the frames
// are cut into chunks of AVG_CHUNK that run in parallel: a chunk starts from the sum of the
// (up to depth) rows before it, added in frame order.  That is the value the recurrence holds at
// that frame whenever its additions are exact -- float terms in a double accumulator: always,
// unless a bin's values within depth frames span more than ~2^26 (78 dB); beyond that the two differ
// in the last bits of a double (the recurrence carries its own rounding history, the restart does
// not).  131 072 rows: 2.1 M rows/s as one chain per bin, HBM-bound in chunks.
constexpr int AVG_CHUNK = 128;
__global__ __launch_bounds__(256) void avg_cum_kernel(const float *__restrict__ psd, long long nframes,
                                                      int bins, int n_out, int depth, int minbin,
                                                      int maxbin, double *__restrict__ avg) {
  const int b = minbin + blockIdx.x * 256 + threadIdx.x;
  if (b >= maxbin) return;
  const long long f0 = (long long)blockIdx.y * AVG_CHUNK;
  const long long f1 = f0 + AVG_CHUNK < nframes ? f0 + AVG_CHUNK : nframes;
  double cum = 0.0;
  for (long long g = f0 > depth ? f0 - depth : 0; g < f0; g++) cum += (double)psd[(size_t)g * bins + b];
  long long f = f0;
  for (; f + 8 <= f1; f += 8) {
    float v[8], old[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      v[u] = psd[(size_t)(f + u) * bins + b];
      old[u] = (f + u >= depth) ? psd[(size_t)(f + u - depth) * bins + b] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (f + u < depth) cum += (double)v[u];
      else cum += (double)v[u] - (double)old[u];
      avg[(size_t)(f + u) * n_out + b] = cum;
    }
  }
  for (; f < f1; f++) {
    const float v = psd[(size_t)f * bins + b];
    if (f < depth) cum += (double)v;
    else cum += (double)v - (double)psd[(size_t)(f - depth) * bins + b];
    avg[(size_t)f * n_out + b] = cum;
  }
}

// K5b.  Per-frame reductions over the band and the output normalisation of the three
// modes (avg.c:129-156, 185-215, 248-294).  One block per frame, in place on the avg row.
// mode: 1 sumavg, 2 plain, 3 sumextreme (glfer.h:56-58).
__global__ __launch_bounds__(256) void avg_norm_kernel(const float *__restrict__ psd, int bins, int n_out,
                                                       int depth, int minbin, int maxbin, int mode,
                                                       int max0, double *__restrict__ avg,
                                                       double *__restrict__ ret) {
  __shared__ double r_sum[256], r_max[256], r_min[256], r_var[256];
  __shared__ int r_idx[256], r_cnt[256];
  const long long f = blockIdx.x;
  const int tid = threadIdx.x;
  double *row = avg + (size_t)f * n_out;
  const int eff = (f + 1 < depth) ? (int)(f + 1) : depth;   // effdepth after this frame
  const double init = (double)psd[(size_t)f * bins + minbin];

  double s = 0.0, mx = -1.0e300, mn = 1.0e300;
  int mi = 0x7fffffff;
  for (int b = minbin + tid; b < maxbin; b += 256) {
    const double c = row[b];
    s += c;
    if (c > mx) { mx = c; mi = b; }
    if (c < mn) mn = c;
  }
  r_sum[tid] = s; r_max[tid] = mx; r_min[tid] = mn; r_idx[tid] = mi;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) {
      r_sum[tid] += r_sum[tid + w];
      if (r_max[tid + w] > r_max[tid] || (r_max[tid + w] == r_max[tid] && r_idx[tid + w] < r_idx[tid])) {
        r_max[tid] = r_max[tid + w];
        r_idx[tid] = r_idx[tid + w];
      }
      if (r_min[tid + w] < r_min[tid]) r_min[tid] = r_min[tid + w];
    }
    __syncthreads();
  }
  // running max starts at psd[minbin] (avg.c:111,163,224) and only a strictly larger sum moves it
  double top = init;
  int peak = -1;
  if (r_max[0] > init) { top = r_max[0]; peak = r_idx[0]; }
  const double low = (r_min[0] < 1.0) ? r_min[0] : 1.0;        // avg.c:165,192-193
  const double span = (double)(maxbin - minbin - 1);
  double spec;
  if (mode == 2) spec = (r_sum[0] - top) / (span * (double)(eff + 1));   // avg.c:147
  else spec = (r_sum[0] - top) / span;                                    // avg.c:199,260
  __syncthreads();

  double var = 0.0;
  int cnt = 0;
  for (int b = tid; b < n_out; b += 256) {
    double out;
    if (b < minbin || b >= maxbin) {
      out = 1e-15;
    } else {
      const double c = row[b];
      if (mode == 2) {
        out = c / (double)(eff + 1);                                      // avg.c:155
      } else if (mode == 3) {
        out = max0 ? (c - low) / (top - low) : c / spec;                  // avg.c:209-212
      } else {
        if (c - spec > 0) {                                               // avg.c:272-284
          out = max0 ? (c - spec) / (top - spec) : c / spec;
          if (b != peak) { var += (c / spec) * (c / spec); cnt++; }
        } else {
          out = 1e-15;
        }
      }
    }
    row[b] = out;
  }
  r_var[tid] = var; r_cnt[tid] = cnt;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) { r_var[tid] += r_var[tid + w]; r_cnt[tid] += r_cnt[tid + w]; }
    __syncthreads();
  }
  if (tid == 0) {
    double *o = ret + (size_t)f * 4;
    o[0] = (mode == 2) ? spec : top / spec;                               // avg.c:158,218,297
    o[1] = (double)peak;
    o[2] = (mode == 1) ? r_var[0] / (double)r_cnt[0] : 0.0;               // avg.c:294
    o[3] = (double)eff;
  }
}

// K5, fused: the sliding sums AND the per-frame normalisation in one pass.  avg_cum_kernel writes
// the sums (8 B per bin) for avg_norm_kernel to read back, reduce and overwrite: 28 B of HBM traffic
// per bin where the output alone is 8.  Here a block owns a chunk of AVG_CHUNK frames and the WHOLE
// band: thread t keeps the double sums of bins minbin + t + 256 j in registers (same recurrence,
// same restart from a direct sum as avg_cum_kernel: bit-identical sums), walks the frames in
// order, and for each frame reduces its sums over the block (wavefront shuffles + four partials,
// one barrier; the partials alternate between two LDS slots, so no second barrier), normalises and
// stores the row: 4 B read (+4 B of the row leaving the window, an L2 hit for small depths) and
// 8 B written per bin.  The next frame's samples are requested before the current one is reduced.
#ifndef GLFER_AVG_NT
#define GLFER_AVG_NT 1    /* +1..3 % */
#endif
#ifndef GLFER_AVG_AHEAD
#define GLFER_AVG_AHEAD 1    /* frames requested ahead of the one being reduced: 1 or 2 (measured: no difference, see the kernel) */
#endif
#if GLFER_AVG_ABL & 1
#define GLFER_AVG_STORE(dst, val) do { const double v_ = (val); if (v_ == 1.2345e-300) (dst) = v_; } while (0)
#else
#if GLFER_AVG_NT
#define GLFER_AVG_STORE(dst, val) __builtin_nontemporal_store((double)(val), &(dst))   /* rows are written once and not read by this kernel */
#else
#define GLFER_AVG_STORE(dst, val) (dst) = (val)
#endif
#endif
// (the double-precision wavefront reductions by DPP -- dpp_f64, lane63_f64, wave_sum_max_min -- live in div_exact.hpp too)
// (Divisor -- a / d for many a and one d, correctly rounded -- lives in div_exact.hpp: the periodogram kernel's own average uses it too)

// RING: the last `depth` rows of the block's bins are kept in LDS (hist[f mod depth][j][tid], every
// thread its own words: no synchronisation), so the row that leaves the sliding sum is not read from
// memory a second time -- measured without it: 16 161 B read per 8 196-B row (rocprofv3 FETCH_SIZE),
// the second read comes from HBM, not L2 (1024 blocks x depth rows do not stay there beside the
// 16 KB/row write stream).  The launcher uses it while depth x BPT KB fit the 64 KB dynamic limit.
// MAP: the averaged row is not stored but mapped at once (main_window_draw's loop over avgdata.avg,
// g_main.c:1186-1236) -- what leaves the kernel is the column's RGB bytes and levbuf shorts, 3 + 2 B
// per bin, instead of 8 B per bin written here and read again by map_kernel.  The levels of every
// column are known beforehand (they come from compute_floor of the PSD rows, not from the average:
// g_main.c:1109-1139).  A thread's bins are 256 apart and a pixel is 3 bytes, so the (colour index,
// dB short) words of a column go through LDS (two buffers: frame f's words are written after frame f's
// reduction barrier and stored as 12-byte / 8-byte pieces after frame f+1's -- no barrier of their
// own).  Frames [fbeg, nframes) are walked; the sums reach back before fbeg (a tile of a longer batch).
#ifndef GLFER_AVGMAP_STAGE
#define GLFER_AVGMAP_STAGE 0     /* 1: the averaged bins of a frame wait in LDS for a rolled mapping loop (159 VGPRs instead of 191, three wavefronts per SIMD without the ring) -- measured SLOWER, 70-73 against 79 M rows/s: the kernel is bound by the instructions it issues, not by what hides them */
#endif
#ifndef GLFER_AVGMAP_RING_KB
#define GLFER_AVGMAP_RING_KB 76  /* the ring of the last `depth` rows is kept in LDS while everything fits this many KB */
#endif
#ifndef GLFER_AVGMAP_ABL
#define GLFER_AVGMAP_ABL 0    /* timing ablations (results wrong): 1 columns not stored, 2 bins not mapped, 4 no colour table, 8 levels not loaded */
#endif
struct AvgMapArgs {
  const float *levels;             // [nframes - fbeg][4], display_max / display_min first
  const unsigned char *colortab;   // 768 bytes
  const double *log_thr;
  unsigned char *rgb;              // [nframes - fbeg][bins][3]
  short *lev;                      // [nframes - fbeg][bins] or null
  int scale_log;
  double thr255, one_m_thr;
  long long fbeg;
};

template <int BPT, bool RING, bool MAP, int NT = 256>
__global__ __launch_bounds__(NT) void avg_fused_kernel(const float *__restrict__ psd, long long nframes, int chunk, int bins,
                                                        int n_out, int depth, int minbin, int maxbin, int mode, int max0,
                                                        double *__restrict__ avg, double *__restrict__ ret, AvgMapArgs ma) {
  constexpr int NW = NT / 64;                     // wavefronts of the block
  constexpr bool RET = !MAP;                      // update_avg's return values (band mean, peak bin, variance): not for columns that are only mapped
  __shared__ double p_sum[2][NW], p_max[2][NW], p_min[2][NW], p_var[2][NW];
  __shared__ int p_idx[2][NW], p_cnt[2][NW];
  __shared__ unsigned tab[MAP ? 256 : 1];
  __shared__ unsigned char vtab[2][MAP ? 256 : 1];   // logarithmic scales: dB short l0 + i -> colour index, per frame parity (display_map.hpp)
  __shared__ int vtab_ok[2];
  extern __shared__ float dyn_lds[];             // MAP: [2][n_out] words of (colour index << 16 | dB short); RING: [depth][BPT][NT]
  unsigned *const pix = reinterpret_cast<unsigned *>(dyn_lds);
  // MAP: a frame's averaged bins wait in LDS (every thread its own BPT slots: no synchronisation) for a
  // ROLLED mapping loop -- unrolled over a register array the mapping of BPT bins at once takes the
  // kernel to 191 VGPRs, two wavefronts per SIMD, for a chain of frames that has nothing else to hide behind
  const size_t pix_words = MAP ? 2 * (((size_t)n_out + 1) & ~(size_t)1) : 0;       // an even count: the doubles behind it stay aligned
  double *const stage = reinterpret_cast<double *>(dyn_lds + pix_words);
  float *const hist = dyn_lds + pix_words + (MAP && GLFER_AVGMAP_STAGE ? 2 * (size_t)BPT * NT : 0);
  const int tid = threadIdx.x, wave = tid >> 6;
  const long long f0 = (MAP ? ma.fbeg : 0) + (long long)blockIdx.x * chunk;
  const long long f1 = f0 + chunk < nframes ? f0 + chunk : nframes;
  if constexpr (MAP)
    if (tid < 256) tab[tid] = (unsigned)ma.colortab[3 * tid] | ((unsigned)ma.colortab[3 * tid + 1] << 8) | ((unsigned)ma.colortab[3 * tid + 2] << 16);
  const double inv_one_m_thr = MAP ? 1.0 / ma.one_m_thr : 0.0;
  // column fr's words -> its RGB bytes and shorts: pixel i is bin n_out-1-i; four pixels per thread and piece
  auto emit = [&](long long fr) {
#if GLFER_AVGMAP_ABL & 1
    if (fr != -12345) return;
#endif
    const unsigned *pw = pix + (size_t)(fr & 1) * (pix_words / 2);
    const size_t col = (size_t)(fr - ma.fbeg);
    unsigned char *orow = ma.rgb + col * (size_t)n_out * 3;
    short *lrow = ma.lev ? ma.lev + col * (size_t)n_out : nullptr;
    const int n4 = n_out & ~3;
    for (int i = 4 * tid; i < n4; i += 4 * NT) {
      unsigned c[4];
      short l[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const unsigned w = pw[n_out - 1 - i - u];
        c[u] = tab[w >> 16];
        l[u] = (short)(w & 0xffffu);
      }
      const unsigned o[3] = {c[0] | (c[1] << 24), (c[1] >> 8) | (c[2] << 16), (c[2] >> 16) | (c[3] << 8)};
      __builtin_memcpy(orow + 3 * (size_t)i, o, 12);
      if (lrow) __builtin_memcpy(lrow + i, l, 8);
    }
    for (int i = n4 + tid; i < n_out; i += NT) {
      const unsigned w = pw[n_out - 1 - i], c = tab[w >> 16];
      orow[3 * i] = (unsigned char)c;
      orow[3 * i + 1] = (unsigned char)(c >> 8);
      orow[3 * i + 2] = (unsigned char)(c >> 16);
      if (lrow) lrow[i] = (short)(w & 0xffffu);
    }
  };
  const int b0 = minbin + tid;
  double cum[BPT];
#pragma unroll
  for (int j = 0; j < BPT; j++) cum[j] = 0.0;
  for (long long g = f0 > depth ? f0 - depth : 0; g < f0; g++) {
    const float *r = psd + (size_t)g * bins;
    float *h = hist + ((size_t)(g % depth) * BPT) * NT + tid;
#pragma unroll
    for (int j = 0; j < BPT; j++)
      if (b0 + NT * j < maxbin) {
        const float x = r[b0 + NT * j];
        cum[j] += (double)x;
        if constexpr (RING) h[NT * j] = x;
      }
  }
  // The next frame's samples are requested before the current one is reduced (AHEAD = 1).  Two frames
  // ahead (AHEAD = 2: two buffers taken in turn, a buffer fetched into again as soon as its frame has
  // gone into the sums) was built to see whether a block that walks its frames in order is short of
  // requests in flight: no -- update_avg 172-181 M rows/s and the average-and-map waterfall 74-75 either way.
  constexpr int AHEAD = GLFER_AVG_AHEAD;
  float v[AHEAD][BPT], old[AHEAD][BPT];
  auto fetch = [&](long long f, auto buf) {
    constexpr int CUR = decltype(buf)::value;
    const float *r = psd + (size_t)f * bins;
    const float *ro = psd + (size_t)(f >= depth ? f - depth : 0) * bins;
#pragma unroll
    for (int j = 0; j < BPT; j++) {
      const bool in = b0 + NT * j < maxbin;
      v[CUR][j] = in ? r[b0 + NT * j] : 0.0f;
      if constexpr (!RING) old[CUR][j] = (in && f >= depth) ? ro[b0 + NT * j] : 0.0f;
    }
  };
  fetch(f0, std::integral_constant<int, 0>{});
  if constexpr (AHEAD == 2) {
    if (f0 + 1 < f1) fetch(f0 + 1, std::integral_constant<int, 1>{});
  }
  const double span = (double)(maxbin - minbin - 1);
  const Divisor by_full_depth((double)(depth + 1));          // the plain average's divisor once the window is full (avg.c:138-139,155)
  const Divisor by_span(span), by_span_full_depth(span * (double)(depth + 1));   // the band mean's divisors: the same every frame
  auto frame = [&](const long long f, auto cur) {
    constexpr int CUR = decltype(cur)::value;
    const int par = (int)(f & 1);
    const int eff = (f + 1 < depth) ? (int)(f + 1) : depth;   // effdepth after this frame
    RowScale rs{};
    DbTable dt{};
    if constexpr (MAP) {
      // the column's levels and (logarithmic scales) its table of colour indices, ready before the
      // frame's barrier; the buffer was last read two frames ago, before the barrier in between
      const float *lv = ma.levels + (size_t)(f - ma.fbeg) * 4;
#if GLFER_AVGMAP_ABL & 8
      rs = row_scale(-20.0f + (float)f1 * 1e-9f, -80.0f);
#else
      rs = row_scale(lv[0], lv[1]);
#endif
      dt = db_table(rs, ma.thr255);
#if GLFER_AVGMAP_ABL & 4
      if (tid == 255) vtab_ok[par] = 1;
      if (tid > 256) {
#else
      if (tid < 256) {
#endif
        bool above = false;
        if (ma.scale_log) vtab[par][tid] = (unsigned char)colour_index((float)(short)(dt.l0 + tid), rs, ma.thr255, ma.one_m_thr, inv_one_m_thr, above);
        if (tid == 255) vtab_ok[par] = ma.scale_log && dt.low_ok && above;
      }
    }
    const double init = (double)psd[(size_t)f * bins + minbin];
    double s = 0.0, mx = -1.0e300, mn = 1.0e300;
    int mi = 0x7fffffff;
    // Windows of up to four rows (the reference's default depth, glfer.c:295-296) with the ring: the sum is taken DIRECTLY,
    // ((p[f-3] + p[f-2]) + p[f-1]) + p[f] in double, oldest row first -- the additions, in the order, that the periodogram
    // kernel's own average makes (spectro16h.hip AVG: it keeps those rows in registers), so that the two agree double for double
    // even where a bin spans more than 2^26 within the window and the additions round (round 5; the running sum below and this
    // one differ there in the last bits, and neither is the reference's own rounding history).
    const bool direct = RING && depth <= 4;                      // (the same in every thread)
    if constexpr (RING) {                                      // row f - depth leaves the sum, row f takes its slot
      float *h = hist + ((size_t)(f % depth) * BPT) * NT + tid;
      if (direct) {
        const float *h1 = hist + ((size_t)((f + depth - 1) % depth) * BPT) * NT + tid;   // row f - 1
        const float *h2 = hist + ((size_t)((f + 2 * depth - 2) % depth) * BPT) * NT + tid;   // row f - 2
        const float *h3 = hist + ((size_t)((f + 3 * depth - 3) % depth) * BPT) * NT + tid;   // row f - 3
        const bool k1 = depth >= 2 && f >= 1, k2 = depth >= 3 && f >= 2, k3 = depth >= 4 && f >= 3;
#pragma unroll
        for (int j = 0; j < BPT; j++)
          if (b0 + NT * j < maxbin) {
            double c = (k3 ? (double)h3[NT * j] : 0.0) + (k2 ? (double)h2[NT * j] : 0.0);   // (adding +0.0 to a sum of non-negative bins is exact)
            c += k1 ? (double)h1[NT * j] : 0.0;
            cum[j] = c + (double)v[CUR][j];
            h[NT * j] = v[CUR][j];
          }
      } else {
#pragma unroll
        for (int j = 0; j < BPT; j++)
          if (b0 + NT * j < maxbin) {
            old[CUR][j] = h[NT * j];
            h[NT * j] = v[CUR][j];
          }
      }
    }
#pragma unroll
    for (int j = 0; j < BPT; j++) {
      if (b0 + NT * j < maxbin) {
        if (direct) {
        } else if (f < depth) cum[j] += (double)v[CUR][j];
        else cum[j] += (double)v[CUR][j] - (double)old[CUR][j];
        const double c = cum[j];
        s += c;
        if constexpr (RET) {
          if (c > mx) { mx = c; mi = b0 + NT * j; }
        } else {
          mx = c > mx ? c : mx;
        }
        if (c < mn) mn = c;
      }
    }
    if (f + AHEAD < f1) fetch(f + AHEAD, cur);                 // into the buffer just read: in flight under this frame and the next
    // the band statistics: the normalising modes divide by them, the return values are made of them;
    // the plain average of a column that is only mapped (no return values) needs neither
    const bool band_stats = RET || mode != 2;                                  // the same in every thread
    if (band_stats) {
#if !(GLFER_AVG_ABL & 2)
      wave_sum_max_min<RET>(s, mx, mi, mn);
#endif
      if ((tid & 63) == 0) { p_sum[par][wave] = s; p_max[par][wave] = mx; p_min[par][wave] = mn; p_idx[par][wave] = mi; }
    }
#if !(GLFER_AVG_ABL & 4)
    __syncthreads();
#endif
    double r_sum = 0.0, r_max = -1.0e300, r_min = 1.0e300;
    int r_idx = 0x7fffffff;
    if (band_stats) {
      r_sum = p_sum[par][0], r_max = p_max[par][0], r_min = p_min[par][0];
      r_idx = p_idx[par][0];
#pragma unroll
      for (int w = 1; w < NW; w++) {
        r_sum += p_sum[par][w];
        if (p_max[par][w] > r_max || (p_max[par][w] == r_max && p_idx[par][w] < r_idx)) { r_max = p_max[par][w]; r_idx = p_idx[par][w]; }
        if (p_min[par][w] < r_min) r_min = p_min[par][w];
      }
    }
    // running max starts at psd[minbin] (avg.c:111,163,224) and only a strictly larger sum moves it
    double top = init;
    int peak = -1;
    if (r_max > init) { top = r_max; peak = r_idx; }
    const double low = (r_min < 1.0) ? r_min : 1.0;            // avg.c:165,192-193
    double spec;
    if (mode == 2) spec = eff == depth ? by_span_full_depth(r_sum - top) : (r_sum - top) / (span * (double)(eff + 1));   // avg.c:147
    else spec = by_span(r_sum - top);                                    // avg.c:199,260

    double *row = MAP ? nullptr : avg + (size_t)f * n_out;
    unsigned *prow = pix + (size_t)par * (pix_words / 2);
    if constexpr (MAP) {
      if (f > f0) emit(f - 1);                                 // every thread is past this frame's barrier: column f-1's words are complete
    }
    // one averaged bin: to the row, or kept for the mapping below (one copy of it for all the modes)
    double outv[(MAP && !GLFER_AVGMAP_STAGE) ? BPT : 1];
    auto put = [&](int j, int b, double val) {
      if constexpr (MAP) {
        if constexpr (GLFER_AVGMAP_STAGE) stage[j * NT + tid] = val;
        else outv[j] = val;
      } else {
        GLFER_AVG_STORE(row[b], val);
      }
    };
    double var = 0.0;
    int cnt = 0;
    // the frame's divisors (the same in every lane); one loop per mode, so that a bin's code is its
    // mode's alone (left inside the loop, the mode tests came out as ~130 branches per frame)
    if (mode == 2) {
      const Divisor by_depth = eff == depth ? by_full_depth : Divisor((double)(eff + 1));   // (a true division only while the window fills)
#pragma unroll
      for (int j = 0; j < BPT; j++) {
        const int b = b0 + NT * j;
        if (b < maxbin) put(j, b, by_depth(cum[j]));                        // avg.c:155
      }
    } else if (mode == 3) {
      const Divisor by_spec(spec), by_range(top - low);
#pragma unroll
      for (int j = 0; j < BPT; j++) {
        const int b = b0 + NT * j;
        if (b < maxbin) put(j, b, max0 ? by_range(cum[j] - low) : by_spec(cum[j]));   // avg.c:209-212
      }
    } else {
      const Divisor by_spec(spec), by_range(top - spec);
#pragma unroll
      for (int j = 0; j < BPT; j++) {
        const int b = b0 + NT * j;
        if (b < maxbin) {
          const double c = cum[j];
          double out = 1e-15;
          if (c - spec > 0) {                                             // avg.c:272-284
            const double q = by_spec(c);
            out = max0 ? by_range(c - spec) : q;
            if constexpr (RET) {
              if (b != peak) { var += q * q; cnt++; }
            }
          }
          put(j, b, out);
        }
      }
    }
    // the columns outside the band (avg.c:150-153): usually a few dozen
    if constexpr (MAP) {
      auto map_column = [&](auto table) {
        constexpr bool TABLE = decltype(table)::value;
        auto word = [&](double val) {
          short l;
          unsigned vv;
#if GLFER_AVGMAP_ABL & 2
          return (unsigned)(long long)val;
#endif
          if constexpr (TABLE) {
            float sf;
            l = db_short<double>(val, 1, ma.log_thr, sf);
            vv = vtab[par][db_table_slot(dt, l)];
          } else {
            map_bin<double>(val, ma.scale_log, rs, ma.thr255, ma.one_m_thr, inv_one_m_thr, ma.log_thr, l, vv);
          }
          return (vv << 16) | (unsigned)(unsigned short)l;
        };
        if constexpr (GLFER_AVGMAP_STAGE) {
#pragma unroll 1
          for (int j = 0; j < BPT; j++) {
            const int b = b0 + NT * j;
            if (b < maxbin) prow[b] = word(stage[j * NT + tid]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < BPT; j++) {
            const int b = b0 + NT * j;
            if (b < maxbin) prow[b] = word(outv[j]);
          }
        }
        if (minbin > 0 || maxbin < n_out) {
          const unsigned w = word(1e-15);
          for (int b = tid; b < minbin; b += NT) prow[b] = w;
          for (int b = maxbin + tid; b < n_out; b += NT) prow[b] = w;
        }
      };
      if (vtab_ok[par]) map_column(std::true_type{});
      else map_column(std::false_type{});
    } else {
      for (int b = tid; b < minbin; b += NT) row[b] = 1e-15;
      for (int b = maxbin + tid; b < n_out; b += NT) row[b] = 1e-15;
    }
    if (RET && mode == 1) {
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        var += __shfl_xor(var, o);
        cnt += __shfl_xor(cnt, o);
      }
      if ((tid & 63) == 0) { p_var[par][wave] = var; p_cnt[par][wave] = cnt; }
      __syncthreads();
      var = (p_var[par][0] + p_var[par][1]) + (p_var[par][2] + p_var[par][3]);
      cnt = (p_cnt[par][0] + p_cnt[par][1]) + (p_cnt[par][2] + p_cnt[par][3]);
      if constexpr (NW == 8) {
        var += (p_var[par][4] + p_var[par][5]) + (p_var[par][6] + p_var[par][7]);
        cnt += (p_cnt[par][4] + p_cnt[par][5]) + (p_cnt[par][6] + p_cnt[par][7]);
      }
    }
    if (RET && tid == 0) {
      double *o = ret + (size_t)f * 4;
      o[0] = (mode == 2) ? spec : top / spec;                             // avg.c:158,218,297
      o[1] = (double)peak;
      o[2] = (mode == 1) ? var / (double)cnt : 0.0;                       // avg.c:294
      o[3] = (double)eff;
    }
  };
  for (long long f = f0; f < f1; f += AHEAD) {
    frame(f, std::integral_constant<int, 0>{});
    if constexpr (AHEAD == 2) {
      if (f + 1 < f1) frame(f + 1, std::integral_constant<int, 1>{});
    }
  }
  if constexpr (MAP) {
    if (f1 > f0) {
      __syncthreads();
      emit(f1 - 1);
    }
  }
}

}  // namespace glfer

using namespace glfer;

// pitch: floats from one row to the next (>= bins; cfg.psd_pitch)
extern "C" hipError_t glfer_launch_floor(const float *psd, size_t nframes, int bins, int pitch, int m, float *stats,
                                         hipStream_t st) {
  if (nframes == 0) return hipSuccess;
  if (m < 1 || m > bins || pitch < bins) return hipErrorInvalidValue;
  const unsigned wgrid = (unsigned)((nframes + 3) / 4);          // one wavefront per row, four rows per block
  const long long nf = (long long)nframes;
#define GLFER_FLOOR_WAVE(E)                                                                                        \
  do {                                                                                                             \
    if (bins == 64 * (E - 1) + 1) hipLaunchKernelGGL((floor_wave_kernel<E, true>), dim3(wgrid), dim3(256), 0, st, psd, nf, bins, pitch, m, stats); \
    else hipLaunchKernelGGL((floor_wave_kernel<E, false>), dim3(wgrid), dim3(256), 0, st, psd, nf, bins, pitch, m, stats); \
  } while (0)
  if (bins <= 64 * 3) GLFER_FLOOR_WAVE(3);
  else if (bins <= 64 * 5) GLFER_FLOOR_WAVE(5);
  else if (bins <= 64 * 9) GLFER_FLOOR_WAVE(9);
  else if (bins <= 64 * 17) GLFER_FLOOR_WAVE(17);
  else if (bins <= 64 * 33) GLFER_FLOOR_WAVE(33);
  else if (bins <= 64 * 65) GLFER_FLOOR_WAVE(65);
  else if (bins <= 64 * 129) GLFER_FLOOR_WAVE(129);
#undef GLFER_FLOOR_WAVE
  else {                                         // longer rows than any block size gives: one workgroup per row, the row in LDS
    const size_t shmem = (size_t)bins * sizeof(float) + 256 * sizeof(uint32_t);
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(floor_kernel), shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(floor_kernel, dim3((unsigned)nframes), dim3(256), shmem, st, psd, bins, pitch, m, stats);
  }
  return hipGetLastError();
}

static int bpt_of(int bpt) { return bpt <= 3 ? bpt : (bpt <= 5 ? 5 : (bpt <= 9 ? 9 : (bpt <= 17 ? 17 : 33))); }   // the BPT the fused kernel is built for

extern "C" hipError_t glfer_launch_avg(int mode, const float *psd, size_t nframes, int bins, int n_out,
                                       int depth, int minbin, int maxbin, int max0, double *avg,
                                       double *ret, hipStream_t st) {
  if (nframes == 0) return hipSuccess;
  const int band = maxbin - minbin;
  if (band < 1 || minbin < 0 || maxbin > bins || maxbin > n_out || depth < 1) return hipErrorInvalidValue;
  const long long nf = (long long)nframes;
  // A fused block walks `chunk` frames in order, so the launch has nframes/chunk blocks: 128-frame
  // chunks for long batches, shorter ones (down to 8) to keep ~1000 blocks in flight for short ones.
  // Every chunk restarts from the `depth` rows before it: once that costs more than the chunk itself
  // the two-pass form (parallel over bins as well) is the better one.
  int chunk = AVG_CHUNK;
  while (chunk > 8 && nf / chunk < 1024) chunk /= 2;
  const int bpt = (band + 255) / 256;
  if (bpt <= 33 && depth <= 2 * chunk) {
    const unsigned blocks = (unsigned)((nf + chunk - 1) / chunk);
    const size_t ring_bytes = (size_t)depth * bpt_of(bpt) * 256 * sizeof(float);
    const bool ring = ring_bytes <= 60 * 1024;                 // beside the kernel's static words, under the 64 KB limit
#define GLFER_AVG_FUSED(B)                                                                                              \
  do {                                                                                                                  \
    if (ring) hipLaunchKernelGGL((avg_fused_kernel<B, true, false>), dim3(blocks), dim3(256), ring_bytes, st, psd, nf, chunk, bins, n_out, depth, minbin, maxbin, mode, max0, avg, ret, AvgMapArgs{}); \
    else hipLaunchKernelGGL((avg_fused_kernel<B, false, false>), dim3(blocks), dim3(256), 0, st, psd, nf, chunk, bins, n_out, depth, minbin, maxbin, mode, max0, avg, ret, AvgMapArgs{}); \
  } while (0)
    if (bpt <= 1) GLFER_AVG_FUSED(1);
    else if (bpt <= 2) GLFER_AVG_FUSED(2);
    else if (bpt <= 3) GLFER_AVG_FUSED(3);
    else if (bpt <= 5) GLFER_AVG_FUSED(5);
    else if (bpt <= 9) GLFER_AVG_FUSED(9);
    else if (bpt <= 17) GLFER_AVG_FUSED(17);
    else GLFER_AVG_FUSED(33);
  } else {
    const unsigned chunks = (unsigned)((nframes + AVG_CHUNK - 1) / AVG_CHUNK);
    hipLaunchKernelGGL(avg_cum_kernel, dim3((unsigned)((band + 255) / 256), chunks), dim3(256), 0, st, psd, nf, bins, n_out,
                       depth, minbin, maxbin, avg);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(avg_norm_kernel, dim3((unsigned)nframes), dim3(256), 0, st, psd, bins, n_out, depth, minbin, maxbin,
                       mode, max0, avg, ret);
  }
#undef GLFER_AVG_FUSED
  return hipGetLastError();
}

// update_avg_* and the column mapping in one kernel (avg_fused_kernel<.., MAP>): frames [fbeg, nframes)
// of the batch `psd` (the sums reach back before fbeg).  levels / rgb / lev are indexed from fbeg.
// Returns hipErrorNotSupported where the fused form does not apply (very wide bands, windows much
// deeper than a chunk, rows whose words do not fit LDS): the caller then runs the two stages.
// 256 threads per block whatever the band: the kernel is bound by the instructions it issues (the
// mapping is ~70 per bin, and ~400 per wavefront and frame do not depend on the bins at all --
// reductions, divisors, the out-of-band word), so blocks of 512 threads on a 2049-bin row, meant to
// halve the chain a frame is, came out SLOWER: 52 M rows/s against 66 for the whole waterfall
// (GLFER_AVGMAP_WIDE=1 builds that form).
#ifndef GLFER_AVGMAP_WIDE
#define GLFER_AVGMAP_WIDE 0
#endif
struct AvgMapShape { int chunk, bpt, nt; bool ring; size_t shmem; bool ok; };
static AvgMapShape avgmap_shape(long long walk, int bins, int depth, int minbin, int maxbin) {
  AvgMapShape a{AVG_CHUNK, 0, 256, false, 0, false};
  while (a.chunk > 8 && walk / a.chunk < 1024) a.chunk /= 2;
  if (GLFER_AVGMAP_WIDE && maxbin - minbin >= 1024) a.nt = 512;
  a.bpt = (maxbin - minbin + a.nt - 1) / a.nt;
  const size_t pix_bytes = 2 * (((size_t)bins + 1) & ~(size_t)1) * sizeof(unsigned) +
                           (GLFER_AVGMAP_STAGE ? (size_t)bpt_of(a.bpt) * a.nt * sizeof(double) : 0);   // + the staged bins
  const size_t ring_bytes = (size_t)depth * bpt_of(a.bpt) * a.nt * sizeof(float);
  a.ring = pix_bytes + ring_bytes <= (size_t)GLFER_AVGMAP_RING_KB * 1024;
  a.shmem = pix_bytes + (a.ring ? ring_bytes : 0);
  a.ok = a.bpt >= 1 && a.bpt <= 33 && depth <= 2 * a.chunk && a.shmem <= 150 * 1024;
  return a;
}
extern "C" int glfer_avgmap_applies(size_t walk, int bins, int depth, int minbin, int maxbin) {
  return avgmap_shape((long long)walk, bins, depth, minbin, maxbin).ok ? 1 : 0;
}

extern "C" hipError_t glfer_launch_avgmap(int mode, const float *psd, size_t fbeg, size_t nframes, int bins, int depth,
                                          int minbin, int maxbin, int max0, int scale_log, double thr255,
                                          double one_m_thr, const float *levels, const unsigned char *colortab,
                                          const double *log_thr, unsigned char *rgb, short *lev, hipStream_t st) {
  if (nframes <= fbeg) return hipSuccess;
  const int band = maxbin - minbin;
  if (band < 1 || minbin < 0 || maxbin > bins || depth < 1) return hipErrorInvalidValue;
  const long long nf = (long long)nframes, walk = nf - (long long)fbeg;
  const AvgMapShape shape = avgmap_shape(walk, bins, depth, minbin, maxbin);
  if (!shape.ok) return hipErrorNotSupported;
  const int chunk = shape.chunk, bpt = shape.bpt;
  const bool ring = shape.ring;
  const size_t shmem = shape.shmem;
  const unsigned blocks = (unsigned)((walk + chunk - 1) / chunk);
  const AvgMapArgs ma{levels, colortab, log_thr, rgb, lev, scale_log, thr255, one_m_thr, (long long)fbeg};
#define GLFER_AVGMAP_NT(B, NT)                                                                                          \
  do {                                                                                                                  \
    const void *fn = ring ? reinterpret_cast<const void *>(avg_fused_kernel<B, true, true, NT>)                         \
                          : reinterpret_cast<const void *>(avg_fused_kernel<B, false, true, NT>);                       \
    hipError_t e = allow_dynamic_lds(fn, shmem);                                                                        \
    if (e != hipSuccess) return e;                                                                                      \
    if (ring) hipLaunchKernelGGL((avg_fused_kernel<B, true, true, NT>), dim3(blocks), dim3(NT), shmem, st, psd, nf, chunk, bins, bins, depth, minbin, maxbin, mode, max0, (double *)nullptr, (double *)nullptr, ma); \
    else hipLaunchKernelGGL((avg_fused_kernel<B, false, true, NT>), dim3(blocks), dim3(NT), shmem, st, psd, nf, chunk, bins, bins, depth, minbin, maxbin, mode, max0, (double *)nullptr, (double *)nullptr, ma); \
  } while (0)
#if GLFER_AVGMAP_WIDE
#define GLFER_AVGMAP(B)                                                                                                 \
  do {                                                                                                                  \
    if (shape.nt == 512) GLFER_AVGMAP_NT(B, 512);                                                                       \
    else GLFER_AVGMAP_NT(B, 256);                                                                                       \
  } while (0)
#else
#define GLFER_AVGMAP(B) GLFER_AVGMAP_NT(B, 256)
#endif
  if (bpt <= 1) GLFER_AVGMAP(1);
  else if (bpt <= 2) GLFER_AVGMAP(2);
  else if (bpt <= 3) GLFER_AVGMAP(3);
  else if (bpt <= 5) GLFER_AVGMAP(5);
  else if (bpt <= 9) GLFER_AVGMAP(9);
  else if (bpt <= 17) GLFER_AVGMAP(17);
  else GLFER_AVGMAP(33);
#undef GLFER_AVGMAP_NT
#undef GLFER_AVGMAP
  return hipGetLastError();
}

// avgdata->cum alone (the shims hand it back to the caller's avg_data_t)
extern "C" hipError_t glfer_launch_avg_cum(const float *psd, size_t nframes, int bins, int n_out, int depth,
                                           int minbin, int maxbin, double *cum, hipStream_t st) {
  if (nframes == 0) return hipSuccess;
  const int band = maxbin - minbin;
  if (band < 1 || minbin < 0 || maxbin > bins || maxbin > n_out || depth < 1) return hipErrorInvalidValue;
  const unsigned chunks = (unsigned)((nframes + AVG_CHUNK - 1) / AVG_CHUNK);
  hipLaunchKernelGGL(avg_cum_kernel, dim3((unsigned)((band + 255) / 256), chunks), dim3(256), 0, st, psd, (long long)nframes,
                     bins, n_out, depth, minbin, maxbin, cum);
  return hipGetLastError();
}
