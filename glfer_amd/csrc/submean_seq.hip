// submean_seq.hip -- per-hop means in the REFERENCE'S OWN ORDER (fft.c:88-92):
//     sig_mean = 0.0;  for (i = 0; i < n_eff; i++) sig_mean += audio_buf[i];  sig_mean /= n_eff;
// a float accumulated sample after sample.  A parallel (tree) sum is the more accurate one, but it
// is not the reference's: on a hop with a DC level comparable to the signal the sequential sum
// drifts by up to ~n*eps/4 (relative) -- the roundings of k*dc + dc all fall the same way for long
// runs of k -- and the residual step that leaves in the frame shows at the low bins (2e-5 of the
// row maximum for dc = rms at N = 4096, tests/test_gpu_round3.py).  cfg.sub_mean =
// GLFER_SUBMEAN_EXACT asks for these means; the estimator then reads a corrected copy of the
// stream made with them (glfer_hip.cpp submean_scratch).
//
// Shape: the chain is sequential per hop, so a LANE walks a hop and 64 hops walk side by side in a
// wavefront.  The samples come in coalesced (a wavefront's load = 64 consecutive samples of ONE
// hop) and are transposed through a wavefront-private LDS tile [64 hops][TILE samples], row stride
// TILE + 1 words: the writes are consecutive words, the transposed reads hit 64 different banks...
// twice over (64 lanes, 32 banks): two passes, the floor for 64 dwords.  No barrier: one wavefront
// owns the tile (the LDS queue orders its own writes and reads).  HBM-bound by design: the stream
// is read once more (4 B per sample); the chain itself, 4-5 clocks per add with four wavefronts per
// SIMD side by side, is an order of magnitude below that.
#include "stockham16.hpp"

namespace glfer {

template <int FMT, int TILE>
__global__ __launch_bounds__(256) void hop_means_seq_kernel(const void *in, float *means, int H, long long nhops) {
  constexpr unsigned esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  __shared__ float tile_all[4][64 * (TILE + 1)];
  const unsigned l = threadIdx.x & 63u;
  const unsigned wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float *tile = tile_all[wv];
  const long long hop0 = ((long long)blockIdx.x * 4 + wv) * 64;          // this wavefront's 64 hops
  if (hop0 >= nhops) return;
  const int rows = (int)(nhops - hop0 < 64 ? nhops - hop0 : 64);
  // one descriptor over the wavefront's hops; rows past the last hop and samples past H read 0 and
  // are never summed (the lane's loop bound, the row bound below)
  const long long span = (long long)rows * H * esz;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char *>(reinterpret_cast<const char *>(in)) + hop0 * (long long)H * esz, 0,
      (unsigned)(span > 0x7fffffffLL ? 0x7fffffffLL : span), 0x00020000);
  float s = 0.0f;
  float v[64];
  auto fetch = [&](int k0) {                         // row r of the tile: samples k0 + l of hop r
    // everything in the VGPR offset (the descriptor's range check covers it): rows past the last
    // hop read 0 and are never stored; samples past H are never summed (the chain's bound below)
    const unsigned base = k0 + (int)l < H ? (unsigned)(k0 + (int)l) * esz : 0x80000000u;
#pragma unroll
    for (int r = 0; r < 64; r++) v[r] = buf_sample<FMT>(rs, base + (unsigned)r * (unsigned)H * esz, 0u);
  };
  static_assert(TILE == 64, "a wavefront's load covers one row of the tile");
  fetch(0);
  for (int k0 = 0; k0 < H; k0 += TILE) {
#pragma unroll
    for (int r = 0; r < 64; r++) tile[r * (TILE + 1) + l] = v[r];
    if (k0 + TILE < H) fetch(k0 + TILE);             // the next tile's loads fly under this tile's chain
    const int kn = H - k0 < TILE ? H - k0 : TILE;
    const float *row = tile + l * (TILE + 1);
    if (kn == TILE) {
#pragma unroll
      for (int k = 0; k < TILE; k++) s += row[k];    // fft.c:89-91: one float add per sample, in order
    } else {
      for (int k = 0; k < kn; k++) s += row[k];
    }
  }
  if ((int)l < rows) means[hop0 + l] = s / (float)H; // fft.c:92: float /= int
}

// The same chains with FEWER hops per wavefront (round 4).  The kernel above puts 64 hops side by side, so a
// launch over n hops has n/64 wavefronts: the 2^18 hops of a whole C3 stream fill the chip, the 4096 hops of a
// 64 MiB piece of it (glfer_hip.cpp runs the means piece by piece in front of the estimator, so that the
// estimator's read of the piece comes out of the 256 MiB Infinity Cache) would be 64 wavefronts on 256 CUs.  Lane
// efficiency is not what bounds this kernel -- one add per sample: with 4 of 64 lanes on the chains a CU still
// sums 26 B/clk, twice its share of HBM -- bytes in flight are.  Here a wavefront owns HPW = 16 or 4 hops and a
// tile of [HPW][SPT] samples, HPW * SPT = 4096: sixteen loads of 64 lanes x 4 consecutive samples (a hop's 256
// samples per instruction: 16 KB in flight per wavefront whatever HPW), the next tile's loads issued before this
// tile's chains run; lane r < HPW walks row r with one ds_read_b128 per four adds (row stride SPT + 4 words: the
// HPW lanes' 16-byte reads fall in different banks).  Persistent: `gridDim.x` blocks walk the groups with a
// stride, so the launcher decides how many CUs the pass may occupy beside an estimator kernel that needs the
// vector ALU.  H must be a multiple of SPT and the stream 4-sample aligned (else: the kernel above).
template <int FMT, int HPW>
__global__ __launch_bounds__(256) void hop_means_tiled_kernel(const void *in, float *means, int H, long long nhops) {
  constexpr unsigned esz = FMT == GLFER_FMT_F32 ? 4 : (FMT == GLFER_FMT_S16 ? 2 : 1);
  constexpr int SPT = 4096 / HPW, RS = SPT + 4, LPH = SPT / 256;      // loads per hop and tile
  static_assert(HPW == 16 || HPW == 4, "16 loads of 256 samples per tile");
  typedef float v4f32 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float tile_all[4][HPW * RS];
  const unsigned l = threadIdx.x & 63u;
  const unsigned wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float *tile = tile_all[wv];
  const long long ngroups = (nhops + HPW - 1) / HPW;
  const long long wstride = (long long)gridDim.x * 4;
  const long long g0 = (long long)blockIdx.x * 4 + wv;
  if (g0 >= ngroups) return;
  const int tpg = H / SPT;                                            // tiles per group
  const long long ntiles = ((ngroups - g0 + wstride - 1) / wstride) * tpg;
  v4f32 v[16];
  auto fetch = [&](long long i) {                                     // tile i of this wavefront
    const long long g = g0 + (i / tpg) * wstride;
    const int k0 = (int)(i % tpg) * SPT;
    const long long hop0 = g * HPW;
    const int rows = (int)(nhops - hop0 < HPW ? nhops - hop0 : HPW);
    // a descriptor over the group's hops: rows past the last hop read 0 (and are never stored)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(in)) + hop0 * (long long)H * esz, 0, (unsigned)((long long)rows * H * esz), 0x00020000);
    const unsigned lane_off = (unsigned)(k0 + 4 * (int)l) * esz;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const unsigned off = lane_off + ((unsigned)(j / LPH) * (unsigned)H + (unsigned)(j % LPH) * 256u) * esz;
      if constexpr (FMT == GLFER_FMT_F32) {
        v[j] = __builtin_bit_cast(v4f32, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0u, GLFER_X_LOAD_AUX));
      } else if constexpr (FMT == GLFER_FMT_S16) {
        typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
        const v2u32 q = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0u, GLFER_X_LOAD_AUX);
        v[j].x = __uint_as_float(q.x);                                  // raw until the tile is written: nothing waits for a load here
        v[j].y = __uint_as_float(q.y);
      } else {
        v[j].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0u, GLFER_X_LOAD_AUX));
      }
    }
  };
  auto to_tile = [&] {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      v4f32 q;
      if constexpr (FMT == GLFER_FMT_F32) {
        q = v[j];
      } else if constexpr (FMT == GLFER_FMT_S16) {                      // wav_fmt.c:111-114
        const int a = (int)__float_as_uint(v[j].x), b = (int)__float_as_uint(v[j].y);
        q = v4f32{(float)(short)(a & 0xffff) / 32768.0f, (float)(a >> 16) / 32768.0f, (float)(short)(b & 0xffff) / 32768.0f, (float)(b >> 16) / 32768.0f};
      } else {                                                          // wav_fmt.c:106-108
        const unsigned a = __float_as_uint(v[j].x);
        q = v4f32{((float)(a & 0xffu) - 128.0f) / 128.0f, ((float)((a >> 8) & 0xffu) - 128.0f) / 128.0f,
                  ((float)((a >> 16) & 0xffu) - 128.0f) / 128.0f, ((float)(a >> 24) - 128.0f) / 128.0f};
      }
      *reinterpret_cast<v4f32 *>(tile + (j / LPH) * RS + (j % LPH) * 256 + 4 * (int)l) = q;
    }
  };
  float s = 0.0f;
  fetch(0);
  for (long long i = 0; i < ntiles; i++) {
    to_tile();
    if (i + 1 < ntiles) fetch(i + 1);                                   // the next tile's loads fly under this tile's chains
    if (l < (unsigned)HPW) {
      const v4f32 *row = reinterpret_cast<const v4f32 *>(tile + l * RS);
#pragma unroll 8
      for (int k = 0; k < SPT / 4; k++) {                               // fft.c:89-91: one float add per sample, in order
        const v4f32 q = row[k];
        s += q.x;
        s += q.y;
        s += q.z;
        s += q.w;
      }
      if ((int)(i % tpg) == tpg - 1) {
        const long long hop = (g0 + (i / tpg) * wstride) * HPW + l;
        if (hop < nhops) means[hop] = s / (float)H;                     // fft.c:92: float /= int
        s = 0.0f;
      }
    }
  }
}

}  // namespace glfer

using namespace glfer;

// hpw: 16 or 4 (the tiled kernel; needs H % (4096 / hpw) == 0 and the stream aligned to four samples), anything else: the
// 64-hops-per-wavefront kernel.  blocks: the tiled kernel's grid (0: one block per four groups, i.e. not persistent).
extern "C" hipError_t glfer_launch_hop_means_tiled(const void *in, float *means, int H, long long nhops, int fmt, int hpw, unsigned blocks,
                                                   hipStream_t st) {
  if (nhops <= 0) return hipSuccess;
  const unsigned esz = fmt == GLFER_FMT_F32 ? 4 : (fmt == GLFER_FMT_S16 ? 2 : 1);
  if ((hpw != 16 && hpw != 4) || H % (4096 / hpw) != 0 || (reinterpret_cast<uintptr_t>(in) & (4u * esz - 1u)) != 0 ||
      (long long)hpw * H * esz > 0x7fffffffLL)
    return hipErrorInvalidValue;
  const long long ngroups = (nhops + hpw - 1) / hpw;
  long long grid = (ngroups + 3) / 4;
  if (blocks && grid > blocks) grid = blocks;
#define GLFER_MEANS_TILED(F, W) hipLaunchKernelGGL((hop_means_tiled_kernel<F, W>), dim3((unsigned)grid), dim3(256), 0, st, in, means, H, nhops)
  switch (fmt) {
    case GLFER_FMT_F32: if (hpw == 16) GLFER_MEANS_TILED(GLFER_FMT_F32, 16); else GLFER_MEANS_TILED(GLFER_FMT_F32, 4); break;
    case GLFER_FMT_S16: if (hpw == 16) GLFER_MEANS_TILED(GLFER_FMT_S16, 16); else GLFER_MEANS_TILED(GLFER_FMT_S16, 4); break;
    case GLFER_FMT_U8: if (hpw == 16) GLFER_MEANS_TILED(GLFER_FMT_U8, 16); else GLFER_MEANS_TILED(GLFER_FMT_U8, 4); break;
    default: return hipErrorInvalidValue;
  }
#undef GLFER_MEANS_TILED
  return hipGetLastError();
}

extern "C" hipError_t glfer_launch_hop_means_seq(const void *in, float *means, int H, long long nhops, int fmt, hipStream_t st) {
  if (nhops <= 0) return hipSuccess;
  const unsigned grid = (unsigned)((nhops + 255) / 256);
  switch (fmt) {
    case GLFER_FMT_F32: hipLaunchKernelGGL((hop_means_seq_kernel<GLFER_FMT_F32, 64>), dim3(grid), dim3(256), 0, st, in, means, H, nhops); break;
    case GLFER_FMT_S16: hipLaunchKernelGGL((hop_means_seq_kernel<GLFER_FMT_S16, 64>), dim3(grid), dim3(256), 0, st, in, means, H, nhops); break;
    case GLFER_FMT_U8: hipLaunchKernelGGL((hop_means_seq_kernel<GLFER_FMT_U8, 64>), dim3(grid), dim3(256), 0, st, in, means, H, nhops); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

