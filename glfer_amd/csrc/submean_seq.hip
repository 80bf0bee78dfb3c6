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

}  // namespace glfer

using namespace glfer;

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

