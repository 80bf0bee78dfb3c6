// mixbench.hip -- what does HBM give a kernel that does nothing but stream, at the READ : WRITE mix of the estimator
// kernels?  One "unit" = RD KB read + WR KB written by one wavefront (16-byte accesses, 1 KB per wave-instruction,
// everything contiguous and aligned: the best case), units dealt out to a persistent grid.  The written value depends
// on the values read.  Mixes: C1 2:2, C2 4:8, C3 16:8, C3 at 75 % overlap 4:8, C4 64:32, plus read-only, write-only, copy.
// A second store shape writes the unit as the product does: 4-byte stores, 256 B per wave-instruction, rows of 2049 floats
// (dense: every row starts 4 bytes further off a 256-byte boundary than the one before; or at a pitch of 2112 floats).
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/mixbench tools/mixbench.hip ; tools/bin/mixbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// RD, WR: KB per unit.  ROWS: the unit is written as WR/8 rows of 2049 floats with dword stores (WR a multiple of 8)
template <int RD, int WR, bool ROWS, int AUX, int PITCH = 2049>
__global__ __launch_bounds__(256) void mix_kernel(const float *in, float *out, long long units) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  for (long long u = wave; u < units; u += nwaves) {
    v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (RD > 0) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in + (size_t)u * (RD * 256)), 0, RD * 1024, 0x00020000);
      v4f v[RD];
#pragma unroll
      for (int j = 0; j < RD; j++) v[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, 0));
#pragma unroll
      for (int j = 0; j < RD; j++) acc += v[j];
    } else {
      acc = v4f{(float)u, 1.0f, 2.0f, 3.0f};
    }
    if constexpr (WR > 0 && !ROWS) {
      const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)u * (WR * 256), 0, WR * 1024, 0x00020000);
#pragma unroll
      for (int j = 0; j < WR; j++) {
        const v4f o = acc + (float)j;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o), ws, lane * 16, j * 1024, AUX);
      }
    } else if constexpr (WR > 0) {
      constexpr int NR = WR / 8;
      const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)u * (NR * PITCH), 0, NR * PITCH * 4, 0x00020000);
#pragma unroll
      for (int r = 0; r < NR; r++) {
#pragma unroll
        for (int j = 0; j < 33; j++)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc.x + (float)j), ws, r * (PITCH * 4) + lane * 4, j * 256, AUX);   // (past bin 2048: into the next row's start / the padding, or dropped)
      }
    }
    if constexpr (WR == 0) {
      if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[u] = acc.x;   // keep the loads live
    }
  }
}

template <int RD, int WR, bool ROWS, int AUX, int PITCH = 2049>
static void run(const char *name, const float *in, float *out, size_t in_bytes, size_t out_bytes, int blocks_per_cu) {
  long long units = 1LL << 40;
  if (RD > 0) units = (long long)(in_bytes / (RD * 1024));
  if (WR > 0) {
    const long long wu = (long long)(out_bytes / (ROWS ? (WR / 8) * PITCH * 4 + PITCH * 4 : WR * 1024));
    units = wu < units ? wu : units;
  }
  const int grid = 256 * blocks_per_cu;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 6; rep++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((mix_kernel<RD, WR, ROWS, AUX, PITCH>), dim3(grid), dim3(256), 0, 0, in, out, units);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep > 0 && ms < best) best = ms;
  }
  const double rd = (double)units * RD * 1024, wr = (double)units * (ROWS ? (WR / 8) * 8196.0 : WR * 1024.0);
  printf("%-34s blocks/CU %d  %7.3f ms  read %6.0f GB/s  write %6.0f GB/s  total %6.0f GB/s = %.3f of 8 TB/s\n", name, blocks_per_cu, best,
         rd / best / 1e6, wr / best / 1e6, (rd + wr) / best / 1e6, (rd + wr) / best / 1e6 / 8000.0);
  CK(hipEventDestroy(a));
  CK(hipEventDestroy(b));
}

int main() {
  const size_t in_bytes = 4ULL << 30, out_bytes = 8ULL << 30;
  float *in, *out;
  CK(hipMalloc(&in, in_bytes));
  CK(hipMalloc(&out, out_bytes + (1 << 20)));
  CK(hipMemset(in, 0, in_bytes));
  CK(hipMemset(out, 0, out_bytes));
  for (int bpc : {4, 8}) {
    run<8, 0, false, 0>("read only", in, out, in_bytes, out_bytes, bpc);
    run<0, 8, false, 0>("write only", in, out, in_bytes, out_bytes, bpc);
    run<0, 8, false, 2>("write only, non-temporal", in, out, in_bytes, out_bytes, bpc);
    run<8, 8, false, 0>("copy 1:1 (8 KB : 8 KB)", in, out, in_bytes, out_bytes, bpc);
    run<2, 2, false, 0>("C1 mix 2 KB : 2 KB", in, out, in_bytes, out_bytes, bpc);
    run<4, 8, false, 0>("C2 mix 4 KB : 8 KB", in, out, in_bytes, out_bytes, bpc);
    run<4, 8, true, 0>("C2 mix, rows of 2049, dword stores", in, out, in_bytes, out_bytes, bpc);
    run<4, 8, true, 0, 2112>("C2 mix, rows at a pitch of 2112", in, out, in_bytes, out_bytes, bpc);
    run<16, 8, false, 0>("C3 mix 16 KB : 8 KB", in, out, in_bytes, out_bytes, bpc);
    run<16, 8, true, 0>("C3 mix, rows of 2049, dword stores", in, out, in_bytes, out_bytes, bpc);
  }
  CK(hipFree(in));
  CK(hipFree(out));
  return 0;
}
