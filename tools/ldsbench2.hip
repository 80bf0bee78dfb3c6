// ldsbench2.hip -- LDS exchange throughput in the kernels' own shape: 256-thread blocks, k blocks
// resident per CU (k wavefronts per SIMD), every wavefront issues its 16 (re,im) entries as 8
// ds_write2_b64, then the block barriers, reads 16 entries back (ds_read_b64) and barriers again.
// Reports LDS-pipe bytes per clock per CU for the write phase + read phase together and the time
// per exchange, to tell whether the exchange is bound by the pipe or by per-wavefront issue.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f32 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void xch_kernel(float *out, int iters) {
  extern __shared__ v2f32 buf[];                      // 4352 entries used; the rest only sets occupancy
  const int t = threadIdx.x;
  const int k = t & 15;
  v2f32 *w0 = buf + 17 * t;                           // exchange 0 writes: 16 consecutive entries
  v2f32 *w1 = buf + 17 * (t - k) + k;                 // exchange 1 writes: stride 17
  const v2f32 *r = buf + t + (t >> 4);                // reads: stride 272
  v2f32 v[16];
#pragma unroll
  for (int q = 0; q < 16; q++) v[q] = v2f32{(float)(t + q), 1.0f};
  for (int it = 0; it < iters; it++) {
    if (MODE != 2) {
#pragma unroll
      for (int q = 0; q < 16; q++) (((it & 1) ? w1 : w0))[q * ((it & 1) ? 17 : 1)] = v[q];
    }
    if (MODE != 3) __syncthreads();
    if (MODE != 1) {
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const v2f32 x = r[q * 272];
        v[q].x += x.x;
        v[q].y += x.y;
      }
    }
    if (MODE != 3) __syncthreads();
  }
  float s = 0;
#pragma unroll
  for (int q = 0; q < 16; q++) s += v[q].x + v[q].y;
  out[blockIdx.x * 256 + t] = s;
}

int main() {
  float *d_out;
  CK(hipMalloc((void **)&d_out, 4096 * 256 * 4));
  const int iters = 4000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const char *name, auto kern, int bpc) -> int {
    const size_t shmem = (size_t)(160 * 1024 / bpc) - 512;          // forces exactly bpc blocks per CU
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const int grid = 256 * bpc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, 0, d_out, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, 0, d_out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double clk = ms * 1e-3 * 2.2e9;
    const double bytes_cu = (double)bpc * iters * 256.0 * 16 * 8;  // per direction
    printf("%-34s %d blocks/CU: %.3f ms  %.0f clk per exchange and block  %.1f B/clk/CU per direction\n", name, bpc, ms,
           clk / iters, bytes_cu / clk);
    return 0;
  };
  for (int bpc = 1; bpc <= 4; bpc++) run("write+barrier+read+barrier", xch_kernel<0>, bpc);
  for (int bpc = 1; bpc <= 4; bpc++) run("write+barrier+barrier (no reads)", xch_kernel<1>, bpc);
  for (int bpc = 1; bpc <= 4; bpc++) run("barrier+read+barrier (no writes)", xch_kernel<2>, bpc);
  for (int bpc = 1; bpc <= 4; bpc++) run("write+read, no barriers", xch_kernel<3>, bpc);
  return 0;
}
