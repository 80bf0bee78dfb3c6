// ldsbench.hip -- cost of the exchange's LDS access patterns (256 lanes, 16 x 8-byte entries per
// lane, as in the Stockham kernels).  Each pattern is issued 16 x per iteration by every wave of
// 2 or 3 resident blocks per CU; reports LDS-array cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f32 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int PAT, bool WRITE>
__global__ __launch_bounds__(256) void lds_kernel(float *out, int iters, unsigned long long *clk) {
  __shared__ v2f32 buf[4352 + 64];
  const int t = threadIdx.x;
  for (int i = t; i < 4352 + 64; i += 256) buf[i] = v2f32{(float)i, 1.0f};
  __syncthreads();
  int base, stride;
  if (PAT == 0) { base = 17 * t; stride = 1; }                              // exchange 0 write: 16 consecutive entries per lane, lanes 17 apart
  else if (PAT == 1) { const int k = t & 15; base = 17 * (t - k) + k; stride = 17; }   // exchange 1 write
  else if (PAT == 2) { base = t + (t >> 4); stride = 272; }                  // exchange read
  else if (PAT == 3) { base = t; stride = 256; }                             // plain lane-contiguous
  else if (PAT == 4) { base = 16 * t; stride = 1; }                          // unpadded 16 consecutive per lane (conflicts)
  else { base = 18 * t; stride = 1; }                                        // +2 per 16 padding
  v2f32 acc = v2f32{0.0f, 0.0f};
  v2f32 v = v2f32{(float)t, 2.0f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < 16; q++) {
      if (WRITE) {
        buf[base + q * stride] = v;
      } else {
        const v2f32 r = *(volatile v2f32 *)&buf[base + q * stride];
        acc += r;
      }
    }
    if (WRITE) v.x += 1.0f;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  out[blockIdx.x * 256 + t] = acc.x + acc.y + buf[(t * 7) % 4352].x;
  if (blockIdx.x == 0 && t == 0) clk[0] = t1 - t0;
}

int main() {
  float *d_out;
  unsigned long long *d_clk;
  CK(hipMalloc((void **)&d_out, 256 * 8 * 256 * 4));
  CK(hipMalloc((void **)&d_clk, 8));
  const int iters = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const char *name, auto kern, int bpc) -> int {
    const int grid = 256 * bpc * 4;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_out, 10, d_clk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_out, iters, d_clk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk;
    CK(hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost));
    // per CU: grid/256 blocks in sequence-ish; LDS instrs per CU = (grid/256) * 4 waves * iters * 16
    const double instr_per_cu = (double)grid / 256.0 * 4 * iters * 16;
    const double cu_clk = ms * 1e-3 * 2.2e9;
    printf("%-44s: %.3f ms  ~%.1f clk per wave-instruction per CU (at 2.2 GHz), %.1f B/clk/CU; one wave: %.1f ticks/instr\n", name, ms,
           cu_clk / instr_per_cu, 512.0 * instr_per_cu / cu_clk, (double)clk / (iters * 16.0));
    return 0;
  };
  // LDS per block ~35 KB -> 4 blocks/CU fit; 256 threads -> occupancy by LDS
  run("write  exch0 (17t+q)", lds_kernel<0, true>, 1);
  run("write  exch1 (17(t-k)+k+17q)", lds_kernel<1, true>, 1);
  run("write  lane-contiguous (t+256q)", lds_kernel<3, true>, 1);
  run("write  unpadded 16t+q", lds_kernel<4, true>, 1);
  run("write  +2 pad 18t+q", lds_kernel<5, true>, 1);
  run("read   exchange (t+t/16+272m)", lds_kernel<2, false>, 1);
  run("read   lane-contiguous", lds_kernel<3, false>, 1);
  run("read   exch0 pattern 17t+q", lds_kernel<0, false>, 1);
  return 0;
}
