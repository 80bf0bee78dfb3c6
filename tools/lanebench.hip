// lanebench.hip -- issue cost of the cross-lane primitives on gfx950, per wave64 instruction, with
// 8 waves/SIMD resident (all CUs): v_mov_b32_dpp (quad_perm / row_ror / row_half_mirror),
// v_permlane16_swap, v_permlane32_swap, ds_swizzle, ds_bpermute, against v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void lane_kernel(float *out, int iters) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = (float)(threadIdx.x * 16 + i);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      if constexpr (MODE == 0) {
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[i + 1]));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i + 1]) : "v"(v[i]));
      } else if constexpr (MODE == 1) {
        asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[i + 1]));
        asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i + 1]) : "v"(v[i]));
      } else if constexpr (MODE == 2) {
        asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[i + 1]));
        asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(v[i + 1]) : "v"(v[i]));
      } else if constexpr (MODE == 3) {
        asm volatile("v_mov_b32_dpp %0, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[i + 1]));
        asm volatile("v_mov_b32_dpp %0, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v[i + 1]) : "v"(v[i]));
      } else if constexpr (MODE == 4) {
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[i + 1]));
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i + 1]), "+v"(v[i]));
      } else if constexpr (MODE == 5) {
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[i + 1]));
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v[i + 1]), "+v"(v[i]));
      } else if constexpr (MODE == 6) {
        asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(BITMASK_PERM,\"0000p\")\n s_waitcnt lgkmcnt(0)" : "=v"(v[i]) : "v"(v[i + 1]));
        asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(BITMASK_PERM,\"0000p\")\n s_waitcnt lgkmcnt(0)" : "=v"(v[i + 1]) : "v"(v[i]));
      } else if constexpr (MODE == 7) {      // add with a DPP operand
        asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[i + 1]));
        asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(v[i + 1]) : "v"(v[i]));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float *d_out;
  CK(hipMalloc((void **)&d_out, 2048 * 256 * 4));
  const int iters = 4000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const char *name, auto kern) -> int {
    hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, d_out, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, d_out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: 8 waves x iters x 16 instructions
    const double instr_per_simd = 8.0 * iters * 16.0;
    printf("%-36s: %.3f ms  %.2f clk per wave-instruction per SIMD at 2.2 GHz (dependent pairs, 8 waves/SIMD)\n", name, ms,
           ms * 1e-3 * 2.2e9 / instr_per_simd);
    return 0;
  };
  run("v_fma_f32", lane_kernel<0>);
  run("v_mov_b32_dpp quad_perm", lane_kernel<1>);
  run("v_mov_b32_dpp row_ror:8", lane_kernel<2>);
  run("v_mov_b32_dpp row_half_mirror", lane_kernel<3>);
  run("v_add_f32_dpp row_ror:8", lane_kernel<7>);
  run("v_permlane32_swap_b32", lane_kernel<4>);
  run("v_permlane16_swap_b32", lane_kernel<5>);
  run("ds_swizzle_b32 (+wait)", lane_kernel<6>);
  return 0;
}
