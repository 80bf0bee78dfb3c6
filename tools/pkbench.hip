// pkbench.hip -- issue rate of packed f32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) against
// v_fma_f32 on gfx950, per wave64 instruction and SIMD, at 1, 2, 4 and 8 resident waves per SIMD.
// 16 independent chains per wave (dependency distance 16 instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void pk_kernel(float *out, int iters, float c0, float c1) {
  float2v v[16];
  float2v c = {c0, c1};
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = float2v{(float)(threadIdx.x * 16 + i), (float)i};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if constexpr (MODE == 0) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].x) : "v"(c.x), "v"(c.y));
      } else if constexpr (MODE == 1) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
      } else if constexpr (MODE == 2) {
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
      } else if constexpr (MODE == 3) {
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
      } else if constexpr (MODE == 4) {   // op_sel swizzle + neg: (-wi, wi) * (b.y, b.x) + t
        asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(v[i]) : "v"(c));
      } else if constexpr (MODE == 5) {   // SGPR pair as the constant
        asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(v[i]) : "s"(c));
      } else if constexpr (MODE == 6) {   // two plain fmas with an SGPR constant
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i].x) : "s"(c0));
      } else if constexpr (MODE == 7) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(c.x));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += v[i].x + v[i].y;
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
    printf("%-44s", name);
    for (int wps : {1, 2, 4, 8}) {
      const int grid = 256 * wps;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_out, 200, 1.0f, 0.5f);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_out, iters, 1.0f, 0.5f);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double instr_per_simd = (double)wps * iters * 16.0;
      printf("  %dw: %.2f ns", wps, ms * 1e6 / instr_per_simd);
    }
    printf("   (ns per wave-instruction per SIMD)\n");
    return 0;
  };
  run("v_fma_f32", pk_kernel<0>);
  run("v_add_f32", pk_kernel<7>);
  run("v_fma_f32 sgpr const", pk_kernel<6>);
  run("v_pk_fma_f32", pk_kernel<1>);
  run("v_pk_mul_f32", pk_kernel<2>);
  run("v_pk_add_f32", pk_kernel<3>);
  run("v_pk_fma_f32 op_sel+neg", pk_kernel<4>);
  run("v_pk_fma_f32 op_sel+neg, sgpr pair", pk_kernel<5>);
  return 0;
}
