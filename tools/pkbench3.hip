// pkbench3.hip -- issue cost of the f32 VALU encodings the butterflies compile to, per wave64
// instruction and SIMD, at 2 and 8 resident wavefronts per SIMD: VOP2 (v_add/v_mul/v_fmac), VOP2 with
// a 32-bit literal (v_fmamk/v_fmaak), VOP3 v_fma_f32 with three registers, with an inline constant
// and a neg modifier, and with an SGPR operand.  16 independent chains per wavefront.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float c0, float c1) {
  float v[16], c = c0 + (float)threadIdx.x * 1e-9f, d = c1 + (float)threadIdx.x * 1e-9f;
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = (float)(threadIdx.x * 16 + i) * 1e-6f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if constexpr (MODE == 0) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(v[i]) : "v"(c));
      else if constexpr (MODE == 1) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(v[i]) : "v"(c));
      else if constexpr (MODE == 2) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
      else if constexpr (MODE == 3) asm volatile("v_fmamk_f32 %0, %0, 0x3f6c835e, %1" : "+v"(v[i]) : "v"(d));
      else if constexpr (MODE == 4) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f6c835e" : "+v"(v[i]) : "v"(d));
      else if constexpr (MODE == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
      else if constexpr (MODE == 6) asm volatile("v_fma_f32 %0, %1, 2.0, -%0" : "+v"(v[i]) : "v"(c));
      else if constexpr (MODE == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "s"(c0), "v"(d));
      else if constexpr (MODE == 8) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(v[i]) : "s"(c0), "v"(d));
      else if constexpr (MODE == 9) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(v[i]) : "v"(c));
      else if constexpr (MODE == 10) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
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
  const int iters = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const char *name, auto kern) -> int {
    printf("%-44s", name);
    for (int wps : {1, 2, 3, 4, 8}) {
      const int grid = 256 * wps;
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_out, iters, 1.0f, 0.5f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("  %dw: %.2f", wps, best * 1e6 / ((double)wps * iters * 16.0));
    }
    printf("   ns per wave-instruction per SIMD\n");
    return 0;
  };
  run("warm-up (v_add_f32_e32)", k<0>);
  run("v_add_f32_e32  (VOP2)", k<0>);
  run("v_sub_f32_e32  (VOP2)", k<9>);
  run("v_mul_f32_e32  (VOP2)", k<1>);
  run("v_fmac_f32_e32 (VOP2)", k<2>);
  run("v_fmac_f32_e32 sgpr src0", k<8>);
  run("v_fmamk_f32    (VOP2 + literal)", k<3>);
  run("v_fmaak_f32    (VOP2 + literal)", k<4>);
  run("v_fma_f32 v,v,v (VOP3)", k<5>);
  run("v_fma_f32 -v,v,v (VOP3 neg)", k<10>);
  run("v_fma_f32 v,2.0,-v (VOP3 inline+neg)", k<6>);
  run("v_fma_f32 s,v,v (VOP3 sgpr)", k<7>);
  return 0;
}
