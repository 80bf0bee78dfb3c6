// pkbench2.hip -- a radix-16 register DFT + per-lane twiddle multiply, scalar form (fft_inreg.hpp: 6-FMA
// butterflies on separate re/im arrays) against a packed form (v_pk_fma_f32 on (re,im) register pairs,
// 3 packed ops per butterfly, twiddles in SGPR pairs), at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../glfer_amd/csrc/fft_inreg.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using namespace glfer;
typedef float f2 __attribute__((ext_vector_type(2)));

template <int R, int K, int PA, int PB, int LEN>
__device__ __forceinline__ void bfly_pk(f2 (&z)[LEN]) {
  const f2 a = z[PA], b = z[PB];
  if constexpr (K == 0) {
    z[PA] = a + b;
    z[PB] = a - b;
  } else if constexpr (4 * K == R) {   // w = -i: t = (bi, -br)
    f2 x, y;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(x) : "v"(a), "v"(b));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(y) : "v"(a), "v"(b));
    z[PA] = x; z[PB] = y;
  } else {
    constexpr cplx64 u = unit_root(K, R);
    const f2 W = {float(u.c), float(-u.s)};
    f2 t, x;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(t) : "s"(W), "v"(b), "v"(a));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(x) : "s"(W), "v"(b), "v"(t));
    z[PA] = x;
    z[PB] = __builtin_elementwise_fma(f2{2.0f, 2.0f}, a, -x);
  }
}
template <int R, int S, int OFF, int LEN>
__device__ __forceinline__ void dit_pk(f2 (&z)[LEN]) {
  if constexpr (R >= 2) {
    dit_pk<R / 2, 2 * S, OFF, LEN>(z);
    dit_pk<R / 2, 2 * S, OFF + S, LEN>(z);
    static_for<0, R / 2>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int pa = OFF + 2 * S * brev(k, R / 2);
      bfly_pk<R, k, pa, pa + S, LEN>(z);
      if constexpr ((k % 4) == 3) __builtin_amdgcn_sched_barrier(0);
    });
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float2 *tw, int iters) {
  const int t = threadIdx.x;
  float2 twl[4];
  for (int i = 0; i < 4; i++) twl[i] = tw[i * 256 + t];
  if constexpr (MODE == 0) {
    float re[16], im[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { re[i] = (float)(t + i) * 1e-3f; im[i] = (float)(t - i) * 1e-3f; }
    for (int it = 0; it < iters; it++) {
      dit<16, 1, 0, 16>(re, im);
#pragma unroll
      for (int i = 1; i < 16; i++) {
        const float2 w = twl[(i + it) & 3];
        const float r = re[i] * w.x - im[i] * w.y, m = re[i] * w.y + im[i] * w.x;
        re[i] = r * 0.25f; im[i] = m * 0.25f;
      }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += re[i] + im[i];
    out[blockIdx.x * 256 + t] = s;
  } else {
    f2 z[16];
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = f2{(float)(t + i) * 1e-3f, (float)(t - i) * 1e-3f};
    for (int it = 0; it < iters; it++) {
      dit_pk<16, 1, 0, 16>(z);
#pragma unroll
      for (int i = 1; i < 16; i++) {
        const float2 w2 = twl[(i + it) & 3];
        const f2 w = f2{w2.x, w2.y} * 0.25f;
        f2 r, x;
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(z[i]), "v"(w));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(x) : "v"(z[i]), "v"(w), "v"(r));
        z[i] = x;
      }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += z[i].x + z[i].y;
    out[blockIdx.x * 256 + t] = s;
  }
}

int main() {
  float *d_out; float2 *d_tw;
  CK(hipMalloc((void **)&d_out, 2048 * 256 * 4));
  CK(hipMalloc((void **)&d_tw, 4096 * 8));
  static float2 tw[4096];
  for (int i = 0; i < 4096; i++) tw[i] = make_float2(cosf(i * 1e-3f), sinf(i * 1e-3f));
  CK(hipMemcpy(d_tw, tw, sizeof(tw), hipMemcpyHostToDevice));
  // correctness of the packed form against the scalar form: same output sums
  static float h0[256], h1[256];
  hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, d_out, d_tw, 3);
  CK(hipMemcpy(h0, d_out, sizeof(h0), hipMemcpyDeviceToHost));
  hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, d_out, d_tw, 3);
  CK(hipMemcpy(h1, d_out, sizeof(h1), hipMemcpyDeviceToHost));
  double md = 0, mx = 0;
  for (int i = 0; i < 256; i++) { md = fmax(md, fabs((double)h0[i] - h1[i])); mx = fmax(mx, fabs((double)h0[i])); }
  printf("packed vs scalar: max |diff| %.3g of max %.3g\n", md, mx);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 3000;
  auto run = [&](const char *name, auto kern) -> int {
    printf("%-28s", name);
    for (int wps : {1, 2, 3, 4, 8}) {
      const int grid = 256 * wps;
      float best = 1e9f;
      for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_out, d_tw, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("  %dw: %6.1f ns", wps, best * 1e6 / ((double)wps * iters));
    }
    printf("   (ns per radix-16 round per SIMD)\n");
    return 0;
  };
  run("scalar (6-FMA butterflies)", k<0>);
  run("packed (v_pk_fma_f32)", k<1>);
  run("scalar (6-FMA butterflies)", k<0>);
  run("packed (v_pk_fma_f32)", k<1>);
  return 0;
}
