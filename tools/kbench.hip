// kbench.hip -- standalone kernel microbenchmarks (developer tool, not part of the library).
//   1. VALU issue rates on gfx950: v_fma_f32 vs v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 at
//      1 and 2 waves per SIMD -- decides whether the FFT core should be written packed.
//   2. the fused spectrogram kernel (N=4096) built for 1 and 2 waves/SIMD, timed with HIP
//      events on synthetic audio and cross-checked against a float64 DFT on the host.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -I../glfer_amd/csrc kbench.hip \
//        ../glfer_amd/csrc/host_tables.cpp -o kbench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define GLFER_NO_LAUNCHERS
#include "spectro2.hip"
#include "spectro16.hip"
#include "spectro16_v3_snapshot.hip"
#include "host_tables.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void valu_kernel(float *out, int iters, float seed, unsigned long long *clk) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  // 16 independent accumulator chains
  v2f acc[16];
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = v2f{seed + i, seed - i};
  v2f a = v2f{1.0000001f, 0.9999999f}, b = v2f{1e-7f, -1e-7f};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if constexpr (MODE == 0) {          // 2 scalar FMAs
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(a.x), "v"(b.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(a.y), "v"(b.y));
      } else if constexpr (MODE == 1) {   // 1 packed FMA
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      } else if constexpr (MODE == 2) {   // 1 packed MUL
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
      } else if constexpr (MODE == 3) {   // 1 packed ADD
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
      } else if constexpr (MODE == 4) {   // 2 scalar ADDs
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i].x) : "v"(b.x));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i].y) : "v"(b.y));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = __builtin_amdgcn_s_memtime() - t0;        // shader clocks
    clk[1] = __builtin_amdgcn_s_memrealtime() - r0;    // 100 MHz ticks
  }
}

template <int MODE>
static void run_valu(const char *name, int blocks_per_cu, float *d_out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = 256 * blocks_per_cu;
  static unsigned long long *d_clk = nullptr;
  if (!d_clk) CK(hipMalloc((void **)&d_clk, 16));
  hipLaunchKernelGGL(valu_kernel<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 100, 1.0f, d_clk);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(valu_kernel<MODE>, dim3(grid), dim3(256), 0, 0, d_out, iters, 1.0f, d_clk);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // lane-results per second: each loop iteration produces 32 float results per lane
  const double lane_ops = (double)grid * 256 * (double)iters * 32.0;
  unsigned long long clk[2];
  CK(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost));
  const double mhz = (double)clk[0] / (double)clk[1] * 100.0;
  printf("valu %-10s waves/SIMD=%d  %.3f ms  %.2f T lane-results/s  clock %.0f MHz  -> %.1f lanes/clk/SIMD\n", name,
         blocks_per_cu, ms, lane_ops / ms / 1e9, mhz, lane_ops / (ms * 1e-3) / (256.0 * 4 * mhz * 1e6));
}

static std::vector<float> synth(size_t n) {
  std::vector<float> x(n);
  unsigned long long s = 0x9E3779B97F4A7C15ull;
  for (size_t i = 0; i < n; i++) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const double u = (double)(s >> 11) / 9007199254740992.0 - 0.5;
    x[i] = (float)(0.5 * sin(2 * M_PI * 1000.0 * i / 48000.0) + 0.25 * sin(2 * M_PI * 7350.5 * i / 48000.0) + 0.17 * u);
  }
  return x;
}

int main(int argc, char **argv) {
  const int nframes = argc > 1 ? atoi(argv[1]) : 65536;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);

  float *d_out;
  CK(hipMalloc((void **)&d_out, 256 * 8 * 256 * sizeof(float)));
  const int wlist[] = {1, 2, 3, 4, 6, 8};
  for (int w : wlist) {
    run_valu<0>("2x v_fma", w, d_out);
    if (w <= 2) run_valu<1>("v_pk_fma", w, d_out);
    run_valu<4>("2x v_add", w, d_out);
  }

  // ---- fused kernel, N=4096, MTM T=5 (NW=2.5, kmax=4), overlap 0
  const int N = 4096, T = 5, NP = 3, H = N, P = N / 2 + 1;
  std::vector<double> tapers((size_t)T * N), sig(T);
  if (!glfer::make_dpss(N, T - 1, 2.5, tapers.data(), sig.data())) { printf("dpss failed\n"); return 1; }
  std::vector<float> taps((size_t)2 * NP * N, 0.0f), tw((size_t)2 * 64 * 64);
  for (int j = 0; j < T; j++) {
    const double sc = sqrt(1.0 / (2.0 * N * (1.0 + sig[j])));
    for (int i = 0; i < N; i++) taps[(size_t)j * N + i] = (float)(tapers[(size_t)j * N + i] * sc);
  }
  glfer::make_twiddles(N, 64, tw.data());
  std::vector<float> x = synth((size_t)nframes * H);
  float *d_x, *d_taps, *d_psd1, *d_psd2;
  float2 *d_tw;
  CK(hipMalloc((void **)&d_x, x.size() * 4));
  CK(hipMalloc((void **)&d_taps, taps.size() * 4));
  CK(hipMalloc((void **)&d_tw, tw.size() * 4));
  CK(hipMalloc((void **)&d_psd1, (size_t)nframes * P * 4));
  CK(hipMalloc((void **)&d_psd2, (size_t)nframes * P * 4));
  CK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_taps, taps.data(), taps.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));

  SpectroParams sp = {};
  sp.stream = d_x; sp.frame0 = 0; sp.nframes = nframes; sp.H = H; sp.R = N - H; sp.npairs = NP;
  sp.history_mode = 0; sp.fmt = GLFER_FMT_F32; sp.taps = d_taps; sp.tw = d_tw; sp.spec_unscale = 1.0f;
  const unsigned grid = (nframes + 3) / 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int variant = 1; variant <= 2; variant++) {
    sp.psd = variant == 1 ? d_psd1 : d_psd2;
    for (int rep = 0; rep < 4; rep++) {
      CK(hipEventRecord(e0));
      if (variant == 1) hipLaunchKernelGGL((glfer::spectro2_kernel<64, GLFER_FMT_F32, false, 1>), dim3(grid), dim3(256), 0, 0, sp);
      else hipLaunchKernelGGL((glfer::spectro2_kernel<64, GLFER_FMT_F32, false, 2>), dim3(grid), dim3(256), 0, 0, sp);
      CK(hipGetLastError());
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double fps = nframes / (ms * 1e-3);
      printf("spectro2<64> WPS=%d rep %d: %.3f ms  %.2f Mframes/s  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)\n", variant, rep,
             ms, fps / 1e6, fps * (4.0 * H + 4.0 * P) / 1e9, fps * (4.0 * H + 4.0 * P) / 8e12 * 100);
    }
  }
  // ---- v3: 16 points per lane, 256 lanes per frame, Stockham radix-16 with LDS exchange
  std::vector<float> tw16((size_t)2 * glfer::make_twiddles16(12, nullptr) * (N / 16));
  glfer::make_twiddles16(12, tw16.data());
  float2 *d_tw16;
  float *d_psd3;
  CK(hipMalloc((void **)&d_tw16, tw16.size() * 4));
  CK(hipMalloc((void **)&d_psd3, (size_t)nframes * P * 4));
  CK(hipMemcpy(d_tw16, tw16.data(), tw16.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> taps_il((size_t)2 * NP * N, 0.0f);
  for (int j = 0; j < T; j++)
    for (int i = 0; i < N; i++) {
      const int TT = N / 16, tt = i % TT, mm = i / TT;
      taps_il[(size_t)(j / 2) * N * 2 + ((size_t)(mm / 2) * TT + tt) * 4 + (size_t)(mm & 1) * 2 + (j & 1)] = taps[(size_t)j * N + i];
    }
  float *d_taps_il;
  CK(hipMalloc((void **)&d_taps_il, taps_il.size() * 4));
  CK(hipMemcpy(d_taps_il, taps_il.data(), taps_il.size() * 4, hipMemcpyHostToDevice));
  SpectroParams sq = sp;
  sq.tw = d_tw16;
  sq.psd = d_psd3;
  for (int variant = 2; variant <= 3; variant++) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
      CK(hipEventRecord(e0));
      if (variant == 2) hipLaunchKernelGGL((glfer_v3::spectro16_kernel<12, GLFER_FMT_F32, false, 2>), dim3(nframes), dim3(256), 0, 0, sq);
      else hipLaunchKernelGGL((glfer_v3::spectro16_kernel<12, GLFER_FMT_F32, false, 3>), dim3(nframes), dim3(256), 0, 0, sq);
      (void)0;
      CK(hipGetLastError());
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0) best = std::min(best, ms);
    }
    printf("v3-snapshot<12> WPS=%d grid=%6d: %.3f ms  %.2f Mframes/s\n", variant, nframes, best, nframes / (best * 1e-3) / 1e6);
  }
  // v4 (persistent, prefetch): WPS sweep at the persistent grid, then timing ablations at WPS=3
  sq.taps = d_taps_il;      // from here on: the current kernel, interleaved taper pairs
  auto time16 = [&](const char *label, auto launch) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
      CK(hipEventRecord(e0));
      launch();
      CK(hipGetLastError());
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0) best = std::min(best, ms);
    }
    const double fps = nframes / (best * 1e-3);
    printf("spectro16<12> %-28s: %.3f ms  %.2f Mframes/s  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)\n", label, best, fps / 1e6,
           fps * (4.0 * H + 4.0 * P) / 1e9, fps * (4.0 * H + 4.0 * P) / 8e12 * 100);
  };
  const unsigned gp = (unsigned)std::min(nframes, 256 * 12);
#define K16(WPS, ABL, LAY, STG) glfer::spectro16_kernel<12, GLFER_FMT_F32, false, WPS, ABL, LAY, STG>
  time16("WPS=2 lay0 grid=2048", [&] { hipLaunchKernelGGL((K16(2, 0, 0, 0)), dim3(2048), dim3(256), 0, 0, sq); });
  time16("WPS=3 lay0 grid=3072", [&] { hipLaunchKernelGGL((K16(3, 0, 0, 0)), dim3(gp), dim3(256), 0, 0, sq); });
  time16("WPS=3 lay1 grid=3072", [&] { hipLaunchKernelGGL((K16(3, 0, 1, 0)), dim3(gp), dim3(256), 0, 0, sq); });
  time16("WPS=3 lay0 stagger 8", [&] { hipLaunchKernelGGL((K16(3, 0, 0, 8)), dim3(gp), dim3(256), 0, 0, sq); });
  time16("WPS=3 lay0 stagger 24", [&] { hipLaunchKernelGGL((K16(3, 0, 0, 24)), dim3(gp), dim3(256), 0, 0, sq); });
  time16("WPS=2 lay0 stagger 16", [&] { hipLaunchKernelGGL((K16(2, 0, 0, 16)), dim3(2048), dim3(256), 0, 0, sq); });
  time16("WPS=4 lay0 grid=4096", [&] { hipLaunchKernelGGL((K16(4, 0, 0, 0)), dim3(4096), dim3(256), 0, 0, sq); });
  time16("WPS=3 lay0 ABL1 no LDS", [&] { hipLaunchKernelGGL((K16(3, 1, 0, 0)), dim3(gp), dim3(256), 0, 0, sq); });
  time16("WPS=3 lay0 ABL2 no bfly", [&] { hipLaunchKernelGGL((K16(3, 2, 0, 0)), dim3(gp), dim3(256), 0, 0, sq); });
  time16("WPS=3 lay0 ABL3 no gather", [&] { hipLaunchKernelGGL((K16(3, 3, 0, 0)), dim3(gp), dim3(256), 0, 0, sq); });
  time16("WPS=3 lay0 grid=3072 again", [&] { hipLaunchKernelGGL((K16(3, 0, 0, 0)), dim3(gp), dim3(256), 0, 0, sq); });
  {
    std::vector<float> a((size_t)64 * P), b((size_t)64 * P);
    CK(hipMemcpy(a.data(), d_psd2, a.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), d_psd3, b.size() * 4, hipMemcpyDeviceToHost));
    double mx = 0, md = 0;
    for (size_t i = 0; i < a.size(); i++) { mx = fmax(mx, fabs(a[i])); md = fmax(md, fabs(a[i] - b[i])); }
    printf("spectro16 vs spectro2, first 64 frames: max|d|/max = %.3e\n", md / mx);
  }
  // ---- correctness: frame 0 and frame 7 against a float64 DFT
  std::vector<float> g1(2 * P), g2(2 * P);
  double worst = 0;
  for (int fi = 0; fi < 2; fi++) {
    const int f = fi ? 7 : 0;
    CK(hipMemcpy(g1.data(), d_psd1 + (size_t)f * P, P * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(g2.data(), d_psd3 + (size_t)f * P, P * 4, hipMemcpyDeviceToHost));
    std::vector<double> ref(P, 0.0);
    for (int j = 0; j < T; j++) {
      for (int k = 0; k < P; k += 37) {       // sampled bins
        double re = 0, im = 0;
        for (int n = 0; n < N; n++) {
          const double v = tapers[(size_t)j * N + n] * (double)x[(size_t)f * H + n];
          const double ang = -2.0 * M_PI * (double)((long long)k * n % N) / N;
          re += v * cos(ang); im += v * sin(ang);
        }
        ref[k] += (re * re + im * im) / N / (1.0 + sig[j]);
      }
    }
    double mx = 0;
    for (int k = 0; k < P; k += 37) mx = fmax(mx, ref[k]);
    for (int k = 0; k < P; k += 37) {
      worst = fmax(worst, fabs(g1[k] - ref[k]) / mx);
      worst = fmax(worst, fabs(g2[k] - ref[k]) / mx);
    }
  }
  printf("max |gpu-ref64|/max(ref) over sampled bins, both variants: %.3e\n", worst);
  return worst < 1e-5 ? 0 : 2;
}
