// xbench.hip -- multitaper (BASELINE config 3: N=4096, NW=2.5, 5 tapers, overlap 0) kernel
// comparison: the packed kernel (spectro16.hip, 3 transforms per frame) against the shared-odd-
// taper kernel (spectro16x.hip, 2.5 per frame) and its build variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-slp-vectorize -Iglfer_amd/csrc tools/xbench.hip \
//         glfer_amd/csrc/host_tables.cpp -o tools/bin/xbench && tools/bin/xbench [nframes]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define GLFER_NO_LAUNCHERS
#include "spectro16.hip"
#include "spectro16x.hip"
#include "spectro16y.hip"
#include "host_tables.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
  const int nframes = argc > 1 ? atoi(argv[1]) : 262144;
  constexpr int LOGN = 12, N = 1 << LOGN, H = N, P = N / 2 + 1, T = 5, NP = 3, TT = N / 16;
  std::vector<double> tapers((size_t)T * N), sig(T);
  if (!glfer::make_dpss(N, T - 1, 2.5, tapers.data(), sig.data())) { printf("dpss failed\n"); return 1; }
  std::vector<float> taps((size_t)2 * NP * N, 0.0f), xt(N);
  for (int j = 0; j < T; j++)
    for (int i = 0; i < N; i++) {
      const int t = i % TT, m = i / TT;
      taps[(size_t)(j / 2) * N * 2 + ((size_t)(m / 2) * TT + t) * 4 + (size_t)(m & 1) * 2 + (j & 1)] =
          (float)(tapers[(size_t)j * N + i] * sqrt(1.0 / (2.0 * N * (1.0 + sig[j]))));
    }
  for (int i = 0; i < N; i++) {
    const int t = i % TT, m = i / TT;
    xt[((size_t)(m / 4) * TT + t) * 4 + (size_t)(m & 3)] = (float)(tapers[(size_t)(T - 1) * N + i] * sqrt(1.0 / (4.0 * N * (1.0 + sig[T - 1]))));
  }
  std::vector<float> tw((size_t)2 * glfer::make_twiddles16(LOGN, nullptr) * TT);
  glfer::make_twiddles16(LOGN, tw.data());
  const size_t ns = (size_t)nframes * H;
  std::vector<float> x(ns);
  unsigned s = 12345;
  for (size_t i = 0; i < ns; i++) { s = s * 1664525u + 1013904223u; x[i] = (float)((s >> 8) * (1.0 / 16777216.0) - 0.5) + 0.3f * sinf(0.01f * (float)i); }
  for (size_t i = (size_t)3 * H; i < (size_t)4 * H; i++) x[i] *= 1e-4f;   // a quiet frame next to loud ones
  float *d_x, *d_taps, *d_xt, *d_psd1, *d_psd2;
  float2 *d_tw;
  CK(hipMalloc((void **)&d_x, ns * 4));
  CK(hipMalloc((void **)&d_taps, taps.size() * 4));
  CK(hipMalloc((void **)&d_xt, xt.size() * 4));
  CK(hipMalloc((void **)&d_tw, tw.size() * 4));
  CK(hipMalloc((void **)&d_psd1, (size_t)nframes * P * 4));
  CK(hipMalloc((void **)&d_psd2, (size_t)nframes * P * 4));
  CK(hipMemcpy(d_x, x.data(), ns * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_taps, taps.data(), taps.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_xt, xt.data(), xt.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
  SpectroParams sp = {};
  sp.stream = d_x; sp.nframes = nframes; sp.H = H; sp.R = N - H; sp.npairs = NP; sp.fmt = GLFER_FMT_F32;
  sp.taps = d_taps; sp.tw = d_tw; sp.xtaps = d_xt; sp.spec_unscale = 1.0f;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timeit = [&](const char *label, auto launch) -> int {
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
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
    printf("%-44s: %.3f ms  %.1f Mframes/s  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)\n", label, best, fps / 1e6,
           fps * (4.0 * H + 4.0 * P) / 1e9, fps * (4.0 * H + 4.0 * P) / 8e12 * 100);
    return 0;
  };
  sp.psd = d_psd1;
  timeit("packed, WPS=3, grid 3072", [&] { hipLaunchKernelGGL((glfer::spectro16_kernel<LOGN, GLFER_FMT_F32, false, 3>), dim3(3072), dim3(256), 0, 0, sp); });
  sp.psd = d_psd2;
#define KX(WPS) glfer::spectro16x_kernel<LOGN, GLFER_FMT_F32, WPS>
  timeit("shared odd taper, WPS=3, grid 3072", [&] { hipLaunchKernelGGL((KX(3)), dim3(3072), dim3(256), 0, 0, sp); });
  timeit("shared odd taper, WPS=3, grid 1536", [&] { hipLaunchKernelGGL((KX(3)), dim3(1536), dim3(256), 0, 0, sp); });
  timeit("shared odd taper, WPS=3, grid 768", [&] { hipLaunchKernelGGL((KX(3)), dim3(768), dim3(256), 0, 0, sp); });
  timeit("shared odd taper, WPS=2, grid 2048", [&] { hipLaunchKernelGGL((KX(2)), dim3(2048), dim3(256), 0, 0, sp); });
  timeit("shared odd taper, WPS=2, grid 512", [&] { hipLaunchKernelGGL((KX(2)), dim3(512), dim3(256), 0, 0, sp); });
  timeit("shared odd taper, WPS=3, grid 3072 again", [&] { hipLaunchKernelGGL((KX(3)), dim3(3072), dim3(256), 0, 0, sp); });
  timeit("shared odd taper, WPS=3, grid 6144", [&] { hipLaunchKernelGGL((KX(3)), dim3(6144), dim3(256), 0, 0, sp); });
  timeit("shared odd taper, WPS=3, grid 12288", [&] { hipLaunchKernelGGL((KX(3)), dim3(12288), dim3(256), 0, 0, sp); });
  sp.psd = d_psd1;
  timeit("packed, WPS=3, grid 6144", [&] { hipLaunchKernelGGL((glfer::spectro16_kernel<LOGN, GLFER_FMT_F32, false, 3>), dim3(6144), dim3(256), 0, 0, sp); });
  timeit("packed, WPS=3, grid 12288", [&] { hipLaunchKernelGGL((glfer::spectro16_kernel<LOGN, GLFER_FMT_F32, false, 3>), dim3(12288), dim3(256), 0, 0, sp); });
  timeit("packed, WPS=3, grid 24576", [&] { hipLaunchKernelGGL((glfer::spectro16_kernel<LOGN, GLFER_FMT_F32, false, 3>), dim3(24576), dim3(256), 0, 0, sp); });
  timeit("packed, WPS=3, grid 65536", [&] { hipLaunchKernelGGL((glfer::spectro16_kernel<LOGN, GLFER_FMT_F32, false, 3>), dim3(65536), dim3(256), 0, 0, sp); });
  timeit("packed, WPS=3, grid 4608", [&] { hipLaunchKernelGGL((glfer::spectro16_kernel<LOGN, GLFER_FMT_F32, false, 3>), dim3(4608), dim3(256), 0, 0, sp); });
  sp.psd = d_psd2;
  timeit("shared odd taper, WPS=3, grid 24576", [&] { hipLaunchKernelGGL((KX(3)), dim3(24576), dim3(256), 0, 0, sp); });
  timeit("shared odd taper, WPS=3, grid 65536", [&] { hipLaunchKernelGGL((KX(3)), dim3(65536), dim3(256), 0, 0, sp); });
  sp.psd = d_psd2;
  {
    auto ky = glfer::spectro16y_kernel<GLFER_FMT_F32>;
    const size_t shy = (size_t)glfer::LaunchY::LDS_WORDS * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(ky), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shy));
    for (unsigned g : {2048u, 8192u, 16384u, 32768u, 65536u, 131072u}) {
      char label[64];
      snprintf(label, sizeof label, "two frames/wavefront (y), grid %u", g);
      timeit(label, [&] { hipLaunchKernelGGL(ky, dim3(g), dim3(256), shy, 0, sp); });
    }
  }
  {
    auto k1 = glfer::spectro16y_kernel<GLFER_FMT_F32, 1>;
    const size_t shy = (size_t)glfer::LaunchY::LDS_WORDS * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shy));
    auto k0 = glfer::spectro16y_kernel<GLFER_FMT_F32, 0>;
    for (int rep = 0; rep < 2; rep++) {
      timeit("y, grid 8192", [&] { hipLaunchKernelGGL(k0, dim3(8192), dim3(256), shy, 0, sp); });
      timeit("y, grid 8192, ABL: no shared round", [&] { hipLaunchKernelGGL(k1, dim3(8192), dim3(256), shy, 0, sp); });
    }
  }
  std::vector<float> a((size_t)256 * P), b((size_t)256 * P);
  CK(hipMemcpy(a.data(), d_psd1, a.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), d_psd2, b.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int f = 0; f < 256; f++) {
    double mx = 0, md = 0;
    for (int k = 0; k < P; k++) { mx = fmax(mx, fabs(a[(size_t)f * P + k])); md = fmax(md, fabs(a[(size_t)f * P + k] - b[(size_t)f * P + k])); }
    if (f == 3 || f == 2) printf("frame %d: peak %.3e  max|d|/peak %.3e\n", f, mx, md / mx);
    worst = fmax(worst, md / mx);
  }
  printf("shared vs packed, first 256 frames: worst per-frame max|d|/peak = %.3e\n", worst);
  return 0;
}
