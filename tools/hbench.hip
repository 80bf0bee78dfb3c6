// hbench.hip -- periodogram (single taper) kernel comparison at BASELINE config 2 (N=4096,
// 75 % overlap): the packed N-point kernel (spectro16.hip, npairs = 1) against the real-input
// N/2-point kernel (spectro16h.hip) and its build variants.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-slp-vectorize -Iglfer_amd/csrc tools/hbench.hip \
//         glfer_amd/csrc/host_tables.cpp -o /tmp/hbench && /tmp/hbench [nframes]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define GLFER_NO_LAUNCHERS
#include "spectro16.hip"
#include "spectro16h.hip"
#include "host_tables.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
  const int nframes = argc > 1 ? atoi(argv[1]) : 1 << 20;
  constexpr int LOGN = 12, N = 1 << LOGN, H = N / 4, P = N / 2 + 1, TH = N / 32;
  std::vector<float> win(N);
  glfer::make_window(0, N, win.data());
  // packed form: taps[0][m/2][t][4], second taper zero
  std::vector<float> taps((size_t)2 * N, 0.0f), htaps(N), hrot(2 * TH);
  for (int i = 0; i < N; i++) {
    const int T = N / 16, t = i % T, m = i / T;
    taps[((size_t)(m / 2) * T + t) * 4 + (size_t)(m & 1) * 2] = (float)(win[i] * sqrt(1.0 / (2.0 * N)));
  }
  for (int m = 0; m < 16; m++)
    for (int t = 0; t < TH; t++)
      for (int e = 0; e < 2; e++)
        htaps[((size_t)(m / 2) * TH + t) * 4 + (size_t)(m & 1) * 2 + e] = (float)(win[2 * (t + TH * m) + e] * sqrt(1.0 / (4.0 * N)));
  for (int t = 0; t < TH; t++) { hrot[2 * t] = (float)cos(2 * M_PI * t / N); hrot[2 * t + 1] = (float)sin(2 * M_PI * t / N); }
  std::vector<float> tw((size_t)2 * glfer::make_twiddles16(LOGN, nullptr) * (N / 16)), htw((size_t)2 * glfer::make_twiddles16(LOGN - 1, nullptr) * TH);
  glfer::make_twiddles16(LOGN, tw.data());
  glfer::make_twiddles16(LOGN - 1, htw.data());
  // the real-input kernel takes only frames that lie wholly inside the stream (the library's
  // launcher sends the first R/H frames to the packed kernel): start at frame R/H = 3
  constexpr int F0 = (N - H) / H;
  const size_t ns = (size_t)(nframes + F0) * H;
  std::vector<float> x(ns);
  unsigned s = 12345;
  for (size_t i = 0; i < ns; i++) { s = s * 1664525u + 1013904223u; x[i] = (float)((s >> 8) * (1.0 / 16777216.0) - 0.5) + 0.3f * sinf(0.01f * (float)i); }
  float *d_x, *d_taps, *d_htaps, *d_psd1, *d_psd2;
  float2 *d_tw, *d_htw, *d_hrot;
  CK(hipMalloc((void **)&d_x, ns * 4));
  CK(hipMalloc((void **)&d_taps, taps.size() * 4));
  CK(hipMalloc((void **)&d_htaps, htaps.size() * 4));
  CK(hipMalloc((void **)&d_tw, tw.size() * 4));
  CK(hipMalloc((void **)&d_htw, htw.size() * 4));
  CK(hipMalloc((void **)&d_hrot, hrot.size() * 4));
  CK(hipMalloc((void **)&d_psd1, (size_t)nframes * P * 4));
  CK(hipMalloc((void **)&d_psd2, (size_t)nframes * P * 4));
  CK(hipMemcpy(d_x, x.data(), ns * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_taps, taps.data(), taps.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_htaps, htaps.data(), htaps.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_htw, htw.data(), htw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_hrot, hrot.data(), hrot.size() * 4, hipMemcpyHostToDevice));
  SpectroParams sp = {};
  sp.stream = d_x; sp.frame0 = F0; sp.nframes = nframes; sp.H = H; sp.R = N - H; sp.npairs = 1; sp.fmt = GLFER_FMT_F32;
  sp.taps = d_taps; sp.tw = d_tw; sp.htaps = d_htaps; sp.htw = d_htw; sp.hrot = d_hrot; sp.spec_unscale = 1.0f;
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
    printf("%-40s: %.3f ms  %.1f Mframes/s  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)\n", label, best, fps / 1e6,
           fps * (4.0 * H + 4.0 * P) / 1e9, fps * (4.0 * H + 4.0 * P) / 8e12 * 100);
    return 0;
  };
  sp.psd = d_psd1;
  timeit("packed N-point, WPS=3, grid 3072", [&] { hipLaunchKernelGGL((glfer::spectro16_kernel<LOGN, GLFER_FMT_F32, false, 3>), dim3(3072), dim3(256), 0, 0, sp); });
  sp.psd = d_psd2;
#define KH(WPS, VAR) glfer::spectro16h_kernel<LOGN, GLFER_FMT_F32, WPS, VAR>
  timeit("real-input N/2, WPS=3 var0, grid 3072", [&] { hipLaunchKernelGGL((KH(3, 0)), dim3(3072), dim3(256), 0, 0, sp); });
  timeit("real-input N/2, WPS=2 var0, grid 2048", [&] { hipLaunchKernelGGL((KH(2, 0)), dim3(2048), dim3(256), 0, 0, sp); });
  timeit("real-input N/2, WPS=3 var1, grid 3072", [&] { hipLaunchKernelGGL((KH(3, 1)), dim3(3072), dim3(256), 0, 0, sp); });
  timeit("real-input N/2, WPS=4 var1, grid 4096", [&] { hipLaunchKernelGGL((KH(4, 1)), dim3(4096), dim3(256), 0, 0, sp); });
  timeit("real-input N/2, WPS=3 var2, grid 3072", [&] { hipLaunchKernelGGL((KH(3, 2)), dim3(3072), dim3(256), 0, 0, sp); });
  timeit("real-input N/2, WPS=4 var2, grid 4096", [&] { hipLaunchKernelGGL((KH(4, 2)), dim3(4096), dim3(256), 0, 0, sp); });
  timeit("real-input N/2, WPS=3 var2, grid 768", [&] { hipLaunchKernelGGL((KH(3, 2)), dim3(768), dim3(256), 0, 0, sp); });
  timeit("real-input N/2, WPS=3 var2, grid 1536", [&] { hipLaunchKernelGGL((KH(3, 2)), dim3(1536), dim3(256), 0, 0, sp); });
  timeit("real-input N/2, WPS=3 var2, grid 6144", [&] { hipLaunchKernelGGL((KH(3, 2)), dim3(6144), dim3(256), 0, 0, sp); });
  timeit("real-input N/2, WPS=3 var2, grid 12288", [&] { hipLaunchKernelGGL((KH(3, 2)), dim3(12288), dim3(256), 0, 0, sp); });
  timeit("real-input N/2, WPS=3 var0 again", [&] { hipLaunchKernelGGL((KH(3, 0)), dim3(3072), dim3(256), 0, 0, sp); });
  std::vector<float> a((size_t)256 * P), b((size_t)256 * P);
  CK(hipMemcpy(a.data(), d_psd1, a.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), d_psd2, b.size() * 4, hipMemcpyDeviceToHost));
  double mx = 0, md = 0;
  for (size_t i = 0; i < a.size(); i++) { mx = fmax(mx, fabs(a[i])); md = fmax(md, fabs(a[i] - b[i])); }
  printf("real-input vs packed, first 256 frames: max|d|/max = %.3e\n", md / mx);
  return 0;
}
