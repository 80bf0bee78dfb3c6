// stampbench.hip -- where does a round go?  Runs spectro16xl_kernel (C3: N=4096, 5 tapers) with
// GLFER_STAMP recording the shader clock of wave 0 of block 0 at every phase boundary of its
// full rounds, and prints the average duration of each phase in steady state.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-slp-vectorize -Iglfer_amd/csrc tools/stampbench.hip \
//         glfer_amd/csrc/host_tables.cpp -o tools/bin/stampbench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ unsigned long long *g_stamps;          // [rounds][16]
__device__ int g_round;
#define GLFER_STAMP(id)                                                                              \
  do {                                                                                               \
    if (blockIdx.x == 8 && threadIdx.x == 0) {                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      const unsigned long long c_ = __builtin_amdgcn_s_memtime();                                    \
      if ((id) == 0) g_round++;                                                                      \
      if (g_round > 0 && g_round <= 64) g_stamps[(g_round - 1) * 16 + (id)] = c_;                    \
      __builtin_amdgcn_sched_barrier(0);                                                             \
    }                                                                                                \
  } while (0)
#define GLFER_NO_LAUNCHERS
#ifdef STAMP_Y
#include "spectro16y.hip"
#else
#include "spectro16xl.hip"
#endif
#include "host_tables.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
  const int nframes = argc > 1 ? atoi(argv[1]) : 262144;
  constexpr int LOGN = 12, N = 1 << LOGN, H = N, P = N / 2 + 1, T = 5, NP = 2, TT = N / 16;
  std::vector<double> tapers((size_t)T * N), sig(T);
  if (!glfer::make_dpss(N, T - 1, 2.5, tapers.data(), sig.data())) { printf("dpss failed\n"); return 1; }
  std::vector<float> lt((size_t)NP * 8 * TT * 2 + 8 * TT);
  for (int j = 0; j < T; j++) {
    const bool last = j == T - 1;
    const double sc = sqrt(1.0 / ((last ? 4.0 : 2.0) * N * (1.0 + sig[j])));
    for (int m = 0; m < 8; m++)
      for (int t = 0; t < TT; t++) {
        const float v = (float)(tapers[(size_t)j * N + t + TT * m] * sc);
        if (last) lt[(size_t)NP * 8 * TT * 2 + (size_t)m * TT + t] = v;
        else lt[(((size_t)(j / 2) * 8 + m) * TT + t) * 2 + (j & 1)] = v;
      }
  }
  std::vector<float> tw((size_t)2 * glfer::make_twiddles16(LOGN, nullptr) * TT);
  glfer::make_twiddles16(LOGN, tw.data());
  const size_t ns = (size_t)nframes * H;
  std::vector<float> x(ns);
  unsigned s = 12345;
  for (size_t i = 0; i < ns; i++) { s = s * 1664525u + 1013904223u; x[i] = (float)((s >> 8) * (1.0 / 16777216.0) - 0.5); }
  float *d_x, *d_lt, *d_psd;
  float2 *d_tw;
  unsigned long long *d_st;
  CK(hipMalloc((void **)&d_x, ns * 4));
  CK(hipMalloc((void **)&d_lt, lt.size() * 4));
  CK(hipMalloc((void **)&d_tw, tw.size() * 4));
  CK(hipMalloc((void **)&d_psd, (size_t)nframes * P * 4));
  CK(hipMalloc((void **)&d_st, 64 * 16 * 8));
  CK(hipMemset(d_st, 0, 64 * 16 * 8));
  CK(hipMemcpy(d_x, x.data(), ns * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_lt, lt.data(), lt.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof d_st));
  SpectroParams sp = {};
  sp.stream = d_x; sp.nframes = nframes; sp.H = H; sp.R = 0; sp.npairs = NP + 1; sp.fmt = GLFER_FMT_F32;
  sp.tw = d_tw; sp.ltaps = d_lt; sp.psd = d_psd;
#ifdef STAMP_Y
  // spectro16y reads the pair tables ([pair][m/2][lane][4]) and the last taper ([m/4][lane][4]) from memory
  std::vector<float> gt((size_t)2 * (NP + 1) * N, 0.0f), xt(N);
  for (int j = 0; j < T; j++)
    for (int i = 0; i < N; i++) {
      const int t = i % TT, m = i / TT;
      gt[(size_t)(j / 2) * N * 2 + ((size_t)(m / 2) * TT + t) * 4 + (size_t)(m & 1) * 2 + (j & 1)] =
          (float)(tapers[(size_t)j * N + i] * sqrt(1.0 / (2.0 * N * (1.0 + sig[j]))));
    }
  for (int i = 0; i < N; i++) {
    const int t = i % TT, m = i / TT;
    xt[((size_t)(m / 4) * TT + t) * 4 + (size_t)(m & 3)] = (float)(tapers[(size_t)(T - 1) * N + i] * sqrt(1.0 / (4.0 * N * (1.0 + sig[T - 1]))));
  }
  float *d_gt, *d_xt;
  CK(hipMalloc((void **)&d_gt, gt.size() * 4));
  CK(hipMalloc((void **)&d_xt, xt.size() * 4));
  CK(hipMemcpy(d_gt, gt.data(), gt.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_xt, xt.data(), xt.size() * 4, hipMemcpyHostToDevice));
  sp.taps = d_gt; sp.xtaps = d_xt;
  const size_t shmem = (size_t)glfer::LaunchY::LDS_WORDS * 8;
  auto kern = glfer::spectro16y_kernel<GLFER_FMT_F32>;
  const char *names[16] = {"round start", "A: form z + pass0 (+writes)", "B: pass0 (+writes, prefetch)", "post-write barrier 0", "reads A+B, barrier",
                           "A: twiddle + pass1 (+writes)", "B: pass1 (+writes)", "post-write barrier 1", "reads A+B, barrier",
                           "A: twiddle + pass2", "", "", "", "", "", "B: pass2, accumulate both"};
  const int NORD = 10;
  const int order[11] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15};
#else
  const size_t shmem = glfer::LaunchXL<LOGN>::lds_bytes(NP);
  auto kern = glfer::spectro16xl_kernel<LOGN, GLFER_FMT_F32>;
  const char *names[16] = {"round start", "form z + pass0 butterflies", "pre-write barrier 0", "writes 0 issued", "post-write barrier 0",
                           "pass1 (reads+twiddle+bfly)", "pre-write barrier 1", "writes 1 issued", "post-write barrier 1",
                           "pass2 (reads+twiddle+bfly)", "", "", "", "", "", "accumulate / separate+store"};
  const int NORD = 10;
  const int order[11] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15};
#endif
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; rep++) {
    int zero = 0;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_round), &zero, sizeof zero));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(2048), dim3(256), shmem, 0, sp);
    CK(hipGetLastError());
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("launch %d: %.3f ms  %.1f Mframes/s (stamped build)\n", rep, ms, nframes / (ms * 1e-3) / 1e6);
  }
  std::vector<unsigned long long> st(64 * 16);
  CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
#ifdef STAMP_Y
  const int period = 3, nkinds = 2;                  // dual, dual, shared
  const char *kind_name[2] = {"dual round (two frames interleaved)", "shared round (one transform for both frames' last taper)"};
  const char *names_s[16] = {"round start", "form z + pass0 butterflies", "(barrier slot unused)", "writes 0 issued", "post-write barrier 0",
                             "reads, barrier, pass1", "(barrier slot unused)", "writes 1 issued", "post-write barrier 1",
                             "reads, barrier, pass2", "", "", "", "", "", "separate + store"};
#else
  const int period = 1, nkinds = 1;
  const char *kind_name[1] = {"round"};
#endif
  for (int kind = 0; kind < nkinds; kind++) {
    double sum[16] = {0};
    int cnt = 0;
    for (int r = 9; r < 57; r++) {                    // steady state
      const bool is_shared = period == 3 && r % 3 == 2;
      if ((kind == 1) != is_shared) continue;
      bool ok = true;
      for (int i = 0; i <= NORD; i++) ok = ok && st[r * 16 + order[i]] != 0;
      if (!ok) continue;
      for (int i = 1; i <= NORD; i++) sum[order[i]] += (double)(st[r * 16 + order[i]] - st[r * 16 + order[i - 1]]);
      cnt++;
    }
    if (!cnt) continue;
    double tot = 0;
    for (int i = 1; i <= NORD; i++) tot += sum[order[i]] / cnt;
    printf("wave 0 of block 8, %s: %d averaged; s_memtime ticks.\n", kind_name[kind], cnt);
    for (int i = 1; i <= NORD; i++) {
#ifdef STAMP_Y
      const char *nm = kind == 1 ? names_s[order[i]] : names[order[i]];
#else
      const char *nm = names[order[i]];
#endif
      printf("  -> %-30s %9.1f  (%4.1f%%)\n", nm, sum[order[i]] / cnt, 100.0 * sum[order[i]] / cnt / tot);
    }
    printf("  total %.1f ticks\n", tot);
  }
  printf("NOTE: every stamp costs a memory round trip of its own (a few hundred ticks); stamp 0 more.\n");
  return 0;
}
