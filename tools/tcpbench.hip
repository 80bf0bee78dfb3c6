// tcpbench.hip -- does a wavefront's L2-HIT load wait behind OTHER wavefronts' HBM-miss loads in the CU's vector L1 (TCP)?
// The multitaper kernels read their taper tables (L2 hits, needed at every round start) beside the sample stream (HBM misses).
// Workgroups of 4 wavefronts, 2 per CU: wavefronts 0-1 time single 16-byte loads from a 64 KB table (issue, s_waitcnt vmcnt(0),
// shader clock before and after), wavefronts 2-3 stream 16-byte loads (MODE 0: nothing; 1: from a 1 MB region = L2 hits;
// 2: from a 4 GB region = HBM misses), DEPTH loads in flight.  Prints the timed loads' mean / median / p90 latency in clocks.
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/tcpbench tools/tcpbench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void tcp_kernel(const float *table, const float *big, size_t big_quads, unsigned *lat, float *sink, int iters) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned gw = blockIdx.x * 4 + wv;
  if (wv < 2) {                                            // timed L2-hit loads
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(table), 0, 64 * 1024, 0x00020000);
    v4f acc = {0, 0, 0, 0};
    unsigned long long tot = 0;
    for (int i = 0; i < iters; i++) {
      const unsigned off = ((gw * 7919u + i * 104729u) & 63u) * 1024u + lane * 16u;
      __builtin_amdgcn_s_waitcnt(0);
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc += v;
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_waitcnt(0);
      const unsigned d = (unsigned)(t1 - t0);
      if (lane == 0 && i >= 8) lat[(size_t)gw * iters + i] = d;
      tot += d;
      // ~400 clocks of arithmetic between timed loads (a butterfly stage's worth)
      float w = acc.x;
#pragma unroll
      for (int k = 0; k < 100; k++) w = __builtin_fmaf(w, 1.0001f, 0.5f);
      acc.x = w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) sink[gw] = acc.x;
  } else if (MODE != 0) {                                  // streaming loads
    const size_t region = MODE == 1 ? (size_t)(1 << 16) : big_quads;     // quads
    const v4f *src = reinterpret_cast<const v4f *>(big);
    size_t pos = ((size_t)gw * 64 * DEPTH * 977) % region;
    v4f acc = {0, 0, 0, 0};
    for (int i = 0; i < iters * 2; i++) {
      v4f v[DEPTH];
#pragma unroll
      for (int j = 0; j < DEPTH; j++) v[j] = src[(pos + (size_t)j * 64 + lane) % region];
#pragma unroll
      for (int j = 0; j < DEPTH; j++) acc += v[j];
      pos = (pos + (size_t)64 * DEPTH * 4099) % region;
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) sink[gw] = acc.x;
  }
}

template <int MODE, int DEPTH>
static void run(const char *name, const float *table, const float *big, size_t big_quads, unsigned *lat, float *sink) {
  const int blocks = 512, iters = 256;
  CK(hipMemset(lat, 0, (size_t)blocks * 4 * iters * 4));
  hipLaunchKernelGGL((tcp_kernel<MODE, DEPTH>), dim3(blocks), dim3(256), 0, 0, table, big, big_quads, lat, sink, iters);
  CK(hipDeviceSynchronize());
  std::vector<unsigned> h((size_t)blocks * 4 * iters);
  CK(hipMemcpy(h.data(), lat, h.size() * 4, hipMemcpyDeviceToHost));
  std::vector<unsigned> v;
  for (unsigned x : h) if (x) v.push_back(x);
  std::sort(v.begin(), v.end());
  double s = 0;
  for (unsigned x : v) s += x;
  printf("%-46s timed loads %7zu  mean %7.0f  median %6u  p90 %6u  p99 %6u  (s_memtime ticks)\n", name, v.size(), s / v.size(), v[v.size() / 2],
         v[v.size() * 9 / 10], v[v.size() * 99 / 100]);
}

int main() {
  const size_t big_bytes = 4ULL << 30;
  float *table, *big, *sink;
  unsigned *lat;
  CK(hipMalloc(&table, 64 * 1024));
  CK(hipMalloc(&big, big_bytes));
  CK(hipMalloc(&sink, 1 << 20));
  CK(hipMalloc(&lat, 512ULL * 4 * 256 * 4));
  CK(hipMemset(table, 0, 64 * 1024));
  CK(hipMemset(big, 0, big_bytes));
  const size_t q = big_bytes / 16;
  for (int rep = 0; rep < 2; rep++) {
    run<0, 4>("neighbours idle", table, big, q, lat, sink);
    run<1, 4>("neighbours stream L2 hits, 4 in flight", table, big, q, lat, sink);
    run<2, 4>("neighbours stream HBM misses, 4 in flight", table, big, q, lat, sink);
    run<2, 16>("neighbours stream HBM misses, 16 in flight", table, big, q, lat, sink);
  }
  return 0;
}
