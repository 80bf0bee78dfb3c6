// rdbench.hip -- how fast can rows of 2049 floats (8196 B, 4-byte aligned only) be streamed out of
// HBM into registers, one row per wavefront, as the per-column kernels (compute_floor, update_avg,
// the display map) do?  Variants of the load shape; every variant reduces the row to its maximum so
// that the loads stay live.  hipcc --offload-arch=gfx950 -O3 -o tools/bin/rdbench tools/rdbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// V0: dword per lane, lane + 64 j (the product's shape)
template <int WPB, int AUX>
__global__ __launch_bounds__(64 * WPB) void rd_dword(const float *psd, long long rows, int bins, float *out) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (r >= rows) return;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(psd + (size_t)r * bins), 0, bins * 4, 0x00020000);
  float v[33];
#pragma unroll
  for (int j = 0; j < 33; j++) v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, j * 256, AUX));
  float m = 0.0f;
#pragma unroll
  for (int j = 0; j < 33; j++) m = fmaxf(m, v[j]);
  m = wave_max(m);
  if (lane == 0) out[r] = m;
}

// V1: dwordx4 per lane (row start is 4-byte aligned only), 4 lane + 256 j; the descriptor's range check drops the overhang
template <int WPB, int AUX>
__global__ __launch_bounds__(64 * WPB) void rd_x4(const float *psd, long long rows, int bins, float *out) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (r >= rows) return;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(psd + (size_t)r * bins), 0, bins * 4, 0x00020000);
  v4f v[9];
#pragma unroll
  for (int j = 0; j < 9; j++) v[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, AUX));
  float m = 0.0f;
#pragma unroll
  for (int j = 0; j < 9; j++) m = fmaxf(fmaxf(m, fmaxf(v[j].x, v[j].y)), fmaxf(v[j].z, v[j].w));
  m = wave_max(m);
  if (lane == 0) out[r] = m;
}

// V2: dwordx4 from the 16-byte aligned address at or below the row start (the whole batch is one
// aligned array): lane reads aligned quads; the first and last quad hold neighbours' bins, masked
template <int WPB, int AUX>
__global__ __launch_bounds__(64 * WPB) void rd_x4a(const float *psd, long long rows, int bins, float *out) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (r >= rows) return;
  const size_t first = (size_t)r * bins;                 // element index of the row's first bin
  const size_t base = first & ~(size_t)3;                // aligned quad at or below it
  const int skip = (int)(first - base);                  // 0..3 foreign elements in front
  const size_t total = (size_t)rows * bins;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(psd + base), 0,
      (unsigned)((total - base < (size_t)(bins + 4) ? total - base : (size_t)(bins + 4)) * 4), 0x00020000);
  v4f v[9];
#pragma unroll
  for (int j = 0; j < 9; j++) v[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, AUX));
  float m = 0.0f;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    const int e = 4 * lane + 256 * j - skip;             // row index of component x
    m = fmaxf(m, (e >= 0 && e < bins) ? v[j].x : 0.0f);
    m = fmaxf(m, (e + 1 >= 0 && e + 1 < bins) ? v[j].y : 0.0f);
    m = fmaxf(m, (e + 2 >= 0 && e + 2 < bins) ? v[j].z : 0.0f);
    m = fmaxf(m, (e + 3 >= 0 && e + 3 < bins) ? v[j].w : 0.0f);
  }
  m = wave_max(m);
  if (lane == 0) out[r] = m;
}

// V3: flat copy-like read, no rows at all: the ceiling for a grid of this shape
template <int AUX>
__global__ __launch_bounds__(256) void rd_flat(const float *psd, size_t nquads, float *out) {
  const size_t stride = (size_t)gridDim.x * 256;
  float m = 0.0f;
  const v4f *q = reinterpret_cast<const v4f *>(psd);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nquads; i += stride * 4) {
    v4f a = __builtin_nontemporal_load(q + i);
    v4f b = i + stride < nquads ? __builtin_nontemporal_load(q + i + stride) : v4f{0, 0, 0, 0};
    v4f c = i + 2 * stride < nquads ? __builtin_nontemporal_load(q + i + 2 * stride) : v4f{0, 0, 0, 0};
    v4f d = i + 3 * stride < nquads ? __builtin_nontemporal_load(q + i + 3 * stride) : v4f{0, 0, 0, 0};
    m = fmaxf(m, fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
    m = fmaxf(m, fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
    m = fmaxf(m, fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w)));
    m = fmaxf(m, fmaxf(fmaxf(d.x, d.y), fmaxf(d.z, d.w)));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int *>(out), __float_as_int(m));
}

template <class F>
static void timeit(const char *name, size_t bytes, F launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0.0f;
  const int reps = 6;
  for (int i = 0; i < reps; i++) {
    CK(hipEventRecord(a));
    launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best;
    sum += ms;
  }
  printf("%-44s avg %.3f ms  %6.0f GB/s   best %.3f ms  %6.0f GB/s\n", name, sum / reps, bytes / (sum / reps * 1e-3) / 1e9, best,
         bytes / (best * 1e-3) / 1e9);
}

int main() {
  const long long rows = 131072;
  const int bins = 2049;
  const size_t n = (size_t)rows * bins;
  float *psd, *out;
  CK(hipMalloc(&psd, n * 4 + 64));
  CK(hipMalloc(&out, rows * 4));
  std::vector<float> h(n);
  for (size_t i = 0; i < n; i++) h[i] = (float)((i * 2654435761u) & 0xffff) * (1.0f / 65536.0f);
  CK(hipMemcpy(psd, h.data(), n * 4, hipMemcpyHostToDevice));
  const size_t bytes = n * 4;
#define RUN(K, WPB, ...) timeit(#K " wpb=" #WPB, bytes, [&] { hipLaunchKernelGGL(K, dim3((unsigned)((rows + WPB - 1) / WPB)), dim3(64 * WPB), 0, 0, psd, rows, bins, out); })
  RUN((rd_dword<4, 0>), 4);
  RUN((rd_dword<4, 2>), 4);
  RUN((rd_dword<1, 0>), 1);
  RUN((rd_dword<2, 0>), 2);
  RUN((rd_dword<8, 0>), 8);
  RUN((rd_x4<4, 0>), 4);
  RUN((rd_x4<4, 2>), 4);
  RUN((rd_x4<1, 0>), 1);
  RUN((rd_x4<8, 0>), 8);
  RUN((rd_x4a<4, 0>), 4);
  RUN((rd_x4a<4, 2>), 4);
  RUN((rd_x4a<1, 0>), 1);
  for (int g : {1024, 2048, 4096, 8192, 16384})
    timeit(("rd_flat grid " + std::to_string(g)).c_str(), bytes, [&] { hipLaunchKernelGGL(rd_flat<0>, dim3(g), dim3(256), 0, 0, psd, n / 4, out); });
  return 0;
}
