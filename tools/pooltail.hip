// pooltail.hip -- the TAIL of a stream-ordered 2 GiB scratch per call (malloc + touch + free + sync, release threshold
// raised): how often does a call take far longer than the median, with (a) hipMallocAsync/hipFreeAsync per call and
// (b) a second small allocation in the same call, (c) one cached block reused (no allocation calls at all)?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void touch(char *p, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i * 4096 < n) p[i * 4096] = 1; }
int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipMemPool_t pool;
  uint64_t keep = (uint64_t)8 << 30;
  CK(hipDeviceGetDefaultMemPool(&pool, 0));
  CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
  const size_t bytes = (size_t)2200 << 20;
  char *cached = nullptr;
  CK(hipMalloc((void **)&cached, bytes));
  for (int mode = 0; mode < 3; mode++) {
    std::vector<double> ts;
    for (int it = 0; it < 400; it++) {
      auto t0 = std::chrono::steady_clock::now();
      char *p = cached, *q = nullptr;
      if (mode < 2) CK(hipMallocAsync((void **)&p, bytes + (size_t)(it % 2) * 65536, st));
      if (mode == 1) CK(hipMallocAsync((void **)&q, 2 << 20, st));
      hipLaunchKernelGGL(touch, dim3((unsigned)((bytes / 4096 + 255) / 256)), dim3(256), 0, st, p, bytes);
      if (mode == 1) CK(hipFreeAsync(q, st));
      if (mode < 2) CK(hipFreeAsync(p, st));
      CK(hipStreamSynchronize(st));
      auto t1 = std::chrono::steady_clock::now();
      if (it >= 3) ts.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::vector<double> s = ts;
    std::sort(s.begin(), s.end());
    const double med = s[s.size() / 2];
    int slow = 0;
    for (double t : ts) slow += t > 5 * med;
    printf("%s: median %.1f us, max %.1f us, calls over 5x the median: %d of %zu\n",
           mode == 0 ? "malloc/free per call" : (mode == 1 ? "malloc/free per call + a 2 MiB one" : "one cached block"), med, s.back(), slow, ts.size());
  }
  return 0;
}
