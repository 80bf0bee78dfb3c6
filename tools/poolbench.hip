// poolbench.hip -- what a stream-ordered allocation of S bytes costs per call (malloc + free + sync),
// with the default pool's release threshold at its default (0) and raised.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void touch(char *p, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i * 4096 < n) p[i * 4096] = 1; }
int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
      hipMemPool_t pool;
      uint64_t keep = (uint64_t)8 << 30;
      CK(hipDeviceGetDefaultMemPool(&pool, 0));
      CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
      uint64_t got = 0;
      CK(hipMemPoolGetAttribute(pool, hipMemPoolAttrReleaseThreshold, &got));
      printf("release threshold now %llu\n", (unsigned long long)got);
    }
    for (size_t mb : {1, 64, 512, 2048}) {
      const size_t bytes = mb << 20;
      double tot = 0;
      for (int it = 0; it < 6; it++) {
        auto t0 = std::chrono::steady_clock::now();
        char *p = nullptr;
        CK(hipMallocAsync((void **)&p, bytes, st));
        hipLaunchKernelGGL(touch, dim3((unsigned)((bytes / 4096 + 255) / 256)), dim3(256), 0, st, p, bytes);
        CK(hipFreeAsync(p, st));
        CK(hipStreamSynchronize(st));
        auto t1 = std::chrono::steady_clock::now();
        if (it) tot += std::chrono::duration<double, std::micro>(t1 - t0).count();
      }
      printf("pass %d (%s)  %5zu MiB: %9.1f us per malloc+touch+free+sync\n", pass, pass ? "threshold 8 GiB" : "default", mb, tot / 5);
    }
  }
  return 0;
}
