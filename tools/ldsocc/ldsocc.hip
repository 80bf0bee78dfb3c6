// Where do one-wavefront workgroups with a given dynamic LDS size land?  Each workgroup records its XCC / SE / CU / SIMD (HW_ID, XCC_ID
// registers), spins for a fixed time so that the whole grid is resident at once, and the host prints how many share a CU and a SIMD.
// build: hipcc --offload-arch=gfx950 -O2 -o ldsocc ldsocc.hip ; run: ./ldsocc <lds bytes> <workgroups>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void where(unsigned *out, long long spin) {
  extern __shared__ float lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  lds[threadIdx.x] = (float)hw;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
}

int main(int argc, char **argv) {
  const size_t lds = argc > 1 ? atol(argv[1]) : 22028;
  const int wgs = argc > 2 ? atoi(argv[2]) : 1792;
  hipFuncSetAttribute((const void *)where, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, where, 64, lds);
  unsigned *d;
  hipMalloc(&d, 8 * wgs);
  hipMemset(d, 0xff, 8 * wgs);
  hipLaunchKernelGGL(where, dim3(wgs), dim3(64), lds, 0, d, 20000000LL);   // 0.2 s at 100 MHz
  hipDeviceSynchronize();
  std::vector<unsigned> h(2 * wgs);
  hipMemcpy(h.data(), d, 8 * wgs, hipMemcpyDeviceToHost);
  std::map<unsigned, int> per_cu, per_simd;
  for (int i = 0; i < wgs; i++) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const unsigned cukey = xcc << 16 | se << 8 | sh << 4 | cu;
    per_cu[cukey]++;
    per_simd[cukey << 2 | simd]++;
  }
  std::map<int, int> hist_cu, hist_simd;
  for (auto &kv : per_cu) hist_cu[kv.second]++;
  for (auto &kv : per_simd) hist_simd[kv.second]++;
  printf("lds %zu B, %d workgroups, occupancy API says %d per CU; CUs seen %zu, SIMDs seen %zu\n", lds, wgs, occ, per_cu.size(), per_simd.size());
  printf("  workgroups per CU  :");
  for (auto &kv : hist_cu) printf("  %d x%d", kv.first, kv.second);
  printf("\n  workgroups per SIMD:");
  for (auto &kv : hist_simd) printf("  %d x%d", kv.first, kv.second);
  printf("\n");
  return 0;
}
