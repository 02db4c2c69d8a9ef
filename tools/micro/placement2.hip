// Where do the TWO wavefronts of N two-wave workgroups land?  Per SIMD: how many wave-0s (the step kernel's main wavefronts) and
// wave-1s (its helper wavefronts) it hosts while all workgroups are resident.  hipcc --offload-arch=gfx950 placement2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(128) void k(unsigned *out, int spin) {
  extern __shared__ double sm[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  double a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0000001 + 1e-9;
  sm[threadIdx.x] = a;
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { out[4 * blockIdx.x + 2 * wv] = hw; out[4 * blockIdx.x + 2 * wv + 1] = xcc + (sm[0] == 12345.0); }
}
int main(int argc, char **argv) {
  int n = argc > 1 ? atoi(argv[1]) : 1024, lds = argc > 2 ? atoi(argv[2]) : 34816;
  unsigned *d; (void)hipMalloc(&d, n * 16);
  (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(k, dim3(n), dim3(128), lds, 0, d, 20000);
  (void)hipDeviceSynchronize();
  std::vector<unsigned> h(4 * n); (void)hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
  std::map<unsigned, int> m0, m1;  // SIMD -> number of wave-0s / wave-1s
  int same_simd = 0, same_cu = 0;
  for (int i = 0; i < n; ++i) {
    unsigned id[2];
    for (int w = 0; w < 2; ++w) {
      unsigned hw = h[4 * i + 2 * w], xcc = h[4 * i + 2 * w + 1] & 0xf;
      unsigned s = (hw >> 4) & 3, c = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      id[w] = ((((xcc << 12) | (se << 8) | (sh << 4) | c)) << 2) | s;
      (w == 0 ? m0 : m1)[id[w]]++;
    }
    same_simd += id[0] == id[1];
    same_cu += (id[0] >> 2) == (id[1] >> 2);
  }
  std::map<std::pair<int, int>, int> hist;
  std::map<unsigned, int> all;
  for (auto &p : m0) all[p.first] = 0;
  for (auto &p : m1) all[p.first] = 0;
  for (auto &p : all) hist[{m0.count(p.first) ? m0[p.first] : 0, m1.count(p.first) ? m1[p.first] : 0}]++;
  printf("n=%d two-wave workgroups, lds=%d: %zu SIMDs used; both wavefronts of a workgroup on one CU: %d, on one SIMD: %d\n", n, lds,
         all.size(), same_cu, same_simd);
  printf(" SIMDs by (wave-0s, wave-1s) hosted:");
  for (auto &p : hist) printf("  (%d,%d) x %d", p.first.first, p.first.second, p.second);
  printf("\n first 8 WGs (simd of wave 0, simd of wave 1 within the CU):");
  for (int i = 0; i < 8; ++i) printf(" (%u,%u)", (h[4 * i] >> 4) & 3, (h[4 * i + 2] >> 4) & 3);
  printf("\n");
  return 0;
}
