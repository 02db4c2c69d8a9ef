// Where does the dispatcher put N one-wave workgroups?  Prints the histogram of waves per SIMD and per CU
// (HW_ID / XCC_ID of every workgroup while all of them are resident).  hipcc --offload-arch=gfx950 placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(64) void k(unsigned *out, int spin) {
  extern __shared__ double sm[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  double a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0000001 + 1e-9;
  sm[threadIdx.x] = a;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc + (sm[0] == 12345.0); }
}
int main(int argc, char **argv) {
  int n = argc > 1 ? atoi(argv[1]) : 1024, lds = argc > 2 ? atoi(argv[2]) : 19840;
  unsigned *d; hipMalloc(&d, n * 8);
  hipLaunchKernelGGL(k, dim3(n), dim3(64), lds, 0, d, 4000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(2 * n); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, int> simd, cu;
  for (int i = 0; i < n; ++i) {
    unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    unsigned s = (hw >> 4) & 3, c = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    unsigned cuid = (xcc << 12) | (se << 8) | (sh << 4) | c;
    cu[cuid]++; simd[(cuid << 2) | s]++;
  }
  std::map<int, int> hs, hc;
  for (auto &p : simd) hs[p.second]++;
  for (auto &p : cu) hc[p.second]++;
  printf("n=%d lds=%d: %zu CUs, %zu SIMDs used\n waves/CU histogram:", n, lds, cu.size(), simd.size());
  for (auto &p : hc) printf(" %dx%d", p.first, p.second);
  printf("\n waves/SIMD histogram:");
  for (auto &p : hs) printf(" %dx%d", p.first, p.second);
  printf("\n first 16 WGs (xcc,se,sh,cu,simd):");
  for (int i = 0; i < 16; ++i) { unsigned hw = h[2*i]; printf(" (%u,%u,%u,%u,%u)", h[2*i+1]&0xf, (hw>>13)&7, (hw>>12)&1, (hw>>8)&0xf, (hw>>4)&3); }
  printf("\n");
  return 0;
}
