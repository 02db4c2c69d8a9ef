// what DPP wave_shr:1 (0x138) delivers on gfx950, once and twice
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *o) {
  int v = threadIdx.x + 100;
  int a = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true);
  int b = __builtin_amdgcn_update_dpp(0, a, 0x138, 0xF, 0xF, true);
  o[threadIdx.x] = a;
  o[64 + threadIdx.x] = b;
}
int main() {
  int *d, h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int r = 0; r < 2; ++r) { for (int i = 0; i < 64; ++i) printf("%d ", h[64 * r + i]); printf("\n"); }
  return 0;
}
