// Micro-measurements behind DESIGN §2a: (1) the rate of s_memtime (the phase-stamp clock) against the 100 MHz
// wall_clock64(); (2) the launch-to-launch period of dependent kernels in a hipGraph for an empty kernel, a kernel
// that only touches its LDS allocation, and one that spins a fixed number of shader cycles, at the headline grid
// (1024 workgroups x 128 threads, 34 KB LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void clock_rate(long long *out, int spin) {
  long long w0 = wall_clock64();
  long long m0 = __builtin_amdgcn_s_memtime();
  long long c0 = clock64();
  while (__builtin_amdgcn_s_memtime() - m0 < spin) {}
  long long w1 = wall_clock64();
  long long m1 = __builtin_amdgcn_s_memtime();
  long long c1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = w1 - w0; out[1] = m1 - m0; out[2] = c1 - c0; }
}
__global__ void empty_kernel(double *p) { if (p == (double *)1) p[0] = 0; }
__global__ void lds_kernel(double *p) {
  extern __shared__ double sh[];
  sh[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (sh[(threadIdx.x + 1) & 127] < 0) p[0] = 0;
}
__global__ void spin_kernel(double *p, int cycles) {
  long long m0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - m0 < cycles) {}
  if (p == (double *)1) p[0] = 0;
}

template <class F> static int graph_period(const char *name, F launch, int K) {
  hipStream_t s; CK(hipStreamCreate(&s));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < K; ++i) launch(s);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("%-44s %8.3f us per kernel (graph of %d)\n", name, 1000.0 * ms / K, K);
  // eager, same stream
  for (int i = 0; i < 50; ++i) launch(s);
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(a, s)); for (int i = 0; i < K; ++i) launch(s); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
  CK(hipEventElapsedTime(&ms, a, b));
  printf("%-44s %8.3f us per kernel (eager x %d)\n", name, 1000.0 * ms / K, K);
  return 0;
}

int main() {
  long long *d; CK(hipMalloc(&d, 64));
  for (int spin : {100000, 1000000}) {
    hipLaunchKernelGGL(clock_rate, dim3(1), dim3(64), 0, 0, d, spin);
    long long h[3]; CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
    printf("spin %d s_memtime ticks: wall_clock64 (100 MHz) %lld -> %.2f us; s_memtime %.1f MHz; clock64 %.1f MHz\n", spin, h[0],
           h[0] / 100.0, h[1] / (h[0] / 100.0), h[2] / (h[0] / 100.0));
  }
  double *p = nullptr;
  CK(hipFuncSetAttribute((const void *)lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  const int K = 500;
  graph_period("empty 1024 x 128", [&](hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(128), 0, s, p); }, K);
  graph_period("empty 1024 x 64", [&](hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(64), 0, s, p); }, K);
  graph_period("LDS 34 KB 1024 x 128", [&](hipStream_t s) { hipLaunchKernelGGL(lds_kernel, dim3(1024), dim3(128), 34176, s, p); }, K);
  graph_period("LDS 34 KB 2048 x 128", [&](hipStream_t s) { hipLaunchKernelGGL(lds_kernel, dim3(2048), dim3(128), 34176, s, p); }, K);
  for (int cyc : {1000, 2000, 4000}) {
    char nm[64]; snprintf(nm, 64, "spin %d s_memtime ticks 1024 x 128", cyc);
    graph_period(nm, [&](hipStream_t s) { hipLaunchKernelGGL(spin_kernel, dim3(1024), dim3(128), 0, s, p, cyc); }, K);
  }
  return 0;
}
