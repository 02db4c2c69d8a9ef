// Host cost of feeding two independent kernel chains (two streams / two graph branches) with ~8 us kernels:
// alternating launch by launch, in blocks of B launches per stream, from one graph with two branches.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void spin_kernel(double *p, int cycles) {
  long long m0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - m0 < cycles) {}
  if (p == (double *)1) p[0] = 0;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s[2]; CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
  const int K = 2000;
  for (int cyc : {2000, 20000, 40000}) {
    for (int blockn : {1, 4, 16, 64}) {
      CK(hipDeviceSynchronize());
      double t0 = now_us();
      for (int k = 0; k < K; k += blockn)
        for (int c = 0; c < 2; ++c)
          for (int j = 0; j < blockn; ++j) hipLaunchKernelGGL(spin_kernel, dim3(512), dim3(128), 34176, s[c], nullptr, cyc);
      double t1 = now_us();
      CK(hipDeviceSynchronize());
      double t2 = now_us();
      printf("spin %5d cycles (%.1f us)  2 streams x %d launches in blocks of %2d: host enqueue %.2f us/launch, total %.2f us per step (2 launches)\n",
             cyc, cyc / 2400.0, K, blockn, (t1 - t0) / (2 * K), (t2 - t0) / K);
    }
    CK(hipDeviceSynchronize());
    double t0 = now_us();
    for (int k = 0; k < K; ++k) hipLaunchKernelGGL(spin_kernel, dim3(1024), dim3(128), 34176, s[0], nullptr, cyc);
    double t1 = now_us();
    CK(hipDeviceSynchronize());
    double t2 = now_us();
    printf("spin %5d cycles  1 stream x %d launches of the whole grid: host enqueue %.2f us/launch, total %.2f us per step\n", cyc, K,
           (t1 - t0) / K, (t2 - t0) / K);
    // one graph, two branches
    hipGraph_t g; hipGraphExec_t ge; hipEvent_t f, j;
    CK(hipEventCreateWithFlags(&f, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
    CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeGlobal));
    CK(hipEventRecord(f, s[0])); CK(hipStreamWaitEvent(s[1], f, 0));
    for (int k = 0; k < 500; ++k)
      for (int c = 0; c < 2; ++c) hipLaunchKernelGGL(spin_kernel, dim3(512), dim3(128), 34176, s[c], nullptr, cyc);
    CK(hipEventRecord(j, s[1])); CK(hipStreamWaitEvent(s[0], j, 0));
    CK(hipStreamEndCapture(s[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s[0])); CK(hipStreamSynchronize(s[0]));
    t0 = now_us();
    CK(hipGraphLaunch(ge, s[0])); CK(hipStreamSynchronize(s[0]));
    t2 = now_us();
    printf("spin %5d cycles  graph with 2 branches x 500: total %.2f us per step (2 kernels)\n", cyc, (t2 - t0) / 500);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
