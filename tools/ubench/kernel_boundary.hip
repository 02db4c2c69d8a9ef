// What a kernel boundary costs at the headline grid (1024 workgroups x 128 threads, 34 KB LDS each) when every
// workgroup also moves the step kernel's records: spin S shader cycles; read R bytes; write W bytes (plain / nontemporal /
// write-through sc0 sc1 stores).  Period per kernel of a 500-kernel hipGraph minus the spin = boundary + memory cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void work(const double *__restrict__ in, double *__restrict__ out, int n_in, int n_out, int cycles) {
  extern __shared__ double sh[];
  long long m0 = __builtin_amdgcn_s_memtime();
  double acc = 0;
  const double *src = in + (size_t)blockIdx.x * n_in;
  for (int i = threadIdx.x; i < n_in; i += blockDim.x) acc += src[i];
  sh[threadIdx.x] = acc;
  while (__builtin_amdgcn_s_memtime() - m0 < cycles) {}
  double *dst = out + (size_t)blockIdx.x * n_out;
  for (int i = threadIdx.x; i < n_out; i += blockDim.x) {
    const double v = sh[threadIdx.x] + i;
    if (MODE == 0) dst[i] = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, &dst[i]);
    else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(&dst[i]), "v"(v) : "memory");
  }
}

template <class F> static float graph_period(F launch, int K) {
  hipStream_t s; hipStreamCreate(&s);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < K; ++i) launch(s, i);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  hipEventRecord(a, s); hipGraphLaunch(ge, s); hipEventRecord(b, s); hipStreamSynchronize(s);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(s);
  return 1000.0f * ms / K;
}

int main() {
  const int WG = 1024, K = 500;
  double *a, *b;
  CK(hipMalloc(&a, (size_t)WG * 4096 * 8)); CK(hipMalloc(&b, (size_t)WG * 4096 * 8));
  CK(hipMemset(a, 0, (size_t)WG * 4096 * 8)); CK(hipMemset(b, 0, (size_t)WG * 4096 * 8));
  const int cyc = 30000;
  printf("spin %d cycles (%.2f us at 2.25 GHz); per-kernel period in a %d-kernel graph, ping-pong buffers\n", cyc, cyc / 2250.0, K);
  struct { int rd, wr; } cases[] = {{0, 0}, {400, 0}, {0, 400}, {400, 400}, {400, 800}, {1600, 1600}};
  for (auto c : cases) {
    float t0 = graph_period([&](hipStream_t s, int i) { hipLaunchKernelGGL(work<0>, dim3(WG), dim3(128), 34176, s, (i & 1) ? b : a, (i & 1) ? a : b, c.rd, c.wr, cyc); }, K);
    float t1 = graph_period([&](hipStream_t s, int i) { hipLaunchKernelGGL(work<1>, dim3(WG), dim3(128), 34176, s, (i & 1) ? b : a, (i & 1) ? a : b, c.rd, c.wr, cyc); }, K);
    float t2 = graph_period([&](hipStream_t s, int i) { hipLaunchKernelGGL(work<2>, dim3(WG), dim3(128), 34176, s, (i & 1) ? b : a, (i & 1) ? a : b, c.rd, c.wr, cyc); }, K);
    printf("read %5d B/wg (%4.1f MB)  write %5d B/wg (%4.1f MB):  plain %6.2f us   nontemporal %6.2f us   sc0 sc1 %6.2f us\n", c.rd * 8,
           c.rd * 8.0 * WG / 1e6, c.wr * 8, c.wr * 8.0 * WG / 1e6, t0, t1, t2);
  }
  return 0;
}
