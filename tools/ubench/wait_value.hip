// hipStreamWaitValue64 as the exchange's wait (round 4): can the command processor follow a counter that a RUNNING kernel
// bumps with device atomics — without a wait kernel, i.e. without a CU, a wavefront slot or a register?
//   memory kinds for the counter: plain hipMalloc, hipExtMallocWithFlags(hipMallocSignalMemory), hipDeviceMallocFinegrained,
//   hipDeviceMallocUncached, pinned host memory
//   producer: a persistent kernel (1024 x 128 threads, 34 KB LDS: the shape of the two-wavefront step loop) that "steps" every
//   `cycles` shader cycles and lets thread 0 of every workgroup add 1 to the counter per step
//   consumer: a second stream with  wait(counter >= (k+1) * blocks) -> stamp kernel (100 MHz wall clock)  for every step k
// Reported: return codes, how late after the producer's own stamp of step k the consumer's stamp lands, and the producer's
// step period with and without the consumer beside it.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(128) void producer(unsigned long long *ctr, long long *stamps, int steps, int cycles, int scope_sys) {
  extern __shared__ double sm[];
  if (threadIdx.x == 0) sm[0] = 0;
  for (int k = 0; k < steps; ++k) {
    const long long m0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - m0 < cycles) {}
    __syncthreads();
    if (threadIdx.x == 64) {  // the helper wavefront's lane 0 counts the workgroup in
      if (scope_sys) __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (blockIdx.x == gridDim.x - 1) stamps[k] = (long long)__builtin_amdgcn_s_memrealtime();
    }
  }
}
__global__ void stamp(long long *out) {
  if (threadIdx.x == 0) *out = (long long)__builtin_amdgcn_s_memrealtime();
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t sp, sc;
  CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
  const int steps = 64, blocks = 1024;
  long long *pst, *cst;
  CK(hipMalloc(&pst, steps * 8));
  CK(hipMalloc(&cst, steps * 8));
  const char *names[] = {"hipMalloc", "signal memory", "fine-grained device", "uncached device", "pinned host"};
  for (int kind = 0; kind < 5; ++kind) {
    unsigned long long *ctr = nullptr;
    hipError_t e = hipSuccess;
    if (kind == 0) e = hipMalloc(&ctr, 64);
    if (kind == 1) e = hipExtMallocWithFlags((void **)&ctr, 8, hipMallocSignalMemory);
    if (kind == 2) e = hipExtMallocWithFlags((void **)&ctr, 64, hipDeviceMallocFinegrained);
    if (kind == 3) e = hipExtMallocWithFlags((void **)&ctr, 64, hipDeviceMallocUncached);
    if (kind == 4) e = hipHostMalloc((void **)&ctr, 64, hipHostMallocCoherent);
    if (e != hipSuccess) {
      printf("[%s] allocation failed: %s\n", names[kind], hipGetErrorString(e));
      (void)hipGetLastError();
      continue;
    }
    const int sys = (kind == 1 || kind == 4) ? 1 : 0;
    for (int cycles : {20000, 32000}) {
      for (int with_consumer = 0; with_consumer < 2; ++with_consumer) {
        if (kind == 1 || kind == 4) *ctr = 0; else CK(hipMemset(ctr, 0, 8));
        CK(hipMemset(pst, 0, steps * 8));
        CK(hipMemset(cst, 0, steps * 8));
        CK(hipDeviceSynchronize());
        const double t0 = now_us();
        hipLaunchKernelGGL(producer, dim3(blocks), dim3(128), 34176, sp, ctr, pst, steps, cycles, sys);
        bool ok = true;
        if (with_consumer) {
          for (int k = 0; k < steps; ++k) {
            e = hipStreamWaitValue64(sc, ctr, (unsigned long long)(k + 1) * blocks, hipStreamWaitValueGte, ~0ull);
            if (e != hipSuccess) {
              printf("[%s] hipStreamWaitValue64 refused: %s\n", names[kind], hipGetErrorString(e));
              (void)hipGetLastError();
              ok = false;
              break;
            }
            hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, sc, cst + k);
          }
        }
        const double t1 = now_us();
        CK(hipStreamSynchronize(sp));
        const double t2 = now_us();
        if (with_consumer && ok) {
          // bounded wait for the consumer (a wait that never fires must not hang the box)
          double tw = now_us();
          while (hipStreamQuery(sc) == hipErrorNotReady && now_us() - tw < 2e6) {}
          if (hipStreamQuery(sc) == hipErrorNotReady) {
            printf("[%s] consumer still waiting 2 s after the producer ended: the wait never fired (counter = %llu)\n", names[kind],
                   (kind == 1 || kind == 4) ? *ctr : 0ull);
            // release it: write the final value from the host side
            unsigned long long big = ~0ull >> 1;
            if (kind == 1 || kind == 4) *ctr = big; else (void)hipMemcpy(ctr, &big, 8, hipMemcpyHostToDevice);
            (void)hipStreamSynchronize(sc);
            continue;
          }
        }
        std::vector<long long> hp(steps), hc(steps);
        CK(hipMemcpy(hp.data(), pst, steps * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hc.data(), cst, steps * 8, hipMemcpyDeviceToHost));
        const double period = (hp[steps - 1] - hp[4]) / 100.0 / (steps - 5);
        double lag_sum = 0, lag_max = 0;
        int nl = 0;
        if (with_consumer && ok)
          for (int k = 4; k < steps; ++k) {
            const double lag = (hc[k] - hp[k]) / 100.0;
            lag_sum += lag;
            if (lag > lag_max) lag_max = lag;
            ++nl;
          }
        printf("[%s] spin %d cycles, consumer %d: producer period %.2f us/step, launch total %.1f us, host enqueue %.1f us", names[kind],
               cycles, with_consumer, period, t2 - t0, t1 - t0);
        if (nl) printf(", consumer stamp lags the last workgroup's count by %.2f us (mean) %.2f us (max)", lag_sum / nl, lag_max);
        printf("\n");
      }
    }
    if (kind == 4) (void)hipHostFree(ctr); else (void)hipFree(ctr);
  }
  return 0;
}
