// v_mfma_f64_4x4x4_4b_f64 on gfx950: operand / result lane maps (probed with one-hot A and power-of-two B) and the
// issue cost of dependent and independent chains.  Result: see profiles/r02d_ubench_mfma_f64_4x4x4.txt.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void probe(double *out) {  // out[la][half][lane]
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int half = 0; half < 2; ++half) {
      const double a = lane == la ? 1.0 : 0.0;
      const double b = (lane / 32 == half) ? ldexp(1.0, lane % 32) : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      out[(la * 2 + half) * 64 + lane] = d;
    }
}
__global__ void timing(long long *out, double *sink) {
  const int lane = threadIdx.x;
  double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < 256; ++i) {  // dependent chain
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < 256; ++i) {  // four independent accumulators
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
  }
  long long t2 = __builtin_amdgcn_s_memtime();
  double f0 = a, f1 = b, f2 = a + b, f3 = a - b;
#pragma unroll 1
  for (int i = 0; i < 256; ++i) {  // dependent v_fma_f64 chain
    f0 = fma(f0, a, b); f0 = fma(f0, a, b); f0 = fma(f0, a, b); f0 = fma(f0, a, b);
  }
  long long t3 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < 256; ++i) {  // four independent v_fma_f64 chains
    f0 = fma(f0, a, b); f1 = fma(f1, a, b); f2 = fma(f2, a, b); f3 = fma(f3, a, b);
  }
  long long t4 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; }
  sink[lane] = c0 + c1 + c2 + c3 + f0 + f1 + f2 + f3;
}
int main() {
  double *d; CK(hipMalloc(&d, 64 * 2 * 64 * 8));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  std::vector<double> h(64 * 2 * 64);
  CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
  // pairs[la] = list of (lb, out lane)
  printf("A lane la multiplies B lane lb into D lane ld:\n");
  for (int la = 0; la < 64; ++la) {
    printf("la %2d:", la);
    for (int half = 0; half < 2; ++half)
      for (int ld = 0; ld < 64; ++ld) {
        double v = h[(la * 2 + half) * 64 + ld];
        if (v == 0) continue;
        unsigned long long bits = (unsigned long long)v;
        for (int k = 0; k < 32; ++k)
          if (bits >> k & 1) printf(" (lb %2d -> ld %2d)", half * 32 + k, ld);
      }
    printf("\n");
  }
  long long *t; double *sink; CK(hipMalloc(&t, 64)); CK(hipMalloc(&sink, 512));
  hipLaunchKernelGGL(timing, dim3(1), dim3(64), 0, 0, t, sink);
  hipLaunchKernelGGL(timing, dim3(1), dim3(64), 0, 0, t, sink);
  long long ht[4]; CK(hipMemcpy(ht, t, 32, hipMemcpyDeviceToHost));
  printf("cycles per instruction (1024 each): mfma dependent %.1f, mfma 4 independent %.1f, v_fma_f64 dependent %.1f, v_fma_f64 4 independent %.1f\n",
         ht[0] / 1024.0, ht[1] / 1024.0, ht[2] / 1024.0, ht[3] / 1024.0);
  return 0;
}
