// What a LONE wavefront pays per instruction on gfx950 (one wavefront per SIMD: the regime of the Ant kernels at 4096 / 8192
// environments): dependent and independent f64 FMA chains, the 8-lane DPP sum of tds_oct.hip, v_cndmask, LDS read round trips.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o lone_wave_latency lone_wave_latency.hip && ./lone_wave_latency
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP 256
// (the clock read takes the chain's value as an operand: the VALU chain is complete when it is read, and starts after it)
__device__ __forceinline__ unsigned long long clk(double &pin) {
  unsigned long long t;
  asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(pin)::"memory");
  return t;
}
__device__ __forceinline__ unsigned long long clk(int &pin) {
  unsigned long long t;
  asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(pin)::"memory");
  return t;
}
template <int CTRL>
__device__ __forceinline__ double dppmov(double v) {
  const int l = __double2loint(v), h = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, h, CTRL, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, l, CTRL, 0xF, 0xF, true));
}
__global__ void k(double *out, unsigned long long *t, const double *in) {
  __shared__ double lds[1024];
  double a = in[threadIdx.x], b = in[64 + threadIdx.x], c = in[128 + threadIdx.x], d = in[192 + threadIdx.x];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = in[i & 255];
  __syncthreads();
  unsigned long long t0, t1;
  // 0: dependent fma chain
  t0 = clk(a);
#pragma unroll
  for (int i = 0; i < REP; ++i) a = __builtin_fma(a, b, c);
  t1 = clk(a);
  if (threadIdx.x == 0) t[0] = t1 - t0;
  // 1: four independent fma chains (REP each)
  double a1 = a, a2 = b, a3 = c, a4 = d;
  t0 = clk(a1);
#pragma unroll
  for (int i = 0; i < REP; ++i) {
    a1 = __builtin_fma(a1, b, c);
    a2 = __builtin_fma(a2, b, c);
    a3 = __builtin_fma(a3, b, c);
    a4 = __builtin_fma(a4, b, c);
  }
  asm volatile("" : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4));
  a1 += a2 + a3 + a4;
  t1 = clk(a1);
  if (threadIdx.x == 0) t[1] = t1 - t0;
  a = a1;
  // 2: dependent 8-lane sums (3 x (2 dpp movs + add))
  t0 = clk(a);
#pragma unroll
  for (int i = 0; i < REP / 4; ++i) {
    a += dppmov<0xB1>(a);
    a += dppmov<0x4E>(a);
    a += dppmov<0x141>(a);
  }
  t1 = clk(a);
  if (threadIdx.x == 0) t[2] = t1 - t0;
  // 3: dependent add chain
  t0 = clk(a);
#pragma unroll
  for (int i = 0; i < REP; ++i) a = a + b;
  t1 = clk(a);
  if (threadIdx.x == 0) t[3] = t1 - t0;
  // 4: dependent max/min chain
  t0 = clk(a);
#pragma unroll
  for (int i = 0; i < REP / 2; ++i) {
    a = fmax(a, b);
    a = fmin(a, c);
  }
  t1 = clk(a);
  if (threadIdx.x == 0) t[4] = t1 - t0;
  // 5: dependent LDS reads (pointer chase through values)
  int idx = threadIdx.x;
  t0 = clk(idx);
#pragma unroll
  for (int i = 0; i < 64; ++i) idx = ((int)lds[idx & 1023]) & 1023;
  t1 = clk(idx);
  if (threadIdx.x == 0) t[5] = t1 - t0;
  // 6: six independent ds_read2_b64-like loads + wait, repeated (a row fetch)
  double acc = 0;
  t0 = clk(idx);
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const double *r = lds + ((idx + 12 * i) & 511);
    double s = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) s += r[j];
    acc += s;
    asm volatile("" : "+v"(acc));
  }
  t1 = clk(acc);
  if (threadIdx.x == 0) t[6] = t1 - t0;
  // 7: dependent f64 mul chain
  t0 = clk(a);
#pragma unroll
  for (int i = 0; i < REP; ++i) a = a * b;
  t1 = clk(a);
  if (threadIdx.x == 0) t[7] = t1 - t0;
  // 8: dependent v_cndmask pairs
  t0 = clk(a);
#pragma unroll
  for (int i = 0; i < REP; ++i) a = (a > b) ? c : a + 0.0 * i;
  t1 = clk(a);
  if (threadIdx.x == 0) t[8] = t1 - t0;
  out[threadIdx.x] = a + acc + idx;
}
int main() {
  double *in, *out;
  unsigned long long *t;
  hipMalloc(&in, 1024 * 8);
  hipMalloc(&out, 64 * 8);
  hipMalloc(&t, 16 * 8);
  double h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = 1.0 + 1e-9 * i;
  for (int i = 0; i < 256; ++i) h[i] = (double)((i * 37) & 1023);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, t, in);
  hipDeviceSynchronize();
  unsigned long long ht[16];
  hipMemcpy(ht, t, sizeof(ht), hipMemcpyDeviceToHost);
  printf("cycles per op, one wavefront alone on its SIMD (s_memtime):\n");
  printf("  dependent v_fma_f64 chain           %.1f\n", ht[0] / (double)REP);
  printf("  4 independent v_fma_f64 chains      %.1f per fma\n", ht[1] / (double)(4 * REP));
  printf("  dependent 8-lane DPP sum            %.1f per sum (6 dpp movs + 3 adds)\n", ht[2] / (double)(REP / 4));
  printf("  dependent v_add_f64 chain           %.1f\n", ht[3] / (double)REP);
  printf("  dependent v_max/v_min_f64 chain     %.1f\n", ht[4] / (double)REP);
  printf("  dependent LDS read (b64) round trip %.1f\n", ht[5] / 64.0);
  printf("  fetch of a 12-double row + use      %.1f per row\n", ht[6] / 32.0);
  printf("  dependent v_mul_f64 chain           %.1f\n", ht[7] / (double)REP);
  printf("  dependent cmp + cndmask chain       %.1f\n", ht[8] / (double)REP);
  return 0;
}
