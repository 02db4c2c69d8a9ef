// tds_lanes.h — register / cross-lane helpers shared by the step kernels (tds_kernels.hip, tds_quad.hip):
// small fixed-size algebra on registers, reciprocal / rsqrt with Newton steps, DPP moves inside and across the 16-lane
// rows of a wavefront (row shifts, rotations, broadcasts; gfx950's v_permlane16_swap for 32-lane groups), lane-group
// sums, the reference's plane_space and matrix_to_quat.  Everything is __device__ __forceinline__ in an anonymous
// namespace: include it from a .hip translation unit.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------------
// small fixed-size algebra on registers (everything fully unrolled; no runtime-indexed arrays)
// ------------------------------------------------------------------------------------------
// Record stores (y records, obs records, visual poses) are ORDINARY write-back stores.  Until round 4 they were streaming
// (non-temporal) stores: a record's lines are not written in one go — the line that holds the end of the state and the first
// visual pose is written in two halves half a step apart, the obs ring's 240-byte records share lines between workgroups —
// and a streamed partial line leaves the L2 before its other half arrives: HBM write traffic 7.68 MB per Ant x 4096 step for
// 6.23 MB of payload.  Write-back stores let the halves meet in the L2: 6.08 MB per step, traffic / algorithmic bytes 1.24 ->
// 1.02 at 1000 steps and 1.34 -> 1.09 on the 20-step command, for +0.7 % of step time
// (profiles/r04_ab_slots11_plain_stores.txt; -DTDS_STREAMING_STORES brings the streaming stores back).
#ifdef TDS_STREAMING_STORES
#define TDS_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define TDS_NT_STORE(v, p) (*(p) = (v))
#endif
template <typename T>
__device__ __forceinline__ void cross3(const T *a, const T *b, T *o) {
  const T x = a[1] * b[2] - a[2] * b[1];
  const T y = a[2] * b[0] - a[0] * b[2];
  const T z = a[0] * b[1] - a[1] * b[0];
  o[0] = x;
  o[1] = y;
  o[2] = z;
}
template <typename T>
__device__ __forceinline__ T dot3(const T *a, const T *b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
template <typename T>
__device__ __forceinline__ void mat3_mulv(const T *m, const T *v, T *o) {
  const T x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  const T y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  const T z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  o[0] = x;
  o[1] = y;
  o[2] = z;
}
// cofactor inverse (tiny_matrix3x3.h:539-559)
template <typename T>
__device__ __forceinline__ void mat3_inverse(const T *m, T *o) {
  const T c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
  const T s = T(1) / (m[0] * c0 + m[1] * c1 + m[2] * c2);
  o[0] = c0 * s; o[1] = (m[2] * m[7] - m[1] * m[8]) * s; o[2] = (m[1] * m[5] - m[2] * m[4]) * s;
  o[3] = c1 * s; o[4] = (m[0] * m[8] - m[2] * m[6]) * s; o[5] = (m[2] * m[3] - m[0] * m[5]) * s;
  o[6] = c2 * s; o[7] = (m[1] * m[6] - m[0] * m[7]) * s; o[8] = (m[0] * m[4] - m[1] * m[3]) * s;
}
template <typename T>
__device__ __forceinline__ void mat3_mul(const T *a, const T *b, T *o) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) o[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c];
}
// symmetric 3x3 stored as (xx, xy, xz, yy, yz, zz)
template <typename T>
__device__ __forceinline__ void sym3_mulv(const T *s, const T *v, T *o) {
  const T x = s[0] * v[0] + s[1] * v[1] + s[2] * v[2];
  const T y = s[1] * v[0] + s[3] * v[1] + s[4] * v[2];
  const T z = s[2] * v[0] + s[4] * v[1] + s[5] * v[2];
  o[0] = x;
  o[1] = y;
  o[2] = z;
}
// 1/x: hardware reciprocal estimate + ONE Newton-Raphson step.  Measured on MI355X over 2^20 random
// operands: v_rcp_f64 alone 4.6e-8 relative error, one step 2.1e-15, two steps 1.1e-16 — one step is
// 9 orders below the 1e-6 parity tolerance and sits on the serial pivot chain of the LDL^T.
// 3 dependent instructions instead of the ~12 of an IEEE division; operands here are pivots /
// diagonal entries in the normal range, no denormal or infinity handling needed.
template <typename T>
__device__ __forceinline__ T rcp_full(T x);
template <>
__device__ __forceinline__ double rcp_full<double>(double d) {
  double r = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}
template <>
__device__ __forceinline__ float rcp_full<float>(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  return __builtin_fmaf(r, e, r);
}
// 1/sqrt(x): hardware estimate + two Newton-Raphson steps (the estimate is good to ~2^-26 in double, each step squares
// the error: ~1e-16 after the second).  x > 0 in the normal range (1 + trace of a rotation matrix and its like).
// sqrt(x) = x * r and c / sqrt(x) = c * r replace an IEEE square root AND an IEEE division (~60 instructions) by ~10.
template <typename T>
__device__ __forceinline__ T rsqrt_full(T x);
template <>
__device__ __forceinline__ double rsqrt_full<double>(double x) {
  double r = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  r = __builtin_fma(r, __builtin_fma(-hx * r, r, 0.5), r);
  r = __builtin_fma(r, __builtin_fma(-hx * r, r, 0.5), r);
  return r;
}
template <>
__device__ __forceinline__ float rsqrt_full<float>(float x) {
  float r = __builtin_amdgcn_rsqf(x);
  const float hx = 0.5f * x;
  r = __builtin_fmaf(r, __builtin_fmaf(-hx * r, r, 0.5f), r);
  return r;
}
template <typename T>
__device__ __forceinline__ T atan2_t(T y, T x);
template <>
__device__ __forceinline__ double atan2_t<double>(double y, double x) { return atan2(y, x); }
template <>
__device__ __forceinline__ float atan2_t<float>(float y, float x) { return atan2f(y, x); }
template <typename T>
__device__ __forceinline__ T asin_t(T a);
template <>
__device__ __forceinline__ double asin_t<double>(double a) { return asin(a); }
template <>
__device__ __forceinline__ float asin_t<float>(float a) { return asinf(a); }
template <typename T>
__device__ __forceinline__ void sincos_t(T a, T *s, T *c);
template <>
__device__ __forceinline__ void sincos_t<double>(double a, double *s, double *c) {
  sincos(a, s, c);
}
template <>
__device__ __forceinline__ void sincos_t<float>(float a, float *s, float *c) {
  sincosf(a, s, c);
}
template <typename T>
__device__ __forceinline__ T max_t(T a, T b) {
  return a > b ? a : b;
}
template <typename T>
__device__ __forceinline__ T min_t(T a, T b) {
  return a < b ? a : b;
}
template <>
__device__ __forceinline__ double max_t<double>(double a, double b) {
  return fmax(a, b);
}
template <>
__device__ __forceinline__ double min_t<double>(double a, double b) {
  return fmin(a, b);
}
template <typename T>
__device__ __forceinline__ T sqrt_t(T a);
template <>
__device__ __forceinline__ double sqrt_t<double>(double a) {
  return sqrt(a);
}
template <>
__device__ __forceinline__ float sqrt_t<float>(float a) {
  return sqrtf(a);
}

// v + (v moved by a DPP cross-lane pattern inside each row of 16 lanes); VALU latency, no LDS
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
  const int l = __double2loint(v), h = __double2hiint(v);
  // old = 0 / bound_ctrl: row rotations have no invalid source lane, and the mov needs no register copy
  const int lo = __builtin_amdgcn_update_dpp(0, l, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, h, CTRL, 0xF, 0xF, true);
  return v + __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int b = __float_as_int(v);
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, b, CTRL, 0xF, 0xF, true));
}
// value of the neighbouring lane inside a 16-lane DPP row (VALU move, no LDS): FROM_NEXT: lane i
// receives lane i+1 (row_shl:1), else lane i receives lane i-1 (row_shr:1); 0 at the row boundary.
template <bool FROM_NEXT>
__device__ __forceinline__ double dpp_neighbour(double v) {
  constexpr int CTRL = FROM_NEXT ? 0x101 : 0x111;
  const int l = __double2loint(v), h = __double2hiint(v);
  const int lo = __builtin_amdgcn_update_dpp(0, l, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, h, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <bool FROM_NEXT>
__device__ __forceinline__ float dpp_neighbour(float v) {
  constexpr int CTRL = FROM_NEXT ? 0x101 : 0x111;
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// lane i receives lane i + D of its 16-lane DPP row (row_shl:D), zero where that lane does not exist
template <int D>
__device__ __forceinline__ double dpp_shl(double v) {
  const int l = __double2loint(v), h = __double2hiint(v);
  const int lo = __builtin_amdgcn_update_dpp(0, l, 0x100 + D, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, h, 0x100 + D, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int D>
__device__ __forceinline__ float dpp_shl(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + D, 0xF, 0xF, true));
}
// lane i receives lane i - D of its 16-lane DPP row (row_shr:D), zero where that lane does not exist
template <int D>
__device__ __forceinline__ double dpp_shr(double v) {
  const int l = __double2loint(v), h = __double2hiint(v);
  const int lo = __builtin_amdgcn_update_dpp(0, l, 0x110 + D, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, h, 0x110 + D, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int D>
__device__ __forceinline__ float dpp_shr(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + D, 0xF, 0xF, true));
}
// lane i receives lane i - D, rows of 16 lanes or not: G == 16 (an environment is one DPP row): row_shr:D; wider lane
// groups: wave_shr:1 (gfx9: shifts across the whole wavefront), D times.  What the first D lanes receive is garbage.
template <int D, int G>
__device__ __forceinline__ double seg_shr(double v) {
  if constexpr (G == 16) {
    return dpp_shr<D>(v);
  } else {
    int l = __double2loint(v), h = __double2hiint(v);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      l = __builtin_amdgcn_update_dpp(0, l, 0x138, 0xF, 0xF, true);
      h = __builtin_amdgcn_update_dpp(0, h, 0x138, 0xF, 0xF, true);
    }
    return __hiloint2double(h, l);
  }
}
template <int D, int G>
__device__ __forceinline__ float seg_shr(float v) {
  if constexpr (G == 16) {
    return dpp_shr<D>(v);
  } else {
    int b = __float_as_int(v);
#pragma unroll
    for (int i = 0; i < D; ++i) b = __builtin_amdgcn_update_dpp(0, b, 0x138, 0xF, 0xF, true);
    return __int_as_float(b);
  }
}
// lane i receives lane i + D (row_shl:D inside an environment's one DPP row, wave_shl:1 D times for wider lane groups)
template <int D, int G>
__device__ __forceinline__ double seg_shl(double v) {
  if constexpr (G == 16) {
    return dpp_shl<D>(v);
  } else {
    int l = __double2loint(v), h = __double2hiint(v);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      l = __builtin_amdgcn_update_dpp(0, l, 0x130, 0xF, 0xF, true);
      h = __builtin_amdgcn_update_dpp(0, h, 0x130, 0xF, 0xF, true);
    }
    return __hiloint2double(h, l);
  }
}
template <int D, int G>
__device__ __forceinline__ float seg_shl(float v) {
  if constexpr (G == 16) {
    return dpp_shl<D>(v);
  } else {
    int b = __float_as_int(v);
#pragma unroll
    for (int i = 0; i < D; ++i) b = __builtin_amdgcn_update_dpp(0, b, 0x130, 0xF, 0xF, true);
    return __int_as_float(b);
  }
}
// the same shift, lanes without a source (the first D of every row) receive 1 instead of 0 (the low word of 1.0 is 0:
// only the high word needs a pre-set destination)
template <int D>
__device__ __forceinline__ double dpp_shr_one(double v) {
  const int l = __double2loint(v), h = __double2hiint(v);
  const int lo = __builtin_amdgcn_update_dpp(0, l, 0x110 + D, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0x3FF00000, h, 0x110 + D, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int D>
__device__ __forceinline__ float dpp_shr_one(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0x3F800000, __float_as_int(v), 0x110 + D, 0xF, 0xF, false));
}
// 32 payload bits carried through an LDS slot of the compute scalar (no arithmetic on them)
template <typename T>
__device__ __forceinline__ T bits_to_scalar(unsigned b);
template <>
__device__ __forceinline__ double bits_to_scalar<double>(unsigned b) {
  return __hiloint2double(0, (int)b);
}
template <>
__device__ __forceinline__ float bits_to_scalar<float>(unsigned b) {
  return __int_as_float((int)b);
}
template <typename T>
__device__ __forceinline__ unsigned scalar_to_bits(T v);
template <>
__device__ __forceinline__ unsigned scalar_to_bits<double>(double v) {
  return (unsigned)__double2loint(v);
}
template <>
__device__ __forceinline__ unsigned scalar_to_bits<float>(float v) {
  return (unsigned)__float_as_int(v);
}

// 32-lane groups span two 16-lane DPP rows (rows 0|1 and 2|3 of the wavefront).  gfx950's
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second:
// r[0] = {a.row0, b.row0, a.row2, b.row2}, r[1] = {a.row1, b.row1, a.row3, b.row3} — a VALU move, no
// LDS round trip (ds_bpermute / __shfl).
// the partner row's value: lane i of row 0 receives lane i of row 1 and vice versa
__device__ __forceinline__ int swap_rows_b32(int x, bool upper_row) {
  const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
  return (int)(upper_row ? r[0] : r[1]);
}
__device__ __forceinline__ double other_row(double v, bool upper_row) {
  return __hiloint2double(swap_rows_b32(__double2hiint(v), upper_row), swap_rows_b32(__double2loint(v), upper_row));
}
__device__ __forceinline__ float other_row(float v, bool upper_row) {
  return __int_as_float(swap_rows_b32(__float_as_int(v), upper_row));
}
// lane SRC (0..31) of each 32-lane group to all its 32 lanes.  A source in the group's FIRST row: row broadcast, then
// lane 15 of rows 0 / 2 into every lane of rows 1 / 3 (row_bcast:15 under row mask 0b1010) — two moves per 32 bits.  A
// source in the SECOND row has no DPP control that reaches back: row broadcast, then that row copied over its partner
// with a row swap (v_permlane16_swap and the copies its two-operand form needs).
template <int SRC>
__device__ __forceinline__ int bcast32_b32(int x) {
  const int t = __builtin_amdgcn_update_dpp(0, x, 0x150 + (SRC & 15), 0xF, 0xF, true);
  if constexpr (SRC < 16) {
    return __builtin_amdgcn_update_dpp(t, t, 0x142, 0xA, 0xF, false);
  } else {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)t, (unsigned)t, false, false);
    return (int)r[1];
  }
}
template <int SRC>
__device__ __forceinline__ double bcast32(double v) {
  // (one 64-bit row broadcast — row_newbcast is the DPP control the 64-bit ALU takes — then the second step per half)
  const double t = __builtin_amdgcn_update_dpp(0.0, v, 0x150 + (SRC & 15), 0xF, 0xF, true);
  const int l = __double2loint(t), h = __double2hiint(t);
  if constexpr (SRC < 16) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(h, h, 0x142, 0xA, 0xF, false),
                            __builtin_amdgcn_update_dpp(l, l, 0x142, 0xA, 0xF, false));
  } else {
    const auto rl = __builtin_amdgcn_permlane16_swap((unsigned)l, (unsigned)l, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap((unsigned)h, (unsigned)h, false, false);
    return __hiloint2double((int)rh[1], (int)rl[1]);
  }
}
template <int SRC>
__device__ __forceinline__ float bcast32(float v) {
  return __int_as_float(bcast32_b32<SRC>(__float_as_int(v)));
}
// sum over the G lanes of an environment; every lane receives the total.  Strides 8,4,2,1 are
// row rotations (DPP row_ror), wider strides go through the LDS crossbar (ds_bpermute).
template <typename T, int G>
__device__ __forceinline__ T group_sum(T v) {
  v = dpp_add<0x128>(v);  // row_ror:8
  v = dpp_add<0x124>(v);  // row_ror:4
  v = dpp_add<0x122>(v);  // row_ror:2
  v = dpp_add<0x121>(v);  // row_ror:1
  if constexpr (G == 32) {
    v += other_row(v, (threadIdx.x & 16) != 0);
  } else {
#pragma unroll
    for (int m = 16; m < G; m <<= 1) v += __shfl_xor(v, m, G);
  }
  return v;
}

// value held by lane SRC of the environment's lane group, delivered to every lane of the group.
// All sources live in lanes 0..NDP-1; for NDP <= 16 that is one 16-lane DPP row -> row_newbcast
// (a VALU move, no LDS round trip); wider groups go through ds_bpermute.
template <int SRC>
__device__ __forceinline__ double dpp_bcast(double v) {
  // row_newbcast is the one DPP control the 64-bit ALU takes (gfx90a on): ONE v_mov_b64_dpp instead of two 32-bit moves
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + SRC, 0xF, 0xF, true);
}
template <int SRC>
__device__ __forceinline__ float dpp_bcast(float v) {
  const int b = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(0, b, 0x150 + SRC, 0xF, 0xF, true));
}
template <typename T, int G, int NDP, int SRC>
__device__ __forceinline__ T lane_bcast(T v) {
  if constexpr (NDP <= 16)
    return dpp_bcast<SRC & 15>(v);
  else if constexpr (G == 32)
    return bcast32<SRC>(v);
  else
    return __shfl(v, SRC, G);
}
// compile-time loop helper (the DPP control must be an immediate)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// reference: src/mb_constraint_solver.hpp:506-520 (plane_space incl. its k = sqrt(a) and p[2] quirks; the host-side
// twin for the fixed plane normal is tds_plane_space in tds_device_model.h)
template <typename T>
__device__ __forceinline__ void plane_space_dev(const T *n, T *p, T *q) {
  const T n_sqr = n[2] * n[2];
  const bool gt = n_sqr > T(0.5);
  const T a = n[1] * n[1] + (gt ? n_sqr : n[0] * n[0]);
  const T k = sqrt_t<T>(a);
  p[0] = gt ? T(0) : -n[1] * k;
  p[1] = gt ? -n[2] * k : n[0] * k;
  p[2] = n[1] * k;
  q[0] = gt ? a * k : -n[2] * p[1];
  q[1] = gt ? -n[0] * p[2] : n[2] * p[0];
  q[2] = gt ? n[0] * p[1] : a * k;
}

// reference: src/math/tiny/tiny_matrix3x3.h:432-465 (getRotation, right-associative build:
// off-diagonal differences transposed w.r.t. Bullet, w negated)
template <typename T>
__device__ __forceinline__ void matrix_to_quat(const T *m, T *q) {
  const T trace = m[0] + m[4] + m[8];
  T t0, t1, t2, t3;
  if (trace < T(0)) {
    // i = index of the largest diagonal element, (j,k) cyclic successors
    const int i = m[0] < m[4] ? (m[4] < m[8] ? 2 : 1) : (m[0] < m[8] ? 2 : 0);
    // select without runtime-indexed arrays
    T mii, mjj, mkk, mjk, mkj, mij, mji, mik, mki;
    if (i == 0) {
      mii = m[0]; mjj = m[4]; mkk = m[8]; mjk = m[5]; mkj = m[7]; mij = m[1]; mji = m[3]; mik = m[2]; mki = m[6];
    } else if (i == 1) {
      mii = m[4]; mjj = m[8]; mkk = m[0]; mjk = m[6]; mkj = m[2]; mij = m[5]; mji = m[7]; mik = m[3]; mki = m[1];
    } else {
      mii = m[8]; mjj = m[0]; mkk = m[4]; mjk = m[1]; mkj = m[3]; mij = m[6]; mji = m[2]; mik = m[7]; mki = m[5];
    }
    // (s = sqrt(a), ti = s / 2, then s = 0.5 / s:  with r = 1 / sqrt(a):  ti = a r / 2,  s = r / 2)
    const T a_ = ((mii - mjj) - mkk) + T(1);
    const T r_ = rsqrt_full<T>(a_);
    const T ti = a_ * r_ * T(0.5);
    const T s = T(0.5) * r_;
    t3 = (mjk - mkj) * s;
    const T tj = (mij + mji) * s;
    const T tk = (mik + mki) * s;
    if (i == 0) { t0 = ti; t1 = tj; t2 = tk; }
    else if (i == 1) { t1 = ti; t2 = tj; t0 = tk; }
    else { t2 = ti; t0 = tj; t1 = tk; }
  } else {
    const T a_ = trace + T(1);
    const T r_ = rsqrt_full<T>(a_);
    t3 = a_ * r_ * T(0.5);
    const T s = T(0.5) * r_;
    t0 = (m[5] - m[7]) * s;
    t1 = (m[6] - m[2]) * s;
    t2 = (m[1] - m[3]) * s;
  }
  q[0] = t0;
  q[1] = t1;
  q[2] = t2;
  q[3] = -t3;
}


}  // namespace
