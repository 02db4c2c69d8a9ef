// tds_kernels.hip — the MI355X (gfx950 / CDNA4) step kernel.
//
// One launch advances N independent environments by one step of the reference's
//   PD -> forward_dynamics (ABA) -> integrate_euler_qdd -> World::step (plane contacts +
//   MLCP / projected Gauss-Seidel) -> integrate_euler -> pack
// (reference: examples/environments/locomotion_contact_simulation.h:151-304).
//
// Mapping.  An environment owns G lanes of a 64-wide wavefront (G = 64, 32 or 16; 64/G
// environments per wavefront, one wavefront per workgroup) and a private LDS region.  Inside an
// environment:  lane == link for the tree sweeps, lane == dof for the joint-space vectors,
// lane == row for the constraint rows, lane == contact point for the narrowphase.  Nothing is
// spilled to scratch and the only HBM traffic is the x record in and the y record out, both
// read / written by consecutive lanes (env-major records == coalesced for lane-per-component).
//
// Formulation (differs from the reference's, results agree to round-off; parity is enforced by
// tests/ against oracle/ and the golden vectors):
//   * all spatial quantities are expressed in WORLD coordinates about the world origin, so the
//     tree sweeps need no 6x6 congruence transforms:  IA_parent += Ia,  pA_parent += pa;
//     (the reference transforms link-to-parent with dense 6x6x6 products,
//      src/dynamics/forward_dynamics.hpp:187-189, src/dynamics/mass_matrix.hpp:45-46)
//   * rigid-body inertias are kept as (I_sym[6], h[3], m) = 10 numbers; only the articulated
//     inertia needs the full symmetric 6x6 (I[6], H[9], M[6]);
//   * CRBA composite inertias ride along the ABA backward sweep;
//   * M = L D L^T (no square roots), B = M^-1 J^T by two triangular solves per constraint row;
//   * PGS runs on  w = M^-1 J^T x  instead of A = J M^-1 J^T:  delta_i = J_i.w - (J_i.B_i) x_i,
//     so A (51x51 for Ant) is never formed and the final  qd -= M^-1 J^T p  is simply  qd -= w;
//   * rows of separated contacts (distance >= 0) are identically zero in the reference
//     (keep_all_points_, src/mb_constraint_solver.hpp:285-291) and yield x = 0, so only
//     penetrating contacts are materialised, in the reference's row order.
#include <hip/hip_runtime.h>

#include "tds_device_model.h"
#include "tds_kernels.h"

namespace {

// ------------------------------------------------------------------------------------------
// small fixed-size algebra on registers (everything fully unrolled; no runtime-indexed arrays)
// ------------------------------------------------------------------------------------------
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
template <typename T>
__device__ __forceinline__ void mat3_tmulv(const T *m, const T *v, T *o) {  // m^T v
  const T x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  const T y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  const T z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  o[0] = x;
  o[1] = y;
  o[2] = z;
}
template <typename T>
__device__ __forceinline__ T rcp_full(T x) {
  return T(1) / x;
}
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
__device__ __forceinline__ T sqrt_t(T a);
template <>
__device__ __forceinline__ double sqrt_t<double>(double a) {
  return sqrt(a);
}
template <>
__device__ __forceinline__ float sqrt_t<float>(float a) {
  return sqrtf(a);
}

// sum over the G lanes of an environment; every lane receives the total
template <typename T, int G>
__device__ __forceinline__ T group_sum(T v) {
#pragma unroll
  for (int m = 1; m < G; m <<= 1) v += __shfl_xor(v, m, G);
  return v;
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
    T s = sqrt_t<T>(((mii - mjj) - mkk) + T(1));
    const T ti = s * T(0.5);
    s = T(0.5) / s;
    t3 = (mjk - mkj) * s;
    const T tj = (mij + mji) * s;
    const T tk = (mik + mki) * s;
    if (i == 0) { t0 = ti; t1 = tj; t2 = tk; }
    else if (i == 1) { t1 = ti; t2 = tj; t0 = tk; }
    else { t2 = ti; t0 = tj; t1 = tk; }
  } else {
    T s = sqrt_t<T>(trace + T(1));
    t3 = s * T(0.5);
    s = T(0.5) / s;
    t0 = (m[5] - m[7]) * s;
    t1 = (m[6] - m[2]) * s;
    t2 = (m[1] - m[3]) * s;
  }
  q[0] = t0;
  q[1] = t1;
  q[2] = t2;
  q[3] = -t3;
}

// ------------------------------------------------------------------------------------------
// the step kernel
// ------------------------------------------------------------------------------------------
// TDS_STAMP: phase-boundary timestamps (shader clock) of workgroup 0, PROF builds only
#define TDS_STAMP(k)                                                        \
  do {                                                                      \
    if (PROF) {                                                             \
      if (blockIdx.x == 0 && threadIdx.x == 0) prof[k] = (long long)__builtin_amdgcn_s_memtime(); \
    }                                                                       \
  } while (0)

template <typename T, int G, bool PROF>
__global__ __launch_bounds__(64) void tds_step_kernel(const DevModel<T> *__restrict__ mdl, TdsLds L,
                                                      const T *x_in, T *__restrict__ y_out,
                                                      const T *__restrict__ actions, T *x_feedback /* may alias x_in */,
                                                      T *__restrict__ obs_out, long long *prof, int n_envs) {
  extern __shared__ __align__(16) unsigned char tds_smem_raw[];
  T *const sm = reinterpret_cast<T *>(tds_smem_raw);
  constexpr int EPW = 64 / G;
  const int lane = threadIdx.x & (G - 1);
  const int grp = threadIdx.x / G;
  const int env = blockIdx.x * EPW + grp;
  const bool valid = env < n_envs;
  T *const E = sm + grp * L.stride;

  const int nl = mdl->num_links, nq = mdl->dof_q, nd = mdl->dof_qd;
  const int in_dim = mdl->input_dim, out_dim = mdl->output_dim, adim = mdl->action_dim;
  const int NLp = L.NLp, NDs = L.NDs;
  const T dt = mdl->dt;

  TDS_STAMP(0);
  // ---- A. x record -> LDS (coalesced: consecutive lanes, consecutive doubles) ---------------
  T *const xr = E + L.xrec;
  for (int i = lane; i < in_dim; i += G) xr[i] = valid ? x_in[(size_t)env * in_dim + i] : T(0);
  if (actions != nullptr && valid)
    for (int i = lane; i < adim; i += G) xr[nq + nd + i] = actions[(size_t)env * adim + i];
  __syncthreads();

  // ---- lane == link: constants -------------------------------------------------------------
  const int li = lane;
  const bool isl = li < nl;
  const int lsafe = isl ? li : 0;
  const int parent = isl ? mdl->parent[lsafe] : -1;
  const int level = isl ? mdl->level[lsafe] : -1;
  const int jt = isl ? mdl->joint_type[lsafe] : TDS_JOINT_FIXED;
  const int di = isl ? mdl->qd_index[lsafe] : -1;  // == q_index (1-DoF joints only)
  const T q = di >= 0 ? xr[di] : T(0);
  const T qd = di >= 0 ? xr[nq + di] : T(0);

  // ---- PD controller (locomotion_contact_simulation.h:168-258) or direct torque -------------
  T tau = T(0);
  if (mdl->step_mode == TDS_STEP_LOCOMOTION) {
    const int ai = isl ? mdl->act_index[lsafe] : -1;
    if (ai >= 0) {
      const int var = nq + nd + adim;
      const T kp = xr[var], kd = xr[var + 1], max_force = xr[var + 2];
      T a = xr[nq + nd + ai];
      const T lim = mdl->action_limit;
      a = a < lim ? a : lim;       // Algebra::min(clamped_action, ACTION_LIMIT)
      a = a > -lim ? a : -lim;     // Algebra::max(clamped_action, -ACTION_LIMIT)
      const T q_des = mdl->init_pose[lsafe] + a;
      T f = kp * (q_des - q) + kd * (T(0) - qd);
      f = f > -max_force ? f : -max_force;
      f = f < max_force ? f : max_force;
      tau = f;
    }
  } else if (di >= 0) {
    tau = xr[nq + nd + di];
  }
  // joint stiffness / damping (forward_dynamics.hpp:122-123)
  if (isl) tau -= mdl->stiffness[lsafe] * q + mdl->damping[lsafe] * qd;

  TDS_STAMP(1);
  // ---- B. jcalc: X_parent = X_T * X_J(q)   (link.hpp:229-287) -------------------------------
  T Sl[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) Sl[k] = isl ? mdl->S[k][lsafe] : T(0);
  T Rp[9], tp[3];
  {
    T RT[9], tT[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) RT[k] = isl ? mdl->X_T[k][lsafe] : T(k % 4 == 0);
#pragma unroll
    for (int k = 0; k < 3; ++k) tT[k] = isl ? mdl->X_T[9 + k][lsafe] : T(0);
    const bool rev = jt >= TDS_JOINT_REVOLUTE_X && jt <= TDS_JOINT_REVOLUTE_AXIS;
    const bool pris = jt >= TDS_JOINT_PRISMATIC_X && jt <= TDS_JOINT_PRISMATIC_AXIS;
    T RJ[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)};
    T tJ[3] = {T(0), T(0), T(0)};
    if (pris) {  // translation = S.bottom * q (unit axis for _X/_Y/_Z)
      tJ[0] = Sl[3] * q;
      tJ[1] = Sl[4] * q;
      tJ[2] = Sl[5] * q;
    }
    T sn, cs;
    sincos_t<T>(jt == TDS_JOINT_REVOLUTE_AXIS ? q * T(0.5) : q, &sn, &cs);
    if (rev) {
      if (jt == TDS_JOINT_REVOLUTE_X) {  // tiny_matrix3x3.h:218-234
        RJ[4] = cs; RJ[5] = -sn; RJ[7] = sn; RJ[8] = cs;
      } else if (jt == TDS_JOINT_REVOLUTE_Y) {
        RJ[0] = cs; RJ[2] = sn; RJ[6] = -sn; RJ[8] = cs;
      } else if (jt == TDS_JOINT_REVOLUTE_Z) {
        RJ[0] = cs; RJ[1] = -sn; RJ[3] = sn; RJ[4] = cs;
      } else {  // axis-angle quaternion with the UNNORMALISED axis (link.hpp:256-261,
                // tiny_quaternion.h:178-183, tiny_matrix3x3.h:315-340)
        const T d = sqrt_t<T>(Sl[0] * Sl[0] + Sl[1] * Sl[1] + Sl[2] * Sl[2]);
        const T sh = sn / d;
        const T qx = Sl[0] * sh, qy = Sl[1] * sh, qz = Sl[2] * sh, qw = cs;
        const T s2 = T(2) / (qx * qx + qy * qy + qz * qz + qw * qw);
        const T xs = qx * s2, ys = qy * s2, zs = qz * s2;
        const T wx = qw * xs, wy = qw * ys, wz = qw * zs;
        const T xx = qx * xs, xy = qx * ys, xz = qx * zs;
        const T yy = qy * ys, yz = qy * zs, zz = qz * zs;
        RJ[0] = T(1) - (yy + zz); RJ[1] = xy - wz; RJ[2] = xz + wy;
        RJ[3] = xy + wz; RJ[4] = T(1) - (xx + zz); RJ[5] = yz - wx;
        RJ[6] = xz - wy; RJ[7] = yz + wx; RJ[8] = T(1) - (xx + yy);
      }
    }
    mat3_mul(RT, RJ, Rp);  // transform.hpp:123-131
    T r[3];
    mat3_mulv(RT, tJ, r);
    tp[0] = tT[0] + r[0];
    tp[1] = tT[1] + r[1];
    tp[2] = tT[2] + r[2];
  }

  TDS_STAMP(2);
  // ---- C. top-down sweep: X_world, world motion axis s, velocity v  (kinematics.hpp:64-97) ---
  T *const Xw = E + L.Xw;    // [12][NLp]
  T *const swd = E + L.swd;  // [6][NDs]   per dof
  T *const vv = E + L.v;     // [6][NLp]   v, later a
  T R[9], p[3], sw[6], vJ[6], v[6];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = T(0);
#pragma unroll
  for (int k = 0; k < 3; ++k) p[k] = T(0);
#pragma unroll
  for (int k = 0; k < 6; ++k) sw[k] = vJ[k] = v[k] = T(0);
  const int nlev = mdl->num_levels;
  for (int lev = 0; lev < nlev; ++lev) {
    if (level == lev) {
      T Rq[9], pq[3], vq[6];
      if (parent >= 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rq[k] = Xw[k * NLp + parent];
#pragma unroll
        for (int k = 0; k < 3; ++k) pq[k] = Xw[(9 + k) * NLp + parent];
#pragma unroll
        for (int k = 0; k < 6; ++k) vq[k] = vv[k * NLp + parent];
      } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rq[k] = mdl->base_R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) pq[k] = mdl->base_t[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) vq[k] = T(0);
      }
      mat3_mul(Rq, Rp, R);
      T r[3];
      mat3_mulv(Rq, tp, r);
      p[0] = pq[0] + r[0];
      p[1] = pq[1] + r[1];
      p[2] = pq[2] + r[2];
      // s = X_world.apply_inverse(S) = (R w, R v + p x (R w))   (transform.hpp:232-243)
      mat3_mulv(R, Sl, sw);
      mat3_mulv(R, Sl + 3, sw + 3);
      T c[3];
      cross3(p, sw, c);
      sw[3] += c[0];
      sw[4] += c[1];
      sw[5] += c[2];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        vJ[k] = sw[k] * qd;
        v[k] = vq[k] + vJ[k];
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) Xw[k * NLp + li] = R[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) Xw[(9 + k) * NLp + li] = p[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) vv[k * NLp + li] = v[k];
      if (di >= 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) swd[k * NDs + di] = sw[k];
      }
    }
    __syncthreads();
  }

  TDS_STAMP(3);
  // ---- D. bias terms and world-frame inertias (kinematics.hpp:96-132, inertia.hpp:121-130) ---
  T *const IAs = E + L.IA;  // [21][NLp]  I(6) H(9) M(6)
  T *const pAs = E + L.pA;  // [6][NLp]
  T *const Ics = E + L.Ic;  // [10][NLp]  I(6) h(3) m
  T cb[6];                  // c = v x vJ
  {
    cross3(v, vJ, cb);
    T c1[3], c2[3];
    cross3(v, vJ + 3, c1);
    cross3(v + 3, vJ, c2);
    cb[3] = c1[0] + c2[0];
    cb[4] = c1[1] + c2[1];
    cb[5] = c1[2] + c2[2];
  }
  if (isl) {
    const T m = mdl->mass[lsafe];
    T com[3], Il[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) com[k] = mdl->com[k][lsafe];
#pragma unroll
    for (int k = 0; k < 9; ++k) Il[k] = mdl->inertia[k][lsafe];
    T cw[3];
    mat3_mulv(R, com, cw);
    cw[0] += p[0];
    cw[1] += p[1];
    cw[2] += p[2];
    // R Il R^T (symmetric)
    T RI[9], Iw[9];
    mat3_mul(R, Il, RI);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Iw[3 * r + c] = RI[3 * r] * R[3 * c] + RI[3 * r + 1] * R[3 * c + 1] + RI[3 * r + 2] * R[3 * c + 2];
    const T c2 = dot3(cw, cw);
    T Is[6];  // I = Icom + m (|c|^2 1 - c c^T)
    Is[0] = Iw[0] + m * (c2 - cw[0] * cw[0]);
    Is[1] = T(0.5) * (Iw[1] + Iw[3]) - m * cw[0] * cw[1];
    Is[2] = T(0.5) * (Iw[2] + Iw[6]) - m * cw[0] * cw[2];
    Is[3] = Iw[4] + m * (c2 - cw[1] * cw[1]);
    Is[4] = T(0.5) * (Iw[5] + Iw[7]) - m * cw[1] * cw[2];
    Is[5] = Iw[8] + m * (c2 - cw[2] * cw[2]);
    const T h[3] = {m * cw[0], m * cw[1], m * cw[2]};
    // I v = (I w + h x v_lin, m v_lin - h x w)
    T Iv[6], t3[3];
    sym3_mulv(Is, v, Iv);
    cross3(h, v + 3, t3);
    Iv[0] += t3[0];
    Iv[1] += t3[1];
    Iv[2] += t3[2];
    cross3(h, v, t3);
    Iv[3] = m * v[3] - t3[0];
    Iv[4] = m * v[4] - t3[1];
    Iv[5] = m * v[5] - t3[2];
    // pA = v x* (I v) = (w x n + v x f, w x f)      (f_ext = 0 after clear_forces)
    T pa0[6], u3[3];
    cross3(v, Iv, pa0);
    cross3(v + 3, Iv + 3, u3);
    pa0[0] += u3[0];
    pa0[1] += u3[1];
    pa0[2] += u3[2];
    cross3(v, Iv + 3, pa0 + 3);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      IAs[k * NLp + li] = Is[k];
      Ics[k * NLp + li] = Is[k];
      pAs[k * NLp + li] = pa0[k];
    }
    // H = [h]x
    IAs[6 * NLp + li] = T(0);
    IAs[7 * NLp + li] = -h[2];
    IAs[8 * NLp + li] = h[1];
    IAs[9 * NLp + li] = h[2];
    IAs[10 * NLp + li] = T(0);
    IAs[11 * NLp + li] = -h[0];
    IAs[12 * NLp + li] = -h[1];
    IAs[13 * NLp + li] = h[0];
    IAs[14 * NLp + li] = T(0);
    // M = m 1
    IAs[15 * NLp + li] = m;
    IAs[16 * NLp + li] = T(0);
    IAs[17 * NLp + li] = T(0);
    IAs[18 * NLp + li] = m;
    IAs[19 * NLp + li] = T(0);
    IAs[20 * NLp + li] = m;
    Ics[6 * NLp + li] = h[0];
    Ics[7 * NLp + li] = h[1];
    Ics[8 * NLp + li] = h[2];
    Ics[9 * NLp + li] = m;
  }
  __syncthreads();

  TDS_STAMP(4);
  // ---- E. bottom-up sweep: ABA articulated inertia / bias, CRBA composite inertia -----------
  //      (forward_dynamics.hpp:50-216, mass_matrix.hpp:39-56) in world coordinates
  T U[6], Dinv = T(0), uu = T(0);
  T Fc[6];  // CRBA: F_i = Ic_i s_i
#pragma unroll
  for (int k = 0; k < 6; ++k) U[k] = Fc[k] = T(0);
  const bool want_crba = mdl->has_plane != 0;
  for (int lev = nlev - 1; lev >= 0; --lev) {
    if (level == lev) {
      T I6[6], H9[9], M6[6], pa[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) I6[k] = IAs[k * NLp + li];
#pragma unroll
      for (int k = 0; k < 9; ++k) H9[k] = IAs[(6 + k) * NLp + li];
#pragma unroll
      for (int k = 0; k < 6; ++k) M6[k] = IAs[(15 + k) * NLp + li];
#pragma unroll
      for (int k = 0; k < 6; ++k) pa[k] = pAs[k * NLp + li];
      // U = IA s
      T t3[3];
      sym3_mulv(I6, sw, U);
      mat3_mulv(H9, sw + 3, t3);
      U[0] += t3[0];
      U[1] += t3[1];
      U[2] += t3[2];
      sym3_mulv(M6, sw + 3, U + 3);
      mat3_tmulv(H9, sw, t3);
      U[3] += t3[0];
      U[4] += t3[1];
      U[5] += t3[2];
      const T D = dot3(sw, U) + dot3(sw + 3, U + 3);
      uu = tau - (dot3(sw, pa) + dot3(sw + 3, pa + 3));
      Dinv = di >= 0 ? rcp_full<T>(D) : T(0);  // forward_dynamics.hpp:153
      // Ia = IA - U U^T / D
      T Ub[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) Ub[k] = U[k] * Dinv;
      I6[0] -= U[0] * Ub[0]; I6[1] -= U[0] * Ub[1]; I6[2] -= U[0] * Ub[2];
      I6[3] -= U[1] * Ub[1]; I6[4] -= U[1] * Ub[2]; I6[5] -= U[2] * Ub[2];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) H9[3 * r + c] -= U[r] * Ub[3 + c];
      M6[0] -= U[3] * Ub[3]; M6[1] -= U[3] * Ub[4]; M6[2] -= U[3] * Ub[5];
      M6[3] -= U[4] * Ub[4]; M6[4] -= U[4] * Ub[5]; M6[5] -= U[5] * Ub[5];
      // pa = pA + Ia c + U u / D
      T Iac[6];
      sym3_mulv(I6, cb, Iac);
      mat3_mulv(H9, cb + 3, t3);
      Iac[0] += t3[0];
      Iac[1] += t3[1];
      Iac[2] += t3[2];
      sym3_mulv(M6, cb + 3, Iac + 3);
      mat3_tmulv(H9, cb, t3);
      Iac[3] += t3[0];
      Iac[4] += t3[1];
      Iac[5] += t3[2];
      const T ud = uu * Dinv;
#pragma unroll
      for (int k = 0; k < 6; ++k) pa[k] += Iac[k] + U[k] * ud;
      if (parent >= 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) atomicAdd(&IAs[k * NLp + parent], I6[k]);
#pragma unroll
        for (int k = 0; k < 9; ++k) atomicAdd(&IAs[(6 + k) * NLp + parent], H9[k]);
#pragma unroll
        for (int k = 0; k < 6; ++k) atomicAdd(&IAs[(15 + k) * NLp + parent], M6[k]);
#pragma unroll
        for (int k = 0; k < 6; ++k) atomicAdd(&pAs[k * NLp + parent], pa[k]);
      }
      if (want_crba) {
        T Ic[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) Ic[k] = Ics[k * NLp + li];
        if (parent >= 0) {
#pragma unroll
          for (int k = 0; k < 10; ++k) atomicAdd(&Ics[k * NLp + parent], Ic[k]);
        }
        // F = Ic s = (I w + h x v, m v - h x w)
        sym3_mulv(Ic, sw, Fc);
        cross3(Ic + 6, sw + 3, t3);
        Fc[0] += t3[0];
        Fc[1] += t3[1];
        Fc[2] += t3[2];
        cross3(Ic + 6, sw, t3);
        Fc[3] = Ic[9] * sw[3] - t3[0];
        Fc[4] = Ic[9] * sw[4] - t3[1];
        Fc[5] = Ic[9] * sw[5] - t3[2];
      }
    }
    __syncthreads();
  }

  TDS_STAMP(5);
  // ---- F. top-down sweep: accelerations, qdd  (forward_dynamics.hpp:245-302) ----------------
  //      a overwrites v in LDS (v of every link is already in registers)
  T qdd = T(0);
  for (int lev = 0; lev < nlev; ++lev) {
    if (level == lev) {
      T a[6];
      if (parent >= 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) a[k] = vv[k * NLp + parent];
      } else {  // base acceleration = -gravity (linear part)
        a[0] = a[1] = a[2] = T(0);
        a[3] = -mdl->grav[0];
        a[4] = -mdl->grav[1];
        a[5] = -mdl->grav[2];
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) a[k] += cb[k];
      if (di >= 0) {
        const T Uta = dot3(U, a) + dot3(U + 3, a + 3);
        qdd = Dinv * (uu - Uta);
#pragma unroll
        for (int k = 0; k < 6; ++k) a[k] += sw[k] * qdd;
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) vv[k * NLp + li] = a[k];
    }
    __syncthreads();
  }
  // integrate_euler_qdd: qd += qdd dt  (integrator.hpp:169-181)
  T qd_new = qd + qdd * dt;
  T q_new = q;

  if (mdl->has_plane && mdl->num_cp > 0) {
    if (di >= 0) xr[nq + di] = qd_new;
    TDS_STAMP(6);
    // ---- G. CRBA entries: M_ii = s_i.F_i,  M_ij = F_i.s_j for ancestors j (mass_matrix.hpp:87-109)
    T *const Fs = E + L.F;    // [6][NLp]
    T *const Ms = E + L.M;    // [nd][NDs]  lower: M / LDL^T workspace, strict upper: L^T
    T *const dinv = E + L.dinv;
    for (int i = lane; i < nd * NDs; i += G) Ms[i] = T(0);
    if (isl) {
#pragma unroll
      for (int k = 0; k < 6; ++k) Fs[k * NLp + li] = Fc[k];
    }
    __syncthreads();
    if (di >= 0) Ms[di * NDs + di] = dot3(sw, Fc) + dot3(sw + 3, Fc + 3);
    const int npairs = mdl->num_pairs;
    for (int pi = lane; pi < npairs; pi += G) {
      const int i = mdl->pair_i[pi], j = mdl->pair_j[pi];
      const int dI = mdl->qd_index[i], dJ = mdl->qd_index[j];
      T s = T(0);
#pragma unroll
      for (int k = 0; k < 6; ++k) s += Fs[k * NLp + i] * swd[k * NDs + dJ];
      Ms[dI * NDs + dJ] = s;  // lower triangle only (dJ < dI)
    }
    __syncthreads();

    TDS_STAMP(7);
    // ---- H. M = L D L^T, right-looking, lane == row.  L[r][k] is written to the strict upper
    //         triangle slot [k][r] so that column k stays readable during step k.
    {
      const int r = lane;
      for (int k = 0; k < nd; ++k) {
        const T dk = Ms[k * NDs + k];
        const T inv = rcp_full<T>(dk);
        if (r > k && r < nd) {
          const T l = Ms[r * NDs + k] * inv;
          for (int c = k + 1; c <= r; ++c) Ms[r * NDs + c] -= l * Ms[c * NDs + k];
          Ms[k * NDs + r] = l;
        }
        if (r == k) dinv[k] = inv;
        __syncthreads();
      }
    }

    TDS_STAMP(8);
    // ---- I. narrowphase: plane vs sphere points (world.hpp:206-282, contact_point.hpp:96-125),
    //         compaction of penetrating points in contact order
    T *const cpx = E + L.cp;  // [5][NCPp]: point_on_b (3), distance, link
    const int NCPp = L.NCPp;
    const int ncp = mdl->num_cp;
    int na = 0;
    for (int base = 0; base < ncp; base += G) {
      const int k = base + lane;
      bool act = false;
      T Pb[3] = {T(0), T(0), T(0)}, dist = T(0);
      int lk = -1;
      if (k < ncp) {
        lk = mdl->cp_link[k];
        T Rl[9], pl[3];
        if (lk >= 0) {
#pragma unroll
          for (int c = 0; c < 9; ++c) Rl[c] = Xw[c * NLp + lk];
#pragma unroll
          for (int c = 0; c < 3; ++c) pl[c] = Xw[(9 + c) * NLp + lk];
        } else {
#pragma unroll
          for (int c = 0; c < 9; ++c) Rl[c] = mdl->base_R[c];
#pragma unroll
          for (int c = 0; c < 3; ++c) pl[c] = mdl->base_t[c];
        }
        const T loc[3] = {mdl->cp_local[0][k], mdl->cp_local[1][k], mdl->cp_local[2][k]};
        T ctr[3];
        mat3_mulv(Rl, loc, ctr);
        ctr[0] += pl[0];
        ctr[1] += pl[1];
        ctr[2] += pl[2];
        const T n[3] = {mdl->plane_n[0], mdl->plane_n[1], mdl->plane_n[2]};
        const T rad = mdl->cp_radius[k];
        // t = -(dot(p, -n) + c);  distance = t - r;  point_on_b = p - r n
        const T t = -((-dot3(ctr, n)) + mdl->plane_c);
        dist = t - rad;
        Pb[0] = ctr[0] - rad * n[0];
        Pb[1] = ctr[1] - rad * n[1];
        Pb[2] = ctr[2] - rad * n[2];
        act = dist < T(0);  // collision mask, mb_constraint_solver.hpp:285
      }
      const unsigned long long bal = __ballot(act);
      const unsigned long long mine = (G == 64) ? bal : ((bal >> (grp * G)) & ((1ull << (G & 63)) - 1ull));
      const int pre = __popcll(mine & ((1ull << lane) - 1ull));
      if (act) {
        const int slot = na + pre;
        cpx[0 * NCPp + slot] = Pb[0];
        cpx[1 * NCPp + slot] = Pb[1];
        cpx[2 * NCPp + slot] = Pb[2];
        cpx[3 * NCPp + slot] = dist;
        cpx[4 * NCPp + slot] = T(lk);
      }
      na += __popcll(mine);
    }
    __syncthreads();  // J/B alias the sweep arrays (IA, pA, Ic, v, F): all of those are dead now

    TDS_STAMP(9);
    // ---- J. constraint Jacobian rows (jacobian.hpp:13-83, mb_constraint_solver.hpp:278-388)
    //         row a: normal, na+a: tangent 1, 2na+a: tangent 2;  lane == dof
    T *const Js = E + L.J;  // [3na][NDs]
    T *const Bs = E + L.B;  // [3na][NDs]
    T *const rowb = E + L.rowb;
    T *const rowai = E + L.rowai;
    T *const rowx = E + L.rowx;
    const int nr = 3 * na;
    const T nb[3] = {mdl->nb[0], mdl->nb[1], mdl->nb[2]};
    const T t1[3] = {mdl->t1[0], mdl->t1[1], mdl->t1[2]};
    const T t2[3] = {mdl->t2[0], mdl->t2[1], mdl->t2[2]};
    {
      const int d = lane;
      T sd[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) sd[k] = d < nd ? swd[k * NDs + d] : T(0);
      for (int a = 0; a < na; ++a) {
        const T P[3] = {cpx[0 * NCPp + a], cpx[1 * NCPp + a], cpx[2 * NCPp + a]};
        const int lk = (int)cpx[4 * NCPp + a];
        const unsigned msk = lk >= 0 ? mdl->anc_dofs[lk] : 0u;
        if (d < nd) {
          T col[3] = {T(0), T(0), T(0)};
          if ((msk >> d) & 1u) {  // xs.bottom = st.bottom - point x st.top
            T c[3];
            cross3(P, sd, c);
            col[0] = sd[3] - c[0];
            col[1] = sd[4] - c[1];
            col[2] = sd[5] - c[2];
          }
          Js[a * NDs + d] = dot3(nb, col);
          Js[(na + a) * NDs + d] = dot3(t1, col);
          Js[(2 * na + a) * NDs + d] = dot3(t2, col);
        }
      }
    }
    __syncthreads();

    TDS_STAMP(10);
    // ---- K. per row: b, B = M^-1 J^T (L D L^T solves), 1/(J.B + cfm); lane == row -----------
    const T cfm = mdl->cfm, erp_dt = mdl->erp_over_dt, rest = mdl->restitution;
    for (int r = lane; r < nr; r += G) {
      const T *Jr = Js + r * NDs;
      T *Br = Bs + r * NDs;
      T vrow = T(0);
      for (int d = 0; d < nd; ++d) vrow += Jr[d] * xr[nq + d];
      // rel_vel = vel_a - vel_b = -J qd:  b_n = -(1+e) n.rel_vel - erp dist/dt,  b_t = -t.rel_vel
      T b;
      if (r < na)
        b = (T(1) + rest) * vrow - erp_dt * cpx[3 * NCPp + r];
      else
        b = vrow;
      rowb[r] = b;
      // forward: L z = J_r^T     (L[k][j] lives at Ms[j][k], j < k)
      for (int k = 0; k < nd; ++k) {
        T s = Jr[k];
        for (int j = 0; j < k; ++j) s -= Ms[j * NDs + k] * Br[j];
        Br[k] = s;
      }
      for (int k = 0; k < nd; ++k) Br[k] *= dinv[k];
      // backward: L^T y = z
      for (int k = nd - 1; k >= 0; --k) {
        T s = Br[k];
        for (int j = k + 1; j < nd; ++j) s -= Ms[k * NDs + j] * Br[j];
        Br[k] = s;
      }
      T g = T(0);
      for (int d = 0; d < nd; ++d) g += Jr[d] * Br[d];
      rowai[r] = rcp_full<T>(g + cfm);
      rowai[nr + r] = g;
      rowx[r] = T(0);
    }
    __syncthreads();

    TDS_STAMP(11);
    // ---- L. projected Gauss-Seidel on w = M^-1 J^T x  (mb_constraint_solver.hpp:101-142);
    //         lane == dof holds w_d
    {
      const int d = lane;
      T w = T(0);
      const T mu = mdl->friction;
      const int iters = mdl->pgs_iterations;
      for (int it = 0; it < iters; ++it) {
        for (int r = 0; r < nr; ++r) {
          const T jr = d < nd ? Js[r * NDs + d] : T(0);
          const T br = d < nd ? Bs[r * NDs + d] : T(0);
          const T jw = group_sum<T, G>(jr * w);
          const T x_old = rowx[r];
          const T delta = jw - rowai[nr + r] * x_old;
          T xn = (rowb[r] - delta) * rowai[r];
          T lo, hi;
          if (r < na) {
            lo = T(0);
            hi = T(100000);
          } else {
            const int dep = r < 2 * na ? r - na : r - 2 * na;
            T s = rowx[dep];
            s = s < T(0) ? T(0) : s;
            lo = -mu * s;
            hi = mu * s;
          }
          xn = xn > lo ? xn : lo;  // Algebra::max(x, lo*s)
          xn = xn < hi ? xn : hi;  // Algebra::min(x, hi*s)
          w += br * (xn - x_old);
          __builtin_amdgcn_wave_barrier();
          rowx[r] = xn;
          __builtin_amdgcn_wave_barrier();
        }
      }
      // qd_b -= M_b^-1 J^T p  (mb_constraint_solver.hpp:476-496) -- lane d <-> link with that dof
      if (d < nd) xr[nq + d] -= w;
    }
    __syncthreads();
    if (di >= 0) qd_new = xr[nq + di];
  }

  TDS_STAMP(12);
  // ---- M. integrate_euler: q += qd dt (integrator.hpp:126-131) and pack y -------------------
  q_new = q + qd_new * dt;

  // ---- N. observation record [q | qd (obs[0] = obs[1] = 0) | reward | done]
  //         (ars_vectorized_environment.h:250-289; ant_environment2.h:75-106;
  //          laikago_environment2.h:130-171)
  if (obs_out != nullptr) {
    __syncthreads();
    if (di == 0) xr[nq + nd] = q;  // x_{t-1}; the action slots are dead by now
    if (di >= 0) xr[di] = q_new;
    __syncthreads();
    if (valid) {
      T *const ob = obs_out + (size_t)env * (nq + nd + 2);
      if (di >= 0) {
        ob[di] = di < 2 ? T(0) : q_new;
        ob[nq + di] = qd_new;
      }
      if (lane == 0) {
        T reward = T(0);
        bool done = false;
        const int rm = mdl->reward_mode;
        if (rm == TDS_REWARD_ANT && nq > 2) {
          const T vel_x = (xr[0] - xr[nq + nd]) / dt;
          done = xr[2] < T(0.26);
          reward = done ? T(0) : vel_x;
        } else if (rm == TDS_REWARD_LAIKAGO && nq > 5) {
          // up_dot_world_z = quat_to_matrix(quat_from_euler_rpy(q[3..5]))(2,2)
          // (tiny_quaternion.h set_euler_rpy, tiny_matrix3x3.h:315-340)
          T sp, cp, st, ct, ss, cs2;
          sincos_t<T>(xr[3] * T(0.5), &sp, &cp);
          sincos_t<T>(xr[4] * T(0.5), &st, &ct);
          sincos_t<T>(xr[5] * T(0.5), &ss, &cs2);
          const T qx = sp * ct * cs2 - cp * st * ss;
          const T qy = cp * st * cs2 + sp * ct * ss;
          const T qz = cp * ct * ss - sp * st * cs2;
          const T qw = cp * ct * cs2 + sp * st * ss;
          const T s2 = T(2) / (qx * qx + qy * qy + qz * qz + qw * qw);
          const T up = T(1) - (qx * (qx * s2) + qy * (qy * s2));
          done = (up < T(0.6)) || (xr[2] < T(0.2));
          reward = done ? T(0) : xr[0];
        }
        ob[nq + nd] = reward;
        ob[nq + nd + 1] = done ? T(1) : T(0);
      }
    }
  }

  T *const yo = y_out + (size_t)env * out_dim;
  if (valid) {
    if (di >= 0) {
      yo[di] = q_new;
      yo[nq + di] = qd_new;
      if (x_feedback != nullptr) {
        x_feedback[(size_t)env * in_dim + di] = q_new;
        x_feedback[(size_t)env * in_dim + nq + di] = qd_new;
      }
    }
    const int nv = mdl->num_visuals;
    const int vbase = nq + nd;
    for (int k = lane; k < nv; k += G) {
      const int lk = mdl->vis_link[k];
      T Rl[9], pl[3], Rv[9], pv[3];
#pragma unroll
      for (int c = 0; c < 9; ++c) Rl[c] = Xw[c * NLp + lk];
#pragma unroll
      for (int c = 0; c < 3; ++c) pl[c] = Xw[(9 + c) * NLp + lk];
#pragma unroll
      for (int c = 0; c < 9; ++c) Rv[c] = mdl->vis_X[c][k];
#pragma unroll
      for (int c = 0; c < 3; ++c) pv[c] = mdl->vis_X[9 + c][k];
      T Ro[9], po[3], qo[4];
      mat3_mul(Rl, Rv, Ro);
      mat3_mulv(Rl, pv, po);
      matrix_to_quat(Ro, qo);
      T *o = yo + vbase + 7 * k;
      o[0] = pl[0] + po[0];
      o[1] = pl[1] + po[1];
      o[2] = pl[2] + po[2];
      o[3] = qo[0];
      o[4] = qo[1];
      o[5] = qo[2];
      o[6] = qo[3];
    }
    int tail = vbase;
    if (mdl->pack_visuals) {
      tail = vbase + 7 * nv;
      if (lane == 0) yo[tail] = mdl->base_R[8];  // up_dot_world_z
      tail += 1;
    }
    for (int i = tail + lane; i < out_dim; i += G) yo[i] = T(0);
  }
  TDS_STAMP(13);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host side: LDS layout + launch
// ------------------------------------------------------------------------------------------
template <typename T>
TdsLds tds_make_lds_layout(const DevModel<T> &m) {
  TdsLds L;
  memset(&L, 0, sizeof(L));
  const int nl = m.num_links, nd = m.dof_qd;
  L.NLp = nl;
  L.NDs = nd | 1;  // odd row stride: lane == row accesses hit distinct LDS banks
  const int ncp = m.has_plane ? m.num_cp : 0;
  L.NCPp = ncp > 0 ? ncp : 1;
  const int nr = 3 * ncp;
  int o = 0;
  L.xrec = o; o += m.input_dim;
  L.Xw = o;   o += 12 * L.NLp;
  L.swd = o;  o += 6 * L.NDs;
  L.M = o;    o += ncp ? nd * L.NDs : 0;
  L.dinv = o; o += ncp ? nd : 0;
  L.cp = o;   o += ncp ? 5 * L.NCPp : 0;
  L.rowb = o; o += nr;
  L.rowai = o; o += 2 * nr;
  L.rowx = o; o += nr;
  // union { sweep arrays } / { J, B }
  const int u = o;
  int s = u;
  L.v = s;  s += 6 * L.NLp;
  L.IA = s; s += 21 * L.NLp;
  L.pA = s; s += 6 * L.NLp;
  L.Ic = s; s += 10 * L.NLp;
  L.F = s;  s += 6 * L.NLp;
  int j = u;
  L.J = j; j += nr * L.NDs;
  L.B = j; j += nr * L.NDs;
  o = s > j ? s : j;
  o = (o + 1) & ~1;  // keep 16-byte alignment of every env region for T = double
  L.stride = o;
  return L;
}

template <typename T>
int tds_launch_step(const DevModel<T> *d_model, const DevModel<T> &h_model, const TdsLds &L, int lanes_per_env,
                    const T *x_in, T *y_out, const T *actions, T *x_feedback, T *obs_out, int n_envs,
                    hipStream_t stream, long long *prof) {
  const int epw = 64 / lanes_per_env;
  const int blocks = (n_envs + epw - 1) / epw;
  const size_t shmem = (size_t)L.stride * epw * sizeof(T);
  (void)h_model;
  switch (lanes_per_env) {
    case 64:
      if (prof) hipLaunchKernelGGL((tds_step_kernel<T, 64, true>), dim3(blocks), dim3(64), shmem, stream, d_model, L, x_in, y_out, actions, x_feedback, obs_out, prof, n_envs);
      else hipLaunchKernelGGL((tds_step_kernel<T, 64, false>), dim3(blocks), dim3(64), shmem, stream, d_model, L, x_in, y_out, actions, x_feedback, obs_out, prof, n_envs);
      break;
    case 32:
      if (prof) hipLaunchKernelGGL((tds_step_kernel<T, 32, true>), dim3(blocks), dim3(64), shmem, stream, d_model, L, x_in, y_out, actions, x_feedback, obs_out, prof, n_envs);
      else hipLaunchKernelGGL((tds_step_kernel<T, 32, false>), dim3(blocks), dim3(64), shmem, stream, d_model, L, x_in, y_out, actions, x_feedback, obs_out, prof, n_envs);
      break;
    case 16:
      if (prof) hipLaunchKernelGGL((tds_step_kernel<T, 16, true>), dim3(blocks), dim3(64), shmem, stream, d_model, L, x_in, y_out, actions, x_feedback, obs_out, prof, n_envs);
      else hipLaunchKernelGGL((tds_step_kernel<T, 16, false>), dim3(blocks), dim3(64), shmem, stream, d_model, L, x_in, y_out, actions, x_feedback, obs_out, prof, n_envs);
      break;
    default:
      return -1;
  }
  return (int)hipGetLastError();
}

template <typename T>
int tds_kernel_max_dynamic_lds(int lanes_per_env, int bytes) {
  hipError_t e = hipSuccess;
  switch (lanes_per_env) {
    case 64: e = hipFuncSetAttribute((const void *)tds_step_kernel<T, 64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void *)tds_step_kernel<T, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); break;
    case 32: e = hipFuncSetAttribute((const void *)tds_step_kernel<T, 32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void *)tds_step_kernel<T, 32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); break;
    case 16: e = hipFuncSetAttribute((const void *)tds_step_kernel<T, 16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void *)tds_step_kernel<T, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); break;
    default: return -1;
  }
  return (int)e;
}

template TdsLds tds_make_lds_layout<double>(const DevModel<double> &);
template TdsLds tds_make_lds_layout<float>(const DevModel<float> &);
template int tds_launch_step<double>(const DevModel<double> *, const DevModel<double> &, const TdsLds &, int, const double *, double *, const double *, double *, double *, int, hipStream_t, long long *);
template int tds_launch_step<float>(const DevModel<float> *, const DevModel<float> &, const TdsLds &, int, const float *, float *, const float *, float *, float *, int, hipStream_t, long long *);
template int tds_kernel_max_dynamic_lds<double>(int, int);
template int tds_kernel_max_dynamic_lds<float>(int, int);
