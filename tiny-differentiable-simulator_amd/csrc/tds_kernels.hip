// tds_kernels.hip — the MI355X (gfx950 / CDNA4) step kernel.
//
// One launch advances N independent environments by one step of the reference's
//   PD -> forward_dynamics -> integrate_euler_qdd -> World::step (plane contacts +
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
// tests/ against oracle/ and the golden vectors; DESIGN.md "Reformulation"):
//   * all spatial quantities are expressed in WORLD coordinates about the world origin, so nothing is
//     transformed between link frames (the reference transforms link-to-parent with dense 6x6x6
//     products, src/dynamics/forward_dynamics.hpp:187-189, src/dynamics/mass_matrix.hpp:45-46);
//   * rigid and composite inertias are (I_sym[6], h[3], m) = 10 numbers; no 6x6 matrix travels;
//   * forward dynamics goes through the joint-space inertia the contact solve needs anyway:
//     M = L D L^T (no square roots, in registers), qdd = M^-1 (tau - C); the bias forces C ride on the
//     composite-inertia (CRBA) sweep — no articulated-body recursion;
//   * serial chains hand their sweep state from lane to lane with DPP row shifts; the massless
//     "virtual" base chain is collapsed (prefix-product kinematics, one broadcast of the torso's
//     composite inertia / force instead of six tree levels);
//   * constraint rows are stored as z~_r = D^-1/2 L^-1 J_r^T, so A = J M^-1 J^T = Z Z^T is never formed
//     (51x51 for Ant): PGS runs on u~ = sum_r z~_r x_r, delta_i = z~_i.u~ - G_ii x_i, and the final
//     qd -= M^-1 J^T p is one back-substitution L^-T D^-1/2 u~;
//   * rows of separated contacts (distance >= 0) are identically zero in the reference
//     (keep_all_points_, src/mb_constraint_solver.hpp:285-291) and yield x = 0, so only
//     penetrating contacts are materialised, in the reference's row order (wave-uniform slots);
//   * three kernel kinds (template parameter KIND): 0 fixed base with 1-dof joints, 1 floating base (six pseudo
//     links, the reference's base-frame ABA incl. its block inverse), 2 spherical joints (three lanes per joint);
//     JOINT_FIXED links may be folded into their parents on the host (tds_device_model.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <type_traits>

#include "tds_device_model.h"
#include "tds_kernels.h"
#include "tds_lanes.h"
// two-wavefront workgroups of the narrow kernels: the generalised force of the PD block waits in LDS for phase F
// (0: carried in registers)
#ifndef TDS_PARK_W2
#define TDS_PARK_W2 1
#endif
// scope of the peer-store exchange's stores into the other ranks' rings (experiments on one GPU may lower it)
#ifndef TDS_PEER_SCOPE
#define TDS_PEER_SCOPE __HIP_MEMORY_SCOPE_SYSTEM
#endif
#ifndef TDS_PARK_LOOP
#define TDS_PARK_LOOP 0
#endif



namespace {

// ------------------------------------------------------------------------------------------
// constraint-row helpers.  SLAB = false: every row of the wavefront's environments fits the LDS
// row store (the common case) and the code touches LDS only.  SLAB = true: rows >= ZR live in the
// global scratch slab; those accesses go through volatile pointers so that hipcc can never fold
// "LDS row or slab row" into one FLAT access through a selected pointer (that cost 3x on PGS).
// ------------------------------------------------------------------------------------------
// per row (lane == row): b_r, forward substitution L z = J_r^T in registers, G_rr = z.D^-1.z,
// 1/(G_rr + cfm); the row is stored back as z~ = D^-1/2 z so that A_rs = J_r M^-1 J_s^T = z~_r.z~_s
// SPLIT (two-wavefront workgroups, run by the helper wavefront while the main one is still in the forward-dynamics
// solve): qdv holds the velocities BEFORE integrate_euler_qdd, and the b slot receives only J_r . qd_pre; the main
// wavefront completes it with the acceleration part afterwards (tds_row_rhs_finish).
// SPLIT with yt_flag != nullptr: the helper also completes b_r itself, with z~_r still in registers, as soon as the main
// wavefront has published y~ (flag in LDS, polled: the main wavefront is ~2 k cycles ahead at that point) — same
// arithmetic as tds_row_rhs_finish, which the main wavefront then skips.
// LDS flags between the two wavefronts of a workgroup.  Through a `volatile T *` (generic address space: the address-space
// inference leaves volatile accesses alone) every one of them was a FLAT instruction with sc0 sc1 and an `s_waitcnt vmcnt(0)`
// of its own — eight flat stores + waits in the middle of the main wavefront's LDL^T (the hand-over of the first half of L),
// a flat load + wait per poll.  These go to the LDS address space explicitly: ds_read / ds_write, lgkmcnt only.
// (the main wavefront's polls; the helper's stay volatile generic loads: as DS reads that build spilled and lost 5 % —
//  tools/experiments/r05_not_kept.txt)
#ifndef TDS_LDS_PUBLISH
#define TDS_LDS_PUBLISH 1
#endif
#define TDS_AS3 __attribute__((address_space(3)))
// A pointer that was LOADED (a field of the kernel-argument structs read through the laundered segment pointer, an entry of a
// pointer table) has no address space the compiler could know: its accesses are FLAT instructions — both counters, out of
// order with the DS instructions, and a flat LOAD (the action block requested a step ahead) holds the next LDS wait until it
// has returned from memory.  tds_global() says "global memory" (an assumption `neither LDS nor scratch`, which the
// address-space inference pass turns into address space 1 for every access derived from the pointer).  -DTDS_RINGS_GLOBAL=0: the loaded pointers as they are.
#ifndef TDS_RINGS_GLOBAL
#define TDS_RINGS_GLOBAL 1
#endif
template <typename P>
__device__ __forceinline__ P *tds_global(P *p) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (TDS_RINGS_GLOBAL != 0)
    __builtin_assume(!__builtin_amdgcn_is_shared((const void *)p) && !__builtin_amdgcn_is_private((const void *)p));
#endif
  return p;
}
template <int WHO = 1, typename T>
__device__ __forceinline__ T tds_lds_poll(const T *p) {  // WHO: 1 the main wavefront's polls, 2 the helper's
  if constexpr (WHO == 1) return *(const volatile TDS_AS3 T *)p;
  else return *(const volatile T *)p;
}
template <typename T>
__device__ __forceinline__ void tds_lds_flag(T *p, T v) {
  if constexpr (TDS_LDS_PUBLISH != 0) *(volatile TDS_AS3 T *)p = v;
  else *(volatile T *)p = v;
}
template <bool SLAB, typename T, int G, int NDP, bool SPLIT = false>
__device__ __forceinline__ void tds_row_solve(int lane, int NA, int na, int nd, int ZR, int OVR, int NCPp,
                                              T *Zs, T *rws, T *xs, const T *qdv, const T *cpx, const T *Lp,
                                              const T *dvec, volatile T *zov, volatile T *rov, T cfm, T erp_dt,
                                              T rest, const T *yt_flag = nullptr, T dt = T(0),
                                              const T *col_flag = nullptr, const T *Lh = nullptr) {
  constexpr int NDs = NDP + 1;
  // ROW LAYOUT (wave-uniform): NA = largest number of penetrating contacts among the wavefront's
  // environments; row a = normal of contact a, NA + a = tangent 1, 2 NA + a = tangent 2.  An environment
  // with fewer contacts has empty slots a >= na: they are stored as all-zero rows (x = 0, no effect),
  // so that the Gauss-Seidel loop needs no per-environment control flow.  The order among an
  // environment's real rows is the reference's (normals, tangent 1, tangent 2, each by contact).
  const int nrw = 3 * NA;
  for (int r = lane; r < nrw; r += G) {
    const int t = (r >= NA ? 1 : 0) + (r >= 2 * NA ? 1 : 0);
    const int a = r - t * NA;
    const bool real = a < na;
    bool in_lds = true;
    if constexpr (SLAB) in_lds = r < ZR;
    T z[NDP];
    T brow = T(0), ai = T(0), g = T(0);
    if (real) {
      if (in_lds) {
#pragma unroll
        for (int k = 0; k < NDP; ++k) z[k] = Zs[r * NDs + k];
      } else {
        volatile T *const Zr = zov + (size_t)(r - ZR) * NDs;
#pragma unroll
        for (int k = 0; k < NDP; ++k) z[k] = Zr[k];
      }
      T vrow = T(0);
#pragma unroll
      // (padding columns k >= nd: z[k] == 0 exactly, and the velocity slot read for them is a real one (index clamped):
      //  behind the velocities of the record lie slots nobody writes in the straight-line kernels, and 0 x stale LDS is
      //  NaN once in a while — seen twice in ~40 suite runs.  A term under `if (k < nd)` would be a uniform branch with
      //  its own LDS round trip.)
      for (int k = 0; k < NDP; ++k) vrow += z[k] * qdv[k < nd ? k : 0];
      // rel_vel = vel_a - vel_b = -J qd:  b_n = -(1+e) n.rel_vel - erp dist/dt,  b_t = -t.rel_vel
      if constexpr (SPLIT)
        brow = vrow;
      else
        brow = t == 0 ? (T(1) + rest) * vrow - erp_dt * cpx[3 * NCPp + a] : vrow;
      // column-oriented: once z[j] is final, every later z[k] takes its update independently.
      // col_flag (two-wavefront workgroups): the main wavefront is still factorising — it raises the flag to 1 when
      // columns 0 .. NDP/2 - 1 of L are in LDS and to 2 when all of L and D are: the first three quarters of this
      // substitution run in the shadow of the second half of the LDL^T instead of after it.
#pragma unroll
      for (int j = 0; j < NDP - 1; ++j) {
        if constexpr (SPLIT && !SLAB) {
          if (col_flag != nullptr && (j == 0 || j == NDP / 2)) {  // wave-uniform
            const T need = j == 0 ? T(1) : T(2);
            while (__any(tds_lds_poll<2>(col_flag) < need)) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (compile-time order of the LDS reads only)
            __builtin_amdgcn_wave_barrier();
          }
        }
#pragma unroll
        for (int k = j + 1; k < NDP; ++k) {
          // (two-wavefront pipeline: the first NDP/2 columns arrive early in a row-major copy, stride NDP/2)
          const T lkj = (SPLIT && !SLAB && Lh != nullptr && j < NDP / 2) ? Lh[k * (NDP / 2) + j] : Lp[(k * (k - 1)) / 2 + j];
          z[k] -= lkj * z[j];
        }
      }
#pragma unroll
      for (int k = 0; k < NDP; ++k) {
        z[k] *= dvec[NDP + k];
        g += z[k] * z[k];
      }
      ai = rcp_full<T>(g + cfm);
    }
    // (measured and not kept, -DTDS_ROWS_EARLY in round 5's experiments: the row, G_rr and 1 / (G_rr + cfm) stored BEFORE the wait
    //  for the main wavefront's y~ — seventeen LDS stores off the helper's stretch between that flag and barrier (3) on paper,
    //  11.79 against 11.60 us per step in the same process)
    if constexpr (SPLIT && !SLAB) {
      if (yt_flag != nullptr) {  // wave-uniform
        while (__any(tds_lds_poll<2>(yt_flag) == T(0))) __builtin_amdgcn_s_sleep(1);
        if (real) {
          const T *const yt = dvec + 3 * NDP;
          T s = T(0);
#pragma unroll
          for (int k = 0; k < NDP; ++k) s += z[k] * yt[k];
          const T vrow = brow + dt * s;
          brow = t == 0 ? (T(1) + rest) * vrow - erp_dt * cpx[3 * NCPp + a] : vrow;
        }
      }
    }
    if (!real) {
#pragma unroll
      for (int k = 0; k < NDP; ++k) z[k] = T(0);
    }
    if (in_lds) {
#pragma unroll
      for (int k = 0; k < NDP; ++k) Zs[r * NDs + k] = z[k];
      rws[r] = brow;
      rws[ZR + r] = ai;
      rws[2 * ZR + r] = g;
    } else if (zov != nullptr) {  // (groups without a live environment have no slab)
      volatile T *const Zr = zov + (size_t)(r - ZR) * NDs;
#pragma unroll
      for (int k = 0; k < NDP; ++k) Zr[k] = z[k];
      rov[r - ZR] = brow;
      rov[OVR + r - ZR] = ai;
      rov[2 * OVR + r - ZR] = g;
    }
  }
}

// The row solve of 32-lane groups with TWO lanes per constraint row: lane r (DPP row 0 of the group) keeps the even
// components z[0], z[2], ... of row r, lane 16 + r (DPP row 1) the odd ones.  The column-oriented substitution needs the
// pivot z[j] on both — one v_permlane16_swap pair per step — and every lane then updates only ITS components: half the
// multiply-adds per lane and, what matters, half the LDS reads of L per wavefront (the 18-dof kernels' row solves are bound
// by the LDS bandwidth of eight wavefronts per CU reading all 153 entries of L each).  Rows of ONE pass
// (3 NA <= 16), every row in LDS, one-wave form; same arithmetic per component as tds_row_solve, the sums over the
// components (J.qd, G_rr) in two halves.
template <typename T, int NDP>
__device__ __forceinline__ void tds_row_solve_half(int lane, int NA, int na, int nd, int ZR, int NCPp, T *Zs, T *rws,
                                                   const T *qdv, const T *cpx, const T *Lp, const T *dvec, T cfm, T erp_dt,
                                                   T rest) {
  constexpr int NDs = NDP + 1;
  constexpr int NH = (NDP + 1) / 2;  // components per lane
  const bool upper = (threadIdx.x & 16) != 0;
  const int p = upper ? 1 : 0;       // parity of this lane's components: k = 2 i + p
  const int r = lane & 15;
  const int nrw = 3 * NA;
  const bool rowl = r < nrw;
  const int rc = rowl ? r : 0;
  const int t = (rc >= NA ? 1 : 0) + (rc >= 2 * NA ? 1 : 0);
  const int a = rc - t * NA;
  const bool real = rowl && a < na;
  T z[NH];
  T vpart = T(0);
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int k = 2 * i + p;
    const int kc = k < NDP ? k : NDP - 1;  // (odd NDP: the last odd slot does not exist — NDP is even here, kept general)
    const T v = Zs[rc * NDs + kc];
    z[i] = (real && k < NDP) ? v : T(0);
    vpart += z[i] * qdv[kc < nd ? kc : 0];
  }
  const T vrow = vpart + other_row(vpart, upper);
  const T brow = t == 0 ? (T(1) + rest) * vrow - erp_dt * cpx[3 * NCPp + a] : vrow;
  // column j: the pivot z[j] to both lanes of the row, then every component k > j of this lane
  static_for<0, NDP - 1>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const T mine = z[j / 2];
    const T theirs = other_row(mine, upper);
    const T zj = (p == (j & 1)) ? mine : theirs;
    static_for<0, NH>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      // k = 2 i + p > j:  2 i > j for both parities; 2 i == j only for the odd lane (k = j + 1); else neither
      if constexpr (2 * i + 1 > j) {
        const int k = 2 * i + p;
        const int kc = k < NDP ? k : NDP - 1;
        const T l = Lp[(kc * (kc - 1)) / 2 + j];
        const bool on = (2 * i > j || p == 1) && k < NDP;
        z[i] -= (on ? l : T(0)) * zj;
      }
    });
  });
  T gpart = T(0);
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int k = 2 * i + p;
    const int kc = k < NDP ? k : NDP - 1;
    z[i] *= dvec[NDP + kc];
    gpart += z[i] * z[i];
  }
  const T g = gpart + other_row(gpart, upper);
  const T ai = real ? rcp_full<T>(g + cfm) : T(0);
  if (rowl) {
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int k = 2 * i + p;
      if (k < NDP) Zs[rc * NDs + k] = z[i];
    }
    if (!upper) {
      rws[rc] = real ? brow : T(0);
      rws[ZR + rc] = ai;
      rws[2 * ZR + rc] = real ? g : T(0);
    }
  }
}

// Right-hand sides of the constraint rows after a SPLIT row solve (lane == row).  The velocity the reference uses is
// the one after integrate_euler_qdd, qd+ = qd + dt qdd with qdd = L^-T D^-1 y (y = L^-1 (tau - C)), hence
//   J_r . qd+ = J_r . qd + dt z~_r . y~,   y~ = D^-1/2 y   (z~_r = D^-1/2 L^-1 J_r^T is what the row store holds)
//   b_n = (1 + e) J_r.qd+ - erp dist / dt,   b_t = J_r.qd+        (mb_constraint_solver.hpp:309-372)
template <bool SLAB, typename T, int G, int NDP>
__device__ __forceinline__ void tds_row_rhs_finish(int lane, int NA, int na, int ZR, int OVR, int NCPp, const T *Zs,
                                                   T *rws, const T *cpx, const T *yt, volatile T *zov, volatile T *rov,
                                                   T dt, T erp_dt, T rest) {
  constexpr int NDs = NDP + 1;
  const int nrw = 3 * NA;
  for (int r = lane; r < nrw; r += G) {
    const int t = (r >= NA ? 1 : 0) + (r >= 2 * NA ? 1 : 0);
    const int a = r - t * NA;
    bool in_lds = true;
    if constexpr (SLAB) in_lds = r < ZR;
    if (a < na) {
      T s = T(0), b0;
      if (in_lds) {
#pragma unroll
        for (int k = 0; k < NDP; ++k) s += Zs[r * NDs + k] * yt[k];
        b0 = rws[r];
      } else {
        volatile T *const Zr = zov + (size_t)(r - ZR) * NDs;
#pragma unroll
        for (int k = 0; k < NDP; ++k) s += Zr[k] * yt[k];
        b0 = rov[r - ZR];
      }
      const T vrow = b0 + dt * s;
      const T b = t == 0 ? (T(1) + rest) * vrow - erp_dt * cpx[3 * NCPp + a] : vrow;
      if (in_lds)
        rws[r] = b;
      else
        rov[r - ZR] = b;
    }
  }
}

// projected Gauss-Seidel (mb_constraint_solver.hpp:101-142) on u~ = sum_r z~_r x_r:
//   delta_i = sum_{j != i} A_ij x_j = z~_i . u~ - G_ii x_i;   lane == dof holds u~_k (returned)
template <bool SLAB, typename T, int G, int NDP>
__device__ __forceinline__ T tds_pgs(int lane, int NA, int ZR, int OVR, int iters, T mu, const T *Zs,
                                     const T *rws, T *xs, volatile const T *zov, volatile const T *rov) {
  constexpr int NDs = NDP + 1;
  const int d = lane;
  const bool dz = d < NDP;
  const int nr = 3 * NA;  // wave-uniform row layout, see tds_row_solve
  T u = T(0);
  for (int it = 0; it < iters; ++it) {
    // software pipeline: everything row r+1 needs that does not depend on row r is loaded before
    // row r's cross-lane reduction, so only the reduction + clamp sit on the dependent chain
    T zn = T(0), bn = T(0), an = T(0), gn = T(0), xon = T(0);
    if (nr > 0) {
      zn = dz ? Zs[d] : T(0);
      bn = rws[0];
      an = rws[ZR];
      gn = rws[2 * ZR];
      xon = it > 0 ? xs[0] : T(0);
    }
    for (int r = 0; r < nr; ++r) {
      const T zr = zn, br = bn, ar = an, gr = gn, x_old = xon;
      const int rn = r + 1;
      if (rn < nr) {
        bool nlds = true;
        if constexpr (SLAB) nlds = rn < ZR;
        if (nlds) {
          zn = dz ? Zs[rn * NDs + d] : T(0);
          bn = rws[rn];
          an = rws[ZR + rn];
          gn = rws[2 * ZR + rn];
        } else if (zov != nullptr) {
          zn = dz ? zov[(size_t)(rn - ZR) * NDs + d] : T(0);
          bn = rov[rn - ZR];
          an = rov[OVR + rn - ZR];
          gn = rov[2 * OVR + rn - ZR];
        } else {  // group without a live environment: zero rows
          zn = bn = an = gn = T(0);
        }
        xon = it > 0 ? xs[rn] : T(0);
      }
      // friction rows scale their box by the normal impulse of the same contact
      // (limit_dependency_, mb_constraint_solver.hpp:417-436); that row is NA or 2 NA rows back
      const bool is_n = r < NA;
      const int dep = r - (r >= NA ? NA : 0) - (r >= 2 * NA ? NA : 0);
      const T sdep = xs[dep];
      const T jw = group_sum<T, G>(zr * u);
      const T delta = jw - gr * x_old;
      T xn = (br - delta) * ar;
      const T sc = sdep < T(0) ? T(0) : sdep;  // where_lt(s, 0, 0, s)
      const T lo = is_n ? T(0) : -mu * sc;
      const T hi = is_n ? T(100000) : mu * sc;
      xn = max_t<T>(xn, lo);  // Algebra::max(x, lo*s)
      xn = min_t<T>(xn, hi);  // Algebra::min(x, hi*s)
      u += zr * (xn - x_old);
      xs[r] = xn;  // (all lanes of the group, same value: see tds_pgs_sweep)
    }
  }
  return u;
}


// The common case of tds_pgs: every row of the wavefront's environments is in LDS.  With the
// wave-uniform row layout the loop has no per-environment control flow at all: empty slots are zero
// rows, the row kind (normal / friction) and the row of the limiting normal impulse are wave-uniform,
// and everything row r + 1 needs is fetched while row r is being reduced, so that the loop-carried
// chain is only: u~ -> dot -> 4 DPP adds -> clamp -> u~.
template <bool FIRST, typename T, int G, int NDP>
__device__ __forceinline__ T tds_pgs_sweep(T u, int lane, int NA, int ZR, T mu, const T *Zs, const T *rws, T *xs) {
  constexpr int NDs = NDP + 1;
  const int dcl = lane < NDP ? lane : NDP - 1;  // lanes >= NDP read a valid slot and discard it
  const bool dz = lane < NDP;
  const int last = ZR - 1;
  T zn = Zs[dcl], bn = rws[0], an = rws[ZR], gn = FIRST ? T(0) : rws[2 * ZR], xon = FIRST ? T(0) : xs[0];
  T xn = T(0), x_n0 = T(0);
  // One loop per row kind (normals, tangent 1, tangent 2; NA rows each): the bounds and the row of the limiting
  // normal impulse are then static per loop instead of selected per row — a lone wavefront issues one
  // instruction per ~4 cycles whatever its type, so the loop's instruction count IS its latency.
  // `normal`: bounds [0, 1e5]; otherwise -/+ mu max(x_normal, 0) with x_normal = xs[a] (limit_dependency_,
  // mb_constraint_solver.hpp:417-436), prefetched one row ahead like the row itself.
  auto rows = [&](auto normal_c, const int r0) {
    constexpr bool normal = decltype(normal_c)::value;
    // (with a single contact slot the normal impulse of THIS sweep comes straight from its register)
    T sn = normal ? T(0) : (NA == 1 ? x_n0 : xs[0]);
    for (int a = 0; a < NA; ++a) {
      const int r = r0 + a;
      const T zr = dz ? zn : T(0);
      const T br = bn, ar = an, gr = gn, x_old = xon, sdep = sn;
      // prefetch row r + 1 (index clamped: the loads are unconditional)
      const int rl = r + 1 < last ? r + 1 : last;
      zn = Zs[rl * NDs + dcl];
      bn = rws[rl];
      an = rws[ZR + rl];
      if constexpr (!FIRST) {
        gn = rws[2 * ZR + rl];
        xon = xs[rl];
      }
      if constexpr (!normal) sn = xs[a + 1 < NA ? a + 1 : a];
      const T jw = group_sum<T, G>(zr * u);
      T delta = jw;
      if constexpr (!FIRST) delta -= gr * x_old;
      xn = (br - delta) * ar;
      if constexpr (normal) {
        xn = max_t<T>(xn, T(0));
        xn = min_t<T>(xn, T(100000));
      } else {
        const T h = mu * (sdep > T(0) ? sdep : T(0));  // where_lt(s, 0, 0, s)
        xn = max_t<T>(xn, -h);  // Algebra::max(x, lo*s)
        xn = min_t<T>(xn, h);   // Algebra::min(x, hi*s)
      }
      if constexpr (FIRST) u += zr * xn; else u += zr * (xn - x_old);
      // every lane of the group holds the same xn and stores it (same address: no bank conflict).  A store
      // predicated on lane == 0 becomes a branch around the DS write, after which the compiler must drain
      // lgkmcnt(0) — the LDS write latency then sits on every iteration of this loop.
      xs[r] = xn;
    }
  };
  rows(std::true_type{}, 0);
  x_n0 = xn;
  rows(std::false_type{}, NA);
  rows(std::false_type{}, 2 * NA);
  return u;
}

template <typename T, int G, int NDP>
__device__ __forceinline__ T tds_pgs_lds(int lane, int NA, int ZR, int iters, T mu, const T *Zs, const T *rws,
                                         T *xs) {
  T u = tds_pgs_sweep<true, T, G, NDP>(T(0), lane, NA, ZR, mu, Zs, rws, xs);
  for (int it = 1; it < iters; ++it) u = tds_pgs_sweep<false, T, G, NDP>(u, lane, NA, ZR, mu, Zs, rws, xs);
  return u;
}

// ------------------------------------------------------------------------------------------
// hand-offs through LDS inside ONE wavefront (TDS_WAVE_SYNC)
// ------------------------------------------------------------------------------------------
// A workgroup is exactly ONE wavefront, and the LDS executes the DS instructions of a wavefront in
// issue order, so cross-lane hand-offs through LDS need no s_barrier and no s_waitcnt drain:
// all that is required is that the compiler keeps the program order of the LDS accesses.  A real
// __syncthreads() would also wait for every outstanding GLOBAL store (the early y writes), which
// single-wave-per-SIMD occupancy cannot hide.
#ifdef TDS_FULL_BARRIER
#define TDS_WAVE_SYNC() __syncthreads()
#else
#define TDS_WAVE_SYNC()                                        \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     \
    __builtin_amdgcn_wave_barrier();                           \
  } while (0)
#endif

// ------------------------------------------------------------------------------------------
// Gram form of the contact solve on the matrix cores (two-wavefront workgroups, 16 lanes per environment, double).
//
// Once the row store holds z~_r = D^-1/2 L^-1 J_r^T (phase K) everything the projected Gauss-Seidel needs is inner
// products of those rows:  A_rs = z~_r . z~_s  (= J_r M^-1 J_s^T)  and the acceleration part of the right-hand sides
// z~_r . y~.  v_mfma_f64_4x4x4_4b_f64 computes four independent 4x4x4 products per instruction; its operand map
// (probed: tools/ubench/mfma_f64_4x4x4.hip) is   A: lane = 16 k + 4 blk + i,  B: lane = 16 k + 4 blk + j,
// D: lane = 16 i + 4 blk + j   — block blk takes the environment in lane group blk, the operands come straight out of
// the environments' LDS regions, and [Z~ | y~] [Z~ | y~]^T lands in a per-environment 16 x 16 buffer with ~50 matrix
// instructions instead of ~12 cross-lane reductions per sweep and 12 more for the right-hand sides.
// The sweep then runs with lane == row and NO reduction at all: every lane keeps its row's  res_s = sum_r A_sr x_r
// up to date (one FMA per updated row, the update broadcast by DPP), so the dependent chain per row is
// sub, mul, max, min, broadcast, fma.  Same iteration as tds_pgs_sweep (mb_constraint_solver.hpp:101-142, rows in the
// reference's order: normals, tangents 1, tangents 2), different association of the sums.
// Needs 3 NA <= 15 (rows on 16 lanes, one column left for y~) and every row in LDS.
// ------------------------------------------------------------------------------------------
#define TDS_GRAM_STRIDE 17  // odd row stride of the 16 x 16 (+1) buffer
#define TDS_GRAM_ZEROS (16 * TDS_GRAM_STRIDE)  // 16 zeros behind the buffer: what masked operand lanes read
// v_max_f64 / v_min_f64 as they are (fmax / fmin would first canonicalise both operands: two more instructions on a
// path whose instruction count is its latency)
__device__ __forceinline__ double tds_vmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double tds_vmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// NA (contact slots of the wavefront) is a template parameter: the tile loops, the row kinds and the DPP sources are
// then all static and the whole solve is branch-free — every LDS read is issued up front instead of one exposed LDS
// round trip per uniform branch.
template <int NDP, int NA>
__device__ __forceinline__ double tds_gram_solve(double *sm, const TdsLds &L, int lane, int na, int ZR, int NCPp,
                                                 int iters, double mu, double dt, double erp_dt, double rest,
                                                 long long *stamp = nullptr, int stamp_at = 0) {
  using T = double;
  static_assert(3 * NA <= 15, "rows on 16 lanes, one column left for y~");
  constexpr int NDs = NDP + 1;
  constexpr int GS = TDS_GRAM_STRIDE;
  constexpr int nr = 3 * NA;          // rows 0..nr-1; column nr carries y~
  constexpr int NTr = (nr + 3) >> 2;  // row tiles
  constexpr int NTc = (nr + 4) >> 2;  // column tiles (incl. the y~ column)
  constexpr int NK = (NDP + 3) >> 2;  // k tiles
  // (diagnostic: one timestamp inside this function, selected by TDS_GRAM_STAMP_AT)
  auto mark = [&](int at) {
    if (stamp != nullptr && at == stamp_at) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0);
      if (blockIdx.x == 0 && threadIdx.x == 0) *stamp = (long long)__builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  const int wl = threadIdx.x & 63;
  const int grp = wl >> 4;
  {
    // ---- [Z~ | y~] [Z~ | y~]^T on the matrix cores
    const int kq = wl >> 4, blk = (wl >> 2) & 3, ij = wl & 3;
    const int eb = blk * L.stride;
    const int zero_at = eb + L.Xw + TDS_GRAM_ZEROS;
    T op[NTc][NK];
#pragma unroll
    for (int t = 0; t < NTc; ++t) {
      const int c = 4 * t + ij;  // row of [Z~ | y~] this lane feeds in tile t
      const int at = (c < nr ? eb + L.Z + c * NDs : (c == nr ? eb + L.dinv + 3 * NDP : zero_at)) + kq;
      const int at_last = (4 * (NK - 1) + kq < NDP) ? at + 4 * (NK - 1) : zero_at;  // (the last k tile may run past NDP)
#pragma unroll
      for (int k = 0; k < NK; ++k) op[t][k] = (4 * k + 3 < NDP) ? sm[at + 4 * k] : sm[at_last];
    }
    mark(4);
    const int out = eb + L.Xw + kq * GS + ij;  // D: row 4 I + (lane >> 4), column 4 J + (lane & 3) of block blk
    T acc[NTr][NTc];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
      for (int I = 0; I < NTr; ++I)
#pragma unroll
        for (int J = 0; J < NTc; ++J)
          acc[I][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(op[I][k], op[J][k], k == 0 ? T(0) : acc[I][J], 0, 0, 0);
    mark(5);
#pragma unroll
    for (int I = 0; I < NTr; ++I)
#pragma unroll
      for (int J = 0; J < NTc; ++J) sm[out + 4 * I * GS + 4 * J] = acc[I][J];
  }
  TDS_WAVE_SYNC();
  mark(1);
  // ---- lane == row
  T *const E = sm + grp * L.stride;
  const T *const Gm = E + L.Xw;
  const T *const rws = E + L.rows;
  const T *const cpx = E + L.cp;
  const T *const Zs = E + L.Z;
  const int s = lane;
  const int t = (s >= NA ? 1 : 0) + (s >= 2 * NA ? 1 : 0);
  const int a = s - t * NA;
  const bool row = s < nr && a < na;
  const int sc = s < nr ? s : 0;
  T Acol[nr];
#pragma unroll
  for (int r = 0; r < nr; ++r) Acol[r] = Gm[s * GS + r];
  const int dcl = lane < NDP ? lane : NDP - 1;
  T zc[nr];  // column dcl of Z~ for u~ below (requested now, used after the sweep)
#pragma unroll
  for (int r = 0; r < nr; ++r) zc[r] = Zs[r * NDs + dcl];
  // right-hand side: J_r . qd+ = J_r . qd + dt z~_r . y~ (see tds_row_rhs_finish)
  const T vrow = rws[sc] + dt * Gm[s * GS + nr];
  const T dist = cpx[3 * NCPp + (a < NCPp ? a : 0)];
  const T b = !row ? T(0) : (t == 0 ? (T(1) + rest) * vrow - erp_dt * dist : vrow);
  const T ar = row ? rws[ZR + sc] : T(0);
  const T gr = row ? rws[2 * ZR + sc] : T(0);
  T x = T(0), res = T(0);
  T lo = T(0), hi = t == 0 ? T(100000) : T(0);
  mark(2);
  for (int it = 0; it < iters; ++it) {
    static_for<0, nr>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      const T delta = it == 0 ? res : res - gr * x;  // sum over the OTHER rows (G_rr x_r taken out again)
      T xn = (b - delta) * ar;
      xn = tds_vmax(xn, lo);
      xn = tds_vmin(xn, hi);
      const T xnr = dpp_bcast<r>(xn);
      const T dxr = it == 0 ? xnr : xnr - dpp_bcast<r>(x);
      res += Acol[r] * dxr;
      x = s == r ? xn : x;
      if constexpr (r < NA) {  // a normal row: its impulse bounds the two friction rows of the same contact
        if (t != 0 && a == r) {
          const T h = mu * (xnr > T(0) ? xnr : T(0));  // where_lt(s, 0, 0, s)
          lo = -h;
          hi = h;
        }
      }
    });
  }
  mark(3);
  // u~ = sum_r z~_r x_r, lane == dof
  T u = T(0);
  static_for<0, nr>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    u += zc[r] * dpp_bcast<r>(x);
  });
  mark(6);
  return lane < NDP ? u : T(0);
}

// ------------------------------------------------------------------------------------------
// the step kernel
// ------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------
// Gram form of the contact solve on the matrix cores (two-wavefront workgroups, 16 lanes per environment, double).
//
// Once the row store holds z~_r = D^-1/2 L^-1 J_r^T (phase K) everything the projected Gauss-Seidel needs is inner
// products of those rows:  A_rs = z~_r . z~_s  (= J_r M^-1 J_s^T)  and the acceleration part of the right-hand sides
// z~_r . y~.  v_mfma_f64_4x4x4_4b_f64 computes four independent 4x4x4 products per instruction; its operand map
// (probed: tools/ubench/mfma_f64_4x4x4.hip) is   A: lane = 16 k + 4 blk + i,  B: lane = 16 k + 4 blk + j,
// D: lane = 16 i + 4 blk + j   — block blk takes the environment in lane group blk, the operands come straight out of
// the environments' LDS regions, and [Z~ | y~] [Z~ | y~]^T lands in a per-environment 16 x 16 buffer with ~50 matrix
// instructions instead of ~12 cross-lane reductions per sweep and 12 more for the right-hand sides.
// The sweep then runs with lane == row and NO reduction at all: every lane keeps its row's  res_s = sum_r A_sr x_r
// up to date (one FMA per updated row, the update broadcast by DPP), so the dependent chain per row is
// sub, mul, max, min, broadcast, fma.  Same iteration as tds_pgs_sweep (mb_constraint_solver.hpp:101-142, rows in the
// reference's order: normals, tangents 1, tangents 2), different association of the sums.
// Needs 3 NA <= 15 (rows on 16 lanes, one column left for y~) and every row in LDS.
// ------------------------------------------------------------------------------------------
#define TDS_GRAM_STRIDE 17  // odd row stride of the 16 x 16 (+1) buffer
#define TDS_GRAM_ZEROS (16 * TDS_GRAM_STRIDE)  // 16 zeros behind the buffer: what masked operand lanes read
template <int NDP>
__device__ __forceinline__ double tds_gram_solve(double *sm, const TdsLds &L, int lane, int NA, int na, int ZR, int NCPp,
                                                 int iters, double mu, double dt, double erp_dt, double rest,
                                                 long long *stamp = nullptr, int stamp_at = 0) {
  using T = double;
  // (diagnostic: one timestamp inside this function, selected by TDS_GRAM_STAMP_AT)
  auto mark = [&](int at) {
    if (stamp != nullptr && at == stamp_at) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0);
      if (blockIdx.x == 0 && threadIdx.x == 0) *stamp = (long long)__builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  constexpr int NDs = NDP + 1;
  constexpr int GS = TDS_GRAM_STRIDE;
  const int wl = threadIdx.x & 63;
  const int grp = wl >> 4;
  const int nr = 3 * NA;          // rows 0..nr-1; column nr carries y~
  const int NTr = (nr + 3) >> 2;  // row tiles
  const int NTc = (nr + 4) >> 2;  // column tiles (incl. the y~ column)
  {
    // ---- [Z~ | y~] [Z~ | y~]^T on the matrix cores
    const int kq = wl >> 4, blk = (wl >> 2) & 3, ij = wl & 3;
    const int eb = blk * L.stride;
    const int zero_at = eb + L.Xw + TDS_GRAM_ZEROS;
    int base[4], base3[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = 4 * t + ij;  // row of [Z~ | y~] this lane feeds in tile t
      const int at = c < nr ? eb + L.Z + c * NDs : (c == nr ? eb + L.dinv + 3 * NDP : zero_at);
      base[t] = at + kq;
      base3[t] = (12 + kq < NDP) ? base[t] : zero_at;  // (k tile 3 runs past the padded dof count)
    }
    T op[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < NTc) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (4 * k >= NDP) op[t][k] = T(0);
          else if (4 * k + 3 < NDP) op[t][k] = sm[base[t] + 4 * k];
          else op[t][k] = sm[base3[t] + (12 + kq < NDP ? 4 * k : 0)];
        }
      }
    const int out = eb + L.Xw + kq * GS + ij;  // D: row 4 I + (lane >> 4), column 4 J + (lane & 3) of block blk
#pragma unroll
    for (int I = 0; I < 4; ++I)
      if (I < NTr) {
#pragma unroll
        for (int J = 0; J < 4; ++J)
          if (J < NTc) {
            T acc = T(0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (4 * k < NDP) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(op[I][k], op[J][k], acc, 0, 0, 0);
            sm[out + 4 * I * GS + 4 * J] = acc;
          }
      }
  }
  TDS_WAVE_SYNC();
  mark(1);
  // ---- lane == row
  T *const E = sm + grp * L.stride;
  const T *const Gm = E + L.Xw;
  const T *const rws = E + L.rows;
  const T *const cpx = E + L.cp;
  const int s = lane;
  const int t = (s >= NA ? 1 : 0) + (s >= 2 * NA ? 1 : 0);
  const int a = s - t * NA;
  const bool row = s < nr && a < na;
  const int sc = s < nr ? s : 0;
  T Acol[15];
#pragma unroll
  for (int r = 0; r < 15; ++r) Acol[r] = r < nr ? Gm[s * GS + r] : T(0);
  // right-hand side: J_r . qd+ = J_r . qd + dt z~_r . y~ (see tds_row_rhs_finish)
  const T vrow = rws[sc] + dt * Gm[s * GS + nr];
  const T b = !row ? T(0) : (t == 0 ? (T(1) + rest) * vrow - erp_dt * cpx[3 * NCPp + a] : vrow);
  const T ar = row ? rws[ZR + sc] : T(0);
  const T gr = row ? rws[2 * ZR + sc] : T(0);
  T x = T(0), res = T(0);
  T lo = T(0), hi = t == 0 ? T(100000) : T(0);
  mark(2);
  for (int it = 0; it < iters; ++it) {
    static_for<0, 15>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      if (r < nr) {  // wave-uniform
        const T delta = res - gr * x;  // sum over the OTHER rows (G_rr x_r taken out again)
        T xn = (b - delta) * ar;
        xn = max_t<T>(xn, lo);
        xn = min_t<T>(xn, hi);
        const T dxr = dpp_bcast<r>(xn - x);
        res += Acol[r] * dxr;
        x = s == r ? xn : x;
        if (r < NA) {  // a normal row: its impulse bounds the two friction rows of the same contact
          const T xr_n = dpp_bcast<r>(xn);
          if (t != 0 && a == r) {
            const T h = mu * (xr_n > T(0) ? xr_n : T(0));  // where_lt(s, 0, 0, s)
            lo = -h;
            hi = h;
          }
        }
      }
    });
  }
  mark(3);
  // u~ = sum_r z~_r x_r, lane == dof
  const T *const Zs = E + L.Z;
  const int dcl = lane < NDP ? lane : NDP - 1;
  T u = T(0);
  static_for<0, 15>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    if (r < nr) u += Zs[r * NDs + dcl] * dpp_bcast<r>(x);
  });
  return lane < NDP ? u : T(0);
}


// The by-value kernel arguments L (LDS layout) and ctl (what the launch does) as the STEP-LOOP builds read them.  Taken
// from the function parameters, every field the loop body uses — and every boolean derived from one — is loop-invariant:
// the compiler loads them all in front of the loop and keeps them live across it, ~60 kernel-argument dwords and ~40
// uniform conditions (two SGPRs each) against 102 SGPRs, i.e. 190 - 230 SGPR spills into VGPR lanes, v_readlane reloads all
// through an issue-bound loop body and, past 192, VGPRs lost to holding them (round 4's review, "What's weak" 3).  The step
// loop therefore re-derives the kernel-argument segment pointer per iteration from a laundered copy — as it does for the
// lane, the lane group and the model pointer — and reads the two structs THROUGH it (constant address space: scalar
// loads, hit the scalar cache), so that a field lives from its first use in an iteration to its last.  The straight-line
// builds keep the parameters.  (The mirror struct below has the kernel's parameter list, hence the segment's layout.)
#define TDS_AS4 __attribute__((address_space(4)))
template <typename T, typename TR>
struct TdsKernArgs {
  const DevModel<T> *mdl;
  TdsLds L;
  const TR *x_in;
  TR *y_out;
  const TR *actions;
  TR *x_feedback;
  TR *obs_out;
  T *ovf;
  long long *prof;
  TdsStepCtl ctl;
  int n_envs;
};
template <bool LOOP, typename S>
struct TdsKaRef {
  using type = const S &;
  static __device__ __forceinline__ type get(const S &param, const TDS_AS4 char *) { return param; }
};
template <typename S>
struct TdsKaRef<true, S> {
  using type = const TDS_AS4 S &;
  static __device__ __forceinline__ type get(const S &, const TDS_AS4 char *at) { return *(const TDS_AS4 S *)at; }
};
template <bool ON, typename P>
__device__ __forceinline__ P tds_ka_val(P param, const TDS_AS4 char *at) {  // a pointer / scalar parameter, by value
  if constexpr (ON) return *(const TDS_AS4 P *)at;
  else return param;
}
#ifndef TDS_KA_RELOAD
#define TDS_KA_RELOAD 1
#endif

// per-group state machine of the in-kernel step loop
#define TDS_MODE_IDLE 0
#define TDS_MODE_RUN 1
#define TDS_MODE_SETTLE 2

// counter-based uniform in [0,1): splitmix64 finaliser of (seed, env, reset count, coordinate).
// Stateless, so a reset is reproducible whatever the launch geometry or order.
__device__ __forceinline__ double tds_uniform01(unsigned long long seed, unsigned env, unsigned count, unsigned j) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (((unsigned long long)env << 32) | (unsigned long long)(count * 64u + j + 1u));
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// q = reset_q + reset_noise * U(-1,1), qd = 0 into the LDS record of one environment; advances its reset counter
// (ant_environment2.h:124-135)
template <typename T, int G, typename CTL>
__device__ __forceinline__ void tds_reset_state(T *xr, const DevModel<T> *mdl, const CTL &ctl, int env, int lane,
                                                int nq, int nd) {
  const unsigned cnt = ctl.reset_count != nullptr ? ctl.reset_count[env] : 0u;
  for (int i = lane; i < nq; i += G) {
    const T u01 = (T)tds_uniform01(ctl.seed, (unsigned)env, cnt, (unsigned)i);
    xr[i] = mdl->reset_q[i] + mdl->reset_noise[i] * ((u01 - T(0.5)) * T(2));
  }
  for (int i = lane; i < nd; i += G) xr[nq + i] = T(0);
  __builtin_amdgcn_wave_barrier();
  if (lane == 0 && ctl.reset_count != nullptr) ctl.reset_count[env] = cnt + 1u;
}

// TDS_STAMP: phase-boundary timestamps (shader clock) of workgroup 0, PROF builds only
// (-DTDS_PROF_LOOP, a profiling build outside the library: the two-wavefront STEP-LOOP kernel with stamps, taken in iteration
//  ctl.flags >> 16 of the launch — the steady state of a step, not the first step of a launch; tools/profile_loop.sh)
#ifdef TDS_PROF_LOOP
#define TDS_PROF_ITER (LOOP ? (int)((unsigned)ctl.flags >> 16) : 0)
#else
#define TDS_PROF_ITER 0
#endif
#define TDS_STAMP(k)                                                        \
  do {                                                                      \
    if (PROF) {                                                             \
      __builtin_amdgcn_sched_barrier(0);                                    \
      __builtin_amdgcn_s_waitcnt(0);                                        \
      if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && tds_iter == TDS_PROF_ITER) \
        prof[(threadIdx.x >> 6) * TDS_NUM_PHASE_STAMPS + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
      if constexpr (W2 && ((k) == 0 || (k) == 13)) { /* 23..27: wall clock (100 MHz) of workgroup 0, last workgroup */ \
        if (threadIdx.x == 0 && blockIdx.x == 0) prof[23 + ((k) == 13)] = (long long)__builtin_amdgcn_s_memrealtime(); \
        if (threadIdx.x == 0) prof[28 + 2 * blockIdx.x + ((k) == 13)] = (long long)__builtin_amdgcn_s_memrealtime(); \
        if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) {              \
          prof[25 + ((k) == 13)] = (long long)__builtin_amdgcn_s_memtime(); \
          if ((k) == 13) prof[27] = (long long)__builtin_amdgcn_s_memrealtime(); \
        }                                                                   \
      }                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                    \
    }                                                                       \
  } while (0)

// TDS_PROBE(id): one extra timestamp (slot 10, free in the two-wavefront form) at the probe selected by
// TDS_GRAM_STAMP_AT=id — for splitting a phase without adding stamps to the record (tools/profile_phases.py)
#define TDS_PROBE(id)                                                       \
  do {                                                                      \
    if (PROF && W2) {                                                       \
      if ((ctl.flags >> 8) == (id)) {                                       \
        __builtin_amdgcn_sched_barrier(0);                                  \
        __builtin_amdgcn_s_waitcnt(0);                                      \
        if (blockIdx.x == 0 && threadIdx.x == 0 && tds_iter == TDS_PROF_ITER) \
          prof[10] = (long long)__builtin_amdgcn_s_memtime();               \
        __builtin_amdgcn_sched_barrier(0);                                  \
      }                                                                     \
    }                                                                       \
  } while (0)

// LOOP = false: exactly one normal step per launch, no reset -> straight-line code (the bench /
// forward_zero path; no loop-carried live ranges).  LOOP = true: the general step loop (substeps,
// auto / forced reset + settle steps) at the price of ~60 more live registers.
// KIND = 0: fixed base, 1-dof joints; 1: floating base; 2: spherical joints (fixed base).  A template parameter,
// not a model flag read at run time — as wave-uniform branches the floating-base blocks cost the fixed-base
// kernels 5 % (measured: Ant x 4096, 23.3 vs 22.2 us per step).
// T = the scalar the kernel computes in, TR = the scalar of the records in HBM (x, y, actions, obs, policy, returns).
// TR == T for the f64 and the f32 builds; <T = double, TR = float> is the "f32 records / f64 arithmetic" build:
// the reference's float ABI (776 B per Ant env-step) with the mass-matrix factorisation kept in double,
// which is what lets the float record of BASELINE config 2 meet the 1e-6 per-step contract.
// LP: 0 = straight-line build, 1 = step-loop build, 2 = step-loop build compiled for two wavefronts per SIMD.
// The step-loop body needs ~300 registers as the compiler likes it (256 VGPR + ~55 AGPR copies: one wavefront per
// SIMD, the faster form up to one wavefront per SIMD, i.e. 4096 Ant environments); held to 256 it spills ~55
// registers to scratch but two wavefronts overlap — measured on MI355X (profiles/r02_rollout_modes.txt): Ant
// rollouts 2.17e8 vs 2.05e8 env-steps/s at 4096 environments, 2.24e8 vs 3.12e8 at 8192.  The launcher picks by
// grid size.
// W2: TWO wavefronts per workgroup (straight-line build of the plain kernels only; 128 threads).  At the headline size
// — 4096 Ant environments = 1024 one-wave workgroups = exactly one wavefront per SIMD — the kernel time is the latency
// of ONE wavefront's dependent instruction stream (a lone wavefront issues an instruction every ~10 cycles).  The
// phases that depend only on the kinematics sweep — narrowphase, visual poses, Jacobian rows — and the per-row
// forward substitutions, which depend only on the factorisation, are taken off that stream by a helper wavefront:
//     wavefront 0:  A load, PD, B jcalc, C kinematics | D inertias, E composite sweep, G mass matrix, H LDL^T |
//                   F forward dynamics | row right-hand sides, L PGS, M/N integrate + pack
//     wavefront 1:  (constants)                        | I narrowphase, M1 visual poses, J Jacobian rows        |
//                   K row solves (forward substitutions) | done
// with three workgroup barriers at the bars.  The LDS groups that alias each other in the one-wave layout are
// disjoint here (their lifetimes now overlap), which costs LDS and is why this form is launched only while the whole
// grid fits the GPU at once (tds_launch_step_impl).
template <typename T, typename TR, int G, int NDP, bool PROF, int LP, int KIND, bool W2 = false>
__global__ __launch_bounds__(W2 ? 128 : 64)
__attribute__((amdgpu_waves_per_eu((LP == 2 || W2 || (LP == 0 && NDP < 24)) ? 2 : 1)))
void tds_step_kernel(const DevModel<T> *__restrict__ mdl_arg, TdsLds L_arg,
                                                      const TR *x_in, TR *__restrict__ y_out,
                                                      const TR *__restrict__ actions, TR *x_feedback /* may alias x_in */,
                                                      TR *__restrict__ obs_out, T *ovf, long long *prof, TdsStepCtl ctl_arg, int n_envs) {
  constexpr bool LOOP = LP != 0;
  // (in front of the step loop the parameters themselves; inside it: see TdsKaRef)
  const TdsLds &L = L_arg;
  const TdsStepCtl &ctl = ctl_arg;
  static_assert(!W2 || ((LP == 0 || LP == 1) && KIND == 0 && NDP < 24),
                "two-wavefront workgroups: plain kernels, straight-line or step-loop");
  // LDS slots behind the x record (xr[in_dim + ...]): 0 x_{t-1}, 1 done; two-wavefront layout: 2, 3 contact counts, 4 "y~ is
  // published", 5 columns of L published; then (step-loop builds with record rings) the step's reward and "its records are out"
  constexpr int RW_SLOT = W2 ? 6 : 2, OUT_SLOT = W2 ? 7 : 3;
  extern __shared__ __align__(16) unsigned char tds_smem_raw[];
  T *const sm = reinterpret_cast<T *>(tds_smem_raw);
  constexpr int EPW = 64 / G;
  const int lane0 = threadIdx.x & (G - 1);
  const int grp0 = (threadIdx.x & 63) / G;
  // which wavefront of the workgroup (scalar: every branch on it is a uniform branch)
  const int wv = W2 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  const bool main_wave = !W2 || wv == 0;
  // The main wavefront is the workgroup's critical path, the helper has slack: where a main and a helper wavefront (of
  // different workgroups: one of each per SIMD when 1024 two-wave workgroups are resident) share a SIMD, the arbiter
  // issues the main one first.  Measured (same process, experiment slots, profiles/r04_ab_slots_ant4096.txt): Ant x 4096
  // ring launches 14.01 -> 13.72 us per step (priority 3; -DTDS_MAIN_PRIO=0 leaves the default arbitration).
#ifndef TDS_MAIN_PRIO
#define TDS_MAIN_PRIO 3
#endif
  if constexpr (W2 && TDS_MAIN_PRIO > 0) {
    if (wv == 0) __builtin_amdgcn_s_setprio(TDS_MAIN_PRIO);
  }

  // ---- A0. the x record (and the fresh actions) are requested from HBM first: their latency runs under the
  //      fetch of the model constants below; dimensions from the kernel arguments, not from the model
  constexpr int XPL = (96 + G - 1) / G;  // record scalars per lane held in registers (longer records: loop in A)
  T xpre[XPL];
  {
    const int env = blockIdx.x * EPW + grp0;
#pragma unroll
    for (int k = 0; k < XPL; ++k) {
      const int i = lane0 + k * G;
      const bool act = actions != nullptr && i >= L.nqnd && i < L.nqnd + L.adim;
      const TR *src = act ? actions + ((size_t)env * L.adim + (i - L.nqnd)) : x_in + ((size_t)env * L.in_dim + i);
      xpre[k] = (main_wave && env < n_envs && i < L.in_dim) ? (T)*src : T(0);
    }
  }

  // ---- in-kernel step loop (LOOP builds): `nsub` normal steps, then (auto / forced reset) the environments that
  //      need it are re-initialised and run `settle_steps` zero-action steps — all inside this launch,
  //      state carried in the LDS record.  mode / left are uniform within a lane group.
  // Only THESE values are carried from one iteration to the next; everything else — the lane's model constants, the
  // LDS region pointers, the record dimensions — is re-derived inside the loop body from per-iteration laundered
  // copies of (lane, group, model pointer), so that the compiler can neither hoist it out of the loop nor keep it
  // live across the back edge (that cost the step-loop build 256 VGPR + ~155 AGPR copies + ~270 spilled SGPRs).
  const int nset = (LOOP && ctl.reset_mode != TDS_RESET_NONE) ? ctl.settle_steps : 0;
  int mode = ((int)(blockIdx.x * EPW + grp0) < n_envs && (!LOOP || ctl.nsub > 0)) ? TDS_MODE_RUN : TDS_MODE_IDLE;
  int left = LOOP ? ctl.nsub : 1;  // normal steps still to run
  int sleft = 0;                   // settle steps still to run (mode SETTLE)
  // rollout mode: return accumulated so far, its step count, "done and not auto-reset" latch
  // (pol: rollout mode.  replay: action replay — tds_hip_step_many as one launch: the block of step k + 1 is requested from
  //  HBM at the top of step k.  pool_r: auto-reset inside a replayed step loop: a done environment takes its next
  //  pre-settled state from the reset pool (tds_api.hip: reset pool) and carries on with the following step — the lane
  //  groups of a launch stay in lock step.  ring_o / ring_y: per-step record rings (tds_hip_step_many_rings): every step of
  //  the launch packs and stores its records.  All of them are derived inside the loop body, per iteration.)
  T next_act = T(0);
  T ret = T(0);
  int cnt = 0;
  bool frozen = false;
  int tds_iter = 0;
  (void)tds_iter;
  if constexpr (LOOP) {
    // prologue of the step-loop build: x record -> LDS, forced reset of the selected environments
    const DevModel<T> *mdl = mdl_arg;
    const int lane = lane0, env = blockIdx.x * EPW + grp0;
    const bool valid = env < n_envs;
    const int nq = mdl->dof_q, nd = mdl->dof_qd, in_dim = mdl->input_dim, adim = mdl->action_dim;
    T *const xr = sm + grp0 * L.stride + L.xrec;
    // The lane's model constants are re-derived inside every iteration BY DESIGN (see above: nothing but the loop state may
    // stay live across the back edge) — which made every iteration begin with a round trip to L2 for the lane's small
    // constants (in front of the PD block) and another one for X_T (in front of jcalc).  Step-loop launches of the plain kernels
    // keep them in a table of the WORKGROUP in LDS instead (TdsLds::cw rows of [G] scalars behind the environments'
    // regions, filled here once per launch by the first lane group; the host grants the rows only where they do not cost a
    // workgroup per CU).  Measured (experiment slots, same process: profiles/r04_ab_slots3_lds_consts.txt): Ant x 4096 ring
    // launches 13.66 -> 13.46 (small constants) -> 13.08 us per step (+ X_T); mass / centre of mass / inertia / motion axis
    // on top change nothing (they are consumed phases later: their latency was hidden already).
    if constexpr (KIND == 0) {
      if (L.cw != 0) {  // wave-uniform (kernel argument)
        constexpr int CWL = TDS_CW_LANE(sizeof(T));
        T *const CW = sm + EPW * L.stride;
        if (main_wave && grp0 == 0) {
          const int ls = lane < mdl->num_links ? lane : 0;
          const bool il = lane < mdl->num_links;
          CW[0 * G + lane] = mdl->init_pose[ls];
          CW[1 * G + lane] = mdl->stiffness[ls];
          CW[2 * G + lane] = mdl->damping[ls];
          const int par = il ? mdl->parent[ls] : -1, lev = il ? mdl->level[ls] : -1;
          const int jtt = il ? mdl->joint_type[ls] : TDS_JOINT_FIXED, dii = il ? mdl->qd_index[ls] : -1;
          const int cfl = il ? mdl->chain_flags[ls] : 0, msl = il ? mdl->lc_slot[ls] : -1;
          const int psl = (il && par >= 0) ? mdl->lc_slot[par] : -1, aci = il ? mdl->act_index[ls] : -1;
          const unsigned lo = (unsigned)((par + 1) & 255) | ((unsigned)((lev + 1) & 255) << 8) | ((unsigned)(jtt & 255) << 16) |
                              ((unsigned)((dii + 1) & 255) << 24);
          const unsigned hi = (unsigned)(cfl & 255) | ((unsigned)((msl + 1) & 255) << 8) | ((unsigned)((psl + 1) & 255) << 16) |
                              ((unsigned)((aci + 1) & 255) << 24);
          if constexpr (sizeof(T) == 8) {
            CW[3 * G + lane] = (T)__hiloint2double((int)hi, (int)lo);
          } else {
            CW[3 * G + lane] = bits_to_scalar<T>(lo);
            CW[4 * G + lane] = bits_to_scalar<T>(hi);
          }
          if (L.cw >= CWL + TDS_CW_XT) {
#pragma unroll
            for (int k = 0; k < 12; ++k) CW[(CWL + k) * G + lane] = mdl->X_T[k][ls];
          }
        }
        if constexpr (W2) __syncthreads();
        else TDS_WAVE_SYNC();
      }
    }
    if (main_wave) {  // (a helper wavefront first touches the record behind barrier (1) of the first step)
#pragma unroll
      for (int k = 0; k < XPL; ++k) {
        const int i = lane + k * G;
        if (i < in_dim) xr[i] = xpre[k];
      }
      for (int i = lane + XPL * G; i < in_dim; i += G) {
        const bool act = actions != nullptr && i >= nq + nd && i < nq + nd + adim;
        xr[i] = !valid ? T(0) : act ? (T)actions[(size_t)env * adim + (i - nq - nd)] : (T)x_in[(size_t)env * in_dim + i];
      }
    }
    TDS_WAVE_SYNC();
    bool finished0 = false;  // forced reset with zero settle steps: nothing to simulate
    if (ctl.reset_mode == TDS_RESET_FORCED && valid && (ctl.mask == nullptr || ctl.mask[env] != 0)) {
      tds_reset_state<T, G>(xr, mdl, ctl, env, lane, nq, nd);
      if (nset > 0) {
        mode = TDS_MODE_SETTLE;
        sleft = nset;
      } else {
        finished0 = true;
      }
    }
    TDS_WAVE_SYNC();
    if (finished0) {
      const bool raw = (ctl.flags & TDS_CTL_RESET_CALL) != 0 && mdl->reset_obs_raw_xy != 0;
      for (int i = lane; i < nq + nd; i += G) {
        if (obs_out != nullptr) obs_out[(size_t)env * (nq + nd + 2) + i] = (TR)((i < 2 && !raw) ? T(0) : xr[i]);
        if (x_feedback != nullptr) x_feedback[(size_t)env * in_dim + i] = (TR)xr[i];
      }
    }
  }

  for (;;) {  // ================================ step loop ================================
  if constexpr (LOOP) {
    if (!__any(mode != TDS_MODE_IDLE)) break;
  }
  int lane_l = lane0, grp_l = grp0;
  // (the model pointer is laundered as a GLOBAL-address-space pointer: laundered as a generic one — round 4 — the address
  //  space was lost with the provenance and every model constant of a step-loop build came in through a FLAT load, which
  //  counts on vmcnt AND lgkmcnt and returns out of order with the DS instructions: every LDS wait behind one became
  //  lgkmcnt(0).  -DTDS_MDL_GLOBAL=0: the generic pointer, 1: global, 2 (default): constant address space, see below;
  //  same-process A/B, Ant x 4096 / x 8192, us per step: 12.10 / 21.02 -> 12.04 / 20.56 -> 11.58 / 19.91)
#ifndef TDS_MDL_GLOBAL
#define TDS_MDL_GLOBAL 2
#endif
  // (TDS_MDL_GLOBAL = 2: the CONSTANT address space — the model is read-only for the kernel — so that loads at uniform
  //  addresses through the laundered pointer are scalar loads again (the header fields every iteration starts with: as
  //  vector loads a round trip to L2 at the top of every step), as they are in the straight-line builds, whose pointer is the
  //  `const __restrict__` kernel argument itself)
#if TDS_MDL_GLOBAL == 2
#define TDS_MDL_AS 4
#else
#define TDS_MDL_AS 1
#endif
  const __attribute__((address_space(TDS_MDL_AS))) DevModel<T> *mdl_g = (const __attribute__((address_space(TDS_MDL_AS))) DevModel<T> *)mdl_arg;
  const DevModel<T> *mdl_f = mdl_arg;
  // (the 32-dof build is at one wavefront per SIMD whatever is done and fares better with the lane constants
  //  hoisted into AGPR copies — 256 + 130 registers, no scratch, against 256 + 256 + 876 B of scratch: only the
  //  model pointer is laundered there)
  if constexpr (TDS_MDL_GLOBAL != 0) {
    if constexpr (LOOP && NDP < 32) asm volatile("" : "+v"(lane_l), "+v"(grp_l), "+s"(mdl_g));
    if constexpr (LOOP && NDP >= 32) asm volatile("" : "+s"(mdl_g));
  } else {
    if constexpr (LOOP && NDP < 32) asm volatile("" : "+v"(lane_l), "+v"(grp_l), "+s"(mdl_f));
    if constexpr (LOOP && NDP >= 32) asm volatile("" : "+s"(mdl_f));
  }
  const DevModel<T> *const mdl = TDS_MDL_GLOBAL != 0 ? (const DevModel<T> *)mdl_g : mdl_f;
  const int lane = lane_l, grp = grp_l;
  // ---- the kernel arguments of this iteration (step-loop builds: read through a laundered segment pointer, see TdsKaRef)
  constexpr bool KA = LOOP && TDS_KA_RELOAD != 0;
  const TDS_AS4 char *ka_seg = (const TDS_AS4 char *)__builtin_amdgcn_kernarg_segment_ptr();
  if constexpr (KA) asm volatile("" : "+s"(ka_seg));
  using TdsKA = TdsKernArgs<T, TR>;
  typename TdsKaRef<KA, TdsLds>::type L = TdsKaRef<KA, TdsLds>::get(L_arg, ka_seg + __builtin_offsetof(TdsKA, L));
  typename TdsKaRef<KA, TdsStepCtl>::type ctl = TdsKaRef<KA, TdsStepCtl>::get(ctl_arg, ka_seg + __builtin_offsetof(TdsKA, ctl));
  // (likewise the pointer parameters and the environment count: taken from the parameters, every `!= nullptr` of the loop body
  //  is a launch-invariant condition the compiler evaluates in front of the loop and keeps — two SGPRs each — in lanes of a
  //  spill register: 48 SGPR spills, 97 v_readlane per iteration in the headline build.  -DTDS_KA_PTRS=0: the parameters)
#ifndef TDS_KA_PTRS
#define TDS_KA_PTRS 1
#endif
  constexpr bool KP = KA && TDS_KA_PTRS != 0;
  const TR *const ka_x_in = tds_ka_val<KP, const TR *>(x_in, ka_seg + __builtin_offsetof(TdsKA, x_in));
  TR *const ka_y_out = tds_ka_val<KP, TR *>(y_out, ka_seg + __builtin_offsetof(TdsKA, y_out));
  const TR *const ka_actions = tds_ka_val<KP, const TR *>(actions, ka_seg + __builtin_offsetof(TdsKA, actions));
  TR *const ka_x_feedback = tds_ka_val<KP, TR *>(x_feedback, ka_seg + __builtin_offsetof(TdsKA, x_feedback));
  TR *const ka_obs_out = tds_ka_val<KP, TR *>(obs_out, ka_seg + __builtin_offsetof(TdsKA, obs_out));
  T *const ka_ovf = tds_ka_val<KP, T *>(ovf, ka_seg + __builtin_offsetof(TdsKA, ovf));
  const int ka_n_envs = tds_ka_val<KP, int>(n_envs, ka_seg + __builtin_offsetof(TdsKA, n_envs));
  const TR *const x_in = tds_global(ka_x_in);
  TR *__restrict__ const y_out = tds_global(ka_y_out);
  const TR *__restrict__ const actions = tds_global(ka_actions);
  TR *const x_feedback = tds_global(ka_x_feedback);
  TR *__restrict__ const obs_out = tds_global(ka_obs_out);
  T *const ovf = tds_global(ka_ovf);
  const int n_envs = ka_n_envs;
  // (the launch-wide conditions, from THIS iteration's arguments — shadowing the prologue's)
  const int nset = (LOOP && ctl.reset_mode != TDS_RESET_NONE) ? ctl.settle_steps : 0;
  const bool pol = LOOP && ctl.policy != nullptr;
  const bool replay = LOOP && ctl.act_pool != nullptr && ctl.policy == nullptr;
  const bool pool_r = LOOP && ctl.pool != nullptr && ctl.policy == nullptr;
  const bool ring_o = LOOP && ctl.obs_ring != nullptr;  // wave-uniform (kernel arguments)
  const bool ring_y = LOOP && ctl.y_ring != nullptr;
  const int env = blockIdx.x * EPW + grp;
  const bool valid = env < n_envs;
  T *const E = sm + grp * L.stride;

  const int nl = mdl->num_links, nq = mdl->dof_q, nd = mdl->dof_qd;
  const int in_dim = mdl->input_dim, out_dim = mdl->output_dim, adim = mdl->action_dim;
  constexpr int NDs = NDP + 1;  // odd row stride of every [row][dof] array
  const T dt = mdl->dt;

  // ---- lane == link: constants -------------------------------------------------------------
  const int li = lane;
  const bool isl = li < nl;
  const int lsafe = isl ? li : 0;
  // (step-loop launches: from the workgroup's constant table in LDS where the host granted one, see the prologue)
  const bool cwt = LOOP && KIND == 0 && L.cw != 0;  // wave-uniform
  constexpr int CWL = TDS_CW_LANE(sizeof(T));
  const T *const CW = sm + EPW * L.stride;
  // (read through an LDS-address-space pointer: as generic pointers the compiler merged "table or model" into ONE flat load
  //  through a selected pointer — table reads that went the flat path, out of order with the DS instructions)
#ifndef TDS_CW_LDS
#define TDS_CW_LDS 1
#endif
#if TDS_CW_LDS
  const TDS_AS3 T *const CWl = (const TDS_AS3 T *)CW;
#else
  const T *const CWl = CW;
#endif
  int parent, level, jt, di;
  unsigned cw_hi = 0u;
  if (cwt) {
    unsigned cw_lo;
    if constexpr (sizeof(T) == 8) {
      const double pk = (double)CWl[3 * G + lane];
      cw_lo = (unsigned)__double2loint(pk);
      cw_hi = (unsigned)__double2hiint(pk);
    } else {
      cw_lo = scalar_to_bits<T>(CWl[3 * G + lane]);
      cw_hi = scalar_to_bits<T>(CWl[4 * G + lane]);
    }
    parent = (int)(cw_lo & 255u) - 1;
    level = (int)((cw_lo >> 8) & 255u) - 1;
    jt = (int)((cw_lo >> 16) & 255u);
    di = (int)(cw_lo >> 24) - 1;
  } else {
    parent = isl ? mdl->parent[lsafe] : -1;
    level = isl ? mdl->level[lsafe] : -1;
    jt = isl ? mdl->joint_type[lsafe] : TDS_JOINT_FIXED;
    di = isl ? mdl->qd_index[lsafe] : -1;  // == q_index (1-DoF joints only)
  }
  // Floating base (DevModel::is_floating): lanes 0..5 are the base's pseudo links and the dofs are numbered
  // joints first, base last; the q / qd RECORD keeps the reference's order
  // q = [quat xyzw | pos | joints], qd = [omega | v | joints]  ->  record indices of this lane's coordinate
  // Spherical joints (DevModel::num_spherical): three lanes per joint, types TDS_JOINT_SPH0/1/2, one frame; the
  // q record holds the joint's quaternion (4 coordinates) at q_rec of the first lane.
  constexpr bool fl = KIND == 1;
  constexpr bool sph = KIND == 2;
  constexpr bool gen = KIND == 1 || KIND == 2 || KIND == 4;  // (the q / qd records are addressed through index tables)
  // KIND 3: worlds of SEVERAL articulated bodies (<= TDS_MAX_BODIES; fixed bases, 1-dof joints) in one lane group: the
  // links of body b + 1 sit behind those of body b, the joint-space inertia is block diagonal (the LDL^T and every solve
  // go through as they are), and one MORE contact pass per body pair a < b handles the contacts between the bodies, in
  // the reference's order (world.hpp:206-282, 293-366).
  // KIND 4: ... with FLOATING bases among the bodies: a floating body's six pseudo links (see KIND 1) sit in front of
  // its links, its dofs are numbered joints first, base last, so that every diagonal block of the factorisation ends in
  // its base's 6 x 6 Schur complement; the floating-base quirks of the reference are applied body by body.
  constexpr bool flm = KIND == 4;
  constexpr bool two = KIND == 3 || KIND == 4;
  // contact solve in Gram form on the matrix cores (tds_gram_solve): two-wavefront workgroups of 16-lane environments
  constexpr bool GRAM = W2 && !LOOP && G == 16 && NDP <= 16 && std::is_same<T, double>::value;  // (opt-in; straight-line form only)
  // two-wavefront workgroups, narrow kernels: no barrier between the LDL^T and the helper's row solves — L reaches the
  // helper in two halves through LDS flags (tds_row_solve), the contact counts reach this wavefront the same way
  constexpr bool PIPE = W2 && NDP <= 16;
  const int bod = (two && isl) ? mdl->body_of_link[lsafe] : 0;  // my link's body
  const int njd = mdl->nj;          // joint dofs (== nd on a fixed base)
  const int fbk = (flm && isl) ? mdl->fb_k[lsafe] : (fl && isl && li < 6 ? li : -1);  // pseudo link k of a floating base
  const int fbq = (flm && isl) ? mdl->fb_q[lsafe] : 0;  // ... its body's quaternion in the q record (position at + 4)
  const bool froot = fbk >= 0;                   // base pseudo link
  // lane == dof role: the first of the six base dofs of my dof's body (-1: its base is fixed), whether my dof is one of
  // them, and where that body's quaternion sits in the q record
  const int dbase0 = flm ? (lane < nd ? mdl->dof_base0[lane] : -1) : (fl ? njd : -1);
  const int dfbq = flm ? (lane < nd ? mdl->dof_fbq[lane] : -1) : 0;
  const bool bdof = flm ? dfbq >= 0 : (fl && lane >= njd && lane < nd);
  const bool sph_lane = sph && jt >= TDS_JOINT_SPH0;
  const int qri = gen ? (isl ? mdl->q_rec[lsafe] : -1) : di;    // (-1: this lane owns no coordinate)
  const int qdri = gen ? (isl ? mdl->qd_rec[lsafe] : -1) : di;
  const int rec_d = gen ? mdl->dof_rec[lane < nd ? lane : 0] : lane;  // lane == dof role
  // Serial chains (parent == lane - 1, the common case for URDF-derived trees) hand their sweep
  // state from lane to lane with DPP row shifts; only the other parent/child links go through the
  // per-link LDS records (see DESIGN.md "chain hand-over").
  int cflags, my_slot, par_slot, act_i;  // chain flags; my (v, a0) side record, my parent's; my action
  T init_pose_l, stiff_l, damp_l;
  if (cwt) {
    cflags = (int)(cw_hi & 255u);
    my_slot = (int)((cw_hi >> 8) & 255u) - 1;
    par_slot = (int)((cw_hi >> 16) & 255u) - 1;
    act_i = (int)(cw_hi >> 24) - 1;
    init_pose_l = CWl[0 * G + lane];
    stiff_l = CWl[1 * G + lane];
    damp_l = CWl[2 * G + lane];
  } else {
    cflags = isl ? mdl->chain_flags[lsafe] : 0;
    my_slot = isl ? mdl->lc_slot[lsafe] : -1;
    par_slot = (isl && parent >= 0) ? mdl->lc_slot[parent] : -1;
    act_i = isl ? mdl->act_index[lsafe] : -1;
    init_pose_l = mdl->init_pose[lsafe];
    stiff_l = mdl->stiffness[lsafe];
    damp_l = mdl->damping[lsafe];
  }
  // Top of a step of a two-wavefront step loop, BOTH wavefronts: barrier (0).  The main wavefront arrives from its
  // integration (the LDS record holds the state, reward and done flag of the step before), the helper — long waiting — from
  // its late visual poses (it has read X_world of the step before: the main wavefront's kinematics sweep may overwrite it).
  // Behind it the helper stores the END-of-step records of the step before (flush_prev_records) while the main wavefront
  // runs A - C, in what used to be the helper's idle wait for barrier (1) — not between its narrowphase and its Jacobian
  // rows, where every store instruction lengthened its path to barrier (2): with the records also going to seven peers
  // (the peer-store exchange of an 8-GPU run) that was 12 % of the step (profiles/r05_peer_store_cost_one_gpu.txt).
  // Round 4's form (-DTDS_FLUSH_EARLY=0): no barrier, the main wavefront polls the helper's "poses are out" flag.
  // (lgkmcnt only: neither wavefront waits for its outstanding global stores here)
#ifndef TDS_FLUSH_EARLY
#define TDS_FLUSH_EARLY 1
#endif
  if constexpr (W2 && LOOP) {
    if constexpr (TDS_FLUSH_EARLY != 0) {
      if (tds_iter > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      if (main_wave && tds_iter > 0) {  // the helper has read X_world of the previous step (its late visual poses, see there)
        const T *const pf = sm + grp * L.stride + L.xrec + in_dim + 4;
        while (__any(tds_lds_poll(pf) != T(2))) __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  const bool chain_child = (cflags & 1) != 0;      // my parent is lane - 1
  const bool has_chain_child = (cflags & 2) != 0;  // lane + 1 is my child and hands over by DPP
  const bool lds_children = (cflags & 4) != 0;     // I have children that are not lane + 1
  // PD / joint constants of the link (fetched here with the other lane constants: the PD block right
  // after the x record arrives must not start with a round trip to L2)
  const int sphq = (sph && isl) ? mdl->sph_q[lsafe] : -1;
  const T act_lim = mdl->action_limit;
  const int step_mode = mdl->step_mode;
  // per-link model constants (joint axis, X_T, rigid inertia).  The straight-line build fetches
  // them HERE, ahead of the x record, so that their L2 latency overlaps the HBM latency of x; the
  // step-loop build re-fetches them per iteration (keeping ~60 VGPRs live across the loop costs more).
  T Sl[6], mass_l, com_l[3], Il[9], RT[9], tT[3];
  auto load_link_consts = [&](const DevModel<T> *md) {
    if (cwt && L.cw >= CWL + TDS_CW_XT) {  // (X_T from the workgroup's table; the rest is consumed phases later)
#pragma unroll
      for (int k = 0; k < 6; ++k) Sl[k] = md->S[k][lsafe];
      mass_l = md->mass[lsafe];
#pragma unroll
      for (int k = 0; k < 3; ++k) com_l[k] = md->com[k][lsafe];
#pragma unroll
      for (int k = 0; k < 9; ++k) Il[k] = md->inertia[k][lsafe];
#pragma unroll
      for (int k = 0; k < 9; ++k) RT[k] = CWl[(CWL + k) * G + lane];
#pragma unroll
      for (int k = 0; k < 3; ++k) tT[k] = CWl[(CWL + 9 + k) * G + lane];
      return;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) Sl[k] = md->S[k][lsafe];
    mass_l = md->mass[lsafe];
#pragma unroll
    for (int k = 0; k < 3; ++k) com_l[k] = md->com[k][lsafe];
#pragma unroll
    for (int k = 0; k < 9; ++k) Il[k] = md->inertia[k][lsafe];
#pragma unroll
    for (int k = 0; k < 9; ++k) RT[k] = md->X_T[k][lsafe];
#pragma unroll
    for (int k = 0; k < 3; ++k) tT[k] = md->X_T[9 + k][lsafe];
  };
  if constexpr (!LOOP) {
    if (main_wave) load_link_consts(mdl);
  }
  // The same for the constants of the later phases (first narrowphase pass: contact point == lane,
  // visual == lane, mass-matrix row == lane, contact frame and solver scalars): issued here, their
  // L2 / scalar-cache latency is long gone when the phase starts; fetched where they are used, each
  // of those phases began with a dependent round trip to L2.
  int pf_cp_link = -1, pf_vis_link = 0, pf_dof_link = 0;
  unsigned pf_anc = 0u, pf_cp_anc = 0u;
  T pf_cp_loc[3] = {T(0), T(0), T(0)}, pf_cp_rad = T(0), pf_vis_X[12];
  T pf_nb[3], pf_t1[3], pf_t2[3], pf_plane_n[3], pf_plane_c, pf_cfm, pf_erp_dt, pf_rest, pf_mu;
  int pf_iters, pf_ncp, pf_nv;
  auto load_phase_consts = [&](const DevModel<T> *md) {
    pf_ncp = md->has_plane ? md->num_cp : 0;
    pf_nv = md->num_visuals;
    const int kc = lane < pf_ncp ? lane : 0;
    pf_cp_link = md->cp_link[kc];
    pf_cp_anc = md->anc_dofs[pf_cp_link >= 0 ? pf_cp_link : 0];
#pragma unroll
    for (int c = 0; c < 3; ++c) pf_cp_loc[c] = md->cp_local[c][kc];
    pf_cp_rad = md->cp_radius[kc];
    const int kv = lane < pf_nv ? lane : 0;
    pf_vis_link = md->vis_link[kv];
#pragma unroll
    for (int c = 0; c < 12; ++c) pf_vis_X[c] = md->vis_X[c][kv];
    pf_dof_link = md->dof_link[lane < md->dof_qd ? lane : 0];
    pf_anc = md->anc_dofs[pf_dof_link];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      pf_nb[c] = md->nb[c];
      pf_t1[c] = md->t1[c];
      pf_t2[c] = md->t2[c];
      pf_plane_n[c] = md->plane_n[c];
    }
    pf_plane_c = md->plane_c;
    pf_cfm = md->cfm;
    pf_erp_dt = md->erp_over_dt;
    pf_rest = md->restitution;
    pf_mu = md->friction;
    pf_iters = md->pgs_iterations;
  };
  if constexpr (!LOOP) load_phase_consts(mdl);
  TDS_STAMP(0);
  // ---- A. x record -> LDS (coalesced: consecutive lanes, consecutive doubles) ---------------
  T *const xr = E + L.xrec;
  if constexpr (!LOOP) {  // (the step-loop build did this in its prologue; the state then lives in the LDS record)
    if (main_wave) {
#pragma unroll
      for (int k = 0; k < XPL; ++k) {
        const int i = lane + k * G;
        if (i < in_dim) xr[i] = xpre[k];
      }
      for (int i = lane + XPL * G; i < in_dim; i += G) {
        const bool act = actions != nullptr && i >= nq + nd && i < nq + nd + adim;
        xr[i] = !valid ? T(0) : act ? (T)actions[(size_t)env * adim + (i - nq - nd)] : (T)x_in[(size_t)env * in_dim + i];
      }
    }
    TDS_WAVE_SYNC();
  }
  auto reset_state = [&]() { tds_reset_state<T, G>(xr, mdl, ctl, env, lane, nq, nd); };
  const bool live = mode != TDS_MODE_IDLE;
  const bool last_run = mode == TDS_MODE_RUN && left == 1;
  const bool settling = LOOP && mode == TDS_MODE_SETTLE;
  if constexpr (LOOP) {
    if (replay) {  // wave-uniform
      // the action block of step k + 1 is requested from HBM here, at the top of step k, and moved into the (by then
      // dead) action slots of the LDS record in the middle of step k (ahead of phase F) — NOT at the top of step k + 1:
      // on gfx9 loads and stores share one in-order counter, so a wait for this load right behind the record stores
      // that end a step (y / obs rings) would wait for those stores too: a full HBM write latency per step
      if (valid && lane < adim) {
        const int blk = (ctl.act_first + tds_iter + 1) % ctl.act_blocks;
        next_act = (T)tds_global((const TR *)ctl.act_pool)[((size_t)blk * ctl.act_envs + env) * adim + lane];
      }
    }
    // ---- rollout mode: action = W obs + b with the environment's own parameters
    //      (VectorizedEnvironment::policy -> NeuralNetwork::compute, one linear layer with bias, identity:
    //       ars_vectorized_environment.h:165-180,293-300; neural_network.hpp:223-262);
    //      obs = [q | qd] with obs[0] = obs[1] = 0 (ars_vectorized_environment.h:283-288)
    if (pol && __any(mode == TDS_MODE_RUN)) {
      const int od = nq + nd;
      const bool raw_xy = (ctl.flags & 1) != 0 && tds_iter == 0;
      if (mode == TDS_MODE_RUN && lane < adim) {
        const TR *const W = (const TR *)ctl.policy + (size_t)env * (adim * od + adim);
        T acc = (T)W[adim * od + lane];
        for (int o = 0; o < od; ++o) {
          const T ob = (o < 2 && !raw_xy) ? T(0) : xr[o];
          acc += ob * (T)W[lane * od + o];
        }
        xr[nq + nd + lane] = acc;
      }
      TDS_WAVE_SYNC();
    }
  }
  const bool do_reward = last_run || ((pol || pool_r || ring_o) && mode == TDS_MODE_RUN);
  // where this step's y record goes, and whether it is packed at all: the handle's y record for the last normal step
  // of a launch, or — with a y ring — the ring slot of EVERY step
  // one scalar into the obs ring: float or record dtype; streaming store, or (TDS_RING_NOFENCE) a device-scope
  // write-through store that needs no cache write-back to become visible to the exchange
  auto ring_put = [&](size_t idx, T v) {
    const int rf = ctl.ring_flags;
    if (rf & TDS_RING_OBS_F32) {
      float *const p = tds_global((float *)ctl.obs_ring) + idx;
      if (rf & TDS_RING_NOFENCE) __hip_atomic_store(p, (float)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else TDS_NT_STORE((float)v, p);
    } else {
      TR *const p = tds_global((TR *)ctl.obs_ring) + idx;
      if (rf & TDS_RING_NOFENCE) __hip_atomic_store(p, (TR)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else TDS_NT_STORE((TR)v, p);
    }
  };
  // Plain kernels (KIND 0) store the END-of-step records of a ring launch — the state part + tail of the y record, the
  // whole obs record — from the LDS record in the MIDDLE of the following step, behind that step's visual poses: on gfx9
  // loads and stores drain through ONE in-order counter, so the first wait for a load at the top of a step would also
  // wait for record stores issued at the end of the step before — a full HBM write latency on every step.  Behind the
  // visual poses no load is waited for until the next step begins.  (The last step of a launch and environments that
  // the reset pool re-initialises store at once.)
  constexpr bool DEFER = LOOP && KIND == 0;
  // (a y RING may have a record stride beyond output_dim — TdsStepCtl::y_stride, records on 128-byte line boundaries: the
  //  zero padding then runs to the stride, so that a record's last line is written whole)
  //  (straight-line launches pointed at a ring slot — the graph form of tds_hip_step_many_rings — take the stride for y_out)
  //  (the host sets ctl.y_stride on EVERY launch — output_dim where nothing else was asked for: no select here)
  const int ystr = ctl.y_stride;
  auto put_y_state = [&](TR *yo, int yend) {  // q | qd | (visual poses: phase M1) | up.z | zero padding, from the LDS record
    for (int i = lane; i < nq + nd; i += G) TDS_NT_STORE((TR)xr[i], &yo[i]);
    int tail = nq + nd;
    if (mdl->pack_visuals) {
      tail += 7 * mdl->num_visuals;
      if (lane == 0) TDS_NT_STORE((TR)(mdl->base_R[8]), &yo[tail]);  // up_dot_world_z (fixed base)
      tail += 1;
    }
    for (int i = tail + lane; i < yend; i += G) TDS_NT_STORE((TR)(0), &yo[i]);
  };
  // [q | qd with obs[0] = obs[1] = 0 | reward | done] of ring slot `slot` (ars_vectorized_environment.h:250-289): the
  // observation from the LDS record as it is NOW, reward / done from their LDS slots (written by the reward block)
  // Peer-store exchange (TdsStepCtl::peer_arrive): the same scalar goes into the same place of every peer's gathered ring —
  // system-scope write-through stores into memory mapped from the other ranks (over xGMI) — by the wavefront that stores
  // the record anyway (the helper wavefront of a two-wavefront workgroup: off the step's dependent chain).
  auto put_obs = [&](int slot) {
    const size_t at = ((size_t)slot * ctl.obs_envs + env) * (nq + nd + 2);
    const int np = (LOOP && ctl.peer_arrive != nullptr) ? ctl.n_peers : 0;  // wave-uniform (kernel arguments)
    const bool rd_only = (ctl.ring_flags & TDS_RING_PEER_REWARD_DONE) != 0;
    for (int i = lane; i < nq + nd + 2; i += G) {
      const int src = i < nq + nd ? i : (i == nq + nd ? in_dim + RW_SLOT : in_dim + 1);
      const T v = i < 2 ? T(0) : xr[src];
      ring_put(at + i, v);
      if constexpr (LOOP) {
        if (np > 0 && (i >= nq + nd || !rd_only)) {
          for (int pr = 0; pr < np; ++pr) {
            char *const pb = (char *)tds_global(((void *const TDS_AS4 *)(const TDS_AS4 void *)ctl.peer_ring)[pr]) + ctl.peer_off;  // (see put_obs_wide)
            if (ctl.ring_flags & TDS_RING_OBS_F32)
              __hip_atomic_store((float *)pb + (at + i), (float)v, __ATOMIC_RELAXED, TDS_PEER_SCOPE);
            else
              __hip_atomic_store((TR *)pb + (at + i), (TR)v, __ATOMIC_RELAXED, TDS_PEER_SCOPE);
          }
        }
      }
    }
  };
  // Peer-store exchange, the whole WAVEFRONT'S records as one burst (TDS_RING_WIDE: every stride of the ring is a multiple
  // of 8 bytes).  The EPW environments of a wavefront own consecutive records of a slot: EPW x (nq + nd + 2) scalars in a
  // row (Ant, float wire: 480 bytes).  Lane-per-component stores cut that row into 2 EPW pieces of 64 / 56 bytes per
  // destination; here every lane takes 8 bytes of the row — read from the environments' LDS records, converted once — and
  // the row goes out with ONE 8-byte-per-lane store instruction per destination (two on a double wire), whole and in order:
  // what a write-through store into another GPU's memory wants to look like on the fabric.  Destinations: this rank's own
  // block (device scope), then every peer's (system scope), the table's pointers fetched four at a time.
  auto put_obs_wide = [&](int slot) {
    const int w = nq + nd + 2;
    const int wl = threadIdx.x & 63;
    const size_t row0 = ((size_t)slot * ctl.obs_envs + (size_t)blockIdx.x * EPW) * (size_t)w;  // first scalar of the wavefront's row
    const bool f32w = (ctl.ring_flags & TDS_RING_OBS_F32) != 0 || sizeof(TR) == 4;
    const bool rd_only = (ctl.ring_flags & TDS_RING_PEER_REWARD_DONE) != 0;
    const int per_unit = f32w ? 2 : 1;                 // scalars per 8-byte unit
    const int n_units = (EPW * w) / per_unit;
    const int np = ctl.n_peers;
    // (the pointer table is read through the CONSTANT address space — written once at set-up, uniform index: scalar loads.
    //  As vector loads each pointer was fetched right in front of its store, and the wait for it — loads and stores return
    //  through one in-order counter — was a wait for the acknowledgement of the PREVIOUS peer's row: the seven rows of an
    //  8-GPU run went out one after the other, 2.3 us per step.  -DTDS_PEER_TAB_CONST=0)
#ifndef TDS_PEER_TAB_CONST
#define TDS_PEER_TAB_CONST 1
#endif
#if TDS_PEER_TAB_CONST
    const unsigned long long *const TDS_AS4 *tab = (const unsigned long long *const TDS_AS4 *)(const TDS_AS4 void *)ctl.peer_ring;
#else
    const unsigned long long *const *tab = tds_global((const unsigned long long *const *)ctl.peer_ring);
#endif
    for (int u0 = 0; u0 < n_units; u0 += 64) {  // (one pass on a float wire up to 128 scalars per wavefront)
      const int u = u0 + wl;
      const bool on = u < n_units;
      unsigned long long bits = 0ull;
      bool tail = false;  // this unit holds a [reward | done] column (with exchange_fields = 1 a unit travels if ANY of its
                          // columns is one of the two: on a float wire with an odd record width they share units with
                          // observation columns — floating-base and spherical models)
      {
        unsigned lo = 0u, hi = 0u;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if (c < per_unit) {
            int f = on ? u * per_unit + c : 0;
            int e = 0;
#pragma unroll
            for (int k = 1; k < EPW; ++k) e += f >= k * w ? 1 : 0;
            const int i = f - e * w;
            const int src = i < nq + nd ? i : (i == nq + nd ? in_dim + RW_SLOT : in_dim + 1);
            const T v = i < 2 ? T(0) : sm[e * L.stride + L.xrec + src];
            tail = tail || i >= nq + nd;
            if (f32w) {
              const unsigned b = (unsigned)__float_as_int((float)v);
              if (c == 0) lo = b; else hi = b;
            } else {
              const double dv = (double)v;
              lo = (unsigned)__double2loint(dv);
              hi = (unsigned)__double2hiint(dv);
            }
          }
        }
        bits = ((unsigned long long)hi << 32) | (unsigned long long)lo;
      }
      const size_t unit_at = row0 / per_unit + (size_t)u;  // (row0 is a multiple of per_unit: TDS_RING_WIDE)
      if (on) __hip_atomic_store(tds_global((unsigned long long *)ctl.obs_ring) + unit_at, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool to_peers = on && (!rd_only || tail);
      for (int p0 = 0; p0 < np; p0 += 4) {  // (the table is padded to a multiple of four entries)
        const unsigned long long *const b0 = tds_global(tab[p0]), *const b1 = tds_global(tab[p0 + 1]), *const b2 = tds_global(tab[p0 + 2]), *const b3 = tds_global(tab[p0 + 3]);
        const size_t po = (size_t)ctl.peer_off / 8 + unit_at;
        if (to_peers) {
          // (stores through explicitly global pointers: as generic ones they were FLAT stores)
          using G64 = __attribute__((address_space(1))) unsigned long long;
          __hip_atomic_store((G64 *)((unsigned long long *)b0 + po), bits, __ATOMIC_RELAXED, TDS_PEER_SCOPE);
          if (p0 + 1 < np) __hip_atomic_store((G64 *)((unsigned long long *)b1 + po), bits, __ATOMIC_RELAXED, TDS_PEER_SCOPE);
          if (p0 + 2 < np) __hip_atomic_store((G64 *)((unsigned long long *)b2 + po), bits, __ATOMIC_RELAXED, TDS_PEER_SCOPE);
          if (p0 + 3 < np) __hip_atomic_store((G64 *)((unsigned long long *)b3 + po), bits, __ATOMIC_RELAXED, TDS_PEER_SCOPE);
        }
      }
    }
  };
  // the end-of-step records of the PREVIOUS step, from the LDS record (whose state part this step has not touched yet)
  auto flush_prev_records = [&]() {
    if constexpr (DEFER) {
      if ((ring_o || ring_y) && tds_iter > 0) {  // wave-uniform
        const bool mine = valid && mode == TDS_MODE_RUN && xr[in_dim + OUT_SLOT] == T(0);
        if (ring_y && mine)
          put_y_state(tds_global((TR *)ctl.y_ring) + ((size_t)((ctl.y_first + tds_iter - 1) % ctl.y_slots) * ctl.ring_envs + env) * ystr, ystr);
        if (ring_o) {
          // (peer-store exchange: the wavefront's records as one burst where all of its environments store this step)
          if (ctl.peer_arrive != nullptr && (ctl.ring_flags & TDS_RING_WIDE) != 0 && __all(mine))
            put_obs_wide((ctl.obs_first + tds_iter - 1) % ctl.obs_slots);
          else if (mine)
            put_obs((ctl.obs_first + tds_iter - 1) % ctl.obs_slots);
        }
      }
    }
  };
  // the previous step's records are visible device-wide: count this workgroup in (TdsStepCtl::progress — what the exchange
  // of the multi-GPU layer polls, tds_shard.hip).  Called by the wavefront that stored them, half a step later.
  // One counter PER RING SLOT (progress[slot]): the workgroups of a launch run at their own pace — a wavefront whose
  // environments carry more contacts falls steps behind the others over a long launch — so a single running total says
  // nothing about the slowest workgroup; the slot's own counter reaches (uses of the slot) x (workgroups) exactly when
  // EVERY workgroup has stored its records of that step.
  // Peer-store exchange: this workgroup's records of ring slot `pslot` are out — acknowledged by the memory they went to,
  // this rank's and the peers' — so it counts itself in on the slot's arrival counters (two levels, each wrapping at its own
  // count: never reset); the workgroup that completes the slot raises the slot's flag of THIS rank on every rank, its own
  // included, to the launch's sequence number.  Every store of every workgroup was acknowledged before that workgroup's count, and the
  // flag stores are issued after the last count returned: a rank that sees the flag sees the records.
  auto peer_signal = [&](int pslot) {
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    const bool rel = (ctl.ring_flags & TDS_RING_PEER_RELEASE) != 0;  // (A/B switch: tds_kernels.h)
    if (rel) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if ((threadIdx.x & 63) == 0) {
      // two levels (tds_kernels.h: TDS_PEER_SUB): workgroup b on first-level counter b mod SUB, whoever completes one on
      // the second level; every counter wraps at its own count and lives on a line of its own
      constexpr unsigned SUB = TDS_PEER_SUB;
      const unsigned g = gridDim.x, j = blockIdx.x % SUB;
      const unsigned n1 = (g - j + SUB - 1u) / SUB;  // workgroups that count on first-level counter j
      const unsigned n2 = g < SUB ? g : SUB;          // first-level counters in use
      unsigned *const base = ctl.peer_arrive + (size_t)pslot * TDS_PEER_ARRIVE_STRIDE;
      if (atomicInc(base + j * TDS_PEER_LINE, n1 - 1u) == n1 - 1u) {
        if (atomicInc(base + 32 * TDS_PEER_LINE, n2 - 1u) == n2 - 1u) {
          const size_t fi = (size_t)ctl.peer_flag_off + (size_t)pslot * (size_t)ctl.peer_flag_stride;
          if (rel) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "");
          for (int pr = 0; pr <= ctl.n_peers; ++pr)
            __hip_atomic_store(ctl.peer_flags[pr] + fi, ctl.peer_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  };
  auto signal_progress = [&](int back = 1) {  // counts in the records of step tds_iter - back
    if constexpr (LOOP) {
      if (ctl.peer_arrive != nullptr) {  // wave-uniform
        if (tds_iter >= back) peer_signal((ctl.obs_first + tds_iter - back) % ctl.obs_slots);
        return;
      }
    }
    if (ctl.progress != nullptr && tds_iter >= back) {  // wave-uniform
      if (ctl.ring_flags & TDS_RING_NOFENCE)
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the obs ring's write-through stores have reached the L2 / memory
      else
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const int pslot = (ctl.obs_first + tds_iter - back) % ctl.obs_slots;
      if ((threadIdx.x & 63) == 0)
        __hip_atomic_fetch_add(ctl.progress + pslot, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  const bool pack_y = ring_y ? (valid && mode == TDS_MODE_RUN) : (last_run && y_out != nullptr);
  TR *const y_step = ring_y ? tds_global((TR *)ctl.y_ring) + ((size_t)((ctl.y_first + tds_iter) % ctl.y_slots) * ctl.ring_envs + env) * ystr
                            : y_out + (size_t)env * (LOOP ? out_dim : ystr);
  // ---- the phases that a two-wavefront workgroup hands to its helper wavefront, as closures (each derives the LDS
  //      addresses it needs itself: nothing is kept live for them across the phases in between)
  const int NCPp = L.NCPp;
  const int ZR = L.zrows;
  const int OVR = L.ovrows;  // surplus rows available per environment in the slab
  // ---- I. narrowphase: penetrating contact points of this lane group, compacted into cpx; returns their number
  auto phase_I = [&]() -> int {
    T *const Xw = E + L.Xw;
    T *const cpx = E + L.cp;  // [5][NCPp]: point_on_b (3), distance, ancestor-dof mask (bit pattern)
    int na = 0;
  if (pf_ncp > 0) {
      const int ncp = pf_ncp;
      for (int base = 0; base < ncp; base += G) {
        const int k = base + lane;
        bool act = false;
        T Pb[3] = {T(0), T(0), T(0)}, dist = T(0);
        int lk = -1;
        unsigned lk_anc = 0u;
        if (k < ncp) {
          const bool first = base == 0;  // wave-uniform: the first pass was prefetched at kernel start
          lk = pf_cp_link;
          lk_anc = pf_cp_anc;
          if (!first) {
            lk = mdl->cp_link[k];
            lk_anc = mdl->anc_dofs[lk >= 0 ? lk : 0];
          }
          // (the link's record is read unconditionally — index clamped — and a geometry of the BASE takes the base
          //  frame as values afterwards: written as "LDS record or model constant" in two branches, the compiler folds
          //  the two into one FLAT load through a selected pointer — 24 of them on the hot path)
          T Rl[9], pl[3];
          {
            const int lkc = lk >= 0 ? lk : 0;
#pragma unroll
            for (int c = 0; c < 9; ++c) Rl[c] = Xw[lkc * TDS_S1 + c];
#pragma unroll
            for (int c = 0; c < 3; ++c) pl[c] = Xw[lkc * TDS_S1 + 9 + c];
            if (__any(lk < 0)) {  // wave-uniform, rare: a geometry on a base link (-1 - b: base of body b)
              const bool on_base = lk < 0;
              const int bl = (two && on_base) ? -1 - lk : 0;
#pragma unroll
              for (int c = 0; c < 9; ++c) Rl[c] = on_base ? (two ? mdl->base_Rb[bl][c] : mdl->base_R[c]) : Rl[c];
#pragma unroll
              for (int c = 0; c < 3; ++c) pl[c] = on_base ? (two ? mdl->base_tb[bl][c] : mdl->base_t[c]) : pl[c];
            }
          }
          T loc[3] = {pf_cp_loc[0], pf_cp_loc[1], pf_cp_loc[2]};
          T rad = pf_cp_rad;
          if (!first) {
            loc[0] = mdl->cp_local[0][k];
            loc[1] = mdl->cp_local[1][k];
            loc[2] = mdl->cp_local[2][k];
            rad = mdl->cp_radius[k];
          }
          T ctr[3];
          mat3_mulv(Rl, loc, ctr);
          ctr[0] += pl[0];
          ctr[1] += pl[1];
          ctr[2] += pl[2];
          const T n[3] = {pf_plane_n[0], pf_plane_n[1], pf_plane_n[2]};
          // t = -(dot(p, -n) + c);  distance = t - r;  point_on_b = p - r n
          const T t = -((-dot3(ctr, n)) + pf_plane_c);
          dist = t - rad;
          Pb[0] = ctr[0] - rad * n[0];
          Pb[1] = ctr[1] - rad * n[1];
          Pb[2] = ctr[2] - rad * n[2];
          act = live && dist < T(0);  // collision mask, mb_constraint_solver.hpp:285
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
          cpx[4 * NCPp + slot] = bits_to_scalar<T>(lk >= 0 ? lk_anc : 0u);
        }
        na += __popcll(mine);
      }
    }
    return na;
  };
  // the largest contact count among the wavefront's environments (wave-uniform)
  auto wave_max = [&](int n_mine) -> int {
    int NAv = n_mine;
#pragma unroll
    for (int msk = G; msk < 64; msk <<= 1) {
      const int o = __shfl_xor(NAv, msk, 64);
      NAv = o > NAv ? o : NAv;
    }
    return __builtin_amdgcn_readfirstlane(NAv);
  };
  auto phase_M1 = [&](const bool late = false) {  // (late: not behind the narrowphase — the prefetched visual is gone)
    T *const Xw = E + L.Xw;
  // ---- M1. visual poses of y (they use the PRE-step X_world, locomotion_contact_simulation.h:281-299)
    {
      // (late: the record's address is derived again, from a laundered copy of the environment index — carried from the
      //  top of the iteration through the narrowphase, the rows and the row solves it costs the build its last registers)
      int env_l = env;
      const T *Xw_l = Xw;
      if (late) {
        int lane_2 = lane0, grp_2 = grp0;
        asm volatile("" : "+v"(lane_2), "+v"(grp_2));
        env_l = blockIdx.x * EPW + grp_2;
        Xw_l = sm + grp_2 * L.stride + L.Xw;
      }
      TR *const yo = !late ? y_step
                           : (ring_y ? tds_global((TR *)ctl.y_ring) + ((size_t)((ctl.y_first + tds_iter) % ctl.y_slots) * ctl.ring_envs + env_l) * ystr
                                     : y_out + (size_t)env_l * (LOOP ? out_dim : ystr));
      const int nv = pf_nv;
      const int vbase = nq + nd;
      if (pack_y) {  // y describes the last normal step of the launch (y ring: every step its own slot)
        for (int k = lane; k < nv; k += G) {
          // wave-uniform: visual == lane was prefetched — at kernel start (straight-line builds), or just ahead of the
          // narrowphase of THIS iteration (step-loop builds below 24 dof: load_phase_consts); the wider step-loop builds
          // fetch their phase constants at the top of the iteration and read the visuals here
          // (not the two-wavefronts-per-SIMD step-loop build: carrying 24 more registers through the narrowphase costs it
          //  116 B of scratch)
          const bool first = !late && (!LOOP || (NDP < 24 && LP != 2)) && k == lane;
          int lk = pf_vis_link;
          if (!first) lk = mdl->vis_link[k];
          T Rl[9], pl[3], Rv[9], pv[3];
#pragma unroll
          for (int c = 0; c < 9; ++c) Rl[c] = Xw_l[lk * TDS_S1 + c];
#pragma unroll
          for (int c = 0; c < 3; ++c) pl[c] = Xw_l[lk * TDS_S1 + 9 + c];
#pragma unroll
          for (int c = 0; c < 9; ++c) Rv[c] = pf_vis_X[c];
#pragma unroll
          for (int c = 0; c < 3; ++c) pv[c] = pf_vis_X[9 + c];
          if (!first) {
#pragma unroll
            for (int c = 0; c < 9; ++c) Rv[c] = mdl->vis_X[c][k];
#pragma unroll
            for (int c = 0; c < 3; ++c) pv[c] = mdl->vis_X[9 + c][k];
          }
          T Ro[9], po[3], qo[4];
          mat3_mul(Rl, Rv, Ro);
          mat3_mulv(Rl, pv, po);
          matrix_to_quat(Ro, qo);
          TR *o = yo + vbase + 7 * k;
          TDS_NT_STORE((TR)(pl[0] + po[0]), &o[0]);
          TDS_NT_STORE((TR)(pl[1] + po[1]), &o[1]);
          TDS_NT_STORE((TR)(pl[2] + po[2]), &o[2]);
          TDS_NT_STORE((TR)qo[0], &o[3]);
          TDS_NT_STORE((TR)qo[1], &o[4]);
          TDS_NT_STORE((TR)qo[2], &o[5]);
          TDS_NT_STORE((TR)qo[3], &o[6]);
          if constexpr (LOOP) {
            // (the LAST step of a launch with a y ring also leaves its record in the handle's y record: what a device
            //  copy of the ring slot behind the launch used to do, at the price of a dispatch per call)
            if (ring_y && last_run && y_out != nullptr) {  // wave-uniform
              TR *o2 = y_out + (size_t)env_l * out_dim + vbase + 7 * k;
              o2[0] = (TR)(pl[0] + po[0]);
              o2[1] = (TR)(pl[1] + po[1]);
              o2[2] = (TR)(pl[2] + po[2]);
              o2[3] = (TR)qo[0];
              o2[4] = (TR)qo[1];
              o2[5] = (TR)qo[2];
              o2[6] = (TR)qo[3];
            }
          }
        }
      }
    }
  };
  // ---- J. constraint Jacobian rows of the wavefront's contact slots (lane == dof)
  auto phase_J = [&](const int na, const int NA) {
    T *const swd = E + L.swd;
    T *const cpx = E + L.cp;
    T *const Zs = E + L.Z;
    volatile T *const zov = (ovf != nullptr && live) ? ovf + (size_t)env * OVR * (NDs + 3) : nullptr;
    const T nb[3] = {pf_nb[0], pf_nb[1], pf_nb[2]};
    const T t1[3] = {pf_t1[0], pf_t1[1], pf_t1[2]};
    const T t2[3] = {pf_t2[0], pf_t2[1], pf_t2[2]};
    {
      // column d of the point Jacobian of contact point P: col = s_lin - P x s_ang  (xs.bottom = st.bottom -
      // point x st.top, jacobian.hpp:56-72); its components along n, t1, t2 are affine in P:
      //   e . col = e . s_lin + P . (e x s_ang)      -> three FMAs per row instead of a cross and a dot
      const int d = lane;
      T sd[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) sd[k] = d < nd ? swd[k * NDs + d] : T(0);
      if ((fl || flm) && bdof) {
        // the reference's point Jacobian takes the base dofs along WORLD axes about the base origin:
        // [ -[r]x | 1 ],  r = point - base position   (jacobian.hpp:39-56)
        const int kb = d - dbase0;
        const T pb[3] = {xr[dfbq + 4], xr[dfbq + 5], xr[dfbq + 6]};
#pragma unroll
        for (int k = 0; k < 6; ++k) sd[k] = T(0);
        if (kb < 3) {
          const T e[3] = {kb == 0 ? T(1) : T(0), kb == 1 ? T(1) : T(0), kb == 2 ? T(1) : T(0)};
          sd[0] = e[0];
          sd[1] = e[1];
          sd[2] = e[2];
          cross3(pb, e, sd + 3);
        } else {
          sd[kb] = T(1);
        }
      }
      T cnv[3], c1v[3], c2v[3];
      cross3(nb, sd, cnv);
      cross3(t1, sd, c1v);
      cross3(t2, sd, c2v);
      const T cn0 = dot3(nb, sd + 3), c10 = dot3(t1, sd + 3), c20 = dot3(t2, sd + 3);
      // software pipeline over the wavefront's contact slots: slot a + 1 is fetched while a is written
      T Pn[3] = {cpx[0], cpx[NCPp], cpx[2 * NCPp]};
      unsigned mskn = scalar_to_bits<T>(cpx[4 * NCPp]);
      const int lastc = NCPp - 1;
      for (int a = 0; a < NA; ++a) {
        const T P[3] = {Pn[0], Pn[1], Pn[2]};
        const unsigned msk = mskn;
        const int an = a + 1 < lastc ? a + 1 : lastc;
        Pn[0] = cpx[0 * NCPp + an];
        Pn[1] = cpx[1 * NCPp + an];
        Pn[2] = cpx[2 * NCPp + an];
        mskn = scalar_to_bits<T>(cpx[4 * NCPp + an]);
        if (d < NDP && a < na) {
          const bool on = d < nd && ((msk >> d) & 1u);
          const T jn = on ? cn0 + dot3(P, cnv) : T(0);
          const T j1 = on ? c10 + dot3(P, c1v) : T(0);
          const T j2 = on ? c20 + dot3(P, c2v) : T(0);
          const int r0 = a, r1 = NA + a, r2 = 2 * NA + a;
          if (r0 < ZR) Zs[r0 * NDs + d] = jn; else zov[(r0 - ZR) * NDs + d] = jn;
          if (r1 < ZR) Zs[r1 * NDs + d] = j1; else zov[(r1 - ZR) * NDs + d] = j1;
          if (r2 < ZR) Zs[r2 * NDs + d] = j2; else zov[(r2 - ZR) * NDs + d] = j2;
        }
      }
    }
  };

  // ---- multi-body worlds (KIND 3): contacts between a geometry of body A and a geometry of body B, body pair by body
  //      pair (world.hpp:206-282; contact_sphere_sphere / contact_capsule_sphere, contact_point.hpp:43-94, 405-438; the
  //      dispatcher's swapped order :478-495).  lane == pair contact point; the penetrating ones are compacted into
  //      pcx [17][NPCp]: point on A (3) | point on B (3) | normal on B (3) | tangent 1 (3) | tangent 2 (3) | distance |
  //      dofs on the two paths base -> link (bit pattern); the pairs' contacts one behind the other.  Returns their
  //      numbers, eight bits per body pair.
  const int NPCp = L.NPCp;
  auto phase_I2 = [&]() -> unsigned long long {
    unsigned long long cnts = 0ull;
    if constexpr (two) {
      T *const Xw = E + L.Xw;
      T *const pcx = E + L.pc;
      int nb_all = 0;  // slots taken by the earlier pairs
      const int nbp = mdl->num_bpairs;
      for (int pr = 0; pr < nbp; ++pr) {
      const int pc0 = mdl->bpair_pc0[pr], npc = mdl->bpair_pc0[pr + 1];
      const int bd_a = mdl->bpair_a[pr], bd_b = mdl->bpair_b[pr];
      int nb2 = 0;
      for (int base = pc0; base < npc; base += G) {
        const int k = base + lane;
        bool act = false;
        T Pa[3] = {T(0), T(0), T(0)}, Pb[3] = {T(0), T(0), T(0)}, nn[3] = {T(0), T(0), T(0)}, dist = T(0);
        unsigned msk = 0u;
        if (k < npc) {
          const int la = mdl->pc_link_a[k], lb = mdl->pc_link_b[k];
          msk = (la >= 0 ? mdl->anc_dofs[la] : 0u) | (lb >= 0 ? mdl->anc_dofs[lb] : 0u);
          T Ra[9], pa[3], Rb[9], pb[3];
          const int lac = la >= 0 ? la : 0, lbc = lb >= 0 ? lb : 0;
#pragma unroll
          for (int c = 0; c < 9; ++c) {
            Ra[c] = Xw[lac * TDS_S1 + c];
            Rb[c] = Xw[lbc * TDS_S1 + c];
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            pa[c] = Xw[lac * TDS_S1 + 9 + c];
            pb[c] = Xw[lbc * TDS_S1 + 9 + c];
          }
          if (__any(la < 0 || lb < 0)) {  // wave-uniform, rare: a geometry on a base
#pragma unroll
            for (int c = 0; c < 9; ++c) {
              Ra[c] = la < 0 ? mdl->base_Rb[bd_a][c] : Ra[c];
              Rb[c] = lb < 0 ? mdl->base_Rb[bd_b][c] : Rb[c];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              pa[c] = la < 0 ? mdl->base_tb[bd_a][c] : pa[c];
              pb[c] = lb < 0 ? mdl->base_tb[bd_b][c] : pb[c];
            }
          }
          const T la3[3] = {mdl->pc_loc_a[0][k], mdl->pc_loc_a[1][k], mdl->pc_loc_a[2][k]};
          const T lb3[3] = {mdl->pc_loc_b[0][k], mdl->pc_loc_b[1][k], mdl->pc_loc_b[2][k]};
          const T ra = mdl->pc_rad_a[k], rb = mdl->pc_rad_b[k];
          T cA[3], cB[3];
          mat3_mulv(Ra, la3, cA);
          mat3_mulv(Rb, lb3, cB);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            cA[c] += pa[c];
            cB[c] += pb[c];
          }
          const bool swp = mdl->pc_swap[k] != 0;
          // contact_sphere_sphere(X, Y): diff = pX - pY, n = diff / |diff|, point_x = pX - rX n, point_y = point_x - dist n;
          // direct: X = A, Y = B.  Swapped (sphere of A, capsule end of B): X = B's end sphere, Y = A's sphere, and the
          // dispatcher exchanges the points and negates the normal afterwards.
          const T *const cX = swp ? cB : cA, *const cY = swp ? cA : cB;
          const T rX = swp ? rb : ra, rY = swp ? ra : rb;
          const T diff[3] = {cX[0] - cY[0], cX[1] - cY[1], cX[2] - cY[2]};
          const T len = sqrt_t<T>(dot3(diff, diff));
          dist = len - (rX + rY);
          const T inv = T(1) / len;
          const T nx[3] = {inv * diff[0], inv * diff[1], inv * diff[2]};
          const T px[3] = {cX[0] - rX * nx[0], cX[1] - rX * nx[1], cX[2] - rX * nx[2]};
          const T py[3] = {px[0] - dist * nx[0], px[1] - dist * nx[1], px[2] - dist * nx[2]};
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            Pa[c] = swp ? py[c] : px[c];
            Pb[c] = swp ? px[c] : py[c];
            nn[c] = swp ? -nx[c] : nx[c];
          }
          act = live && len > T(1e-5) && dist < T(0);  // CONTACT_EPSILON; collision mask, mb_constraint_solver.hpp:285
        }
        const unsigned long long bal = __ballot(act);
        const unsigned long long mine = (G == 64) ? bal : ((bal >> (grp * G)) & ((1ull << (G & 63)) - 1ull));
        const int pre = __popcll(mine & ((1ull << lane) - 1ull));
        if (act) {
          const int slot = nb_all + nb2 + pre;
          T t1[3], t2[3];
          plane_space_dev<T>(nn, t1, t2);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            pcx[(0 + c) * NPCp + slot] = Pa[c];
            pcx[(3 + c) * NPCp + slot] = Pb[c];
            pcx[(6 + c) * NPCp + slot] = nn[c];
            pcx[(9 + c) * NPCp + slot] = t1[c];
            pcx[(12 + c) * NPCp + slot] = t2[c];
          }
          pcx[15 * NPCp + slot] = dist;
          pcx[16 * NPCp + slot] = bits_to_scalar<T>(msk);
        }
        nb2 += __popcll(mine);
      }
      cnts |= (unsigned long long)nb2 << (8 * pr);
      nb_all += nb2;
      }
    }
    return cnts;
  };
  // rows of the contacts between the bodies (lane == dof): the relative velocity is vel_a - vel_b, so a dof of body
  // A enters with -J_a, a dof of body B with +J_b (mb_constraint_solver.hpp:278-388: rows [J_a | J_b], right-hand
  // side from J_a qd_a - J_b qd_b, qd_a += M_a^-1 J_a^T p, qd_b -= M_b^-1 J_b^T p); each body's column is taken at
  // ITS contact point (point_jacobian2(mb_a, link_a, world_point_on_a) / (mb_b, link_b, world_point_on_b))
  auto phase_J2 = [&](const int pr, const int off, const int nb2, const int NB) {
    if constexpr (two) {
      T *const swd = E + L.swd;
      T *const pcx = E + L.pc + off;  // (the pair's contacts sit behind those of the earlier pairs)
      T *const Zs = E + L.Z;
      volatile T *const zov = (ovf != nullptr && live) ? ovf + (size_t)env * OVR * (NDs + 3) : nullptr;
      const int d = lane;
      T sd[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) sd[k] = d < nd ? swd[k * NDs + d] : T(0);
      if (flm && bdof) {  // base dofs of a floating body: WORLD axes about the base origin (see phase J)
        const int kb = d - dbase0;
        const T pb[3] = {xr[dfbq + 4], xr[dfbq + 5], xr[dfbq + 6]};
#pragma unroll
        for (int k = 0; k < 6; ++k) sd[k] = T(0);
        if (kb < 3) {
          const T e[3] = {kb == 0 ? T(1) : T(0), kb == 1 ? T(1) : T(0), kb == 2 ? T(1) : T(0)};
          sd[0] = e[0];
          sd[1] = e[1];
          sd[2] = e[2];
          cross3(pb, e, sd + 3);
        } else {
          sd[kb] = T(1);
        }
      }
      const int bd_a = mdl->bpair_a[pr];
      const bool of_a = d >= mdl->body_dof0[bd_a] && d < mdl->body_dof0[bd_a + 1];  // (a dof of a third body: masked out)
      const T sgn = of_a ? T(-1) : T(1);
      for (int a = 0; a < NB; ++a) {
        if (d < NDP && a < nb2) {
          const int o = of_a ? 0 : 3;
          const T P[3] = {pcx[(o + 0) * NPCp + a], pcx[(o + 1) * NPCp + a], pcx[(o + 2) * NPCp + a]};
          const unsigned msk = scalar_to_bits<T>(pcx[16 * NPCp + a]);
          const bool on = d < nd && ((msk >> d) & 1u);
          T val[3];
#pragma unroll
          for (int e = 0; e < 3; ++e) {
            const T dir[3] = {pcx[(6 + 3 * e + 0) * NPCp + a], pcx[(6 + 3 * e + 1) * NPCp + a], pcx[(6 + 3 * e + 2) * NPCp + a]};
            T cr[3];
            cross3(dir, sd, cr);  // e . col = e . s_lin + P . (e x s_ang)
            val[e] = on ? sgn * (dot3(dir, sd + 3) + dot3(P, cr)) : T(0);
          }
          const int r0 = a, r1 = NB + a, r2 = 2 * NB + a;
          if (r0 < ZR) Zs[r0 * NDs + d] = val[0]; else zov[(r0 - ZR) * NDs + d] = val[0];
          if (r1 < ZR) Zs[r1 * NDs + d] = val[1]; else zov[(r1 - ZR) * NDs + d] = val[1];
          if (r2 < ZR) Zs[r2 * NDs + d] = val[2]; else zov[(r2 - ZR) * NDs + d] = val[2];
        }
      }
    }
  };

  if constexpr (W2) {
    if (wv == 1) {
      // ================= helper wavefront of a two-wavefront workgroup =================
      // (step-loop build: the helper loops along, step by step; besides the contact pipeline it is the RECORDER — visual
      //  poses, and from the LDS record the previous step's y state + obs record — so that per-step records cost the main
      //  wavefront's dependent chain nothing but the reward block)
      if constexpr (LOOP) load_phase_consts(mdl);
      TDS_STAMP(1);
      if constexpr (LOOP) {  // (TDS_RING_SIGNAL_LATE: the records stored in the iteration before this one)
        if (ctl.ring_flags & TDS_RING_SIGNAL_LATE) signal_progress(2);
      }
      if constexpr (LOOP && TDS_FLUSH_EARLY != 0) flush_prev_records();  // (behind barrier (0): see there)
      // (the helper's side of barrier (1) does not wait for its global stores — the previous step's records, and in the
      //  peer-store exchange the rows to every peer, issued just above: __syncthreads() is `s_waitcnt vmcnt(0) lgkmcnt(0)`
      //  + s_barrier, i.e. it held the barrier until the slowest of those stores was acknowledged.  -DTDS_LIGHT_BARRIER1=0)
#ifndef TDS_LIGHT_BARRIER1
#define TDS_LIGHT_BARRIER1 1
#endif
      if constexpr (LOOP && TDS_LIGHT_BARRIER1 != 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else __syncthreads();  // (1) the main wavefront has written the x record, X_world and the motion axes
      TDS_STAMP(2);
      T *const cpx = E + L.cp;
      T *const Zs = E + L.Z;
      T *const rws = E + L.rows;
      T *const xs = E + L.xrow;
      const int na_h = phase_I();
      const bool contacts_h = __any(na_h > 0) != 0;
      const int NA_h = wave_max(na_h);
      const bool split_ok = 3 * NA_h <= ZR;  // (more rows than the LDS store holds: the main wavefront takes the slab path)
      if constexpr (!PIPE) {
        if (lane == 0) {  // contact count of the environment / row slots of the wavefront, for the main wavefront
          xr[in_dim + 2] = bits_to_scalar<T>((unsigned)na_h);
          xr[in_dim + 3] = bits_to_scalar<T>((unsigned)NA_h);
        }
      }
      TDS_STAMP(3);
      // (the visual poses go out behind barrier (3), in this wavefront's idle tail — see there; not with the Gram form of
      //  the contact solve, whose buffer takes X_world's place behind barrier (2))
      if constexpr (!LOOP) {
        if (L.gram_ok) phase_M1();
      }
      if constexpr (!(LOOP && TDS_FLUSH_EARLY != 0)) flush_prev_records();
      if (pack_y && !(DEFER && ring_y)) {  // tail of the y record: up_dot_world_z, zero padding
        TR *const yo = y_step;
        int tail = nq + nd;
        if (mdl->pack_visuals) {
          tail += 7 * mdl->num_visuals;
          if (lane == 0) TDS_NT_STORE((TR)(mdl->base_R[8]), &yo[tail]);
          tail += 1;
        }
        for (int i = tail + lane; i < ystr; i += G) TDS_NT_STORE((TR)(0), &yo[i]);
      }
      TDS_STAMP(4);
      if (contacts_h && split_ok) phase_J(na_h, NA_h);
      TDS_STAMP(5);
      if constexpr (PIPE) {
        // (2') contact list and rows are in LDS: the counts double as the "ready" signal the main wavefront polls
        TDS_WAVE_SYNC();
        if (lane == 0) xr[in_dim + 2] = bits_to_scalar<T>((unsigned)na_h);
        TDS_WAVE_SYNC();
        if (lane == 0) xr[in_dim + 3] = bits_to_scalar<T>((unsigned)NA_h);
      } else {
        __syncthreads();  // (2) the factors L, 1/D are in LDS; the rows and the contact list are visible to the main wavefront
      }
      TDS_STAMP(6);
      // (the row solves end at barrier (3), where the main wavefront arrives from its forward-dynamics solve: from here to
      //  the barrier the helper is as critical as the main wavefront it shares its SIMD with — same priority.  Measured,
      //  profiles/r04_ab_slots7_helper_priority.txt: 13.37 -> 13.21 us per step; the main wavefront was waiting ~2 k cycles
      //  at barrier (3) once it had priority everywhere, profiles/r04f_ant4096_f64_phases.txt.  From the Jacobian rows or
      //  the narrowphase on instead: 12.99 / 13.29 against 12.71, no priorities at all 13.58,
      //  profiles/r04_ab_slots9_helper_from.txt)
      if constexpr (TDS_MAIN_PRIO > 0) __builtin_amdgcn_s_setprio(TDS_MAIN_PRIO);
      if (contacts_h && split_ok)
        tds_row_solve<false, T, G, NDP, true>(lane, NA_h, na_h, nd, ZR, OVR, NCPp, Zs, rws, xs, xr + nq, cpx, E + L.Lp,
                                             E + L.dinv, nullptr, nullptr, pf_cfm, pf_erp_dt, pf_rest,
                                             L.gram_ok ? nullptr : xr + in_dim + 4, dt,
                                             PIPE ? xr + in_dim + 5 : nullptr, PIPE ? E + L.Lh : nullptr);
      TDS_STAMP(7);
      if constexpr (TDS_MAIN_PRIO > 0) __builtin_amdgcn_s_setprio(0);
      if constexpr (LOOP) {  // (the records this wavefront stored behind the visual poses)
        if (!(ctl.ring_flags & TDS_RING_SIGNAL_LATE) || last_run) signal_progress();
      }
      __syncthreads();  // (3) z~ rows and their scalars are final
      TDS_STAMP(8);
      // The visual poses of this step go out HERE, in the helper's idle tail, not between the narrowphase
      // and the Jacobian rows: with the main wavefront at priority the helper's I -> M1 -> J -> K had become the longer
      // path to barrier (3) (profiles/r04f_ant4096_f64_phases.txt: the main wavefront waited 2.1 k cycles there).  X_world
      // stays in LDS until the main wavefront's next kinematics sweep; in the step-loop builds the main wavefront makes sure
      // of that with the flag below at the top of its next iteration (normally long set).  Measured with the priority above
      // (profiles/r04_ab_slots8_m1_late.txt): 13.40 -> 12.86 us per step.  (Round 3 had tried this move without the
      // priorities: no gain — the helper was not the critical path then.)
      if (LOOP || !L.gram_ok) phase_M1(true);
      if constexpr (LOOP) {
        TDS_WAVE_SYNC();
        if (lane == 0) xr[in_dim + 4] = T(2);  // "poses are out" (the slot of the y~ flag, free until the next barrier (1))
      }
      if constexpr (LOOP) {
        // the same bookkeeping the main wavefront does for a plain / replayed step (these launches have no settle steps)
        if (mode == TDS_MODE_RUN && --left == 0) mode = TDS_MODE_IDLE;
        ++tds_iter;
        continue;
      } else {
        return;
      }
    }
  }

  const T q = (di >= 0 && (!gen || (qri >= 0 && !sph_lane))) ? xr[gen ? (qri >= 0 ? qri : 0) : qri] : T(0);
  const T qd = di >= 0 ? xr[nq + qdri] : T(0);

  // ---- PD controller (locomotion_contact_simulation.h:168-258) or direct torque -------------
  // (a closure, evaluated right here under the latency of the constant loads.  Evaluating it just ahead of phase F, where
  //  tau is first needed, was tried for the 18-dof kernels, whose 12 B of scratch are tau carried through the sweeps: the
  //  allocator then spilled 20 B elsewhere — measured with tools/kernel_resources.sh, not kept)
  T tau = T(0);
  constexpr bool LATE_PD = false;
  auto compute_tau = [&]() {
    if (step_mode == TDS_STEP_LOCOMOTION) {
      const int ai = act_i;
      if (ai >= 0) {
        const int var = nq + nd + adim;
        const T kp = xr[var], kd = xr[var + 1], max_force = xr[var + 2];
        T a = settling ? T(0) : xr[nq + nd + ai];  // reset settles with zero action
        const T lim = act_lim;
        a = a < lim ? a : lim;       // Algebra::min(clamped_action, ACTION_LIMIT)
        a = a > -lim ? a : -lim;     // Algebra::max(clamped_action, -ACTION_LIMIT)
        const T q_des = init_pose_l + a;
        T f = kp * (q_des - q) + kd * (T(0) - qd);
        f = f > -max_force ? f : -max_force;
        f = f < max_force ? f : max_force;
        tau = f;
      } else if (sph && ai <= -2) {
        // spherical branch (locomotion_contact_simulation.h:188-226): q_desired = identity, qd_desired = 0;
        // position_error = matrix_to_euler_xyz(quat_to_matrix(inverse(identity) * q_actual)) (matrix_utils.hpp:18-90),
        // lane k of the joint takes component k; the clamped force goes to tau (this lane was kept by the builder:
        // floating base or link index >= 4, :215-221)
        const int qo = sphq;
        const int var = nq + nd + adim;
        const T kp = xr[var], kd = xr[var + 1], max_force = xr[var + 2];
        const T qx = xr[qo], qy = xr[qo + 1], qz = xr[qo + 2], qw = xr[qo + 3];
        const T s2 = T(2) / (qx * qx + qy * qy + qz * qz + qw * qw);  // tiny_matrix3x3.h:315-340
        const T xs = qx * s2, ys = qy * s2, zs = qz * s2;
        const T wx = qw * xs, wy = qw * ys, wz = qw * zs;
        const T xx = qx * xs, xy = qx * ys, xz = qx * zs;
        const T yy = qy * ys, yz = qy * zs, zz = qz * zs;
        // get_matrix_elem(mat, k) = mat(k % 3, k / 3) (matrix_utils.hpp:11-16)
        const T e0 = T(1) - (yy + zz), e3 = xy - wz, e1 = xy + wz, e4 = T(1) - (xx + zz);
        const T fi = xz - wy, e5 = yz + wx, e8 = T(1) - (xx + yy);
        const bool le = fi <= T(1), ge = fi >= T(-1);
        const int k = jt - TDS_JOINT_SPH0;
        T pe;
        if (k == 1) {
          pe = le ? (ge ? asin_t<T>(fi) : -T(1.57079632679489661923)) : T(1.57079632679489661923);
        } else {
          const bool mid = le && ge;
          // k == 0: atan2(-e5, e8) | -atan2(e3, e4) | atan2(e3, e4);   k == 2: atan2(-e1, e0) | 0 | 0
          const T ay = k == 0 ? (mid ? -e5 : e3) : -e1;
          const T ax = k == 0 ? (mid ? e8 : e4) : e0;
          const T a = atan2_t<T>(ay, ax);
          pe = k == 0 ? ((le && !ge) ? -a : a) : (mid ? a : T(0));
        }
        T f = kp * pe + kd * (T(0) - qd);
        f = f > -max_force ? f : -max_force;
        f = f < max_force ? f : max_force;
        tau = f;
      }
    } else if (flm) {  // (torques of the joints of all bodies, one behind the other; base dofs carry none)
      const int ti = isl ? mdl->tau_rec[lsafe] : -1;
      if (ti >= 0) tau = settling ? T(0) : xr[nq + nd + ti];
    } else if (di >= 0 && di < adim) {  // (adim == joint dofs: the base dofs of a floating base carry no torque)
      tau = settling ? T(0) : xr[nq + nd + di];
    }
    // joint stiffness / damping (forward_dynamics.hpp:122-123)
    if (isl) tau -= stiff_l * q + damp_l * qd;
    if constexpr (sph) {
      // spherical joint: tau -= stiffness * quaternion_axis_angle(quat) (forward_dynamics.hpp:70-74,
      // tiny_algebra.hpp:509-527), component k on lane k (q == 0 on these lanes)
      if (sphq >= 0 && stiff_l != T(0)) {
        const T qx = xr[sphq], qy = xr[sphq + 1], qz = xr[sphq + 2], qw = xr[sphq + 3];
        const T qn = sqrt_t<T>(qx * qx + qy * qy + qz * qz);
        const T theta = T(2) * atan2_t<T>(qn, qw);
        // pow(epsilon, 1/4) = 2^-13
        const T sc = qn < T(1.220703125e-4) ? T(1) / (T(0.5) + theta * theta * (T(1) / T(48))) : theta / qn;
        tau -= stiff_l * sc * xr[sphq + (jt - TDS_JOINT_SPH0)];
      }
    }

  };
  if constexpr (!LATE_PD) compute_tau();
  // 18-dof straight-line kernels (Laikago): tau would be the one value carried in registers from here through every
  // sweep to phase F — it waits in LDS instead (a slot of its own per link)
  constexpr bool PARK_TAU = (!LOOP && ((!W2 && NDP > 16 && NDP < 24) || (W2 && NDP <= 16 && TDS_PARK_W2))) ||
                            (LOOP && W2 && NDP <= 16 && TDS_PARK_LOOP);
  if constexpr (PARK_TAU) {
    if (isl) E[L.tau + li] = tau;
  }

  TDS_STAMP(1);
  // ---- B. jcalc: X_parent = X_T * X_J(q)   (link.hpp:229-287) -------------------------------
  if constexpr (LOOP) load_link_consts(mdl);
  if constexpr (LOOP && NDP >= 24) load_phase_consts(mdl);
  T Rp[9], tp[3];
  T sn, cs;  // sin / cos of the lane's joint angle (half angle for REVOLUTE_AXIS); phase C's closed-form root chain uses them
  {
    const bool rev = jt >= TDS_JOINT_REVOLUTE_X && jt <= TDS_JOINT_REVOLUTE_AXIS;
    const bool pris = jt >= TDS_JOINT_PRISMATIC_X && jt <= TDS_JOINT_PRISMATIC_AXIS;
    T RJ[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)};
    T tJ[3] = {T(0), T(0), T(0)};
    if (pris) {  // translation = S.bottom * q (unit axis for _X/_Y/_Z)
      tJ[0] = Sl[3] * q;
      tJ[1] = Sl[4] * q;
      tJ[2] = Sl[5] * q;
    }
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
    if (sph && jt == TDS_JOINT_SPH0) {  // X_J.rotation = quat_to_matrix(q[0..3])  (link.hpp:262-266)
      const int qb = qri >= 0 ? qri : 0;
      const T qx = xr[qb], qy = xr[qb + 1], qz = xr[qb + 2], qw = xr[qb + 3];
      const T s2 = T(2) / (qx * qx + qy * qy + qz * qz + qw * qw);
      const T xs = qx * s2, ys = qy * s2, zs = qz * s2;
      const T wx = qw * xs, wy = qw * ys, wz = qw * zs;
      const T xx = qx * xs, xy = qx * ys, xz = qx * zs;
      const T yy = qy * ys, yz = qy * zs, zz = qz * zs;
      RJ[0] = T(1) - (yy + zz); RJ[1] = xy - wz; RJ[2] = xz + wy;
      RJ[3] = xy + wz; RJ[4] = T(1) - (xx + zz); RJ[5] = yz - wx;
      RJ[6] = xz - wy; RJ[7] = yz + wx; RJ[8] = T(1) - (xx + yy);
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
  T *const Xw = E + L.Xw;    // link-major records [link][TDS_S1]: X_world rot(9) trans(3) | v(6)
  T *const swd = E + L.swd;  // [6][NDs]   per dof
  T *const va = E + L.v;     // side records [slot][TDS_S1]: v(6) | a0(6) of the links with non-chain children
  T R[9], p[3], sw[6], vJ[6], v[6];
  // cb = v x vJ (velocity-product acceleration, kinematics.hpp:96-99); a0 = acceleration the link would
  // have with all joint accelerations zero: a0_i = a0_parent + cb_i, a0_base = -gravity
  T cb[6], a0[6];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = T(0);
#pragma unroll
  for (int k = 0; k < 3; ++k) p[k] = T(0);
#pragma unroll
  for (int k = 0; k < 6; ++k) sw[k] = vJ[k] = v[k] = cb[k] = a0[k] = T(0);
  auto bias_accel = [&]() {  // cb from v, vJ
    cross3(v, vJ, cb);
    T c1[3], c2[3];
    cross3(v, vJ + 3, c1);
    cross3(v + 3, vJ, c2);
    cb[3] = c1[0] + c2[0];
    cb[4] = c1[1] + c2[1];
    cb[5] = c1[2] + c2[2];
  };
  const int nlev = mdl->num_levels;
  const int rkc = mdl->kin_chain_last;  // serial chain 0..rkc from the base in lanes 0..rkc (>= the root joint of E'): -1 = none
  const int eul = KIND == 0 ? mdl->euler_root : 0;  // wave-uniform
  const int leg_len = KIND == 0 ? mdl->leg_len : 0;  // wave-uniform: the links behind the root chain are chains of this length
  if (eul != 0) {
    // ---- The root chain in CLOSED FORM (DevModel::euler_root: links 0..5 = prismatic X, Y, Z, revolute X, Y, Z with
    //      identity X_T — the free motion of the URDF-derived fixed-base robots, the Ant and Laikago).  The chain's six
    //      transforms are  trans(q0, q1, q2) Rx(q3) Ry(q4) Rz(q5);  world pose, motion axis, velocity and bias
    //      acceleration of link 5 (the torso / chassis: the only link of the chain with mass, shapes and children) follow
    //      from 15 lane broadcasts of (q, sin q, cos q, qd) and ~200 independent multiply-adds on every lane — instead of
    //      nine dependent rounds of DPP scans (three 3x3 products deep each) that a lone wavefront sat through at ~10
    //      cycles per instruction.  Same quantities as kinematics.hpp:64-97 produces link by link, other association.
    const T q0 = lane_bcast<T, G, NDP, 0>(q), q1 = lane_bcast<T, G, NDP, 1>(q), q2 = lane_bcast<T, G, NDP, 2>(q);
    const T sx = lane_bcast<T, G, NDP, 3>(sn), cx = lane_bcast<T, G, NDP, 3>(cs);
    const T sy = lane_bcast<T, G, NDP, 4>(sn), cy = lane_bcast<T, G, NDP, 4>(cs);
    const T sz = lane_bcast<T, G, NDP, 5>(sn), cz = lane_bcast<T, G, NDP, 5>(cs);
    // R5 = Rx Ry Rz, revolute axes a3 = e_x, a4 = Rx e_y, a5 = Rx Ry e_z, all through the point P = base + (q0, q1, q2)
    // (base frame == world frame: the host sets euler_root only then — the prismatic axes are the unit vectors, a3 = e_x)
    R[0] = cy * cz;                 R[1] = -cy * sz;                R[2] = sy;
    R[3] = sx * sy * cz + cx * sz;  R[4] = cx * cz - sx * sy * sz;  R[5] = -sx * cy;
    R[6] = sx * sz - cx * sy * cz;  R[7] = cx * sy * sz + sx * cz;  R[8] = cx * cy;
    const T P[3] = {q0 + mdl->base_t[0], q1 + mdl->base_t[1], q2 + mdl->base_t[2]};
    const T A3[3] = {T(1), T(0), T(0)}, A4[3] = {T(0), cx, sx}, A5[3] = {sy, -sx * cy, cx * cy};
    const T C0[3] = {T(1), T(0), T(0)}, C1[3] = {T(0), T(1), T(0)}, C2[3] = {T(0), T(0), T(1)};
    p[0] = P[0]; p[1] = P[1]; p[2] = P[2];
    // this lane's world motion axis: s = (0 | column) for the prismatic links, (A | P x A) for the revolute ones
    {
      const bool pr = li < 3;
      T ang[3], lin[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        ang[k] = pr ? T(0) : (li == 3 ? A3[k] : (li == 4 ? A4[k] : A5[k]));
        lin[k] = li == 0 ? C0[k] : (li == 1 ? C1[k] : C2[k]);
      }
      T c[3];
      cross3(P, ang, c);
      const bool inch6 = isl && li <= 5;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        sw[k] = inch6 ? ang[k] : T(0);
        sw[3 + k] = inch6 ? (pr ? lin[k] : c[k]) : T(0);
      }
    }
    // link 5: v = (W5 | U + P x W5),  a0 = sum_j crm(v_j) vJ_j - g  with  v_j = (W_j | U + P x W_j),  vJ_j = (J_j | P x J_j),
    // J_j = A_j qd_j, W_3 = J_3, W_4 = W_3 + J_4, W_5 = W_4 + J_5  (the prismatic joints contribute nothing: no rotation yet)
    // (scheduling fence: the pose / axis part above is done before the velocity part starts — interleaved, the two
    //  parts' temporaries together cost the narrow kernels their last free registers)
    __builtin_amdgcn_sched_barrier(0);
    {
      const T d0 = lane_bcast<T, G, NDP, 0>(qd), d1 = lane_bcast<T, G, NDP, 1>(qd), d2 = lane_bcast<T, G, NDP, 2>(qd);
      const T d3 = lane_bcast<T, G, NDP, 3>(qd), d4 = lane_bcast<T, G, NDP, 4>(qd), d5 = lane_bcast<T, G, NDP, 5>(qd);
      const T U[3] = {d0, d1, d2};
      const T J3[3] = {A3[0] * d3, A3[1] * d3, A3[2] * d3}, J4[3] = {A4[0] * d4, A4[1] * d4, A4[2] * d4},
              J5[3] = {A5[0] * d5, A5[1] * d5, A5[2] * d5};
      const T W4[3] = {J3[0] + J4[0], J3[1] + J4[1], J3[2] + J4[2]};
      const T W5[3] = {W4[0] + J5[0], W4[1] + J5[1], W4[2] + J5[2]};
      T pJ3[3], pJ4[3], pJ5[3];
      cross3(P, J3, pJ3);
      cross3(P, J4, pJ4);
      cross3(P, J5, pJ5);
      const T pW4[3] = {pJ3[0] + pJ4[0], pJ3[1] + pJ4[1], pJ3[2] + pJ4[2]};
      const T pW5[3] = {pW4[0] + pJ5[0], pW4[1] + pJ5[1], pW4[2] + pJ5[2]};
      const T V3[3] = {U[0] + pJ3[0], U[1] + pJ3[1], U[2] + pJ3[2]};
      const T V4[3] = {U[0] + pW4[0], U[1] + pW4[1], U[2] + pW4[2]};
      const T V5[3] = {U[0] + pW5[0], U[1] + pW5[1], U[2] + pW5[2]};
      // crm(v) vJ = (w x wJ | w x vJ_lin + v_lin x wJ)
      T a45[3], a55[3], t1[3], t2[3], l3[3], l4[3], l5[3];
      cross3(J3, J4, a45);   // W4 x J4 = J3 x J4
      cross3(W4, J5, a55);   // W5 x J5 = W4 x J5
      cross3(J3, pJ3, t1);
      cross3(V3, J3, t2);
      l3[0] = t1[0] + t2[0]; l3[1] = t1[1] + t2[1]; l3[2] = t1[2] + t2[2];
      cross3(W4, pJ4, t1);
      cross3(V4, J4, t2);
      l4[0] = t1[0] + t2[0]; l4[1] = t1[1] + t2[1]; l4[2] = t1[2] + t2[2];
      cross3(W5, pJ5, t1);
      cross3(V5, J5, t2);
      l5[0] = t1[0] + t2[0]; l5[1] = t1[1] + t2[1]; l5[2] = t1[2] + t2[2];
      // (links 0..4: massless, v = a0 = 0 keeps their zero inertia's products finite; with the leg scan below the lanes
      //  of the legs start from the torso's velocity and bias acceleration: their chains hang off it)
      const bool torso = isl && (li == 5 || (leg_len != 0 && li > 5));
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        v[k] = torso ? W5[k] : T(0);
        v[3 + k] = torso ? V5[k] : T(0);
        a0[k] = torso ? a45[k] + a55[k] : T(0);
        a0[3 + k] = torso ? (l3[k] + l4[k] + l5[k]) - mdl->grav[k] : T(0);
      }
    }
    if (leg_len != 0) {
      // ---- The legs as ONE segmented prefix scan: every link behind the torso sits in a serial chain of leg_len
      //      consecutive lanes that hangs off the torso (DevModel::leg_len).  Chain-local prefix products of the joint
      //      transforms (log2(leg_len) rounds of lane shifts), one composition with the torso's pose — which the closed
      //      form above left on EVERY lane —, then prefix sums of the joint velocities and of the velocity-product
      //      accelerations on top of the torso's: the same X_world, s, v, a0 the level loop produces link by link
      //      (kinematics.hpp:64-97), leg_len tree levels and their LDS hand-over from the torso shorter.
      const bool leg = isl && li > 5;
      const int jpos = leg ? ((li - 6) & (leg_len - 1)) : 0;  // my position in my chain
      T Rl[9], pl[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) Rl[k] = Rp[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) pl[k] = tp[k];
      static_for<0, 2>([&](auto dc) {
        constexpr int D = 1 << decltype(dc)::value;
        if (D < leg_len) {  // wave-uniform
          const bool take = jpos >= D;
          T Rq[9], pq[3];
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            const T sh = seg_shr<D, G>(Rl[k]);
            Rq[k] = take ? sh : ((k == 0 || k == 4 || k == 8) ? T(1) : T(0));
          }
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const T sh = seg_shr<D, G>(pl[k]);
            pq[k] = take ? sh : T(0);
          }
          T Rn[9], r[3];
          mat3_mul(Rq, Rl, Rn);
          mat3_mulv(Rq, pl, r);
#pragma unroll
          for (int k = 0; k < 9; ++k) Rl[k] = Rn[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) pl[k] = pq[k] + r[k];
        }
      });
      {
        // into the world: X = X_torso o X_chain  (R, p hold the torso's pose on every lane)
        T Rn[9], r[3];
        mat3_mul(R, Rl, Rn);
        mat3_mulv(R, pl, r);
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = leg ? Rn[k] : R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = leg ? p[k] + r[k] : p[k];
      }
      if (leg) {
        // s = X_world.apply_inverse(S) = (R w, R v + p x (R w))   (transform.hpp:232-243)
        mat3_mulv(R, Sl, sw);
        mat3_mulv(R, Sl + 3, sw + 3);
        T c[3];
        cross3(p, sw, c);
        sw[3] += c[0];
        sw[4] += c[1];
        sw[5] += c[2];
#pragma unroll
        for (int k = 0; k < 6; ++k) vJ[k] = sw[k] * qd;
      }
      {
        T pre[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) pre[k] = leg ? vJ[k] : T(0);
        static_for<0, 2>([&](auto dc) {
          constexpr int D = 1 << decltype(dc)::value;
          if (D < leg_len) {  // wave-uniform
            const bool take = jpos >= D;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              const T sh = seg_shr<D, G>(pre[k]);
              pre[k] += take ? sh : T(0);
            }
          }
        });
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = leg ? v[k] + pre[k] : v[k];
      }
      if (leg) bias_accel();
      {
        T pre[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) pre[k] = leg ? cb[k] : T(0);
        static_for<0, 2>([&](auto dc) {
          constexpr int D = 1 << decltype(dc)::value;
          if (D < leg_len) {  // wave-uniform
            const bool take = jpos >= D;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              const T sh = seg_shr<D, G>(pre[k]);
              pre[k] += take ? sh : T(0);
            }
          }
        });
#pragma unroll
        for (int k = 0; k < 6; ++k) a0[k] = leg ? a0[k] + pre[k] : a0[k];
      }
    }
    if (isl && li <= 5 && lds_children) {  // children other than lane + 1 read my record (link 5: the hips)
#pragma unroll
      for (int k = 0; k < 9; ++k) Xw[li * TDS_S1 + k] = R[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) Xw[li * TDS_S1 + 9 + k] = p[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        va[my_slot * TDS_S1 + k] = v[k];
        va[my_slot * TDS_S1 + 6 + k] = a0[k];
      }
    }
    TDS_WAVE_SYNC();
  } else if (rkc >= 0) {
    // The root chain's world transforms are a prefix product of the local ones along consecutive lanes:
    // inclusive scan with DPP row shifts (ceil(log2(rkc + 1)) <= 4 rounds) instead of rkc+1 tree levels.
    //   (Ra, ta) o (Rb, tb) = (Ra Rb, ta + Ra tb)        (transform.hpp:123-131)
    const bool inch = isl && li <= rkc;
    if (fl) {
      // floating base (kinematics.hpp:35-47): base_X_world = (quat_to_matrix(q[0..3]), q[4..6]) is the frame
      // of all six pseudo links; their motion axes are the base frame's own axes (omega, then v)
      if (inch) {
        const T qx = xr[0], qy = xr[1], qz = xr[2], qw = xr[3];
        const T s2 = T(2) / (qx * qx + qy * qy + qz * qz + qw * qw);  // tiny_matrix3x3.h:315-340
        const T xs = qx * s2, ys = qy * s2, zs = qz * s2;
        const T wx = qw * xs, wy = qw * ys, wz = qw * zs;
        const T xx = qx * xs, xy = qx * ys, xz = qx * zs;
        const T yy = qy * ys, yz = qy * zs, zz = qz * zs;
        R[0] = T(1) - (yy + zz); R[1] = xy - wz; R[2] = xz + wy;
        R[3] = xy + wz; R[4] = T(1) - (xx + zz); R[5] = yz - wx;
        R[6] = xz - wy; R[7] = yz + wx; R[8] = T(1) - (xx + yy);
        p[0] = xr[4];
        p[1] = xr[5];
        p[2] = xr[6];
      }
    } else if (li == 0) {  // link 0 hangs off the base
      mat3_mul(mdl->base_R, Rp, R);
      T r[3];
      mat3_mulv(mdl->base_R, tp, r);
      p[0] = mdl->base_t[0] + r[0];
      p[1] = mdl->base_t[1] + r[1];
      p[2] = mdl->base_t[2] + r[2];
    } else {
#pragma unroll
      for (int k = 0; k < 9; ++k) R[k] = Rp[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) p[k] = tp[k];
    }
    static_for<0, 4>([&](auto dc) {
      constexpr int D = 1 << decltype(dc)::value;
      // No selects: the first D lanes of a row have no source and receive the IDENTITY transform (composing with it
      // is exact), and what the lanes behind the chain pick up does not matter — their R, p are assigned afresh at
      // their own level from the local transform.
      if (D <= rkc && !fl) {  // wave-uniform
        T Rq[9], pq[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rq[k] = (k == 0 || k == 4 || k == 8) ? dpp_shr_one<D>(R[k]) : dpp_shr<D>(R[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) pq[k] = dpp_shr<D>(p[k]);
        T Rn[9], r[3];
        mat3_mul(Rq, R, Rn);
        mat3_mulv(Rq, p, r);
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = Rn[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = pq[k] + r[k];
      }
    });
    TDS_PROBE(11);
    // world motion axes and velocities of the chain: v_i = sum_{j <= i} s_j qd_j (prefix sum)
    if (inch) {
      mat3_mulv(R, Sl, sw);
      mat3_mulv(R, Sl + 3, sw + 3);
      T c[3];
      cross3(p, sw, c);
      sw[3] += c[0];
      sw[4] += c[1];
      sw[5] += c[2];
#pragma unroll
      for (int k = 0; k < 6; ++k) v[k] = vJ[k] = sw[k] * qd;
    }
    static_for<0, 4>([&](auto dc) {
      constexpr int D = 1 << decltype(dc)::value;
      // (no mask: lanes without a source receive 0, the lanes behind the chain assign v afresh at their level)
      if (D <= rkc) {  // wave-uniform
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] += dpp_shr<D>(v[k]);
      }
    });
    TDS_PROBE(12);
    // bias accelerations of the chain: prefix sum of cb on top of the base acceleration -gravity.
    // (floating base: a0 stays zero — the axes move with the base, v x v = 0, and the reference adds gravity
    //  to the base acceleration AFTER the joint accelerations are known, forward_dynamics.hpp:315-319)
    if (inch && !fl) {
      bias_accel();
#pragma unroll
      for (int k = 0; k < 6; ++k) a0[k] = cb[k];
    }
    static_for<0, 4>([&](auto dc) {
      constexpr int D = 1 << decltype(dc)::value;
      if (D <= rkc) {  // wave-uniform
#pragma unroll
        for (int k = 0; k < 6; ++k) a0[k] += dpp_shr<D>(a0[k]);
      }
    });
    if (inch && !fl) {
      a0[3] -= mdl->grav[0];
      a0[4] -= mdl->grav[1];
      a0[5] -= mdl->grav[2];
    }
    if (inch && lds_children) {  // children other than lane + 1 read my record
#pragma unroll
      for (int k = 0; k < 9; ++k) Xw[li * TDS_S1 + k] = R[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) Xw[li * TDS_S1 + 9 + k] = p[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        va[my_slot * TDS_S1 + k] = v[k];
        va[my_slot * TDS_S1 + 6 + k] = a0[k];
      }
    }
    TDS_WAVE_SYNC();
  }
  TDS_PROBE(13);
  for (int lev = leg_len != 0 ? nlev : mdl->kin_lev0; lev < nlev; ++lev) {  // (leg scan: every link is done)
    const bool mine = level == lev && li > rkc;  // (the chain's links are done)
    const bool by_dpp = mine && chain_child;
    const bool by_lds = mine && parent >= 0 && !chain_child;
    // the parent's record: from lane - 1 by DPP (chain children) or from LDS (the others); no zero-filling — every
    // lane of the level is served by one of the two or is a root (below), the other lanes discard what they get.
    T Rq[9], pq[3], vq[6], aq[6];
    const bool any_dpp = __any(by_dpp) != 0, any_lds = __any(by_lds) != 0;
    if (any_dpp) {  // (the moves run on EVERY lane: a DPP source lane must be active)
#pragma unroll
      for (int k = 0; k < 9; ++k) Rq[k] = dpp_neighbour<false>(R[k]);
#pragma unroll
      for (int k = 0; k < 3; ++k) pq[k] = dpp_neighbour<false>(p[k]);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        vq[k] = dpp_neighbour<false>(v[k]);
        aq[k] = dpp_neighbour<false>(a0[k]);
      }
      if (by_lds) {  // both kinds at one level (the Ant's hips: lane 6 follows the torso in lane 5, the other three do not)
#pragma unroll
        for (int k = 0; k < 9; ++k) Rq[k] = Xw[parent * TDS_S1 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) pq[k] = Xw[parent * TDS_S1 + 9 + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          vq[k] = va[par_slot * TDS_S1 + k];
          aq[k] = va[par_slot * TDS_S1 + 6 + k];
        }
      }
    } else {
      const int pl = by_lds ? parent : 0, ps = by_lds ? par_slot : 0;
#pragma unroll
      for (int k = 0; k < 9; ++k) Rq[k] = Xw[pl * TDS_S1 + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) pq[k] = Xw[pl * TDS_S1 + 9 + k];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        vq[k] = va[ps * TDS_S1 + k];
        aq[k] = va[ps * TDS_S1 + 6 + k];
      }
    }
    (void)any_lds;
    if (mine) {
      if (parent < 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rq[k] = two ? mdl->base_Rb[bod][k] : mdl->base_R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) pq[k] = two ? mdl->base_tb[bod][k] : mdl->base_t[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) vq[k] = aq[k] = T(0);
        // base acceleration = -gravity in the body's own base frame (forward_dynamics.hpp:237-243)
        aq[3] = two ? -mdl->gravb[bod][0] : -mdl->grav[0];
        aq[4] = two ? -mdl->gravb[bod][1] : -mdl->grav[1];
        aq[5] = two ? -mdl->gravb[bod][2] : -mdl->grav[2];
        if (flm && fbk == 0) {
          // floating base (kinematics.hpp:35-47): base_X_world = (quat_to_matrix(q[0..3]), q[4..6]) of the body's own
          // share of the q record; it is the frame of all six pseudo links (identity joint transforms behind this one)
          const T qx = xr[fbq], qy = xr[fbq + 1], qz = xr[fbq + 2], qw = xr[fbq + 3];
          const T s2 = T(2) / (qx * qx + qy * qy + qz * qz + qw * qw);  // tiny_matrix3x3.h:315-340
          const T xs = qx * s2, ys = qy * s2, zs = qz * s2;
          const T wx = qw * xs, wy = qw * ys, wz = qw * zs;
          const T xx = qx * xs, xy = qx * ys, xz = qx * zs;
          const T yy = qy * ys, yz = qy * zs, zz = qz * zs;
          Rq[0] = T(1) - (yy + zz); Rq[1] = xy - wz; Rq[2] = xz + wy;
          Rq[3] = xy + wz; Rq[4] = T(1) - (xx + zz); Rq[5] = yz - wx;
          Rq[6] = xz - wy; Rq[7] = yz + wx; Rq[8] = T(1) - (xx + yy);
          pq[0] = xr[fbq + 4];
          pq[1] = xr[fbq + 5];
          pq[2] = xr[fbq + 6];
          aq[3] = aq[4] = aq[5] = T(0);  // (gravity reaches a floating base after the solve, forward_dynamics.hpp:315-319)
        }
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
      bias_accel();
      if (sph && jt >= TDS_JOINT_SPH1) {
        // the three lanes of a spherical joint are ONE joint: c = v x vJ with vJ = S_3d qd (kinematics.hpp:96-97),
        // i.e. over the three lanes  v_parent x (vJ_0 + vJ_1 + vJ_2).  As a chain the lanes would add
        // vJ_0 x vJ_1 + (vJ_0 + vJ_1) x vJ_2 on top: take the earlier lanes' joint velocities (same frame:
        // axis j = column j of R about the frame origin) out of v before the cross product
        T vb[6] = {v[0], v[1], v[2], v[3], v[4], v[5]};
        const int nprev = jt - TDS_JOINT_SPH0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (j < nprev) {
            const T qj = xr[nq + qdri - nprev + j];
            const T ax[3] = {R[j] * qj, R[3 + j] * qj, R[6 + j] * qj};
            T c[3];
            cross3(p, ax, c);
            vb[0] -= ax[0];
            vb[1] -= ax[1];
            vb[2] -= ax[2];
            vb[3] -= c[0];
            vb[4] -= c[1];
            vb[5] -= c[2];
          }
        }
        cross3(vb, vJ, cb);
        T c1[3], c2[3];
        cross3(vb, vJ + 3, c1);
        cross3(vb + 3, vJ, c2);
        cb[3] = c1[0] + c2[0];
        cb[4] = c1[1] + c2[1];
        cb[5] = c1[2] + c2[2];
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) a0[k] = aq[k] + cb[k];
      if (flm && fbk >= 0) {
        // (pseudo links of a floating base: the axes move with the base — no velocity-product acceleration, see KIND 1)
#pragma unroll
        for (int k = 0; k < 6; ++k) a0[k] = T(0);
      }
      if (lds_children) {  // children other than lane + 1 read my record
#pragma unroll
        for (int k = 0; k < 9; ++k) Xw[li * TDS_S1 + k] = R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) Xw[li * TDS_S1 + 9 + k] = p[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          va[my_slot * TDS_S1 + k] = v[k];
          va[my_slot * TDS_S1 + 6 + k] = a0[k];
        }
      }
    }
    if (__any(mine && lds_children)) TDS_WAVE_SYNC();
    if (lev == mdl->kin_lev0) TDS_PROBE(14);
  }
  TDS_PROBE(15);
  // X_world of the remaining links (narrowphase, visual poses) and the world motion axes per dof
  // (leg scan: no leg link has written its record yet, whatever its children — a chain that crosses a DPP row marks the
  //  link in front of the crossing as a parent of LDS children for the later sweeps)
  if (isl && (!lds_children || (leg_len != 0 && li > 5))) {
#pragma unroll
    for (int k = 0; k < 9; ++k) Xw[li * TDS_S1 + k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) Xw[li * TDS_S1 + 9 + k] = p[k];
  }
  if (di >= 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) swd[k * NDs + di] = sw[k];
  }
  TDS_WAVE_SYNC();

  TDS_STAMP(3);
  // ---- I. narrowphase right after the kinematics sweep (it only needs X_world), so that the
  //         LDS holding X_world / v can be recycled by the dynamics sweeps
  // ---- I + M1. narrowphase and visual poses right after the kinematics sweep (they only need X_world), so that the
  //         LDS holding X_world / v can be recycled by the dynamics sweeps.  In a two-wavefront workgroup the helper
  //         wavefront runs them (and the Jacobian rows) while this one goes on with the dynamics.
  int na = 0, NA = 0;
  bool wave_contacts = false;
  unsigned long long pair_cnts = 0ull;  // multi-body worlds: penetrating contacts between the bodies, 8 bits per body pair
  bool any_pairs = false;               // ... any in this wavefront
  if constexpr (W2) {
    if (lane == 0) {
      xr[in_dim + 4] = T(0);  // "y~ is published" (phase F), polled by the helper wavefront
      if constexpr (PIPE) {
        xr[in_dim + 5] = T(0);                               // columns of L published (0, 1 = first half, 2 = all + D)
        xr[in_dim + 3] = bits_to_scalar<T>(0xFFFFFFFFu);     // the helper's contact counts: not there yet
      }
    }
    __syncthreads();  // (1) x record, X_world and the motion axes are in LDS: the helper wavefront starts
    if constexpr (LOOP) load_phase_consts(mdl);
  } else {
    // (step-loop build: the constants of the later phases are fetched only now — one L2 round trip per iteration
    //  instead of ~60 registers held through the kinematics sweep, which is what keeps this build at two
    //  wavefronts per SIMD)
    if constexpr (LOOP && NDP < 24) load_phase_consts(mdl);
    na = phase_I();
    // does any environment of this wavefront have a penetrating contact?  If not, the whole
    // constraint pipeline (CRBA, LDL^T, rows, PGS) is skipped: with keep_all_points_ the reference
    // still solves, but every row is identically zero and leaves qd untouched.
    wave_contacts = __any(na > 0) != 0;
    // wave-uniform constraint-row layout (see tds_row_solve): NA = the largest contact count among the
    // wavefront's environments (computed here, long before phase J needs it)
    NA = wave_max(na);
    if constexpr (two) {
      pair_cnts = phase_I2();
      any_pairs = __any(pair_cnts != 0ull) != 0;
    }
    phase_M1();
    flush_prev_records();
    TDS_WAVE_SYNC();  // X_world / v in LDS are dead from here on (their space is reused)
  }

  // ---- D. world-frame rigid inertias and the forces of the "all joint accelerations zero" motion
  //         (kinematics.hpp:96-132, inertia.hpp:121-130).
  // Forward dynamics is NOT done with the articulated-body recursion here: the joint-space inertia
  // M = L D L^T is needed anyway for the contact solve (phases G, H), and with it
  //        qdd = M^-1 (tau - C),   C_i = s_i . f_i,   f_i = sum over the subtree of (I a0 + v x* I v)
  // costs one substitution, while the bias forces C ride on the composite-inertia sweep as six more
  // numbers per link.  The same q̈ the reference's ABA (forward_dynamics.hpp:11-326) produces, to
  // round-off; a 6x6 articulated inertia never has to travel up the tree.
  // per-link records [link][TDS_S2]: f or F(6) | Ic I(6) h(3) m.  A record is
  // an accumulation target only for links with children that are not lane + 1.
  T *const pAs = E + L.pA;
  T *const Ics = E + L.Ic;
  T Ic[10], fc[6];  // composite inertia (I sym 6 | h | m) and composite bias force of my subtree so far
#pragma unroll
  for (int k = 0; k < 10; ++k) Ic[k] = T(0);
#pragma unroll
  for (int k = 0; k < 6; ++k) fc[k] = T(0);
  if (isl) {
    const T m = mass_l;
    T cw[3];
    mat3_mulv(R, com_l, cw);
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
    // I = Icom + m (|c|^2 1 - c c^T)
    Ic[0] = Iw[0] + m * (c2 - cw[0] * cw[0]);
    Ic[1] = T(0.5) * (Iw[1] + Iw[3]) - m * cw[0] * cw[1];
    Ic[2] = T(0.5) * (Iw[2] + Iw[6]) - m * cw[0] * cw[2];
    Ic[3] = Iw[4] + m * (c2 - cw[1] * cw[1]);
    Ic[4] = T(0.5) * (Iw[5] + Iw[7]) - m * cw[1] * cw[2];
    Ic[5] = Iw[8] + m * (c2 - cw[2] * cw[2]);
    Ic[6] = m * cw[0];
    Ic[7] = m * cw[1];
    Ic[8] = m * cw[2];
    Ic[9] = m;
    const T *const h = Ic + 6;
    // I x = (I w + h x x_lin, m x_lin - h x w)  for x = v and x = a0
    T Iv[6], Ia[6], t3[3];
    sym3_mulv(Ic, v, Iv);
    cross3(h, v + 3, t3);
    Iv[0] += t3[0];
    Iv[1] += t3[1];
    Iv[2] += t3[2];
    cross3(h, v, t3);
    Iv[3] = m * v[3] - t3[0];
    Iv[4] = m * v[4] - t3[1];
    Iv[5] = m * v[5] - t3[2];
    sym3_mulv(Ic, a0, Ia);
    cross3(h, a0 + 3, t3);
    Ia[0] += t3[0];
    Ia[1] += t3[1];
    Ia[2] += t3[2];
    cross3(h, a0, t3);
    Ia[3] = m * a0[3] - t3[0];
    Ia[4] = m * a0[4] - t3[1];
    Ia[5] = m * a0[5] - t3[2];
    // f = I a0 + v x* (I v),  v x* f = (w x n + v_lin x f_lin, w x f_lin)      (f_ext = 0 after clear_forces)
    T u3[3];
    cross3(v, Iv, fc);
    cross3(v + 3, Iv + 3, u3);
    fc[0] += u3[0];
    fc[1] += u3[1];
    fc[2] += u3[2];
    cross3(v, Iv + 3, fc + 3);
#pragma unroll
    for (int k = 0; k < 6; ++k) fc[k] += Ia[k];
    if ((fl || flm) && fbk == 5) {
      // floating base body (kinematics.hpp:52-61): the reference's bias force of the base is ONLY the
      // gyroscopic torque  w x ((R I R^T) w)  with w = qd[0..2] taken as it is, stored as the top of a
      // base-frame force vector (no v x* I v, no gravity).  In world coordinates: (R gyro, 0).
      const int wq = flm ? nq + qdri - 5 : nq;  // (the body's [omega | v] share of the qd record)
      const T w3[3] = {xr[wq], xr[wq + 1], xr[wq + 2]};
      T Iww[3], gy[3];
      mat3_mulv(Iw, w3, Iww);
      cross3(w3, Iww, gy);
      mat3_mulv(R, gy, fc);
      fc[3] = fc[4] = fc[5] = T(0);
    }
    if (lds_children) {
#pragma unroll
      for (int k = 0; k < 6; ++k) pAs[li * TDS_S2 + k] = T(0);
#pragma unroll
      for (int k = 0; k < 10; ++k) Ics[li * TDS_S2 + k] = T(0);
    }
  }
  TDS_WAVE_SYNC();

  TDS_STAMP(4);
  // ---- E. bottom-up sweep: composite inertia (CRBA, mass_matrix.hpp:39-56) and composite bias force
  //         (the backward pass of inverse dynamics) in world coordinates: plain sums up the tree
  T Fc[6], Cb = T(0);  // F_i = Ic_i s_i (mass-matrix phase), C_i = s_i . f_i
#pragma unroll
  for (int k = 0; k < 6; ++k) Fc[k] = T(0);
  auto project = [&](const T *Icx, const T *fx) {  // F = Ic s = (I w + h x v, m v - h x w);  C = s . f
    T t3[3];
    sym3_mulv(Icx, sw, Fc);
    cross3(Icx + 6, sw + 3, t3);
    Fc[0] += t3[0];
    Fc[1] += t3[1];
    Fc[2] += t3[2];
    cross3(Icx + 6, sw, t3);
    Fc[3] = Icx[9] * sw[3] - t3[0];
    Fc[4] = Icx[9] * sw[4] - t3[1];
    Fc[5] = Icx[9] * sw[5] - t3[2];
    Cb = dot3(sw, fx) + dot3(sw + 3, fx + 3);
  };
  // with a root joint (DevModel::root_last) the massless base chain 0..rk-1 is handled after the loop
  const int rk = mdl->root_last;  // == level of that link; -1: none
  // A robot that is ONE serial chain (pendulums, the cartpole): the composites are suffix sums along the lanes —
  // log2 rounds of DPP shifts and one projection for all links at once instead of one level per link
  const bool pure_chain = rk < 0 && mdl->kin_chain_last == nl - 1 && nl > 1;  // wave-uniform
  if (pure_chain) {
    static_for<0, 4>([&](auto dc) {
      constexpr int D = 1 << decltype(dc)::value;
      if (D < nl) {  // (lanes that are no links hold zeros)
#pragma unroll
        for (int k = 0; k < 6; ++k) fc[k] += dpp_shl<D>(fc[k]);
#pragma unroll
        for (int k = 0; k < 10; ++k) Ic[k] += dpp_shl<D>(Ic[k]);
      }
    });
    if (isl) project(Ic, fc);
  }
  // (chains of four: two rounds + one hand-over instead of four levels — Laikago's sweep 5.0 k -> 3.4 k cycles; chains of two,
  //  the Ant: the level loop's one DPP hand-over + one level is the shorter way, measured 3.7 k against 4.3 k)
  const bool leg_sweep = leg_len > 2;  // wave-uniform
  if (leg_sweep) {
    // the legs (DevModel::leg_len: chains of equal length in consecutive lanes, all hanging off link 5): suffix sums
    // along every chain in log2(leg_len) rounds of lane shifts, then the chain heads hand their totals to the torso
    const bool leg = isl && li > 5;
    const int jpos = leg ? ((li - 6) & (leg_len - 1)) : 0;
    static_for<0, 2>([&](auto dc) {
      constexpr int D = 1 << decltype(dc)::value;
      if (D < leg_len) {  // wave-uniform
        const T recv = (leg && jpos + D < leg_len) ? T(1) : T(0);  // (select by multiplication, as below)
#pragma unroll
        for (int k = 0; k < 6; ++k) fc[k] += recv * seg_shl<D, G>(fc[k]);
#pragma unroll
        for (int k = 0; k < 10; ++k) Ic[k] += recv * seg_shl<D, G>(Ic[k]);
      }
    });
    if (leg && jpos == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) atomicAdd(&pAs[5 * TDS_S2 + k], fc[k]);
#pragma unroll
      for (int k = 0; k < 10; ++k) atomicAdd(&Ics[5 * TDS_S2 + k], Ic[k]);
    }
    TDS_WAVE_SYNC();
  }
  for (int lev = (pure_chain || leg_sweep) ? rk : nlev - 1; lev > rk; --lev) {
    const bool mine = level == lev;
    if (mine && lds_children) {  // what the children other than lane + 1 handed over
#pragma unroll
      for (int k = 0; k < 6; ++k) fc[k] += pAs[li * TDS_S2 + k];
#pragma unroll
      for (int k = 0; k < 10; ++k) Ic[k] += Ics[li * TDS_S2 + k];
    }
    const bool to_lds = mine && parent >= 0 && !chain_child;
    // (the projection F = Ic s, C = s . f is NOT done level by level: nothing in the sweep needs it, so every lane
    //  projects its finished composite once, behind the sweep — two projections fewer on the dependent chain)
    if (to_lds) {
#pragma unroll
      for (int k = 0; k < 6; ++k) atomicAdd(&pAs[parent * TDS_S2 + k], fc[k]);
#pragma unroll
      for (int k = 0; k < 10; ++k) atomicAdd(&Ics[parent * TDS_S2 + k], Ic[k]);
    }
    if (__any(mine && chain_child)) {  // wave-uniform: lane i takes over from its child in lane i + 1
      // (select by multiplication: one FMA instead of two 32-bit selects and an add per value)
      const T recv = (has_chain_child && level + 1 == lev) ? T(1) : T(0);
#pragma unroll
      for (int k = 0; k < 6; ++k) fc[k] += recv * dpp_neighbour<true>(fc[k]);
#pragma unroll
      for (int k = 0; k < 10; ++k) Ic[k] += recv * dpp_neighbour<true>(Ic[k]);
    }
    if (__any(to_lds)) TDS_WAVE_SYNC();
  }
  // ---- E'. root joint: the links 0..rk-1 of the base chain are massless, so the composite inertia and
  //      force of link rk pass through them unchanged: F_i = Ic_rk s_i, C_i = s_i . f_rk for i <= rk,
  //      all at once instead of rk+1 more levels
  if (rk >= 0) {  // wave-uniform
    if (li == rk && lds_children) {
#pragma unroll
      for (int k = 0; k < 6; ++k) fc[k] += pAs[li * TDS_S2 + k];
#pragma unroll
      for (int k = 0; k < 10; ++k) Ic[k] += Ics[li * TDS_S2 + k];
    }
    if (rk == 5) {
      // the usual six-link base chain: link 5's totals reach lanes 0..4 by a DPP broadcast (their own composites are
      // zero: massless links) — no LDS round trip
      const bool take = isl && li < 5;
      static_for<0, 6>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const T b = lane_bcast<T, G, NDP, 5>(fc[k]);
        fc[k] = take ? b : fc[k];
      });
      static_for<0, 10>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const T b = lane_bcast<T, G, NDP, 5>(Ic[k]);
        Ic[k] = take ? b : Ic[k];
      });
    } else {
      if (li == rk) {
#pragma unroll
        for (int k = 0; k < 6; ++k) pAs[li * TDS_S2 + k] = fc[k];
#pragma unroll
        for (int k = 0; k < 10; ++k) Ics[li * TDS_S2 + k] = Ic[k];
      }
      TDS_WAVE_SYNC();
      if (isl && li < rk) {
#pragma unroll
        for (int k = 0; k < 10; ++k) Ic[k] = Ics[rk * TDS_S2 + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) fc[k] = pAs[rk * TDS_S2 + k];
      }
      TDS_WAVE_SYNC();
    }
  }
  if (!pure_chain && isl) project(Ic, fc);

  TDS_STAMP(5);
  T qd_new = qd;
  T q_new = q;

  {
    // ---- G. mass-matrix row of dof d, straight into registers (lane == dof == row):
    //         M[d][j] = F_d . s_j for j on the path base -> d   (mass_matrix.hpp:87-109)
    T *const Fs = E + L.F;      // (= pA slot of the link records)
    T *const Lp = E + L.Lp;     // strictly-lower L packed row-major: L[r][j] at r(r-1)/2 + j
    T *const dvec = E + L.dinv; // [3][NDP]: 1/D_k | sqrt(1/D_k) | column scratch (NDP > 16)
    // lane == link == dof for every link (DevModel::dof_identity: the Ant): F, tau - C stay in the lane's registers
    const bool didn = !gen && !two && mdl->dof_identity != 0;  // wave-uniform
    if (!didn) {
      if (isl) {
#pragma unroll
        for (int k = 0; k < 6; ++k) Fs[li * TDS_S2 + k] = Fc[k];
      }
      TDS_WAVE_SYNC();
    }
    T Mr[NDP];
    {
      const int d = lane;
      const bool isd = d < nd;
      const int lk = isd ? pf_dof_link : 0;
      const unsigned anc = isd ? pf_anc : 0u;
      T Fd[6];
      if (didn) {
#pragma unroll
        for (int k = 0; k < 6; ++k) Fd[k] = isd ? Fc[k] : T(0);
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) Fd[k] = isd ? Fs[lk * TDS_S2 + k] : T(0);
      }
      // Columns j >= nd hold whatever the row store left there; they are deselected by the ancestor mask, so the reads
      // need no `j < nd` guard.  That matters: a read under a uniform branch is its own LDS round trip, NDP of them in a
      // row.  Narrow kernels read all NDP x 6 values at once; the wider ones (no registers for that) go three columns per
      // guarded group.
      constexpr int GCH = NDP <= 16 ? NDP : 3;
#pragma unroll
      for (int j0 = 0; j0 < NDP; j0 += GCH) {
        T s[GCH];
#pragma unroll
        for (int jj = 0; jj < GCH; ++jj) s[jj] = T(0);
        if (NDP <= 16 || j0 < nd) {
#pragma unroll
          for (int jj = 0; jj < GCH; ++jj) {
            if (j0 + jj < NDP) {
#pragma unroll
              for (int k = 0; k < 6; ++k) s[jj] += Fd[k] * swd[k * NDs + j0 + jj];
            }
          }
        }
#pragma unroll
        for (int jj = 0; jj < GCH; ++jj) {
          const int j = j0 + jj;
          if (j < NDP) {
            const T sj = ((anc >> j) & 1u) ? s[jj] : T(0);
            Mr[j] = (!isd && j == d) ? T(1) : sj;  // padding rows: identity
          }
        }
      }
      if constexpr (flm) {
        // rows of the base dofs of a floating body (numbered behind the body's joints): against a joint dof j of the
        // same body the composite force is that of the JOINT's link (see KIND 1 below)
        const bool brow = isd && bdof;
        const int jlo = isd ? mdl->dof_joint0[d] : 0;
        T sb[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) sb[k] = brow ? swd[k * NDs + d] : T(0);
#pragma unroll
        for (int j = 0; j < NDP; ++j) {
          if (j < nd) {
            const int lj = mdl->dof_link[j];
            T s = T(0);
#pragma unroll
            for (int k = 0; k < 6; ++k) s += Fs[lj * TDS_S2 + k] * sb[k];
            Mr[j] = (brow && j >= jlo && j < dbase0) ? s : Mr[j];
          }
        }
      }
      if (fl) {  // wave-uniform
        // rows of the base dofs (numbered last): against a joint dof j the composite force is that of the
        // JOINT's link, M[b][j] = s_b . (Ic_j s_j)   (mass_matrix.hpp:111-115, symmetric counterpart)
        const bool brow = isd && d >= njd;
        T sb[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) sb[k] = brow ? swd[k * NDs + d] : T(0);
#pragma unroll
        for (int j = 0; j < NDP; ++j) {
          if (j < njd) {
            const int lj = mdl->dof_link[j];
            T s = T(0);
#pragma unroll
            for (int k = 0; k < 6; ++k) s += Fs[lj * TDS_S2 + k] * sb[k];
            Mr[j] = brow ? s : Mr[j];
          }
        }
      }
    }
    TDS_STAMP(6);
    // ---- H. M = L D L^T entirely in registers (replaces the Cholesky inverse,
    //         tiny_matrix_x.h:240-345).  Right-looking; column k is gathered from the lanes that
    //         own rows k..NDP-1 with cross-lane shuffles, no LDS traffic, no barriers.
    //         Entries above the diagonal of a lane's row are never read by anyone.
    T my_inv = T(1);  // 1 / D_lane
    static_for<0, NDP>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (NDP <= 16) {
        // column k lives in one 16-lane DPP row: gather it with row_newbcast moves (VALU latency)
        const T dk = lane_bcast<T, G, NDP, k>(Mr[k]);
        const T inv = rcp_full<T>(dk);
        const T lr = Mr[k] * inv;
        static_for<k + 1, NDP>([&](auto cc) {
          constexpr int c = decltype(cc)::value;
          const T mck = lane_bcast<T, G, NDP, c>(Mr[k]);
          Mr[c] -= lr * mck;
        });
        my_inv = lane == k ? inv : my_inv;
        Mr[k] = lane > k ? lr : Mr[k];
        if constexpr (PIPE && k == NDP / 2 - 1) {
          // columns 0 .. k of L are final: off they go, every lane its row's first entries at a fixed stride (no
          // address selects in the middle of the factorisation: it has no registers to spare; entries above the
          // diagonal are written too and never read)
          // (volatile stores: kept in program order — data, then flag — without a scheduling barrier in the middle of
          //  the factorisation)
          if constexpr (TDS_LDS_PUBLISH != 0) {
            // (volatile stores into the LDS address space: kept in program order — data, then flag — without a scheduling
            //  barrier in the middle of the factorisation, as ds_write instructions)
            volatile TDS_AS3 T *const Lh = (volatile TDS_AS3 T *)(E + L.Lh + (lane < 16 ? lane : 16) * (NDP / 2));  // (row 16: lanes beyond the rows)
#pragma unroll
            for (int j = 0; j <= k; ++j) Lh[j] = Mr[j];
            // (flag by lane 0, the other lanes hit a scratch slot of their own: an address select, no branch)
            *(volatile TDS_AS3 T *)(lane == 0 ? xr + in_dim + 5 : dvec + 2 * NDP + (lane < NDP ? lane : NDP - 1)) = T(1);
          } else {
            volatile T *const Lh = E + L.Lh + (lane < 16 ? lane : 16) * (NDP / 2);  // (row 16: lanes beyond the rows)
#pragma unroll
            for (int j = 0; j <= k; ++j) Lh[j] = Mr[j];
            // (flag by lane 0, the other lanes hit a scratch slot of their own: an address select, no branch)
            *(volatile T *)(lane == 0 ? xr + in_dim + 5 : dvec + 2 * NDP + (lane < NDP ? lane : NDP - 1)) = T(1);
          }
        }
      } else {
        // wider systems: every lane publishes its column-k entry once, all lanes read the column
        // back as LDS broadcasts (immediate offsets; ds_bpermute needed one address VGPR per source
        // lane and drove the NDP=24 kernels into 140 AGPR + SGPR spills)
        T *const colb = dvec + 2 * NDP;
        if (lane < NDP) colb[lane] = Mr[k];
        TDS_WAVE_SYNC();
        const T dk = colb[k];
        const T inv = rcp_full<T>(dk);
        const T lr = Mr[k] * inv;
        static_for<k + 1, NDP>([&](auto cc) {
          constexpr int c = decltype(cc)::value;
          Mr[c] -= lr * colb[c];
        });
        my_inv = lane == k ? inv : my_inv;
        Mr[k] = lane > k ? lr : Mr[k];
        TDS_WAVE_SYNC();
      }
    });
    if (lane < NDP) {
      dvec[lane] = my_inv;
      dvec[NDP + lane] = sqrt_t<T>(my_inv);
      const int off = (lane * (lane - 1)) / 2;
      // (row `lane` has `lane` entries: the surplus writes go to this lane's slot of the column scratch, which phase F
      //  clears next — an address select per entry instead of a branch around each write)
      T *const dump = dvec + 2 * NDP + lane;
#pragma unroll
      for (int j = 0; j < NDP - 1; ++j) {  // (all of it: this wavefront's own back substitutions read the packed copy)
        T *const dst = j < lane ? &Lp[off + j] : dump;
        *dst = Mr[j];
      }
    }
    if constexpr (PIPE) {
      TDS_WAVE_SYNC();
      if (lane == 0) xr[in_dim + 5] = T(2);
    }
    TDS_STAMP(7);
    bool split_ok = false;  // two-wavefront workgroup: the helper wavefront does the row solves
    if constexpr (W2) {
      if constexpr (PIPE) {
        // (2') no barrier: the helper has long written its contact counts (after its Jacobian rows) — poll them
        const T *const cnt = xr + in_dim + 3;
        while (__any(scalar_to_bits<T>(tds_lds_poll(cnt)) == 0xFFFFFFFFu)) __builtin_amdgcn_s_sleep(1);
        TDS_WAVE_SYNC();
      } else {
        __syncthreads();  // (2) L, 1/D are in LDS for the helper wavefront's row solves; its contact list and rows are visible here
      }
      if constexpr (GRAM) {  // (the sweep groups' storage is free from here on: zeros for tds_gram_solve's masked lanes)
        if (L.gram_ok && lane < 16) E[L.Xw + TDS_GRAM_ZEROS + lane] = T(0);
      }
      na = (int)scalar_to_bits<T>(xr[in_dim + 2]);
      NA = __builtin_amdgcn_readfirstlane((int)scalar_to_bits<T>(xr[in_dim + 3]));
      wave_contacts = NA > 0;
      split_ok = 3 * NA <= ZR;
    }
    // ---- F. forward dynamics: qdd = M^-1 (tau - C) with the factorisation just computed, then
    //         integrate_euler_qdd: qd += qdd dt (integrator.hpp:169-181).  tau - C travels from the link
    //         lanes to the dof lanes through the (now free) column scratch of dvec.
    T *const rhsx = dvec + 2 * NDP;
    if (lane < NDP) rhsx[lane] = T(0);
    if constexpr (LATE_PD) compute_tau();
    if constexpr (LOOP) {
      // (action replay: the NEXT step's action block, requested at the top of this step, goes into the action slots of
      //  the record — nobody reads them any more in this step: the PD block has long turned them into tau)
      if (replay && mode == TDS_MODE_RUN && lane < adim) xr[nq + nd + lane] = next_act;
      // the ring records of the PREVIOUS step left this wavefront half a step ago (behind this step's visual poses)
      if constexpr (!W2) signal_progress();
    }
    TDS_WAVE_SYNC();
    if (!didn) {
      if constexpr (PARK_TAU) {
        if (di >= 0) rhsx[di] = E[L.tau + li] - Cb;
      } else {
        if (di >= 0) rhsx[di] = tau - Cb;
      }
      TDS_WAVE_SYNC();
    }
    {
      const int d = lane;
      T tau_f = tau;
      if constexpr (PARK_TAU) {
        if (didn) tau_f = E[L.tau + (lane < nl ? lane : 0)];
      }
      T yv = didn ? (d < nd ? tau_f - Cb : T(0)) : (d < NDP ? rhsx[d] : T(0));
      // L y = rhs (L unit lower, row d of it in Mr[0..d-1]), column by column
      // (the row mask goes into the multiplier ahead of time: the dependent chain per step is broadcast + FMA, no select)
      //  (narrow kernels only: the wide ones have no registers for a masked copy of the row)
      static_for<0, NDP - 1>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const T yk = lane_bcast<T, G, NDP, k>(yv);
        if constexpr (NDP <= 16) {
          const T lm = d > k ? Mr[k] : T(0);
          yv -= lm * yk;
        } else {
          yv = d > k ? yv - Mr[k] * yk : yv;
        }
      });
      T xv = yv * my_inv;
      if constexpr (W2) {  // y~ = D^-1/2 y for the rows' right-hand sides (completed by the helper wavefront at the
                           // end of its row solves — it polls the flag; LDS executes a wavefront's writes in order)
        if (d < NDP) dvec[3 * NDP + d] = yv * sqrt_t<T>(my_inv);
        TDS_WAVE_SYNC();
        if (lane == 0) xr[in_dim + 4] = T(1);
      }
      const bool jrow = fl ? d < njd : !bdof;  // (the base rows of a floating base keep a_base in the back substitution)
      if (fl || flm) {  // wave-uniform
        // (KIND 4: every lane works on the base block of ITS dof's body; lanes of fixed-base bodies compute on block 0
        //  and discard)
        const int njd = dbase0 >= 0 ? dbase0 : 0;
        const bool has_base = dbase0 >= 0;
        // With the joint dofs eliminated, the base rows read  Sigma a_base = rho:  Sigma = L_b D_b L_b^T (the
        // trailing 6x6 block of the factors) is the articulated inertia of the base and rho = L_b y_b is minus
        // its bias force.  The reference takes  a_base = -base_abi.inv_mul(base_bias_force)  with ITS block
        // inverse (inertia.hpp:302-329: lower-left block taken as -H, exact only for a skew H) — restated as is.
        TDS_WAVE_SYNC();
        if (d < NDP) rhsx[d] = yv;
        TDS_WAVE_SYNC();
        T Lb[6][6], Db[6], yb[6], rho[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          Db[i] = T(1) / dvec[njd + i];
          yb[i] = rhsx[njd + i];
#pragma unroll
          for (int j = 0; j < 6; ++j) Lb[i][j] = j < i ? Lp[((njd + i) * (njd + i - 1)) / 2 + njd + j] : (j == i ? T(1) : T(0));
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          rho[i] = yb[i];
#pragma unroll
          for (int j = 0; j < i; ++j) rho[i] += Lb[i][j] * yb[j];
        }
        T Sg[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) {
            T a = T(0);
#pragma unroll
            for (int m = 0; m <= j; ++m) a += Lb[i][m] * Db[m] * Lb[j][m];
            Sg[i][j] = Sg[j][i] = a;
          }
        T Ai[9], Hb[9], Mb[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            Ai[3 * r + c] = Sg[r][c];
            Hb[3 * r + c] = Sg[r][3 + c];
            Mb[3 * r + c] = Sg[3 + r][3 + c];
          }
        T Ainv[9], t1[9], t2[9], Sc[9], Dc[9], ABD[9];
        mat3_inverse(Ai, Ainv);
        mat3_mul(Hb, Ainv, t1);   // -C Ainv  with C = -H
        mat3_mul(t1, Hb, t2);     // -C Ainv B
#pragma unroll
        for (int k = 0; k < 9; ++k) Sc[k] = Mb[k] + t2[k];  // M - C Ainv B
        mat3_inverse(Sc, Dc);
        mat3_mul(Ainv, Hb, t1);
        mat3_mul(t1, Dc, ABD);    // Ainv B DCAB
        mat3_mul(ABD, Hb, t1);
        mat3_mul(t1, Ainv, t2);   // -(AinvBDCAB C Ainv)
        T I2[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) I2[k] = Ainv[k] - t2[k];
        // a_base = Q rho:  top = I2 rho_top - ABD rho_bot,  bottom = DCAB rho_bot - ABD^T rho_top
        T ab[6], u3[3], w3[3];
        mat3_mulv(I2, rho, u3);
        mat3_mulv(ABD, rho + 3, w3);
        ab[0] = u3[0] - w3[0];
        ab[1] = u3[1] - w3[1];
        ab[2] = u3[2] - w3[2];
        mat3_mulv(Dc, rho + 3, u3);
#pragma unroll
        for (int c = 0; c < 3; ++c) ab[3 + c] = u3[c] - (ABD[c] * rho[0] + ABD[3 + c] * rho[1] + ABD[6 + c] * rho[2]);
#pragma unroll
        for (int i = 0; i < 6; ++i) xv = (has_base && d == njd + i) ? ab[i] : xv;
      }
      // L^T x = D^-1 y with the packed copy of L in LDS (the base rows of a floating base keep a_base)
      // (column d of L^T requested up front, index clamped: a read under `if (d < k)` is a read inside a branch, and
      //  every one of the NDP - 1 dependent steps then waits for its own LDS round trip)
      //  (wide kernels: not the registers for it — they read step by step)
      constexpr bool PRE = NDP <= 16;
      T lt[PRE ? NDP - 1 : 1];
      if constexpr (PRE) {
#pragma unroll
        for (int k = 1; k < NDP; ++k) {
          const T lv = Lp[(k * (k - 1)) / 2 + (d < k ? d : 0)];
          lt[k - 1] = (d < k && jrow) ? lv : T(0);
        }
      }
      // (no registers to hold the whole column: a ring of four reads runs ahead of the dependent chain instead)
      T ring[4];
      if constexpr (!PRE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = NDP - 1 - i;
          ring[i] = k >= 1 ? Lp[(k * (k - 1)) / 2 + (d < k ? d : 0)] : T(0);
        }
      }
      static_for<0, NDP - 1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int k = NDP - 1 - i;
        const T xk = lane_bcast<T, G, NDP, k>(xv);
        if constexpr (PRE) xv -= lt[k - 1] * xk;
        else {
          const T lv = ring[i & 3];
          if constexpr (k - 4 >= 1) ring[i & 3] = Lp[((k - 4) * (k - 5)) / 2 + (d < k - 4 ? d : 0)];
          xv -= ((d < k && jrow) ? lv : T(0)) * xk;
        }
      });
      if (fl && d >= njd + 3 && d < nd) xv += mdl->grav[d - njd - 3];  // forward_dynamics.hpp:315-319
      if (flm && bdof && d >= dbase0 + 3) xv += mdl->gravb[mdl->dof_body[d]][d - dbase0 - 3];
      // integrate_euler_qdd; from here on the velocities live in dof order in the column scratch
      TDS_WAVE_SYNC();
      if (d < NDP) rhsx[d] = d < nd ? xr[nq + rec_d] + xv * dt : T(0);
    }
    TDS_STAMP(8);

    TDS_WAVE_SYNC();  // Z aliases the sweep arrays (f, Ic, F): all of those are dead now
    if constexpr (W2) __syncthreads();  // (3) the helper wavefront's z~ rows, G_rr and 1 / (G_rr + cfm) are final
    TDS_STAMP(9);
    if (wave_contacts) {

    // ---- J. constraint Jacobian rows (jacobian.hpp:13-83, mb_constraint_solver.hpp:278-388)
    //         row a: normal, na+a: tangent 1, 2na+a: tangent 2;  lane == dof
    // Rows 0..ZR-1 of an environment live in LDS; an environment with more than ZR/3 penetrating
    // contacts keeps the surplus rows (and their scalars) in a global scratch slab — rare, slow,
    // exact.  Sizing LDS for the typical contact count instead of the worst case is what lets four
    // workgroups share a CU.
    T *const cpx = E + L.cp;
    T *const Zs = E + L.Z;  // [ZR][NDs]: J rows, overwritten in place by z~ = D^-1/2 L^-1 J^T
    T *const rws = E + L.rows;  // [3][ZR]: b | 1/(G+cfm) | G
    T *const xs = E + L.xrow;   // [3 ncp]: the impulses x of ALL rows stay in LDS (read back within the wave)
    volatile T *const zov = (ovf != nullptr && live) ? ovf + (size_t)env * OVR * (NDs + 3) : nullptr;
    volatile T *const rov = zov != nullptr ? zov + (size_t)OVR * NDs : nullptr;  // [3][OVR]
    const int nr = 3 * NA;
    const T cfm = pf_cfm, erp_dt = pf_erp_dt, rest = pf_rest;
    const bool any_slab = nr > ZR;  // wave-uniform
    bool gram = false;
    if constexpr (W2) {
      // Rows and row solves normally came from the helper wavefront.  More rows than the LDS store holds (rare): this
      // wavefront builds and solves them itself, through the scratch slab — with the SAME split arithmetic, so that
      // an environment's result does not depend on which path its wavefront-mates forced.
      if (!split_ok) {
        phase_J(na, NA);
        TDS_WAVE_SYNC();
        if (OVR > 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        if (any_slab)
          tds_row_solve<true, T, G, NDP, true>(lane, NA, na, nd, ZR, OVR, NCPp, Zs, rws, xs, xr + nq, cpx, Lp, dvec, zov,
                                               rov, cfm, erp_dt, rest);
        else
          tds_row_solve<false, T, G, NDP, true>(lane, NA, na, nd, ZR, OVR, NCPp, Zs, rws, xs, xr + nq, cpx, Lp, dvec, zov,
                                                rov, cfm, erp_dt, rest);
        TDS_WAVE_SYNC();
        if (OVR > 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      }
      if constexpr (GRAM) gram = L.gram_ok && split_ok && NA >= 2 && NA <= 5;  // wave-uniform
      // complete the right-hand sides with the acceleration part (Gram form: a column of the matrix product)
      if (gram || (split_ok && !L.gram_ok)) {  // (split_ok: the helper wavefront has completed them)
      } else if (any_slab)
        tds_row_rhs_finish<true, T, G, NDP>(lane, NA, na, ZR, OVR, NCPp, Zs, rws, cpx, dvec + 3 * NDP, zov, rov, dt, erp_dt, rest);
      else
        tds_row_rhs_finish<false, T, G, NDP>(lane, NA, na, ZR, OVR, NCPp, Zs, rws, cpx, dvec + 3 * NDP, zov, rov, dt, erp_dt, rest);
      TDS_WAVE_SYNC();
      if (OVR > 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    } else {
    phase_J(na, NA);
    TDS_WAVE_SYNC();
    // surplus rows went through global memory: drain the stores before other lanes load them.
    // (a fence, NOT __syncthreads(): hipcc 7.2 miscompiled the <f64,G=64,NDP=24> kernel when an
    //  s_barrier sat in this conditional block — caught by tests/test_hip_parity.py)
    if (OVR > 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    TDS_STAMP(10);

    // ---- K. per row (lane == row): b_r, forward substitution  L z = J_r^T  in registers,
    //         G_rr = z.D^-1.z,  1/(G_rr + cfm);  the row is stored back as z~ = D^-1/2 z so that
    //         A_rs = J_r M^-1 J_s^T = z~_r . z~_s  — one matrix instead of J and M^-1 J^T.
    if (any_slab)
      tds_row_solve<true, T, G, NDP>(lane, NA, na, nd, ZR, OVR, NCPp, Zs, rws, xs, rhsx, cpx, Lp, dvec, zov, rov,
                                     cfm, erp_dt, rest);
    // (18-dof kernels, rows of one pass: two lanes per row — measured, profiles/r04_ab_slots14_laikago_row_half.txt:
    //  laikago_soft x 8192 48.2 -> 47.1 us per step)
    else if (G == 32 && NDP > 16 && NDP < 24 && 3 * NA <= 16 && 3 * NA <= ZR)  // wave-uniform
      tds_row_solve_half<T, NDP>(lane, NA, na, nd, ZR, NCPp, Zs, rws, rhsx, cpx, Lp, dvec, cfm, erp_dt, rest);
    else
      tds_row_solve<false, T, G, NDP>(lane, NA, na, nd, ZR, OVR, NCPp, Zs, rws, xs, rhsx, cpx, Lp, dvec, zov, rov,
                                      cfm, erp_dt, rest);
    TDS_WAVE_SYNC();
    if (OVR > 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    TDS_STAMP(11);

    // ---- L. projected Gauss-Seidel (mb_constraint_solver.hpp:101-142) on u~ = sum_r z~_r x_r:
    //         delta_i = sum_{j != i} A_ij x_j = z~_i . u~ - G_ii x_i;   lane == dof holds u~_k
    {
      const int d = lane;
      const bool dz = d < NDP;
      const T mu = pf_mu;
      const int iters = pf_iters;
      T u;
      if constexpr (GRAM) {
        long long *const gst = PROF && tds_iter == TDS_PROF_ITER ? prof + 10 : nullptr;
        const int gat = ctl.flags >> 8;
        if (!gram)
          u = any_slab ? tds_pgs<true, T, G, NDP>(lane, NA, ZR, OVR, iters, mu, Zs, rws, xs, zov, rov)
                       : tds_pgs_lds<T, G, NDP>(lane, NA, ZR, iters, mu, Zs, rws, xs);
        else if (NA == 2)
          u = tds_gram_solve<NDP, 2>(sm, L, lane, na, ZR, NCPp, iters, mu, dt, erp_dt, rest, gst, gat);
        else if (NA == 3)
          u = tds_gram_solve<NDP, 3>(sm, L, lane, na, ZR, NCPp, iters, mu, dt, erp_dt, rest, gst, gat);
        else if (NA == 4)
          u = tds_gram_solve<NDP, 4>(sm, L, lane, na, ZR, NCPp, iters, mu, dt, erp_dt, rest, gst, gat);
        else
          u = tds_gram_solve<NDP, 5>(sm, L, lane, na, ZR, NCPp, iters, mu, dt, erp_dt, rest, gst, gat);
      } else {
        u = any_slab ? tds_pgs<true, T, G, NDP>(lane, NA, ZR, OVR, iters, mu, Zs, rws, xs, zov, rov)
                     : tds_pgs_lds<T, G, NDP>(lane, NA, ZR, iters, mu, Zs, rws, xs);
      }
      // delta_qd = M^-1 J^T p = L^-T D^-1/2 u~   (mb_constraint_solver.hpp:476-496: qd_b -= delta_qd)
      T w = dz ? u * dvec[NDP + d] : T(0);
      constexpr bool PRE = NDP <= 16;  // (see phase F)
      T lt[PRE ? NDP - 1 : 1];
      if constexpr (PRE) {
#pragma unroll
        for (int k = 1; k < NDP; ++k) {
          const T lv = Lp[(k * (k - 1)) / 2 + (d < k ? d : 0)];
          lt[k - 1] = d < k ? lv : T(0);
        }
      }
      T ring[4];  // (see phase F)
      if constexpr (!PRE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = NDP - 1 - i;
          ring[i] = k >= 1 ? Lp[(k * (k - 1)) / 2 + (d < k ? d : 0)] : T(0);
        }
      }
      static_for<0, NDP - 1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int k = NDP - 1 - i;
        const T wk = lane_bcast<T, G, NDP, k>(w);
        if constexpr (PRE) w -= lt[k - 1] * wk;
        else {
          const T lv = ring[i & 3];
          if constexpr (k - 4 >= 1) ring[i & 3] = Lp[((k - 4) * (k - 5)) / 2 + (d < k - 4 ? d : 0)];
          w -= (d < k ? lv : T(0)) * wk;
        }
      });
      if (d < nd) rhsx[d] -= w;
    }
    }  // wave_contacts
    if constexpr (two) {
      // ---- further contact passes: body a against body b for every pair a < b, each with the velocities the earlier
      //      passes left (World::step resolves the body pairs one after the other: plane-0, plane-1, .., 0-1, 0-2, ..,
      //      1-2, ..; world.hpp:340-352)
      int pair_off = 0;
      for (int pr = 0; any_pairs && pr < mdl->num_bpairs; ++pr) {
      const int nb_pairs = (int)((pair_cnts >> (8 * pr)) & 255ull);
      const int NB_pairs = wave_max(nb_pairs);
      const int my_off = pair_off;
      pair_off += nb_pairs;
      if (NB_pairs > 0) {
        TDS_WAVE_SYNC();
        T *const pcx = E + L.pc + my_off;
        T *const Zs = E + L.Z;
        T *const rws = E + L.rows;
        T *const xs = E + L.xrow;
        volatile T *const zov = (ovf != nullptr && live) ? ovf + (size_t)env * OVR * (NDs + 3) : nullptr;
        volatile T *const rov = zov != nullptr ? zov + (size_t)OVR * NDs : nullptr;
        const T cfm = pf_cfm, erp_dt = pf_erp_dt, rest = pf_rest;
        const bool slab2 = 3 * NB_pairs > ZR;  // wave-uniform
        phase_J2(pr, my_off, nb_pairs, NB_pairs);
        TDS_WAVE_SYNC();
        if (OVR > 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        // (the distance of pair contact a sits at pcx[15 NPCp + a]: the row solve reads cpx[3 NCPp + a])
        if (slab2)
          tds_row_solve<true, T, G, NDP>(lane, NB_pairs, nb_pairs, nd, ZR, OVR, NPCp, Zs, rws, xs, rhsx, pcx + 12 * NPCp,
                                         Lp, dvec, zov, rov, cfm, erp_dt, rest);
        else
          tds_row_solve<false, T, G, NDP>(lane, NB_pairs, nb_pairs, nd, ZR, OVR, NPCp, Zs, rws, xs, rhsx, pcx + 12 * NPCp,
                                          Lp, dvec, zov, rov, cfm, erp_dt, rest);
        TDS_WAVE_SYNC();
        if (OVR > 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        const int d = lane;
        const T u = slab2 ? tds_pgs<true, T, G, NDP>(lane, NB_pairs, ZR, OVR, pf_iters, pf_mu, Zs, rws, xs, zov, rov)
                          : tds_pgs_lds<T, G, NDP>(lane, NB_pairs, ZR, pf_iters, pf_mu, Zs, rws, xs);
        T w = d < NDP ? u * dvec[NDP + d] : T(0);
        constexpr bool PRE = NDP <= 16;  // (see phase F)
        T lt[PRE ? NDP - 1 : 1];
        if constexpr (PRE) {
#pragma unroll
          for (int k = 1; k < NDP; ++k) {
            const T lv = Lp[(k * (k - 1)) / 2 + (d < k ? d : 0)];
            lt[k - 1] = d < k ? lv : T(0);
          }
        }
        T ring[4];  // (see phase F)
        if constexpr (!PRE) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int k = NDP - 1 - i;
            ring[i] = k >= 1 ? Lp[(k * (k - 1)) / 2 + (d < k ? d : 0)] : T(0);
          }
        }
        static_for<0, NDP - 1>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          constexpr int k = NDP - 1 - i;
          const T wk = lane_bcast<T, G, NDP, k>(w);
          if constexpr (PRE) w -= lt[k - 1] * wk;
          else {
            const T lv = ring[i & 3];
            if constexpr (k - 4 >= 1) ring[i & 3] = Lp[((k - 4) * (k - 5)) / 2 + (d < k - 4 ? d : 0)];
            w -= (d < k ? lv : T(0)) * wk;
          }
        });
        if (d < nd) rhsx[d] -= w;
      }
      }  // body pairs
    }
    TDS_WAVE_SYNC();
    if (di >= 0) qd_new = rhsx[di];
  }

  TDS_STAMP(12);
  // ---- M. integrate_euler: q += qd dt (integrator.hpp:126-131) and pack y -------------------
  q_new = q + qd_new * dt;

  // carry the state in the LDS record (slot in_dim keeps x_{t-1} for the Ant reward)
  TDS_WAVE_SYNC();
  T up_z = T(0);  // floating base: base_X_world.rotation(2,2) AFTER the step (integrator.hpp:83)
  if (froot) {
    // base coordinates (integrator.hpp:23-89): quaternion += quat_velocity(quat, omega, dt)
    // (tiny_algebra.hpp:604-614), normalised; position += v dt.  Every pseudo-link lane evaluates the whole
    // quaternion and stores its own component(s): lanes 0..3 the quaternion, lanes 3..5 also the position.
    const T h = T(0.5) * dt;
    const T *const qdv = E + L.dinv + 2 * NDP;  // velocities in dof order (phase F)
    const int wb = flm ? di - fbk : njd;  // first base dof of my body
    const int qb = flm ? fbq : 0;         // its quaternion in the q record
    const T w0 = qdv[wb], w1 = qdv[wb + 1], w2 = qdv[wb + 2];
    const T b0 = xr[qb], b1 = xr[qb + 1], b2 = xr[qb + 2], b3 = xr[qb + 3];
    T n0 = b0 + (b3 * w0 + b2 * w1 - b1 * w2) * h;
    T n1 = b1 + (b3 * w1 + b0 * w2 - b2 * w0) * h;
    T n2 = b2 + (b3 * w2 + b1 * w0 - b0 * w1) * h;
    T n3 = b3 + (-b0 * w0 - b1 * w1 - b2 * w2) * h;
    const T ql = sqrt_t<T>(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3);
    n0 /= ql;
    n1 /= ql;
    n2 /= ql;
    n3 /= ql;
    up_z = T(1) - (n0 * n0 + n1 * n1) * (T(2) / (n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3));
    const T pos_new = fbk >= 3 ? xr[qb + 4 + fbk - 3] + qd_new * dt : T(0);
    __builtin_amdgcn_wave_barrier();
    if (fbk < 4) xr[qb + fbk] = fbk == 0 ? n0 : fbk == 1 ? n1 : fbk == 2 ? n2 : n3;
    if (fbk >= 3) xr[qb + 4 + fbk - 3] = pos_new;
  }
  if (sph_lane) {
    // spherical joint (integrator.hpp:94-123): qd *= pow(joint_damping, 1000 dt), then
    // quat += quat_velocity_spherical(quat, qd, dt) (tiny_algebra.hpp:618-629), normalised — by the first lane
    const T damp = mdl->sph_damping;
    qd_new *= damp;
    if (jt == TDS_JOINT_SPH0) {
      const T *const qdv = E + L.dinv + 2 * NDP;  // velocities in dof order (phase F)
      const T h = T(0.5) * dt;
      const T w0 = qd_new, w1 = qdv[di + 1] * damp, w2 = qdv[di + 2] * damp;
      const T b0 = xr[qri], b1 = xr[qri + 1], b2 = xr[qri + 2], b3 = xr[qri + 3];
      T n0 = b0 + (b3 * w0 + b1 * w2 - b2 * w1) * h;
      T n1 = b1 + (b3 * w1 + b2 * w0 - b0 * w2) * h;
      T n2 = b2 + (b3 * w2 + b0 * w1 - b1 * w0) * h;
      T n3 = b3 + (-b0 * w0 - b1 * w1 - b2 * w2) * h;
      const T ql = sqrt_t<T>(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3);
      xr[qri] = n0 / ql;
      xr[qri + 1] = n1 / ql;
      xr[qri + 2] = n2 / ql;
      xr[qri + 3] = n3 / ql;
    }
  }
  if (di == 0) xr[in_dim] = q;
  if (di >= 0) {
    if (!gen) {
      xr[qri] = q_new;
    } else if (qri >= 0 && !sph_lane) {
      xr[qri] = q_new;
    }
    xr[nq + qdri] = qd_new;
  }
  TDS_WAVE_SYNC();

  // ---- y record (q, qd, up, zero padding; the visual poses went out in M1) of the last normal step
  const int n_y_targets = (LOOP && !DEFER && ring_y && pack_y && last_run && y_out != nullptr) ? 2 : 1;
  for (int yt_i = 0; yt_i < n_y_targets; ++yt_i)
  if (pack_y && !(DEFER && ring_y)) {
    TR *const yo = yt_i == 0 ? y_step : y_out + (size_t)env * out_dim;
    if (gen) {  // the q record is not one coordinate per lane: copy it out as it is
      for (int i = lane; i < nq + nd; i += G) TDS_NT_STORE((TR)(xr[i]), &yo[i]);
    } else if (di >= 0) {
      TDS_NT_STORE((TR)(q_new), &yo[di]);
      TDS_NT_STORE((TR)(qd_new), &yo[nq + di]);
    }
    if constexpr (!W2) {  // (two-wavefront workgroup: the helper wavefront wrote the tail of the record)
      const int nv = mdl->num_visuals;
      int tail = nq + nd;
      if (mdl->pack_visuals) {
        tail += 7 * nv;
        // up_dot_world_z (of body 0; lane 0 is its first pseudo link when its base floats)
        if (lane == 0) TDS_NT_STORE((TR)((fl || (flm && fbk == 0)) ? up_z : (two ? mdl->base_Rb[0][8] : mdl->base_R[8])), &yo[tail]);
        tail += 1;
      }
      const int yend = yt_i == 0 ? ystr : out_dim;
      for (int i = tail + lane; i < yend; i += G) TDS_NT_STORE((TR)(0), &yo[i]);
    }
  }

  // ---- N. reward / done of the last normal step
  //         (ars_vectorized_environment.h:250-289; ant_environment2.h:75-106;
  //          laikago_environment2.h:130-171)
  T reward = T(0);  // (lane 0)
  // (Laikago's reward takes the sine and cosine of three half angles: lanes 0..2 evaluate one each, side by side, instead
  //  of lane 0 running the three evaluations in a row with the other lanes of the wavefront waiting — same function, same
  //  arguments, same bits)
  T rw_s0 = T(0), rw_c0 = T(1), rw_s1 = T(0), rw_c1 = T(1), rw_s2 = T(0), rw_c2 = T(1);
  if (mdl->reward_mode == TDS_REWARD_LAIKAGO && nq > 5 && (last_run || pol || pool_r || ring_o)) {
    sincos_t<T>(lane < 3 ? xr[3 + lane] * T(0.5) : T(0), &rw_s0, &rw_c0);
    rw_s1 = dpp_bcast<1>(rw_s0), rw_c1 = dpp_bcast<1>(rw_c0);
    rw_s2 = dpp_bcast<2>(rw_s0), rw_c2 = dpp_bcast<2>(rw_c0);
  }
  if (do_reward && lane == 0) {
    bool done = false;
    const int rm = mdl->reward_mode;
    if (rm == TDS_REWARD_ANT && nq > 2) {
      const T vel_x = (xr[0] - xr[in_dim]) / dt;
      done = xr[2] < T(0.26);
      reward = done ? T(0) : vel_x;
    } else if (rm == TDS_REWARD_LAIKAGO && nq > 5) {
      // up_dot_world_z = quat_to_matrix(quat_from_euler_rpy(q[3..5]))(2,2)
      // (tiny_quaternion.h set_euler_rpy, tiny_matrix3x3.h:315-340)
      const T sp = rw_s0, cp = rw_c0, st = rw_s1, ct = rw_c1, ss = rw_s2, cs2 = rw_c2;
      const T qx = sp * ct * cs2 - cp * st * ss;
      const T qy = cp * st * cs2 + sp * ct * ss;
      const T qz = cp * ct * ss - sp * st * cs2;
      const T qw = cp * ct * cs2 + sp * st * ss;
      const T s2 = T(2) / (qx * qx + qy * qy + qz * qz + qw * qw);
      const T up = T(1) - (qx * (qx * s2) + qy * (qy * s2));
      done = (up < T(0.6)) || (xr[2] < T(0.2));
      reward = done ? T(0) : xr[0];
    } else if (sph && rm == TDS_REWARD_HUMANOID && nq > 6) {
      // up_dot_world_z = quat_to_matrix(q[3..6])(2,2); done = up < 0.6 || z < 0.8; reward = x
      // (humanoid_environment.h:172-196, tiny_matrix3x3.h:315-340)
      const T qx = xr[3], qy = xr[4], qz = xr[5], qw = xr[6];
      const T s2 = T(2) / (qx * qx + qy * qy + qz * qz + qw * qw);
      const T up = T(1) - (qx * (qx * s2) + qy * (qy * s2));
      done = (up < T(0.6)) || (xr[2] < T(0.8));
      reward = done ? T(0) : xr[0];
    }
    if (obs_out != nullptr && last_run) {
      TR *const ob = obs_out + (size_t)env * (nq + nd + 2);
      ob[nq + nd] = (TR)reward;
      ob[nq + nd + 1] = (done || frozen) ? TR(1) : TR(0);
    }
    if constexpr (LOOP) {
      if (ring_o) xr[in_dim + RW_SLOT] = reward;  // (the obs record is stored from the LDS record: put_obs)
    }
    xr[in_dim + 1] = done ? T(1) : T(0);
  }
  if constexpr (!LOOP) {
    // straight-line build: exactly one normal step, no reset -> the environment is finished here;
    // observation (obs[0] = obs[1] = 0, ars_vectorized_environment.h:283-288) and resident state go
    // out straight from registers
    auto write_state = [&]() {
      if (live && gen) {
        for (int i = lane; i < nq + nd; i += G) {
          if (obs_out != nullptr) obs_out[(size_t)env * (nq + nd + 2) + i] = (TR)(i < 2 ? T(0) : xr[i]);
          if (x_feedback != nullptr) x_feedback[(size_t)env * in_dim + i] = (TR)xr[i];
        }
      } else if (live && di >= 0) {
        if (obs_out != nullptr) {
          TR *const ob = obs_out + (size_t)env * (nq + nd + 2);
          ob[di] = (TR)(di < 2 ? T(0) : q_new);
          ob[nq + di] = (TR)qd_new;
        }
        if (x_feedback != nullptr) {
          x_feedback[(size_t)env * in_dim + di] = (TR)q_new;
          x_feedback[(size_t)env * in_dim + nq + di] = (TR)qd_new;
        }
      }
    };
    if (ctl.pool == nullptr) {  // wave-uniform (kernel argument): the ordinary step — nothing of the pool on its path
      write_state();
    } else {
      // auto_reset_when_done (ars_vectorized_environment.h:262-277) without leaving the straight-line kernel: the
      // reset state of (seed, environment, reset count) — re-initialised AND settled — was computed ahead of time
      // into the environment's pool ring (tds_api.hip: reset pool); a done environment copies it in.  y, reward and
      // done describe the terminal step, observation and resident state the fresh environment.
      TDS_WAVE_SYNC();  // lane 0's done flag
      if (live && xr[in_dim + 1] != T(0)) {
        const unsigned c = ctl.reset_count[env];
        const TR *const src = tds_global((const TR *)ctl.pool) + ((size_t)(c % (unsigned)ctl.pool_depth) * ctl.pool_envs + env) * (nq + nd);
        for (int i = lane; i < nq + nd; i += G) {
          const TR v = src[i];
          if (obs_out != nullptr) obs_out[(size_t)env * (nq + nd + 2) + i] = i < 2 ? TR(0) : v;
          if (x_feedback != nullptr) x_feedback[(size_t)env * in_dim + i] = v;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) ctl.reset_count[env] = c + 1u;
      } else {
        write_state();
      }
    }
  } else {
    TDS_WAVE_SYNC();
    const bool done_now = do_reward && xr[in_dim + 1] != T(0);
    const bool auto_r = ctl.reset_mode == TDS_RESET_AUTO;
    // ring records of this step: stored now — the last step of the launch; an environment the pool re-initialises below
    // (its y record describes the TERMINAL state, which the pool entry is about to overwrite) — or in the middle of the
    // next step (DEFER, see put_y_state)
    const bool ring_step = (ring_o || ring_y) && valid && mode == TDS_MODE_RUN;
    const bool ring_now = ring_step && (!DEFER || last_run || (pool_r && done_now));
    if constexpr (DEFER) {
      if (ring_y && ring_now) put_y_state(y_step, ystr);
      if (ring_y && ring_step && last_run && y_out != nullptr) put_y_state(y_out + (size_t)env * out_dim, out_dim);
    }

    // ---- mode transition of this lane group
    bool finished = false;
    if (mode == TDS_MODE_RUN) {
      --left;
      bool reset_now;
      if (pol) {
        // return bookkeeping of Worker::rollouts (ars_vectorized_worker.h:121-137) on top of
        // VectorizedEnvironment::step (ars_vectorized_environment.h:240-289): the step that ends with
        // done is not counted; without auto-reset the environment stays done for the rest of the rollout
        if (!frozen) {
          if (done_now) {
            frozen = !auto_r;
          } else {
            ret += reward - (T)ctl.shift;
            ++cnt;
          }
        }
        reset_now = done_now && auto_r;
      } else if (pool_r) {
        // auto_reset_when_done (ars_vectorized_environment.h:262-277) after EVERY step of the loop, through the pool:
        // entry (reset count mod depth) of the environment's ring is the state reset() + the settle steps lead to
        if (done_now) {
          const unsigned c = ctl.reset_count[env];
          const TR *const src = tds_global((const TR *)ctl.pool) + ((size_t)(c % (unsigned)ctl.pool_depth) * ctl.pool_envs + env) * (nq + nd);
          for (int i = lane; i < nq + nd; i += G) xr[i] = (T)src[i];
          __builtin_amdgcn_wave_barrier();
          if (lane == 0) ctl.reset_count[env] = c + 1u;
        }
        reset_now = false;
      } else {
        reset_now = left == 0 && done_now && auto_r;
      }
      if (reset_now) {  // auto_reset_when_done (ars_vectorized_environment.h:262-277)
        reset_state();
        if (nset > 0) {
          mode = TDS_MODE_SETTLE;
          sleft = nset;
        }
      }
      if (mode == TDS_MODE_RUN && left == 0) {
        mode = TDS_MODE_IDLE;
        finished = true;
      }
    } else if (mode == TDS_MODE_SETTLE) {
      if (--sleft == 0) {
        if (left > 0) {
          mode = TDS_MODE_RUN;
        } else {
          mode = TDS_MODE_IDLE;
          finished = true;
        }
      }
    }
    if (finished && pol && lane == 0) {
      if (ctl.ret_sum != nullptr) ((TR *)ctl.ret_sum)[env] = (TR)ret;
      if (ctl.ret_steps != nullptr) ctl.ret_steps[env] = cnt;
    }
    TDS_WAVE_SYNC();
    // ---- per-step observation record (ring): [q | qd] with obs[0] = obs[1] = 0 (ars_vectorized_environment.h:283-288) of
    //      the state the NEXT step starts from — after an auto-reset through the pool that is the fresh environment,
    //      while reward / done (written above) describe the step that ended (ars_vectorized_environment.h:262-277)
    if (ring_o) {
      if (ctl.peer_arrive != nullptr && (ctl.ring_flags & TDS_RING_WIDE) != 0 && __all(ring_now))
        put_obs_wide((ctl.obs_first + tds_iter) % ctl.obs_slots);
      else if (ring_now)
        put_obs((ctl.obs_first + tds_iter) % ctl.obs_slots);
    }
    // (peer-store exchange: EVERY step of the launch is counted in — the last one here, by the wavefront that has just stored
    //  it; kernel completion would tell this rank, not the peers)
    if (ring_o && ctl.peer_arrive != nullptr && __any(last_run)) peer_signal((ctl.obs_first + tds_iter) % ctl.obs_slots);
    if constexpr (DEFER) {
      if (ring_step && lane == 0) xr[in_dim + OUT_SLOT] = ring_now ? T(1) : T(0);  // "the records of this step are out"
    }
    // ---- the environment is done with this launch: observation (obs[0] = obs[1] = 0,
    //      ars_vectorized_environment.h:283-288) and resident state
    if (finished) {
      // (a forced reset hands out what the environment's own reset() returns: only Ant's zeroes the base x, y —
      //  ant_environment2.h:162-163 vs laikago_environment2.h:63-116; the step always does, ars_vectorized_environment.h:283-288)
      const bool raw = (ctl.flags & TDS_CTL_RESET_CALL) != 0 && mdl->reset_obs_raw_xy != 0;
      for (int i = lane; i < nq + nd; i += G) {
        if (obs_out != nullptr) obs_out[(size_t)env * (nq + nd + 2) + i] = (TR)((i < 2 && !raw) ? T(0) : xr[i]);
        if (x_feedback != nullptr) x_feedback[(size_t)env * in_dim + i] = (TR)xr[i];
      }
    }
  }
  if constexpr (!LOOP) break;
#ifdef TDS_PROF_LOOP
  TDS_STAMP(13);
#endif
  ++tds_iter;
  }  // ================================ end of the step loop ================================
  TDS_STAMP(13);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host side: LDS layout + launch
// ------------------------------------------------------------------------------------------
// padded dof count = template parameter NDP of the kernel.  Besides the coarse widths (8/16/24/32) the
// widths of the two benchmark robots are instantiated exactly for their natural lane count
// (Ant: 14 dof on 16 lanes, Laikago: 18 dof on 32 lanes): LDL^T and the row solves scale with NDP^2.
#if !defined(TDS_ONLY_F32) && !defined(TDS_ONLY_MIX) && (!defined(TDS_ONLY_KIND) || TDS_ONLY_KIND == 0) && !defined(TDS_ALT)
int tds_padded_dof(int nd, int lanes) {
  if (lanes == 16 && nd > 8 && nd <= 14) return 14;
  if (lanes == 32 && nd > 16 && nd <= 18) return 18;
  return nd <= 8 ? 8 : (nd <= 16 ? 16 : (nd <= 24 ? 24 : 32));
}
#endif

// register-pressure experiments compile ONE (lanes, padded dof) instantiation: -DTDS_DEBUG_ONLY=3218
#ifndef TDS_DEBUG_ONLY
#define TDS_DEBUG_ONLY 0
#endif
constexpr bool tds_instantiate(int key) { return TDS_DEBUG_ONLY == 0 || TDS_DEBUG_ONLY == key; }

template <typename T>
TdsLds tds_make_lds_layout(const DevModel<T> &m, int na_cap, int lanes_per_env, bool w2) {
  TdsLds L;
  memset(&L, 0, sizeof(L));
  const int nl = m.num_links;
  const int ndp = tds_padded_dof(m.dof_qd, lanes_per_env);
  L.NLp = nl;
  L.NDP = ndp;
  L.NDs = ndp + 1;  // odd row stride: lane == row accesses hit distinct LDS banks
  const int ncp = m.has_plane ? m.num_cp : 0;
  L.NCPp = ncp > 0 ? ncp : 1;
  // two-body worlds: the contacts between the bodies are a second pass through the same row store
  const int npc = m.num_bodies >= 2 ? m.num_pc : 0;
  L.NPCp = npc > 0 ? npc : 1;
  const int nct = ncp > npc ? ncp : npc;  // contacts of the larger pass
  if (na_cap <= 0 || na_cap > nct) na_cap = nct;
  L.zrows = 3 * na_cap;            // constraint rows kept in LDS
  L.ovrows = 3 * nct - L.zrows;    // surplus rows per environment (global scratch slab)
  int o = 0;
  // persistent for the whole step
  L.xrec = o; o += m.input_dim + 4 + (w2 ? 4 : 0);  // + x_{t-1}, the done flag, the reward and the "records are out" flag of
                                                    //   the step loop (two-wavefront layout: + 2 .. + 5 are the contact counts
                                                    //   and flags handed between the wavefronts)
  // two pairs with disjoint lifetimes share their storage:
  //   swd  (world motion axes per dof: phases C..J)  |  rows (b, 1/(G+cfm), G per constraint row: K..L)
  //   cp   (contact points: phases I..K)             |  xrow (impulses x of all rows: L)
  {
    const int a = 6 * L.NDs, b = 3 * L.zrows;
    if (npc > 0) {  // (the second contact pass builds its rows from the motion axes after the first one's row scalars)
      L.swd = o; o += a;
      L.rows = o; o += b;
    } else {
      L.swd = o; L.rows = o; o += a > b ? a : b;
    }
  }
  {
    const int a = ncp ? 5 * L.NCPp : 0, b = 3 * nct;
    L.cp = o; L.xrow = o; o += a > b ? a : b;
  }
  L.pc = o;
  if (npc > 0) o += 17 * L.NPCp;  // contact list of the pairs: lives from the narrowphase to the second pass
  L.Lp = o;   o += (ndp * (ndp - 1)) / 2;
  L.Lh = o;   // two-wavefront pipeline (narrow kernels): row-major copy of the first ndp/2 columns of L, 16 + 1 rows
  if (w2 && ndp <= 16) o += 17 * (ndp / 2);  // (+ one row for the lanes of wider groups that own no row)
  L.dinv = o; o += (w2 ? 4 : 3) * ndp;  // 1/D | sqrt(1/D) | column scratch of the wide LDL^T / rhs exchange (| y~)
  L.tau = o;
  if ((ndp > 16 && ndp < 24) || (w2 && ndp <= 16 && TDS_PARK_W2)) o += nl;  // (the generalised forces wait here from the PD block to phase F)
  // three phase groups share one region:
  //   1. kinematics sweep:   per-link records [X_world(12) | v(6)]              stride TDS_S1
  //   2. composite sweep:    per-link records [f or F(6) | Ic(10)] stride TDS_S2
  //   3. constraint rows:    Z[zrows][NDs]
  const int u = o;
  int g1 = u;
  L.Xw = g1; g1 += TDS_S1 * L.NLp;
  L.v = g1; g1 += TDS_S1 * m.num_lc_slots;
  // (two-wavefront workgroups: the helper wavefront reads X_world and writes the rows while the main one sweeps the
  //  inertias — the three groups are laid out one after the other)
  int g2 = w2 ? g1 : u;
  L.IA = g2; L.pA = g2; L.F = g2; L.Ic = g2 + 6; L.a = g2; g2 += TDS_S2 * L.NLp;
  int g3 = w2 ? g2 : u;
  L.Z = g3; g3 += L.zrows * L.NDs;
  // Gram form of the contact solve (tds_gram_solve): its 16 x 17 buffer + 16 zeros reuse the two sweep groups
  // Opt-in (TDS_HIP_GRAM=1): measured 0.4k of 32k cycles better than the z~ sweep at Ant x 4096 (profiles/r02d_gram_mfma.txt),
  // and an environment's low-order bits then depend on whether its wavefront-mates push NA past 5 (sweep) or not (Gram).
  L.gram_ok = (tds_opt_now_flag(TDS_OPT_GRAM) && w2 && lanes_per_env == 16 && ndp <= 16 && m.num_bodies < 2 &&
               L.Z - L.Xw >= TDS_GRAM_ZEROS + 16) ? 1 : 0;
  o = g1 > g2 ? g1 : g2;
  o = o > g3 ? o : g3;
  o = (o + 1) & ~1;  // keep 16-byte alignment of every env region for T = double
  L.stride = o;
  L.in_dim = m.input_dim;
  L.adim = m.action_dim;
  L.nqnd = m.dof_q + m.dof_qd;
  return L;
}

template <typename T, typename TR, int KIND>
int tds_launch_step_impl(const DevModel<T> *d_model, const DevModel<T> &h_model, const TdsLds &L, int lanes_per_env,
                         const TR *x_in, TR *y_out, const TR *actions, TR *x_feedback, TR *obs_out, T *ovf, int n_envs,
                         hipStream_t stream, const TdsStepCtl &ctl, long long *prof, int form) {
  const bool two_waves = (form & TDS_FORM_W2) != 0;
  const int epw = 64 / lanes_per_env;
  const int blocks = (n_envs + epw - 1) / epw;
  // (+ the workgroup's constant table of the step-loop launches, TdsLds::cw)
  const size_t shmem = (size_t)L.stride * epw * sizeof(T) + (size_t)L.cw * lanes_per_env * sizeof(T);
  (void)h_model;
#ifdef TDS_PROF_LOOP
#define TDS_PROF_LOOP_LAUNCH(GG, NN)                                                                         \
    if constexpr (KIND == 0 && NN <= 14) {                                                                    \
      if (two_waves && !simple && prof) {                                                                    \
        hipLaunchKernelGGL((tds_step_kernel<T, TR, GG, (NN < 24 ? NN : 8), true, 1, 0, true>), dim3(blocks), dim3(128), shmem, \
                           stream, d_model, L, x_in, y_out, actions, x_feedback, obs_out, ovf, prof, ctl, n_envs); \
        break;                                                                                               \
      }                                                                                                      \
    }
#else
#define TDS_PROF_LOOP_LAUNCH(GG, NN)
#endif
#define TDS_LAUNCH(GG, NN)                                                                                   \
  do {                                                                                                       \
    if constexpr (KIND == 0 && NN <= 14) {                                                                    \
      if (two_waves && simple && !prof) {                                                                    \
        hipLaunchKernelGGL((tds_step_kernel<T, TR, GG, (NN < 24 ? NN : 8), false, 0, 0, true>), dim3(blocks), dim3(128), shmem, \
                           stream, d_model, L, x_in, y_out, actions, x_feedback, obs_out, ovf, prof, ctl, n_envs); \
        break;                                                                                               \
      }                                                                                                      \
      if constexpr (PROF_BUILDS) {                                                                           \
        if (two_waves && simple && prof) {                                                                   \
          hipLaunchKernelGGL((tds_step_kernel<T, TR, GG, (NN < 24 ? NN : 8), true, 0, 0, true>), dim3(blocks), dim3(128), shmem, \
                             stream, d_model, L, x_in, y_out, actions, x_feedback, obs_out, ovf, prof, ctl, n_envs); \
          break;                                                                                             \
        }                                                                                                    \
      }                                                                                                      \
    }                                                                                                        \
    TDS_PROF_LOOP_LAUNCH(GG, NN)                                                                             \
    if constexpr (KIND == 0 && NN <= 14) {                                                                    \
      if (two_waves && !simple && !prof) {                                                                   \
        hipLaunchKernelGGL((tds_step_kernel<T, TR, GG, (NN < 24 ? NN : 8), false, 1, 0, true>), dim3(blocks), dim3(128), shmem, \
                           stream, d_model, L, x_in, y_out, actions, x_feedback, obs_out, ovf, prof, ctl, n_envs); \
        break;                                                                                               \
      }                                                                                                      \
    }                                                                                                        \
    if (prof) {                                                                                              \
      if constexpr (PROF_BUILDS)                                                                             \
        hipLaunchKernelGGL((tds_step_kernel<T, TR, GG, NN, true, 0, 0>), dim3(blocks), dim3(64), shmem, stream, \
                           d_model, L, x_in, y_out, actions, x_feedback, obs_out, ovf, prof, ctl, n_envs);   \
      else return -2;                                                                                        \
    } else if (simple)                                                                                         \
      hipLaunchKernelGGL((tds_step_kernel<T, TR, GG, NN, false, 0, KIND>), dim3(blocks), dim3(64), shmem, stream, \
                         d_model, L, x_in, y_out, actions, x_feedback, obs_out, ovf, prof, ctl, n_envs);     \
    else if constexpr (NN < 24)                                                                              \
      hipLaunchKernelGGL((tds_step_kernel<T, TR, GG, NN, false, 2, KIND>), dim3(blocks), dim3(64), shmem, stream, \
                         d_model, L, x_in, y_out, actions, x_feedback, obs_out, ovf, prof, ctl, n_envs);     \
    else                                                                                                     \
      hipLaunchKernelGGL((tds_step_kernel<T, TR, GG, NN, false, 1, KIND>), dim3(blocks), dim3(64), shmem, stream, \
                         d_model, L, x_in, y_out, actions, x_feedback, obs_out, ovf, prof, ctl, n_envs);     \
  } while (0)
  // the phase-stamp builds exist for the plain kernels with double records only (a diagnostic: profiles/*_phases.txt)
  constexpr bool PROF_BUILDS = KIND == 0 && sizeof(T) == 8 && sizeof(TR) == 8;
  if (prof && !PROF_BUILDS) return -2;
  // straight-line kernel when the launch is exactly one normal step without any reset
  const bool simple = ctl.nsub == 1 && ctl.reset_mode == TDS_RESET_NONE && ctl.policy == nullptr &&
                      ctl.obs_ring == nullptr && ctl.y_ring == nullptr;  // (record rings: the step-loop builds write them)
  // step-loop build: above one wavefront per SIMD (256 CUs x 4 SIMDs) the two-wavefronts-per-SIMD compilation wins
  // (option loop_occ = 1 / 2 forces either: TDS_FORM_LOOP_OCC*)
  const int loop_force = (form & TDS_FORM_LOOP_OCC2) ? 2 : ((form & TDS_FORM_LOOP_OCC1) ? 1 : 0);
  // (>= 24 padded dof: the straight-line build is itself at one wavefront per SIMD; the two-wave loop build would
  //  spill hundreds of registers there)
  // (round 4: compiled without MachineLICM the two-wavefronts-per-SIMD compilation of the kernels up to 14 padded dof
  //  holds no scratch — 245 VGPR — and is taken at ANY grid size: the one-wavefront-per-SIMD compilation (256 VGPR + 20
  //  AGPR copies) buys nothing there any more, and its <double, double, 16, 8> instantiation does not terminate when
  //  built without the pass — profiles/r04_diag_loop_hang.txt.  Round 5: that compilation is no longer INSTANTIATED
  //  below 14 padded dof — nothing can launch it; option loop_occ = 1 is refused there by tds_api.hip: launch(),
  //  TDS_ERR_UNSUPPORTED, and tests/test_options.py runs the 8-dof loop launches under a watchdog)
  // (round 6: below 24 padded dof only the two-wavefronts-per-SIMD compilation is instantiated — the robots whose batch
  //  sizes made the other one worth its build time, 14 and 18 dof, run in kernels of their own: tds_oct.hip, tds_quad.hip)
  (void)loop_force;
  const int key = lanes_per_env * 100 + L.NDP;
  switch (key) {
#define TDS_CASE(GG, NN) \
  case GG * 100 + NN:    \
    if constexpr (tds_instantiate(GG * 100 + NN)) TDS_LAUNCH(GG, NN); else return -1; \
    break;
    TDS_CASE(16, 8) TDS_CASE(16, 14) TDS_CASE(32, 18) TDS_CASE(16, 16) TDS_CASE(32, 8) TDS_CASE(32, 16) TDS_CASE(32, 24)
    TDS_CASE(32, 32) TDS_CASE(64, 8) TDS_CASE(64, 16)
#undef TDS_CASE
    default:
      return -1;
  }
#undef TDS_LAUNCH
  return (int)hipGetLastError();
}

template <typename T, typename TR, int KIND>
int tds_kernel_max_dynamic_lds_impl(int lanes_per_env, int ndp, int bytes) {
  hipError_t e = hipSuccess;
#define TDS_ATTR(GG, NN)                                                                                        \
  do {                                                                                                          \
    e = hipFuncSetAttribute((const void *)tds_step_kernel<T, TR, GG, NN, false, 0, KIND>,                         \
                            hipFuncAttributeMaxDynamicSharedMemorySize, bytes);                                 \
    if constexpr (NN >= 24) {                                                                                   \
      if (e == hipSuccess)                                                                                      \
        e = hipFuncSetAttribute((const void *)tds_step_kernel<T, TR, GG, NN, false, 1, KIND>,                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);                             \
    } else {                                                                                                    \
      if (e == hipSuccess)                                                                                      \
        e = hipFuncSetAttribute((const void *)tds_step_kernel<T, TR, GG, NN, false, 2, KIND>,                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);                             \
    }                                                                                                           \
    if constexpr (KIND == 0 && sizeof(T) == 8 && sizeof(TR) == 8) {                                             \
      if (e == hipSuccess)                                                                                      \
        e = hipFuncSetAttribute((const void *)tds_step_kernel<T, TR, GG, NN, true, 0, 0>,                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);                             \
    }                                                                                                           \
    if constexpr (KIND == 0 && NN <= 14) {                                                                       \
      if (e == hipSuccess)                                                                                      \
        e = hipFuncSetAttribute((const void *)tds_step_kernel<T, TR, GG, NN, false, 0, 0, true>,                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);                             \
      if (e == hipSuccess)                                                                                      \
        e = hipFuncSetAttribute((const void *)tds_step_kernel<T, TR, GG, NN, false, 1, 0, true>,                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);                             \
    }                                                                                                           \
  } while (0)
  switch (lanes_per_env * 100 + ndp) {
#define TDS_CASE(GG, NN) \
  case GG * 100 + NN:    \
    if constexpr (tds_instantiate(GG * 100 + NN)) TDS_ATTR(GG, NN); \
    break;
    TDS_CASE(16, 8) TDS_CASE(16, 14) TDS_CASE(32, 18) TDS_CASE(16, 16) TDS_CASE(32, 8) TDS_CASE(32, 16) TDS_CASE(32, 24)
    TDS_CASE(32, 32) TDS_CASE(64, 8) TDS_CASE(64, 16)
#undef TDS_CASE
    default: return -1;
  }
#undef TDS_ATTR
  return (int)e;
}

// The file is compiled fifteen times (csrc/Makefile): -DTDS_ONLY_F64 / -DTDS_ONLY_F32 / -DTDS_ONLY_MIX pick the build
// (compute scalar, record scalar) = (double, double) / (float, float) / (double, float), -DTDS_ONLY_KIND=0..4 the
// kernel kind (plain / floating base / spherical joints / several bodies / several bodies with floating bases), so
// that the parts of the kernel set build in parallel.
// (tds_make_lds_layout and tds_padded_dof live in the KIND 0 units.)
#define TDS_INSTANTIATE(TT, TR, KV)                                                                                     \
  template int tds_launch_step_impl<TT, TR, KV>(const DevModel<TT> *, const DevModel<TT> &, const TdsLds &, int,       \
                                                const TR *, TR *, const TR *, TR *, TR *, TT *, int, hipStream_t,     \
                                                const TdsStepCtl &, long long *, int);                                 \
  template int tds_kernel_max_dynamic_lds_impl<TT, TR, KV>(int, int, int);
#if !defined(TDS_ONLY_KIND)
#define TDS_ALL_KINDS 1
#define TDS_ONLY_KIND -1
#endif
#if !defined(TDS_ONLY_F64) && !defined(TDS_ONLY_F32) && !defined(TDS_ONLY_MIX)
#define TDS_ONLY_F64 1
#define TDS_ONLY_F32 1
#define TDS_ONLY_MIX 1
#endif
#if TDS_ONLY_KIND == 0 || defined(TDS_ALL_KINDS)
#define TDS_INSTANTIATE_K0(TT, TR) TDS_INSTANTIATE(TT, TR, 0)
#else
#define TDS_INSTANTIATE_K0(TT, TR)
#endif
#if TDS_ONLY_KIND == 1 || defined(TDS_ALL_KINDS)
#define TDS_INSTANTIATE_K1(TT, TR) TDS_INSTANTIATE(TT, TR, 1)
#else
#define TDS_INSTANTIATE_K1(TT, TR)
#endif
#if TDS_ONLY_KIND == 2 || defined(TDS_ALL_KINDS)
#define TDS_INSTANTIATE_K2(TT, TR) TDS_INSTANTIATE(TT, TR, 2)
#else
#define TDS_INSTANTIATE_K2(TT, TR)
#endif
#if TDS_ONLY_KIND == 3 || defined(TDS_ALL_KINDS)
#define TDS_INSTANTIATE_K3(TT, TR) TDS_INSTANTIATE(TT, TR, 3)
#else
#define TDS_INSTANTIATE_K3(TT, TR)
#endif
#if TDS_ONLY_KIND == 4 || defined(TDS_ALL_KINDS)
#define TDS_INSTANTIATE_K4(TT, TR) TDS_INSTANTIATE(TT, TR, 4)
#else
#define TDS_INSTANTIATE_K4(TT, TR)
#endif
#define TDS_INSTANTIATE_KINDS(TT, TR) \
  TDS_INSTANTIATE_K0(TT, TR) TDS_INSTANTIATE_K1(TT, TR) TDS_INSTANTIATE_K2(TT, TR) TDS_INSTANTIATE_K3(TT, TR) \
  TDS_INSTANTIATE_K4(TT, TR)
#if defined(TDS_ONLY_F64)
#if (TDS_ONLY_KIND == 0 || defined(TDS_ALL_KINDS)) && !defined(TDS_ALT)
template TdsLds tds_make_lds_layout<double>(const DevModel<double> &, int, int, bool);
#endif
TDS_INSTANTIATE_KINDS(double, double)
#endif
#ifdef TDS_ALT
// the slot's entry point (tds_kernels.h: EXPERIMENT SLOTS); *lanes_ndp_key = the one instantiation this unit holds
extern "C" int TDS_ALT_PASTE(tds_alt_launch_, TDS_ALT)(const void *d_model, const void *h_model, const TdsLds *L,
                                                       int lanes_per_env, const void *x_in, void *y_out, const void *actions,
                                                       void *x_feedback, void *obs_out, void *ovf, int n_envs,
                                                       hipStream_t stream, const TdsStepCtl *ctl, int form, int *lanes_ndp_key) {
  if (lanes_ndp_key) *lanes_ndp_key = TDS_DEBUG_ONLY;
  if (!d_model) return 0;  // (query only)
  return tds_launch_step_impl<double, double, 0>((const DevModel<double> *)d_model, *(const DevModel<double> *)h_model, *L,
                                                 lanes_per_env, (const double *)x_in, (double *)y_out, (const double *)actions,
                                                 (double *)x_feedback, (double *)obs_out, (double *)ovf, n_envs, stream, *ctl,
                                                 nullptr, form);
}
#endif
#if defined(TDS_ONLY_MIX)
TDS_INSTANTIATE_KINDS(double, float)
#endif
#if defined(TDS_ONLY_F32)
#if TDS_ONLY_KIND == 0 || defined(TDS_ALL_KINDS)
template TdsLds tds_make_lds_layout<float>(const DevModel<float> &, int, int, bool);
#endif
TDS_INSTANTIATE_KINDS(float, float)
#endif
