// tds_oct.hip — the step kernel of the star-shaped robots with TWO-link legs on 8 lanes per environment (round 6; the
// headline: BASELINE configs 3 and 5, the gym Ant).
//
// The Ant is the same star as Laikago (tds_quad.hip): a root body on the reference's six "virtual" links (prismatic x, y, z,
// revolute x, y, z: tds_device_model.h euler_root) and four legs with nothing between them but the root — here serial
// chains of two links (hip, ankle), every link with a 1-dof joint and a capsule, the root body with a sphere: 17 contact
// points against the plane.  The general kernel gives it a 16-lane group and a dense 14 x 14 register LDL^T; this kernel:
//
//   * 8 lanes per environment = the 8 leg links, one PAIR of lanes per leg (lane = 2 leg + position in the chain); EIGHT
//     environments per wavefront.  The root chain needs no lane: pose, motion axes, velocity and bias acceleration of
//     the root body in closed form on every lane (the formulas of the general kernel's phase C); its rigid inertia
//     redundantly on every lane; the legs' composites reach it by three DPP moves (pair, quad, half-row mirror).
//   * M in LEAVES-FIRST order [leg 0 | leg 1 | leg 2 | leg 3 | root]: four independent 2 x 2 leg blocks, their 6-column
//     couplings to the root, the root's 6 x 6 block.  LDL^T: both lanes of a pair factorise their leg's block (values by
//     quad_perm moves, no LDS), the coupling rows L_c = W D^-1 follow inside the pair, the root's Schur complement
//     S = R - sum_lanes L_c W^T is R in registers (closed form of the six root axes) minus 21 lane-parallel sums over an
//     LDS copy of (L_c | W), factorised redundantly on every lane.  Forward dynamics qdd = M^-1 (tau - C) and the final
//     qd -= M^-1 J^T p are pair-local substitutions plus six 8-lane sums.
//   * a contact's constraint row touches its leg's two dofs and the root's six: z~ = D^-1/2 L^-1 J^T is 8 wide.  The
//     rows are never all in memory: the Gauss-Seidel sweep (mb_constraint_solver.hpp:101-142) visits them in the
//     reference's order — normals, tangents 1, tangents 2, each by contact — in WINDOWS of eight: eight lanes solve the
//     rows of eight consecutive sweep positions (lane == row), the sweep consumes them, the next window follows.  No
//     contact-count cap, no overflow slab: all 17 points of an environment may penetrate at once.
//
// Same arithmetic contract as the general kernel — the reference's env step (locomotion_contact_simulation.h:151-304) to
// round-off, the same quirks (contacts from pre-step transforms with post-integration velocities, plane_space's k,
// unnormalised REVOLUTE_AXIS axes, visual poses that lag q by one step) — pinned by the same tests: a handle takes this
// kernel when its model is such a star (DevModel::oct, tds_device_model.h) and option oct is not 0; option oct = 0 keeps
// the general kernel, and tests/test_oct.py holds the two against each other and against the reference.
//
// Two forms (template parameter LOOP) as in tds_quad.hip: one step per launch, and K steps per launch with the state in
// LDS, a fresh action block per step, per-step record rings, reset-pool entries taken inside the loop and the exchange of
// the multi-GPU layer (tds_shard.hip: progress counters, peer stores).  Two builds of each (template parameter W2): one
// wavefront per workgroup of eight environments, and TWO — a main wavefront on the step's dependent chain and a helper /
// recorder beside it (narrowphase, visual poses, constraint rows, every record store), the form the host grants while
// every workgroup is resident with at most two wavefronts per SIMD (Ant x 4096: one wavefront per SIMD; x 8192: two).
// In-kernel reset + settle, substeps with one action through a policy and on-device rollouts stay with the general
// kernel.  Reference files as in tds_kernels.hip.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "tds_device_model.h"
#include "tds_kernels.h"
#include "tds_lanes.h"

// Phase stamps (a build of its own: -DTDS_OCT_PROF, tools/oct_profile.sh): workgroup TDS_OCT_PROF_WG writes the shader clock
// at the phase boundaries of iteration TDS_OCT_PROF_ITER of a step-loop launch into tds_oct_prof_buf
#ifdef TDS_OCT_PROF
#ifndef TDS_OCT_PROF_WG
#define TDS_OCT_PROF_WG 3
#endif
__device__ unsigned long long tds_oct_prof_buf[32];
// (shader clock, 100 MHz real-time clock) at the top of the first 32 iterations of the stamped workgroup's main wavefront: the
// shader clock's frequency step by step (tools/oct_clock_ramp.py)
__device__ unsigned long long tds_oct_prof_clk[64];
// 100 MHz real-time clock of every workgroup (up to 2048) at its first instruction, at the top of its first step and behind its
// last step: how far apart the workgroups of a launch start and end (tools/oct_clock_ramp.py)
__device__ unsigned long long tds_oct_prof_wg[3 * 2048];
__device__ int tds_oct_prof_iter = 500;
#define OCT_STAMP(k, pin)                                                                       \
  do {                                                                                          \
    unsigned long long t_;                                                                      \
    auto p_ = (pin); /* a value of the phase before: computed before the clock is read */       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(p_)::"memory");        \
    if (prof_on) prof_t[k] = t_;                                                                \
  } while (0)
#else
#define OCT_STAMP(k, pin) do { } while (0)
#endif

// -DTDS_OCT_MARKS: "; OCTMARK <name>" comment lines in the assembly at the phase boundaries (tools/oct_isa_phases.py counts the
// instructions between them: on a lone wavefront the instruction count IS the time)
#ifdef TDS_OCT_MARKS
#define OCT_MARK(name) asm volatile("; OCTMARK " name ::: "memory")
#else
#define OCT_MARK(name) do { } while (0)
#endif

namespace {

#define OCT_SYNC()                                             \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     \
    __builtin_amdgcn_wave_barrier();                           \
  } while (0)

// ---- DPP moves inside the 8-lane group of an environment (two environments share a 16-lane DPP row) ----
template <int CTRL>
__device__ __forceinline__ double oct_dpp(double v) {
  const int l = __double2loint(v), h = __double2hiint(v);
  const int lo = __builtin_amdgcn_update_dpp(0, l, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, h, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float oct_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// the other lane of my pair (quad_perm [1, 0, 3, 2])
template <typename T>
__device__ __forceinline__ T pair_other(T v) { return oct_dpp<0xB1>(v); }
// lane K of my pair to both lanes (quad_perm [0, 0, 2, 2] / [1, 1, 3, 3])
template <int K, typename T>
__device__ __forceinline__ T pair_bcast(T v) { return oct_dpp<(K == 0 ? 0xA0 : 0xF5)>(v); }
// sum over the 8 lanes of an environment, every lane receives the SAME bits: pair (a + b and b + a are the same sum),
// pair of pairs, then the mirror image of the half row (lane i <- lane 7 - i: the other quad, whose four lanes agree)
// (the operand is pinned in a register first: a product handed in must be ROUNDED before the first addition — contracted
//  into it, lane i would add its exact product to lane i ^ 1's rounded one and lane i ^ 1 the other way round: the two
//  "same" sums then differ in the last bit, and with them everything that is redundant on every lane behind them; an
//  environment's result would depend on which lane solves which constraint row, i.e. on its wavefront-mates' contacts)
template <typename T>
__device__ __forceinline__ T oct_sum(T v) {
  asm volatile("" : "+v"(v));
  v += oct_dpp<0xB1>(v);   // quad_perm [1, 0, 3, 2]
  v += oct_dpp<0x4E>(v);   // quad_perm [2, 3, 0, 1]
  v += oct_dpp<0x141>(v);  // row_half_mirror
  return v;
}
// lane K (0..7) of every environment to its 8 lanes: row_newbcast of lane K into the first half of every 16-lane row, of
// lane 8 + K into the second (bank masks: a bank is four lanes)
template <int K>
__device__ __forceinline__ double oct_bcast(double v) {
  const double t = __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xF, 0x3, false);
  return __builtin_amdgcn_update_dpp(t, v, 0x150 + 8 + K, 0xF, 0xC, false);
}
template <int K>
__device__ __forceinline__ float oct_bcast(float v) {
  const int b = __float_as_int(v);
  const int t = __builtin_amdgcn_update_dpp(0, b, 0x150 + K, 0xF, 0x3, false);
  return __int_as_float(__builtin_amdgcn_update_dpp(t, b, 0x150 + 8 + K, 0xF, 0xC, false));
}

// sin and cos of a joint angle: Cody-Waite reduction by pi / 2 in two fused steps (exact for |x| < 1e5: the product k * hi is
// formed exactly inside the FMA and cancels against x) and the fdlibm kernels on [-pi/4, pi/4] (__kernel_sin / __kernel_cos:
// < 1 ulp) — ~35 instructions where the library routine takes ~90 with its branch to the Payne-Hanek reduction; angles
// beyond 1e5 rad (no simulation gets there, but a caller may hand in anything) take the library routine, wave-uniformly
__device__ __forceinline__ void oct_sincos(double x, double *sn, double *cs) {
  const bool big = !(__builtin_fabs(x) < 1.0e5);
  const double k = __builtin_rint(x * 6.36619772367581382433e-01);
  double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);
  r = __builtin_fma(-k, 6.12323399573676603587e-17, r);
  const int q = (int)k;
  const double z = r * r;
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
  ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
  ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
  const double s0 = __builtin_fma(z * r, ps, r);
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
  pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
  pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  const double c0 = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
  const bool swap = (q & 1) != 0;
  const double ss = swap ? c0 : s0, cc = swap ? s0 : c0;
  double s_ = (q & 2) ? -ss : ss, c_ = ((q + 1) & 2) ? -cc : cc;
  // (the lanes beyond 1e5 — or NaN — take the library routine; the OTHER lanes of the wavefront keep their own result: an
  //  environment's bits must not depend on a wavefront-mate that has left the finite range)
  if (__builtin_expect(__any(big), 0)) {
    double s2, c2;
    sincos(x, &s2, &c2);
    s_ = big ? s2 : s_;
    c_ = big ? c2 : c_;
  }
  *sn = s_;
  *cs = c_;
}
__device__ __forceinline__ void oct_sincos(float x, float *sn, float *cs) { sincosf(x, sn, cs); }

// LDS per environment, offsets in scalars of T
struct OctOff {
  int lcw, legf, swl, qdp, fac, win, xs, cp, stride;
};
struct OctLds {
  static constexpr int LCW = 6;   // [8 lanes][Lc(6)]  (W = Lc D, needed for the Schur sums only, borrows the impulses' slots)
  // a window row: z~ leg (2) | z~ root (6) | 0 | 0 | b | 1 / (G + cfm) | G | hip lane of the contact's leg.  (The two zeros: lane j
  // of the sweep reads root entry j — lanes 6, 7 of an environment read them.)  Between its two halves (see help_rows_geom /
  // rows_solve) a row holds the unsolved Jacobian row and the contact's distance instead.
  static constexpr int ZW = 14;
  static constexpr int Z_B = 10, Z_A = 11, Z_G = 12, Z_HL = 13;
  static constexpr int NCP = 17;  // contact points (torso + 2 per leg link), and the stride of the contact list
};
__host__ __device__ inline OctOff oct_layout(int in_dim) {
  OctOff o;
  int at = in_dim + 4;          // x record | x_{t-1} | done | reward | flag / count
  at = (at + 1) & ~1;
  o.lcw = at;  at += 8 * OctLds::LCW;     // L_c of the 8 leg-dof lanes
  o.legf = at; at += 4 * 3;               // per leg: l10 | sqrt(1/d0) | sqrt(1/d1)
  o.swl = at;  at += 8 * 6;               // world motion axis of the 8 leg-dof lanes
  o.qdp = at;  at += 8;                   // leg velocities after integrate_euler_qdd
  o.fac = at;  at += 28;                  // the 21 lane-parallel sums of the Schur complement; then (two-wavefront build) the root
                                          // block's factors for the helper: L_S (15) | sqrt(1/D_S) (6) | root velocities (6)
  o.win = at;  at += 2 * 8 * OctLds::ZW;  // two windows of constraint rows (the second doubles as the hand-over of the links' world
                                          // transforms, 8 x 12, and the root's sines / cosines, 6, from the main wavefront to the helper)
  o.xs = at;   at += 3 * OctLds::NCP + 1; // impulses (before the sweep: W = L_c D of the 8 leg-dof lanes, for the Schur sums)
  o.cp = at;   at += 5 * OctLds::NCP + 1; // contact list: point (3) | distance | owner lane, per slot
  // the environments of a wavefront on different banks: four consecutive ones (half a wavefront) must not meet on an
  // 8-byte bank pair — the stride in 4-byte words a multiple of 8 that is neither 0 nor 32 mod 64
  at = (at + 1) & ~1;
  while ((((at * 2) & 63) % 8) != 0 || ((at * 2) & 63) == 0 || ((at * 2) & 63) == 32) at += 2;
  o.stride = at;
  return o;
}

struct OctKernArgs {
  const void *mdl, *x_in;
  void *y_out;
  const void *actions;
  void *x_feedback, *obs_out;
  TdsStepCtl ctl;
  int n_envs;
  OctOff O;
};
template <bool LOOP>
struct OctCtlRef {
  using type = const TdsStepCtl &;
  static __device__ __forceinline__ type get(const TdsStepCtl &param, const __attribute__((address_space(4))) char *) { return param; }
};
template <>
struct OctCtlRef<true> {
  using type = const __attribute__((address_space(4))) TdsStepCtl &;
  static __device__ __forceinline__ type get(const TdsStepCtl &, const __attribute__((address_space(4))) char *at) {
    return *(const __attribute__((address_space(4))) TdsStepCtl *)at;
  }
};
template <typename P>
__device__ __forceinline__ P *oct_global(P *p) {  // a loaded pointer: not LDS, not scratch (global_ instead of flat_ accesses)
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_assume(!__builtin_amdgcn_is_shared((const void *)p) && !__builtin_amdgcn_is_private((const void *)p));
#endif
  return p;
}

// the constants of the model as the kernel reads them: one table in LDS behind the environments' regions, copied from
// DevModel::oct_tab (laid out by tds_build_oct_table on the host: tds_device_model.h, TDS_OCT_*) at the top of a launch
using TB = TdsOctTab;

// barrier between the two wavefronts of a workgroup (W2: LDS writes done, then s_barrier — neither wavefront waits for its
// outstanding global stores); a compiler-level fence in the one-wave build, where the LDS executes a wavefront's
// instructions in order
#ifdef TDS_OCT_NT
#define OCT_ST(v, p) __builtin_nontemporal_store((v), (p))
#else
#define OCT_ST(v, p) (*(p) = (v))
#endif
#define OCT_BAR()                                                                        \
  do {                                                                                   \
    if constexpr (W2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    \
    else OCT_SYNC();                                                                     \
  } while (0)

// LOOP: K steps per launch (state in LDS).  W2: TWO wavefronts per workgroup of eight environments — the MAIN wavefront walks
// the step's dependent chain (PD, kinematics, inertias, LDL^T, forward dynamics, the Gauss-Seidel sweep, integration,
// reward), the HELPER does what hangs off it (narrowphase, visual poses, the constraint rows window by window, every record
// store, the exchange); in the one-wave build the one wavefront plays both roles in the same order.
// BUILD: 1 one wavefront per workgroup; 2 two, compiled for two wavefronts per SIMD (256 registers); 3 two, compiled for one
// wavefront per SIMD (what a launch of at most two workgroups per compute unit gets anyway: Ant x 4096 — no spills)
// BUILD 4: the two-wavefront build for launches that run BESIDE another launch's wavefronts on a SIMD (the reset pool's refill
// passes beside the chunk they top up, tds_api.hip: pool_run): at most 240 registers — a wavefront of the one-wavefront-per-SIMD
// build holds 272 of a SIMD's 512 — every wavefront at the lowest priority.  (The kernel's body is a device function inlined
// into two kernels, because the register limit is an attribute that takes a literal.)
template <typename T, typename TR, bool LOOP, int BUILD>
__device__ __forceinline__ void oct_body(const DevModel<T> *__restrict__ mdl_arg, const TR *x_in, TR *__restrict__ y_out,
                                         const TR *__restrict__ actions, TR *x_feedback /* may alias x_in */,
                                         TR *__restrict__ obs_out, const TdsStepCtl &ctl_arg, int n_envs, const OctOff &O) {
  extern __shared__ __align__(16) unsigned char tds_oct_smem[];
  T *const sm = reinterpret_cast<T *>(tds_oct_smem);
#ifdef TDS_OCT_PROF
  if (threadIdx.x == 0 && blockIdx.x < 2048) {
    unsigned long long c_;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_)::"memory");
    tds_oct_prof_wg[3 * blockIdx.x] = c_;
  }
#endif
  constexpr bool W2 = BUILD >= 2;
  constexpr int nq = 14, nd = 14, adim = 8, in_dim = nq + nd + adim + 3, w_obs = nq + nd + 2;
  constexpr int NT = W2 ? 128 : 64;
  T *const CT = sm + 8 * O.stride;  // the constant table
  {
    // ---- the constant table (coalesced copy) and A. x record -> LDS, fresh actions over the action slice
    // (every global load of the prologue is issued before the first of them is waited for: as loops of load -> LDS store the
    //  table copy and the record copy were ten dependent round trips, 4.8 us in front of the first step of every launch —
    //  tools/oct_clock_ramp.py)
    const int t = threadIdx.x;
    constexpr int TN = (TB::TOTAL + NT - 1) / NT, XN = (in_dim + 7) / 8;
    T tv[TN], xv[XN];
#pragma unroll
    for (int k = 0; k < TN; ++k) {
      const int i = t + k * NT;
      tv[k] = i < TB::TOTAL ? mdl_arg->oct_tab[i] : T(0);
    }
    const int lane0 = t & 7, grp0 = (t & 63) >> 3, env0 = blockIdx.x * 8 + grp0;
    const bool valid0 = env0 < n_envs && t < 64;
#pragma unroll
    for (int k = 0; k < XN; ++k) {
      const int i = lane0 + 8 * k;
      const bool act = actions != nullptr && i >= nq + nd && i < nq + nd + adim;
      xv[k] = (!valid0 || i >= in_dim) ? T(0) : act ? (T)actions[(size_t)env0 * adim + (i - nq - nd)] : (T)x_in[(size_t)env0 * in_dim + i];
    }
#pragma unroll
    for (int k = 0; k < TN; ++k) {
      const int i = t + k * NT;
      if (i < TB::TOTAL) CT[i] = tv[k];
    }
    if (t < 64) {
      T *const xr = sm + grp0 * O.stride;
#pragma unroll
      for (int k = 0; k < XN; ++k) {
        const int i = lane0 + 8 * k;
        if (i < in_dim) xr[i] = xv[k];
      }
      if (lane0 < 4) xr[in_dim + lane0] = T(0);
    }
    if constexpr (W2) __syncthreads();
    else OCT_SYNC();
  }
  const int nsteps = LOOP ? ctl_arg.nsub : 1;
  T next_act = T(0);  // (step-loop form, main wavefront: the action of the NEXT step, requested a step ahead)
  // ring positions of the current step, carried along as scalars (a remainder by a run-time divisor per step and ring costs
  // ~100 instructions at the top of the main wavefront's chain): action block of step it + 1, y slot and obs slot of step it
  int act_blk = 0, y_slot = 0, o_slot = 0;
  if constexpr (LOOP) {
    if (ctl_arg.act_pool != nullptr && ctl_arg.act_blocks > 0) act_blk = (ctl_arg.act_first + 1) % ctl_arg.act_blocks;
    if (ctl_arg.y_ring != nullptr && ctl_arg.y_slots > 0) y_slot = ctl_arg.y_first % ctl_arg.y_slots;
    if (ctl_arg.obs_ring != nullptr && ctl_arg.obs_slots > 0) o_slot = ctl_arg.obs_first % ctl_arg.obs_slots;
  }
  // (where my lane's action sits in the record: one register across the steps, against a table read in front of the PD block's
  //  read of the action — two LDS round trips in a row at the top of the main wavefront's step)
  const int act_slot = (int)(CT + (threadIdx.x & 7) * TB::LSTR)[TB::ACT];
  for (int it = 0; it < nsteps; ++it) {  // ================================ step loop ================================
  // (nothing but `it`, next_act and act_slot lives across an iteration: lane and kernel-argument segment are laundered)
  const __attribute__((address_space(4))) char *ka_seg = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
  int tid = threadIdx.x;
  if constexpr (LOOP) asm volatile("" : "+s"(ka_seg), "+v"(tid));
  const int wv = W2 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
  const bool is_main = !W2 || wv == 0, is_help = !W2 || wv == 1;  // wave-uniform (the profile build's stamps)
  // Two wavefronts per SIMD (BUILD 2): where a main wavefront shares its SIMD with a helper, the main one — the step's
  // instruction stream — issues first (Ant x 8192, same box: 7.93 / 8.09e8 -> 8.42 / 8.37e8 env-steps/s; no effect at one
  // wavefront per SIMD)
#ifndef TDS_OCT_PRIO
#define TDS_OCT_PRIO 1
#endif
  if constexpr (W2 && TDS_OCT_PRIO != 0) {
    if (BUILD != 4 && wv == 0) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(0);  // (a helper one step in front of a refill pass's wavefronts on its SIMD: measured, nothing)
  }
  (void)is_main;
  (void)is_help;
#ifdef TDS_OCT_PROF
  unsigned long long prof_t[26];
  const bool prof_on = blockIdx.x == TDS_OCT_PROF_WG && it == tds_oct_prof_iter;
  if (it == 0 && tid == 0 && blockIdx.x < 2048) {
    unsigned long long c_;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_)::"memory");
    tds_oct_prof_wg[3 * blockIdx.x + 1] = c_;
  }
  if (blockIdx.x == TDS_OCT_PROF_WG && it < 32 && tid == 0) {
    unsigned long long c0_, c1_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0_), "=s"(c1_)::"memory");
    tds_oct_prof_clk[2 * it] = c0_;
    tds_oct_prof_clk[2 * it + 1] = c1_;
  }
#endif
  const int lane = tid & 7;
  const int grp = (tid & 63) >> 3;
  const int env = blockIdx.x * 8 + grp;
  const bool valid = env < n_envs;
  T *const E = sm + grp * O.stride;
  T *const xr = E;
  const int leg = lane >> 1, pos = lane & 1;
  const int dq = 6 + lane;  // my dof in the q / qd records
  const T *const CL = CT + lane * TB::LSTR;  // my lane's constants
  typename OctCtlRef<LOOP>::type ctl = OctCtlRef<LOOP>::get(ctl_arg, ka_seg + __builtin_offsetof(OctKernArgs, ctl));
  const T dt = CT[TB::SC + TB::DT];
  const int pgs_iters = (int)CT[TB::SC + TB::PGS_ITERATIONS];
  const bool last = it == nsteps - 1;
  OCT_STAMP(0, tid);

  // state shared by the phases of a role (main: kinematics .. integration; helper: narrowphase .. records); what crosses from
  // the main wavefront to the helper goes through LDS in the two-wavefront build and stays in these registers otherwise
  T q, qd, tau;
  T Sl[6], R[9], p[3], sw[6], v[6], a0[6];
  T R5[9], P[3], A4[3], A5[3], pA3[3], pA4[3], pA5[3], v5[6], a5[6];
  T Lc[6], l10, id0, id1, sq0, sq1, my_id, my_sq, Ls[15], ids[6], sq_ids[6], qd_new, qdr_new[6];
  T Cb, Cr[6], It[10];  // (main wavefront: bias forces of my dof / the root dofs, the robot's total inertia — across its barriers)
  T u = T(0), urm = T(0);  // u~ = -dt y~ + sum_r z~_r x_r, distributed: my own dof's leg entry | root entry `lane` (lanes 6, 7: zero)
  T u_init = T(0), urm_init = T(0);
  int na = 0, NA = 0;

  // the root body's frame and the root's revolute axes from the six sines / cosines (kinematics.hpp:64-97; tds_kernels.hip
  // phase C: same formulas): R5, P, A4, A5 and the linear parts P x A of the three revolute axes (A3 = e_x)
  auto root_frame = [&](T sx, T cx, T sy, T cy, T sz, T cz) {
    R5[0] = cy * cz;                 R5[1] = -cy * sz;                R5[2] = sy;
    R5[3] = sx * sy * cz + cx * sz;  R5[4] = cx * cz - sx * sy * sz;  R5[5] = -sx * cy;
    R5[6] = sx * sz - cx * sy * cz;  R5[7] = cx * sy * sz + sx * cz;  R5[8] = cx * cy;
    P[0] = xr[0] + CT[TB::SC + TB::BASE_T];
    P[1] = xr[1] + CT[TB::SC + TB::BASE_T + 1];
    P[2] = xr[2] + CT[TB::SC + TB::BASE_T + 2];
    A4[0] = T(0); A4[1] = cx; A4[2] = sx;
    A5[0] = sy; A5[1] = -sx * cy; A5[2] = cx * cy;
    pA3[0] = T(0); pA3[1] = P[2]; pA3[2] = -P[1];  // P x e_x
    cross3(P, A4, pA4);
    cross3(P, A5, pA5);
  };
  auto times_inertia = [&](const T *I, const T *s, T *F) {  // (I w + h x v, m v - h x w)
    T t3[3];
    sym3_mulv(I, s, F);
    cross3(I + 6, s + 3, t3);
    F[0] += t3[0];
    F[1] += t3[1];
    F[2] += t3[2];
    cross3(I + 6, s, t3);
    F[3] = I[9] * s[3] - t3[0];
    F[4] = I[9] * s[4] - t3[1];
    F[5] = I[9] * s[5] - t3[2];
  };
  auto dot6 = [&](const T *a, const T *b) -> T { return dot3(a, b) + dot3(a + 3, b + 3); };

  auto main_kin = [&]() {
    // ================================ main: PD, jcalc, kinematics ================================
    OCT_MARK("main_top");
    q = xr[dq];
    qd = xr[nq + dq];
    // ---- PD controller (locomotion_contact_simulation.h:168-258); joint stiffness / damping
    tau = T(0);
    {
      const int act_i = act_slot;
      if (act_i >= 0) {
        const int var = nq + nd + adim;
        const T kp = xr[var], kd = xr[var + 1], max_force = xr[var + 2];
        const T act_lim = CT[TB::SC + TB::ACTION_LIMIT];
        T a = xr[nq + nd + act_i];
        a = a < act_lim ? a : act_lim;
        a = a > -act_lim ? a : -act_lim;
        const T q_des = CL[TB::IPOSE] + a;
        T f = kp * (q_des - q) + kd * (T(0) - qd);
        f = f > -max_force ? f : -max_force;
        f = f < max_force ? f : max_force;
        tau = f;
      }
      tau -= CL[TB::STIFF] * q + CL[TB::DAMP] * qd;
      // (tau is next read by the forward dynamics, two barriers on: left to itself the back end moves this line down there and
      //  keeps stiffness and damping alive for it — in the 256-register build through scratch, a memory round trip the main
      //  wavefront waited for behind barrier (2))
      asm volatile("" : "+v"(tau));
    }
    OCT_MARK("main_jcalc");
    // ---- B. jcalc (link.hpp:229-287)
#pragma unroll
    for (int k = 0; k < 6; ++k) Sl[k] = CL[TB::S + k];
    T Rp[9], tp[3];
    {
      // the joint transform: R_J = cos I + sin [n]x + (1 - cos) n n^T about the joint's unit axis (table: n, n n^T; a prismatic
      // joint's angle is multiplied by 0), t_J = S_linear q (zero for a revolute joint) — every joint type of link.hpp:229-287
      // without a branch; X_parent = X_T X_J, and where every X_T rotation is the identity (the Ant) no product at all
      T sn, cs;
      oct_sincos(q * CL[TB::ROTF], &sn, &cs);
      const T c1 = T(1) - cs;
      const T nx = CL[TB::NAX], ny = CL[TB::NAX + 1], nz = CL[TB::NAX + 2];
      T RJ[9];
      RJ[0] = cs + c1 * CL[TB::NN + 0];
      RJ[1] = c1 * CL[TB::NN + 1] - sn * nz;
      RJ[2] = c1 * CL[TB::NN + 2] + sn * ny;
      RJ[3] = c1 * CL[TB::NN + 1] + sn * nz;
      RJ[4] = cs + c1 * CL[TB::NN + 3];
      RJ[5] = c1 * CL[TB::NN + 4] - sn * nx;
      RJ[6] = c1 * CL[TB::NN + 2] - sn * ny;
      RJ[7] = c1 * CL[TB::NN + 4] + sn * nx;
      RJ[8] = cs + c1 * CL[TB::NN + 5];
      const T tJ[3] = {Sl[3] * q, Sl[4] * q, Sl[5] * q};
      if (CT[TB::SC + TB::XT_IDENT] != T(0)) {  // wave-uniform
#pragma unroll
        for (int k = 0; k < 9; ++k) Rp[k] = RJ[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) tp[k] = CL[TB::XT + 9 + k] + tJ[k];
      } else {
        T RT[9], r[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) RT[k] = CL[TB::XT + k];
        mat3_mul(RT, RJ, Rp);
        mat3_mulv(RT, tJ, r);
#pragma unroll
        for (int k = 0; k < 3; ++k) tp[k] = CL[TB::XT + 9 + k] + r[k];
      }
    }
    OCT_MARK("main_rootsincos");
    // ---- C. the root chain in closed form, on every lane.  The three root angles' sines and cosines: lanes 0, 1, 2 of the
    //         environment, broadcast (and, two-wavefront build, handed to the helper behind the links' world transforms)
    {
      T rs, rc;
      oct_sincos(xr[3 + (lane < 3 ? lane : 0)], &rs, &rc);
      const T sx = oct_bcast<0>(rs), cx = oct_bcast<0>(rc);
      const T sy = oct_bcast<1>(rs), cy = oct_bcast<1>(rc);
      const T sz = oct_bcast<2>(rs), cz = oct_bcast<2>(rc);
      if constexpr (W2) {
        if (lane < 3) {
          T *const sc = E + O.win + 8 * OctLds::ZW + 96;
          sc[2 * lane] = rs;
          sc[2 * lane + 1] = rc;
        }
      }
      root_frame(sx, cx, sy, cy, sz, cz);
    }
    OCT_MARK("main_rootvel");
    {
      const T d0 = xr[nq + 0], d1 = xr[nq + 1], d2 = xr[nq + 2], d3 = xr[nq + 3], d4 = xr[nq + 4], d5 = xr[nq + 5];
      const T U[3] = {d0, d1, d2};
      const T J3[3] = {d3, T(0), T(0)}, J4[3] = {A4[0] * d4, A4[1] * d4, A4[2] * d4}, J5[3] = {A5[0] * d5, A5[1] * d5, A5[2] * d5};
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
      T a45[3], a55[3], t1[3], t2[3], l3[3], l4[3], l5[3];
      cross3(J3, J4, a45);
      cross3(W4, J5, a55);
      cross3(J3, pJ3, t1);
      cross3(V3, J3, t2);
      l3[0] = t1[0] + t2[0]; l3[1] = t1[1] + t2[1]; l3[2] = t1[2] + t2[2];
      cross3(W4, pJ4, t1);
      cross3(V4, J4, t2);
      l4[0] = t1[0] + t2[0]; l4[1] = t1[1] + t2[1]; l4[2] = t1[2] + t2[2];
      cross3(W5, pJ5, t1);
      cross3(V5, J5, t2);
      l5[0] = t1[0] + t2[0]; l5[1] = t1[1] + t2[1]; l5[2] = t1[2] + t2[2];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        v5[k] = W5[k];
        v5[3 + k] = V5[k];
        a5[k] = a45[k] + a55[k];
        a5[3 + k] = (l3[k] + l4[k] + l5[k]) - CT[TB::SC + TB::GRAV + k];
      }
    }
    OCT_MARK("main_legs");
    // ---- the legs: the ankle composes its joint transform with the hip's (one step of a segmented scan along the pair),
    //      the root's pose in front; prefix sums of the joint velocities and of the velocity-product accelerations
    {
      const bool take = pos == 1;
      T Rq[9], pq[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const T sh = pair_bcast<0>(Rp[k]);
        Rq[k] = take ? sh : ((k == 0 || k == 4 || k == 8) ? T(1) : T(0));
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const T sh = pair_bcast<0>(tp[k]);
        pq[k] = take ? sh : T(0);
      }
      T Rl[9], pl[3], r[3];
      mat3_mul(Rq, Rp, Rl);
      mat3_mulv(Rq, tp, r);
#pragma unroll
      for (int k = 0; k < 3; ++k) pl[k] = pq[k] + r[k];
      mat3_mul(R5, Rl, R);
      mat3_mulv(R5, pl, r);
      p[0] = P[0] + r[0];
      p[1] = P[1] + r[1];
      p[2] = P[2] + r[2];
      // s = X_world.apply_inverse(S) = (R w, R v + p x (R w))   (transform.hpp:232-243)
      mat3_mulv(R, Sl, sw);
      mat3_mulv(R, Sl + 3, sw + 3);
      T c[3];
      cross3(p, sw, c);
      sw[3] += c[0];
      sw[4] += c[1];
      sw[5] += c[2];
      T vJ[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) vJ[k] = sw[k] * qd;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const T sh = pair_bcast<0>(vJ[k]);
        v[k] = v5[k] + (vJ[k] + (take ? sh : T(0)));
      }
      // cb = v x vJ (kinematics.hpp:96-99)
      T cb[6];
      cross3(v, vJ, cb);
      T c1[3], c2[3];
      cross3(v, vJ + 3, c1);
      cross3(v + 3, vJ, c2);
      cb[3] = c1[0] + c2[0];
      cb[4] = c1[1] + c2[1];
      cb[5] = c1[2] + c2[2];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const T sh = pair_bcast<0>(cb[k]);
        a0[k] = a5[k] + (cb[k] + (take ? sh : T(0)));
      }
    }
    OCT_MARK("main_publish_kin");
    // my world motion axis, for the rows of the contacts (lane-dependent reads in the row windows); two-wavefront build: my
    // link's world transform for the helper's narrowphase and visual poses (the slots of the second row window)
    {
      T *const swl = E + O.swl + lane * 6;
#pragma unroll
      for (int k = 0; k < 6; ++k) swl[k] = sw[k];
      if constexpr (W2) {
        T *const kin = E + O.win + 8 * OctLds::ZW + lane * 12;
#pragma unroll
        for (int k = 0; k < 9; ++k) kin[k] = R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) kin[9 + k] = p[k];
      }
    }
    OCT_MARK("main_kin_end");
    OCT_STAMP(1, sw[5]);
  };

  // where this step's y record goes: the slot of a y ring (every step of a step-loop launch), else the handle's y record
  // (the last step); the last step of a ring launch leaves its record in the handle's y record as well
  const int ystr = ctl.y_stride;
  const int out_dim = (int)CT[TB::SC + TB::OUTPUT_DIM];
  const int nv = (int)CT[TB::SC + TB::NUM_VISUALS];
  // (worked out where a phase stores — three places — from a ring position laundered there: computed once at the top of the
  //  step the two pointers lived across it, in the 256-register build through scratch, reloaded in front of the helper's stores)
  auto y_where = [&](TR *&yo, TR *&yo2, int &yend, int &yend2) {
    int ys = y_slot;
    asm volatile("" : "+s"(ys));
    yo = nullptr;
    yo2 = nullptr;
    yend = ystr;
    yend2 = out_dim;
    if (LOOP && ctl.y_ring != nullptr) {
      yo = oct_global((TR *)ctl.y_ring) + ((size_t)ys * ctl.ring_envs + env) * ystr;
      if (last && y_out != nullptr) yo2 = y_out + (size_t)env * out_dim;
    } else if (last && y_out != nullptr) {
      yo = y_out + (size_t)env * (LOOP ? out_dim : ystr);
      yend = LOOP ? out_dim : ystr;
    }
  };
  // ---- exchange launches of the multi-GPU layer (tds_shard.hip): the records of a step are counted in on the slot's progress
  //      counter (RCCL forms), or on its arrival counters with the flags of every rank raised by the workgroup that
  //      completes the slot (peer-store exchange; see tds_kernels.hip: peer_signal)
  auto signal_slot = [&](int pslot) {
    if (ctl.peer_arrive != nullptr) {
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): every store of this wavefront acknowledged by the memory it went to
      const bool rel = (ctl.ring_flags & TDS_RING_PEER_RELEASE) != 0;  // (A/B switch for the first run on a fabric: tds_kernels.h)
      if (rel) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      if ((tid & 63) == 0) {
        constexpr unsigned SUB = TDS_PEER_SUB;
        const unsigned g = gridDim.x, j = blockIdx.x % SUB;
        const unsigned n1 = (g - j + SUB - 1u) / SUB;  // workgroups that count on first-level counter j
        const unsigned n2 = g < SUB ? g : SUB;          // first-level counters in use
        unsigned *const base = oct_global(ctl.peer_arrive) + (size_t)pslot * TDS_PEER_ARRIVE_STRIDE;
        if (atomicInc(base + j * TDS_PEER_LINE, n1 - 1u) == n1 - 1u) {
          if (atomicInc(base + 32 * TDS_PEER_LINE, n2 - 1u) == n2 - 1u) {
            const size_t fi = (size_t)ctl.peer_flag_off + (size_t)pslot * (size_t)ctl.peer_flag_stride;
            if (rel) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "");
            for (int pr = 0; pr <= ctl.n_peers; ++pr)
              __hip_atomic_store(ctl.peer_flags[pr] + fi, ctl.peer_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
      }
    } else if (ctl.progress != nullptr) {
      if (ctl.ring_flags & TDS_RING_NOFENCE) __builtin_amdgcn_s_waitcnt(0x0f70);  // (write-through record stores)
      else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if ((tid & 63) == 0) __hip_atomic_fetch_add(oct_global(ctl.progress) + pslot, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  // The peer form's count in two halves: the first-level atomic (with return) is ISSUED behind barrier (1) — the records of
  // step it - 1 went out a kinematics phase ago — and its result is looked at in front of the visual poses (help_poses): as
  // one piece there, the helper waited for the atomic's round trip to the L2 between barriers (1b) and (2), where the main
  // wavefront waits for the helper
  unsigned sig_tok = 0u;
  auto signal_issue = [&](int pslot) {
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if ((ctl.ring_flags & TDS_RING_PEER_RELEASE) != 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if ((tid & 63) == 0) {
      constexpr unsigned SUB = TDS_PEER_SUB;
      const unsigned g = gridDim.x, j = blockIdx.x % SUB;
      const unsigned n1 = (g - j + SUB - 1u) / SUB;
      unsigned *const base = oct_global(ctl.peer_arrive) + (size_t)pslot * TDS_PEER_ARRIVE_STRIDE;
      sig_tok = atomicInc(base + j * TDS_PEER_LINE, n1 - 1u);
    }
  };
  auto signal_finish = [&](int pslot) {
    if ((tid & 63) == 0) {
      constexpr unsigned SUB = TDS_PEER_SUB;
      const unsigned g = gridDim.x, j = blockIdx.x % SUB;
      const unsigned n1 = (g - j + SUB - 1u) / SUB;
      const unsigned n2 = g < SUB ? g : SUB;
      unsigned *const base = oct_global(ctl.peer_arrive) + (size_t)pslot * TDS_PEER_ARRIVE_STRIDE;
      if (sig_tok == n1 - 1u) {
        if (atomicInc(base + 32 * TDS_PEER_LINE, n2 - 1u) == n2 - 1u) {
          const size_t fi = (size_t)ctl.peer_flag_off + (size_t)pslot * (size_t)ctl.peer_flag_stride;
          if ((ctl.ring_flags & TDS_RING_PEER_RELEASE) != 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "");
          for (int pr = 0; pr <= ctl.n_peers; ++pr)
            __hip_atomic_store(ctl.peer_flags[pr] + fi, ctl.peer_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  };

  auto help_np = [&]() {
    OCT_MARK("help_np");
    // ================================ helper: narrowphase, visual poses ================================
    if constexpr (LOOP) {
      if (it > 0 && ctl.obs_ring != nullptr && ctl.peer_arrive != nullptr)  // (wave-uniform; in front of the action request: the wait)
        signal_issue((o_slot == 0 ? ctl.obs_slots : o_slot) - 1);
      // The NEXT step's action block is requested here, by the helper (a different block per step: tds_hip_step_many), and goes
      // into the record's action slots in front of the visual poses' stores (help_poses) — the slots are dead since the PD
      // block, in front of barrier (1); the next PD block is behind barrier (0).  Requested by the main wavefront and held
      // until the integration, the value crossed the whole step in a register: the 256-register build spilled it to scratch
      // at the top of the step — a wait for the load itself — and reloaded it on the main wavefront's path
      if (ctl.act_pool != nullptr && it + 1 < nsteps && valid)  // (wave-uniform but for `valid`)
        next_act = (T)oct_global((const TR *)ctl.act_pool)[((size_t)act_blk * ctl.act_envs + env) * adim + lane];
    }
    if constexpr (W2) {  // my link's world transform and the root's sines / cosines, from the main wavefront
      const T *const kin = E + O.win + 8 * OctLds::ZW + lane * 12;
#pragma unroll
      for (int k = 0; k < 9; ++k) R[k] = kin[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) p[k] = kin[9 + k];
      const T *const sc = E + O.win + 8 * OctLds::ZW + 96;
      root_frame(sc[0], sc[1], sc[2], sc[3], sc[4], sc[5]);
    }
    // ---- I. narrowphase (contact_point.hpp:96-161): my link's capsule = two spheres; the root body's sphere on every lane.
    //         Penetrating points in the reference's order (root, then per link +L/2 end, -L/2 end) into the contact list
    {
      T *const cpx = E + O.cp;
      const T n[3] = {CT[TB::SC + TB::PLANE_N], CT[TB::SC + TB::PLANE_N + 1], CT[TB::SC + TB::PLANE_N + 2]};
      const T pc = CT[TB::SC + TB::PLANE_C];
      auto sphere = [&](const T *Rl, const T *pl, const T *loc, T rad, T *pt, T &dist) {
        T ctr[3];
        mat3_mulv(Rl, loc, ctr);
        ctr[0] += pl[0];
        ctr[1] += pl[1];
        ctr[2] += pl[2];
        const T t = -((-dot3(ctr, n)) + pc);
        dist = t - rad;
        pt[0] = ctr[0] - rad * n[0];
        pt[1] = ctr[1] - rad * n[1];
        pt[2] = ctr[2] - rad * n[2];
      };
      T pt0[3], pt1[3], ptt[3], d0, d1, dtt;
      sphere(R, p, CL + TB::CPL0, CL[TB::CPR0], pt0, d0);
      sphere(R, p, CL + TB::CPL1, CL[TB::CPR1], pt1, d1);
      sphere(R5, P, CT + TB::ROOT + TB::R_CPL, CT[TB::ROOT + TB::R_CPR], ptt, dtt);
      const bool act0 = valid && d0 < T(0), act1 = valid && d1 < T(0), actt = valid && dtt < T(0) && CT[TB::ROOT + TB::R_CPR] >= T(0);
      const unsigned long long b0 = __ballot(act0), b1 = __ballot(act1), bt = __ballot(actt);
      const unsigned m0 = (unsigned)((b0 >> (grp * 8)) & 0xFFull), m1 = (unsigned)((b1 >> (grp * 8)) & 0xFFull);
      const unsigned below = (1u << lane) - 1u;
      const int r0 = (actt ? 1 : 0) + __popc(m0 & below) + __popc(m1 & below);
      const int r1 = r0 + (act0 ? 1 : 0);
      if (act0) {
        cpx[0 * OctLds::NCP + r0] = pt0[0];
        cpx[1 * OctLds::NCP + r0] = pt0[1];
        cpx[2 * OctLds::NCP + r0] = pt0[2];
        cpx[3 * OctLds::NCP + r0] = d0;
        cpx[4 * OctLds::NCP + r0] = (T)lane;
      }
      if (act1) {
        cpx[0 * OctLds::NCP + r1] = pt1[0];
        cpx[1 * OctLds::NCP + r1] = pt1[1];
        cpx[2 * OctLds::NCP + r1] = pt1[2];
        cpx[3 * OctLds::NCP + r1] = d1;
        cpx[4 * OctLds::NCP + r1] = (T)lane;
      }
      if (actt && lane == 0) {
        cpx[0 * OctLds::NCP] = ptt[0];
        cpx[1 * OctLds::NCP] = ptt[1];
        cpx[2 * OctLds::NCP] = ptt[2];
        cpx[3 * OctLds::NCP] = dtt;
        cpx[4 * OctLds::NCP] = T(8);
      }
      na = (actt ? 1 : 0) + __popc(m0) + __popc(m1);
      // the largest count among the wavefront's environments (scalar arithmetic on the three ballots)
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int c = __popc((unsigned)((b0 >> (g * 8)) & 0xFFull)) + __popc((unsigned)((b1 >> (g * 8)) & 0xFFull)) + (int)((bt >> (g * 8)) & 1ull);
        NA = c > NA ? c : NA;
      }
      if constexpr (W2) {  // (for the main wavefront's sweep: the spare slot of environment 0's record)
        if ((tid & 63) == 0) sm[in_dim + 3] = (T)NA;
      }
    }
    OCT_STAMP(8, na);
  };
  auto help_poses = [&]() {
    OCT_MARK("help_signal_poses");
    if constexpr (LOOP) {
      // (loads and stores return through one in-order counter on gfx9: this wait sees the helper's record stores of the step
      //  before, issued a narrowphase and a row stage ago, and nothing of this step)
      if (ctl.act_pool != nullptr && !last) xr[nq + nd + lane] = next_act;
      // the records of step it - 1 — stored at the end of the iteration before, long acknowledged by now: the wait costs
      // nothing here, in front of this step's first stores — are counted in
      if (it > 0 && ctl.obs_ring != nullptr) {  // (wave-uniform)
        if (ctl.peer_arrive != nullptr) signal_finish((o_slot == 0 ? ctl.obs_slots : o_slot) - 1);
        else signal_slot((o_slot == 0 ? ctl.obs_slots : o_slot) - 1);
      }
    }
    // ---- M1. visual poses of y, from the PRE-step X_world (locomotion_contact_simulation.h:281-299): visual 1 + lane is
    //          my link's (DevModel::oct checks the order); visual 0 — the root body's — goes out on lane 7
    TR *yo, *yo2;
    int yend, yend2;
    y_where(yo, yo2, yend, yend2);
    (void)yend;
    (void)yend2;
    if (valid && yo != nullptr && nv > 0) {
      auto pose_out = [&](const T *Rl, const T *pl, const T *vx, int k) {
        T Ro[9], po[3], qo[4];
        mat3_mul(Rl, vx, Ro);
        mat3_mulv(Rl, vx + 9, po);
        matrix_to_quat(Ro, qo);
        TR *o = yo + (nq + nd) + 7 * k;
        OCT_ST((TR)(pl[0] + po[0]), &o[0]);
        OCT_ST((TR)(pl[1] + po[1]), &o[1]);
        OCT_ST((TR)(pl[2] + po[2]), &o[2]);
        OCT_ST((TR)qo[0], &o[3]);
        OCT_ST((TR)qo[1], &o[4]);
        OCT_ST((TR)qo[2], &o[5]);
        OCT_ST((TR)qo[3], &o[6]);
        if (yo2 != nullptr) {
          TR *o2 = yo2 + (nq + nd) + 7 * k;
          OCT_ST((TR)(pl[0] + po[0]), &o2[0]);
          OCT_ST((TR)(pl[1] + po[1]), &o2[1]);
          OCT_ST((TR)(pl[2] + po[2]), &o2[2]);
          OCT_ST((TR)qo[0], &o2[3]);
          OCT_ST((TR)qo[1], &o2[4]);
          OCT_ST((TR)qo[2], &o2[5]);
          OCT_ST((TR)qo[3], &o2[6]);
        }
      };
      pose_out(R, p, CL + TB::VIS, 1 + lane);
      if (lane == 7) pose_out(R5, P, CT + TB::ROOT + TB::R_VIS, 0);
    }
    OCT_MARK("help_np_end");
    OCT_STAMP(9, na);
  };

  auto main_dyn = [&]() {
    // ================================ main: inertias, LDL^T, forward dynamics ================================
    // ---- D. world-frame rigid inertia and bias force of my link, and (redundantly on every lane) of the root body
    //         (kinematics.hpp:96-132, inertia.hpp:121-130): I = (Isym 6 | h 3 | m), f = I a0 + v x* I v
    auto rigid = [&](const T *Rl, const T *pl, T m, const T *com, const T *Ib, const T *vl, const T *al, T *Ic, T *fc) {
      T cw[3];
      mat3_mulv(Rl, com, cw);
      cw[0] += pl[0];
      cw[1] += pl[1];
      cw[2] += pl[2];
      T RI[9], Iw[9];
      mat3_mul(Rl, Ib, RI);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Iw[3 * r + c] = RI[3 * r] * Rl[3 * c] + RI[3 * r + 1] * Rl[3 * c + 1] + RI[3 * r + 2] * Rl[3 * c + 2];
      const T c2 = dot3(cw, cw);
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
      T Iv[6], Ia[6], t3[3];
      sym3_mulv(Ic, vl, Iv);
      cross3(h, vl + 3, t3);
      Iv[0] += t3[0];
      Iv[1] += t3[1];
      Iv[2] += t3[2];
      cross3(h, vl, t3);
      Iv[3] = m * vl[3] - t3[0];
      Iv[4] = m * vl[4] - t3[1];
      Iv[5] = m * vl[5] - t3[2];
      sym3_mulv(Ic, al, Ia);
      cross3(h, al + 3, t3);
      Ia[0] += t3[0];
      Ia[1] += t3[1];
      Ia[2] += t3[2];
      cross3(h, al, t3);
      Ia[3] = m * al[3] - t3[0];
      Ia[4] = m * al[4] - t3[1];
      Ia[5] = m * al[5] - t3[2];
      T u3[3];
      cross3(vl, Iv, fc);
      cross3(vl + 3, Iv + 3, u3);
      fc[0] += u3[0];
      fc[1] += u3[1];
      fc[2] += u3[2];
      cross3(vl, Iv + 3, fc + 3);
#pragma unroll
      for (int k = 0; k < 6; ++k) fc[k] += Ia[k];
    };
    OCT_MARK("main_rigid");
    T Ic[10], fc[6];
    rigid(R, p, CL[TB::MASS], CL + TB::COM, CL + TB::INER, v, a0, Ic, fc);
    T ft[6];  // (It, ft) the root body's; below: + the legs' = the whole robot's
    rigid(R5, P, CT[TB::ROOT + TB::R_MASS], CT + TB::ROOT + TB::R_COM, CT + TB::ROOT + TB::R_INER, v5, a5, It, ft);
    OCT_STAMP(2, ft[5]);
    OCT_MARK("main_totals");
    // ---- E. composite inertia / bias force (CRBA, mass_matrix.hpp:39-56): the robot's totals = root + every leg link
    //         (8-lane sums of the rigid values: every lane the same bits); the hip's composite = hip + ankle
#pragma unroll
    for (int k = 0; k < 10; ++k) It[k] += oct_sum(Ic[k]);
#pragma unroll
    for (int k = 0; k < 6; ++k) ft[k] += oct_sum(fc[k]);
    {
      const T recv = pos == 0 ? T(1) : T(0);
#pragma unroll
      for (int k = 0; k < 6; ++k) fc[k] += recv * pair_bcast<1>(fc[k]);
#pragma unroll
      for (int k = 0; k < 10; ++k) Ic[k] += recv * pair_bcast<1>(Ic[k]);
    }
    OCT_MARK("main_FC");
    // F = Ic s, C = s . f of my dof
    T Fc[6];
    times_inertia(Ic, sw, Fc);
    Cb = dot6(sw, fc);
    // the root's revolute axes as (angular | linear); the prismatic ones are (0 | e_k)
    const T ax3[6] = {T(1), T(0), T(0), pA3[0], pA3[1], pA3[2]}, ax4[6] = {A4[0], A4[1], A4[2], pA4[0], pA4[1], pA4[2]},
            ax5[6] = {A5[0], A5[1], A5[2], pA5[0], pA5[1], pA5[2]};
    Cr[0] = ft[3];  // bias forces of the root dofs
    Cr[1] = ft[4];
    Cr[2] = ft[5];
    Cr[3] = dot6(ax3, ft);
    Cr[4] = dot6(ax4, ft);
    Cr[5] = dot6(ax5, ft);
    // ---- G. M in leaves-first order.  My row of my leg's 2 x 2 block (M[i][j] = F_i . s_j for j an ancestor of i or i,
    //         mass_matrix.hpp:87-109) and my coupling to the root dofs
    T Bm0, Bm1, Cc[6];
    {
      T s0[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) s0[k] = pair_bcast<0>(sw[k]);
      Bm0 = dot6(Fc, s0);   // hip lane: M_hh; ankle lane: M_ah
      Bm1 = dot6(Fc, sw);   // ankle lane: M_aa
      Cc[0] = Fc[3];
      Cc[1] = Fc[4];
      Cc[2] = Fc[5];
      Cc[3] = dot6(Fc, ax3);
      Cc[4] = dot6(Fc, ax4);
      Cc[5] = dot6(Fc, ax5);
    }
    OCT_MARK("main_legldl");
    // ---- H. LDL^T.  The leg block on both lanes of the pair
    {
      const T b00 = pair_bcast<0>(Bm0);
      const T b10 = pair_bcast<1>(Bm0), b11 = pair_bcast<1>(Bm1);
      // (1 / sqrt(d) by the hardware estimate + two Newton steps, 1 / d as its square: what a division AND a square root
      //  per pivot cost — ~25 instructions — for 8)
      sq0 = rsqrt_full<T>(b00);
      id0 = sq0 * sq0;
      l10 = b10 * id0;
      sq1 = rsqrt_full<T>(b11 - l10 * b10);
      id1 = sq1 * sq1;
    }
    my_id = pos == 0 ? id0 : id1;
    my_sq = pos == 0 ? sq0 : sq1;
    // the coupling rows: W_hip = C_hip, W_ankle = C_ankle - l10 W_hip;  L_c = W / d
    {
      T W[6];
      const T m1 = pos == 1 ? l10 : T(0);
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        W[r] = Cc[r] - m1 * pair_bcast<0>(Cc[r]);
        Lc[r] = W[r] * my_id;
      }
      T *const lcw = E + O.lcw + lane * OctLds::LCW;
      T *const wl = E + O.xs + lane * 6;  // (the impulses' slots: free until the sweep)
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        lcw[r] = Lc[r];
        wl[r] = W[r];
      }
      if (pos == 0) {
        T *const lf = E + O.legf + leg * 3;
        lf[0] = l10;
        lf[1] = sq0;
        lf[2] = sq1;
      }
    }
    OCT_STAMP(3, Lc[5]);
  };
  auto main_dyn2 = [&]() {
    OCT_MARK("main_schur");
    // the sums sum_lanes L_c[r] W[r'] of the Schur complement, entry e = r (r + 1) / 2 + r' on lane e mod 8 (three passes)
    {
      const T *const lcw = E + O.lcw;
      const T *const wl = E + O.xs;
      T *const Ssum = E + O.fac;
      const int code = (int)CL[TB::SCHUR];  // (which entries this lane sums: table)
#pragma unroll
      for (int pass = 0; pass < 3; ++pass) {
        const int rc = (code >> (6 * pass)) & 63;
        const int r = rc >> 3, rp = rc & 7;
        const int e = (r * (r + 1)) / 2 + rp;
        T acc = T(0);
#pragma unroll
        for (int l = 0; l < 8; ++l) acc += lcw[l * OctLds::LCW + r] * wl[l * 6 + rp];
        Ssum[e] = acc;
      }
    }
    OCT_MARK("main_Rblock");
    const T ax3[6] = {T(1), T(0), T(0), pA3[0], pA3[1], pA3[2]}, ax4[6] = {A4[0], A4[1], A4[2], pA4[0], pA4[1], pA4[2]},
            ax5[6] = {A5[0], A5[1], A5[2], pA5[0], pA5[1], pA5[2]};
    // the root block R[r][r'] = s_r . (It s_r') in registers (packed lower triangle, r (r + 1) / 2 + r'): the prismatic
    // axes are unit vectors, It e_k = (h x e_k | m e_k)
    T Sm[21];
    {
      const T m = It[9];
      T F3[6], F4[6], F5[6];
      times_inertia(It, ax3, F3);
      times_inertia(It, ax4, F4);
      times_inertia(It, ax5, F5);
      Sm[0] = m;
      Sm[1] = T(0); Sm[2] = m;
      Sm[3] = T(0); Sm[4] = T(0); Sm[5] = m;
      Sm[6] = F3[3]; Sm[7] = F3[4]; Sm[8] = F3[5]; Sm[9] = dot6(ax3, F3);
      Sm[10] = F4[3]; Sm[11] = F4[4]; Sm[12] = F4[5]; Sm[13] = dot6(ax4, F3); Sm[14] = dot6(ax4, F4);
      Sm[15] = F5[3]; Sm[16] = F5[4]; Sm[17] = F5[5]; Sm[18] = dot6(ax5, F3); Sm[19] = dot6(ax5, F4); Sm[20] = dot6(ax5, F5);
    }
    OCT_SYNC();
    OCT_MARK("main_ldl6");
    // ... the Schur complement, factorised redundantly on every lane: Ls (strictly lower, row-major packed), 1 / D
    {
      const T *const Ssum = E + O.fac;
#pragma unroll
      for (int e = 0; e < 21; ++e) Sm[e] -= Ssum[e];
      static_for<0, 6>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const T rs = rsqrt_full<T>(Sm[(k * (k + 1)) / 2 + k]);
        const T inv = rs * rs;
        sq_ids[k] = rs;
        ids[k] = inv;
        T col[6];
        static_for<k + 1, 6>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          col[r] = Sm[(r * (r + 1)) / 2 + k];           // S(r, k) before scaling
          Ls[(r * (r - 1)) / 2 + k] = col[r] * inv;     // L(r, k)
        });
        static_for<k + 1, 6>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          static_for<k + 1, r + 1>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            Sm[(r * (r + 1)) / 2 + c] -= Ls[(r * (r - 1)) / 2 + k] * col[c];
          });
        });
      });
    }
    OCT_STAMP(4, Ls[14]);
    OCT_MARK("main_fd");
    // two-wavefront build: the root block's factors for the helper's rows
    if constexpr (W2) {
      T *const fac = E + O.fac;  // (the Schur sums are dead: every lane has read them)
      OCT_SYNC();
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 15; ++k) fac[k] = Ls[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) fac[15 + k] = sq_ids[k];
      }
    }
    OCT_MARK("main_dyn_end");
  };
  // ---- F. forward dynamics, the forward half (forward_dynamics.hpp:11-326 as qdd = M^-1 (tau - C); integrator.hpp:169-181):
  //      y~ = D^-1/2 L^-1 (tau - C) in the leaves-first system [B C^T; C R] = L D L^T — the leg part on the dof lanes, the root
  //      part on every lane.  The backward half is shared with the contact impulse: the constraint rows' right-hand sides
  //      are formed with the velocities BEFORE the step (rows_geom), and the sweep starts from u~ = -dt y~ instead of 0 —
  //      z~_r . (dt y~) is exactly the J_r (dt qdd) the reference's b_r contains — so that ONE back-substitution at the end
  //      gives  qd_new = qd - L^-T D^-1/2 u~  = qd + dt qdd - M^-1 J^T p  (with no contact: the plain forward dynamics).
  auto main_fd = [&]() {
    OCT_MARK("main_fd");
    T y = tau - Cb;
    y -= (pos == 1 ? l10 : T(0)) * pair_bcast<0>(y);
    T yr[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) yr[r] = -Cr[r] - oct_sum(Lc[r] * y);
    static_for<1, 6>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      static_for<0, r>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        yr[r] -= Ls[(r * (r - 1)) / 2 + c] * yr[c];
      });
    });
    u = -dt * (y * my_sq);
    T sel = yr[0] * sq_ids[0];
#pragma unroll
    for (int r = 1; r < 6; ++r) sel = lane == r ? yr[r] * sq_ids[r] : sel;
    urm = lane < 6 ? -dt * sel : T(0);
    u_init = u;
    urm_init = urm;
    OCT_STAMP(5, u);
  };
  // (two-wavefront build, behind barrier (2)) the largest contact count for the main wavefront, the root block's factors for the helper
  auto main_get_count = [&]() { NA = __builtin_amdgcn_readfirstlane((int)sm[in_dim + 3]); };
  auto help_get_factors = [&]() {
    const T *const fac = E + O.fac;
#pragma unroll
    for (int k = 0; k < 15; ++k) Ls[k] = fac[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) sq_ids[k] = fac[15 + k];
  };

  // ---- J, K, L. contacts (wave-uniform: NA = the largest count among the wavefront's environments).  The sweep's order is
  //      the reference's: rows 0 .. NA-1 normals, NA .. 2NA-1 tangents 1, 2NA .. 3NA-1 tangents 2, each by contact (an
  //      environment with fewer contacts has zero rows in its empty slots).  Rows are solved a WINDOW of eight sweep
  //      positions at a time — lane == row, by the helper — and consumed by the main wavefront's sweep; two window buffers:
  //      the helper solves window w + 1 while the main wavefront sweeps window w.
  // window wi of the sweep: Gauss-Seidel iteration wi / nwin, sweep positions w0 .. w0 + 7, row buffer wi & 1; lane == row.
  // Three stages, each as early as its inputs exist (two-wavefront build: all by the helper, beside the main wavefront's chain):
  //   rows_geom  the contact's Jacobian row along the row's direction and the right-hand side c_r from the velocities BEFORE
  //              the step (the kinematics only);
  //   rows_leg   the contact's leg block of z~ = D^-1/2 L^-1 J^T (the leg factors: published at barrier (1b));
  //   rows_root  the root block, G and 1 / (G + cfm) (the root block's factors: barrier (2)).
  auto rows_geom = [&](int wi) {
      OCT_MARK("rows_geom");
      const int nr = 3 * NA, nwin = (nr + 7) >> 3;
      const int w0 = (wi % nwin) * 8;
      T *const row = E + O.win + (wi & 1) * (8 * OctLds::ZW) + lane * OctLds::ZW;
      const T *const cpx = E + O.cp;
      const T *const swl = E + O.swl;
      const int s = w0 + lane;
      const int tk = (s >= NA ? 1 : 0) + (s >= 2 * NA ? 1 : 0);
      const int a = s - tk * NA;
      const bool real = s < nr && a < na;
      const int ac = real ? a : 0;
      // (first batch of LDS reads: the contact and the row's direction — n, t1, t2 lie one behind the other in the table —;
      //  every read unconditional, what a padding row reads is masked out at the end: no branch, no wait inside the batch)
      const T Pc[3] = {cpx[0 * OctLds::NCP + ac], cpx[1 * OctLds::NCP + ac], cpx[2 * OctLds::NCP + ac]};
      const T dist = cpx[3 * OctLds::NCP + ac];
      const T owner = cpx[4 * OctLds::NCP + ac];
      const T *const dir = CT + TB::SC + TB::NB + 3 * tk;
      const T e[3] = {dir[0], dir[1], dir[2]};
      const int ol = real ? (int)owner : 8;    // owner lane; 8: the root body
      const bool on_leg = ol < 8;
      const int hl = on_leg ? (ol & ~1) : 0;   // the hip lane of the contact's leg
      const bool ank = on_leg && (ol & 1);     // the contact sits on the ankle link: both dofs of the leg
      T sh[6], sa[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        sh[c] = swl[hl * 6 + c];
        sa[c] = swl[(hl + 1) * 6 + c];
      }
      const T qh = xr[nq + 6 + hl], qa = xr[nq + 6 + hl + 1];  // velocities before the step
      // column of the point Jacobian along e: e . s_lin + P . (e x s_ang) = e . s_lin + (P x e) . s_ang  (jacobian.hpp:56-72)
      T mo[3];
      cross3(Pc, e, mo);
      const T j0 = dot3(e, sh + 3) + dot3(mo, sh), j1 = dot3(e, sa + 3) + dot3(mo, sa);
      const T z0 = on_leg ? j0 : T(0), z1 = ank ? j1 : T(0);
      T zr[6];
      zr[0] = e[0];
      zr[1] = e[1];
      zr[2] = e[2];
      zr[3] = dot3(e, pA3) + mo[0];
      zr[4] = dot3(e, pA4) + dot3(mo, A4);
      zr[5] = dot3(e, pA5) + dot3(mo, A5);
      const T vrow = ((z0 * qh + z1 * qa) + (zr[0] * xr[nq + 0] + zr[1] * xr[nq + 1])) +
                     ((zr[2] * xr[nq + 2] + zr[3] * xr[nq + 3]) + (zr[4] * xr[nq + 4] + zr[5] * xr[nq + 5]));
      // rel_vel = vel_a - vel_b = -J qd:  b_n = -(1 + e) n.rel_vel - erp dist / dt,  b_t = -t.rel_vel  (the dt qdd share of
      // the velocity: see main_fd)
      const T crow = tk == 0 ? (T(1) + CT[TB::SC + TB::RESTITUTION]) * vrow - CT[TB::SC + TB::ERP_OVER_DT] * dist : vrow;
      row[0] = z0;
      row[1] = z1;
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) row[2 + rr] = real ? zr[rr] : T(0);
      row[8] = T(0);
      row[9] = T(0);
      row[OctLds::Z_B] = real ? crow : T(0);
      row[OctLds::Z_A] = real ? T(1) : T(0);
      row[OctLds::Z_HL] = on_leg ? (T)hl : T(-2);
      OCT_MARK("rows_geom_end");
  };
  auto rows_leg = [&](int wi) {
      OCT_MARK("rows_leg");
      T *const row = E + O.win + (wi & 1) * (8 * OctLds::ZW) + lane * OctLds::ZW;
      const T *const lcw = E + O.lcw;
      T z0 = row[0], z1 = row[1], zr[6];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) zr[rr] = row[2 + rr];
      const T hlv = row[OctLds::Z_HL];
      const int hl = hlv >= T(0) ? (int)hlv : 0;  // (a contact of the root body: z0 = z1 = 0, whatever leg is read)
      T lch[6], lca[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        lch[c] = lcw[hl * OctLds::LCW + c];
        lca[c] = lcw[(hl + 1) * OctLds::LCW + c];
      }
      const T *const lf = E + O.legf + (hl >> 1) * 3;
      const T lf0 = lf[0], lf1 = lf[1], lf2 = lf[2];
      // forward substitution L z = J^T, leaves first: the contact's leg, its share of the root rows; z~_leg = D^-1/2 z
      z1 -= lf0 * z0;
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) row[2 + rr] = zr[rr] - (lch[rr] * z0 + lca[rr] * z1);
      row[0] = z0 * lf1;
      row[1] = z1 * lf2;
  };
  auto rows_root = [&](int wi) {
      OCT_MARK("rows_root");
      T *const row = E + O.win + (wi & 1) * (8 * OctLds::ZW) + lane * OctLds::ZW;
      const T z0 = row[0], z1 = row[1];
      T zr[6];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) zr[rr] = row[2 + rr];
      const bool real = row[OctLds::Z_A] != T(0);
      static_for<1, 6>([&](auto rc) {
        constexpr int rr = decltype(rc)::value;
        static_for<0, rr>([&](auto cc) {
          constexpr int c = decltype(cc)::value;
          zr[rr] -= Ls[(rr * (rr - 1)) / 2 + c] * zr[c];
        });
      });
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) zr[rr] *= sq_ids[rr];
      // G = z~ . z~
      const T g = ((z0 * z0 + z1 * z1) + (zr[0] * zr[0] + zr[1] * zr[1])) + ((zr[2] * zr[2] + zr[3] * zr[3]) + (zr[4] * zr[4] + zr[5] * zr[5]));
      const T ai = real ? rcp_full<T>(g + CT[TB::SC + TB::CFM]) : T(0);
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) row[2 + rr] = zr[rr];
      row[OctLds::Z_A] = ai;
      row[OctLds::Z_G] = g;
      OCT_MARK("rows_root_end");
      OCT_STAMP(10, ai);
  };
  // ---- the sweep over a window (mb_constraint_solver.hpp:101-142), by the main wavefront.  A lone wavefront issues an
  //      instruction every ~5 cycles whatever it depends on (tools/ubench/lone_wave_latency.hip): what a row costs is its
  //      instruction count.  u~ = sum_r z~_r x_r is therefore DISTRIBUTED — lane j holds leg entry j (its own dof's) and root
  //      entry j (lanes 6, 7: the row's zero slots) — so that z~_r . u~ is one multiply, one fma and ONE 8-lane sum; the clamp
  //      runs redundantly on every lane; the normal rows and the friction rows of a window are separate loops (no selects on
  //      the row's kind), the first Gauss-Seidel iteration a separate instance (x starts from 0: no G x_old, no read of the
  //      previous impulse).  A row's operands are requested while the row before is processed (two register sets, the loop
  //      unrolled by two: no copies).
  auto main_sweep = [&](int wi) {
      OCT_MARK("sweep");
      const int nr = 3 * NA, nwin = (nr + 7) >> 3;
      const int pit = wi / nwin;
      const int w0 = (wi - pit * nwin) * 8;
      const T *const Zs = E + O.win + (wi & 1) * (8 * OctLds::ZW);
      T *const xs = E + O.xs;
      const T my_hip = (T)(lane & ~1);
      const T mu = CT[TB::SC + TB::FRICTION], rest = CT[TB::SC + TB::RESTITUTION];
      const int wn = nr - w0 < 8 ? nr - w0 : 8;
      const int k_n = NA - w0 < 0 ? 0 : (NA - w0 < wn ? NA - w0 : wn);  // rows 0 .. k_n - 1 of the window are normal rows
      struct Ops { T zs, zm, b, a, g, hl, sd, xo; };
      auto request = [&](int k, Ops &o, auto firstc, auto normalc) {
        constexpr bool FIRST = decltype(firstc)::value, NORMAL = decltype(normalc)::value;
        const T *const row = Zs + k * OctLds::ZW;
        o.zs = row[pos];        // my dof's leg entry, if the row's contact sits on my leg
        o.zm = row[2 + lane];   // my root entry (lanes 6, 7: zero)
        o.b = row[OctLds::Z_B];
        o.a = row[OctLds::Z_A];
        o.hl = row[OctLds::Z_HL];
        if constexpr (!FIRST) {
          o.g = row[OctLds::Z_G];
          o.xo = xs[w0 + k];
        }
        if constexpr (!NORMAL) o.sd = xs[w0 + k - ((w0 + k >= 2 * NA) ? 2 : 1) * NA];  // the impulse of the contact's normal row
      };
      auto process = [&](int k, const Ops &o, auto firstc, auto normalc) {
        constexpr bool FIRST = decltype(firstc)::value, NORMAL = decltype(normalc)::value;
        const T zl = o.hl == my_hip ? o.zs : T(0);
        T jw;
        // (a normal row's b_r carries (1 + e) J (dt qdd): the e-fold of z~_r . (dt y~) = -z~_r . u~_init on top of what u~ holds)
        if constexpr (NORMAL) jw = oct_sum(zl * (u + rest * u_init) + o.zm * (urm + rest * urm_init));
        else jw = oct_sum(zl * u + o.zm * urm);
        T xn;
        if constexpr (FIRST) xn = (o.b - jw) * o.a;
        else xn = (o.b - (jw - o.g * o.xo)) * o.a;
        if constexpr (NORMAL) {
          xn = max_t<T>(xn, T(0));
          xn = min_t<T>(xn, T(100000));
        } else {
          const T hi = mu * max_t<T>(o.sd, T(0));  // where_lt(s, 0, 0, s)
          xn = max_t<T>(xn, -hi);
          xn = min_t<T>(xn, hi);
        }
        T dx;
        if constexpr (FIRST) dx = xn;
        else dx = xn - o.xo;
        u += zl * dx;
        urm += o.zm * dx;
        xs[w0 + k] = xn;  // (every lane the same value to the same slot)
      };
      auto run = [&](int k0, int k1, auto firstc, auto normalc) {  // rows k0 .. k1 - 1
        if (k0 >= k1) return;
        Ops A, B;
        request(k0, A, firstc, normalc);
        for (int k = k0; k < k1; k += 2) {
          request(k + 1 < k1 ? k + 1 : k, B, firstc, normalc);
          process(k, A, firstc, normalc);
          if (k + 1 < k1) {
            request(k + 2 < k1 ? k + 2 : k + 1, A, firstc, normalc);
            process(k + 1, B, firstc, normalc);
          }
        }
      };
      if (pit == 0) {
        run(0, k_n, std::true_type{}, std::true_type{});
        run(k_n, wn, std::true_type{}, std::false_type{});
      } else {
        run(0, k_n, std::false_type{}, std::true_type{});
        run(k_n, wn, std::false_type{}, std::false_type{});
      }
      OCT_SYNC();
      OCT_MARK("sweep_end");
  };
  auto windows = [&]() -> int { return NA > 0 ? pgs_iters * ((3 * NA + 7) >> 3) : 0; };

  // (the step's done flag on every lane of the environment, for main_pool: under the Ant's rule it never passes through LDS)
  bool done_reg = false, done_in_reg = false;
  auto main_fin = [&]() {
    // ================================ main: impulse, integration, reward ================================
    OCT_MARK("main_fin");
    OCT_STAMP(6, u);
    {
      // qd_new = qd - L^-T D^-1/2 u~: forward dynamics and contact impulse in one back-substitution (see main_fd;
      // mb_constraint_solver.hpp:476-496: qd_b -= M^-1 J^T p)
      T w = u * my_sq;
      T wr[6];
      static_for<0, 6>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        wr[r] = oct_bcast<r>(urm) * sq_ids[r];  // (the root part of u~ back onto every lane)
      });
      static_for<0, 5>([&](auto ic) {
        constexpr int r = 4 - decltype(ic)::value;
        static_for<r + 1, 6>([&](auto cc) {
          constexpr int c = decltype(cc)::value;
          wr[r] -= Ls[(c * (c - 1)) / 2 + r] * wr[c];
        });
      });
#pragma unroll
      for (int r = 0; r < 6; ++r) w -= Lc[r] * wr[r];
      w -= (pos == 0 ? l10 : T(0)) * pair_bcast<1>(w);
      qd_new = qd - w;
#pragma unroll
      for (int r = 0; r < 6; ++r) qdr_new[r] = xr[nq + r] - wr[r];
    }
    OCT_MARK("main_integrate");
    // ---- M. integrate_euler: q += qd dt (integrator.hpp:126-131); the new state into the LDS record
    T qn0, qn2, qo0;  // the new base x, z and the old base x: every lane has them — the Ant's reward reads no LDS
    // (the reward rule's two constants requested here: behind the barrier below they were a round trip of their own)
    const int rm = (int)CT[TB::SC + TB::REWARD_MODE];
    const T inv_dt = CT[TB::SC + TB::INV_DT];
    OCT_SYNC();
    {
      // root coordinates: every lane holds the six velocities; lane 0 stores them and the integrated coordinates
      T qn_root[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) qn_root[r] = xr[r] + qdr_new[r] * dt;
      const T q_old0 = xr[0];
      qn0 = qn_root[0];
      qn2 = qn_root[2];
      qo0 = q_old0;
      OCT_SYNC();
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          xr[r] = qn_root[r];
          xr[nq + r] = qdr_new[r];
        }
      }
      xr[dq] = q + qd_new * dt;
      xr[nq + dq] = qd_new;
      if (lane == 0) {
        xr[in_dim] = q_old0;  // x_{t-1} (the Ant's reward reads it)
        xr[in_dim + 3] = T(0);  // "the y state of this step is out already" (set below for an environment the reset pool re-initialises)
      }
    }
    OCT_SYNC();
    OCT_MARK("main_reward");
    // ---- N. reward / done (ant_environment2.h:75-106; laikago_environment2.h:130-171)
    done_in_reg = rm == TDS_REWARD_ANT;
    if (rm == TDS_REWARD_ANT) {  // (wave-uniform) from registers, redundantly on every lane: (x_t - x_{t-1}) / dt, done: z < 0.26
      const T vel_x = (qn0 - qo0) * inv_dt;
      done_reg = qn2 < T(0.26);
      const T reward = done_reg ? T(0) : vel_x;
      if (lane == 0) {
        xr[in_dim + 1] = done_reg ? T(1) : T(0);
        xr[in_dim + 2] = reward;
      }
    } else {
      T rs = T(0), rc = T(1);
      if (rm == TDS_REWARD_LAIKAGO) sincos_t<T>(lane < 3 ? xr[3 + lane] * T(0.5) : T(0), &rs, &rc);  // (wave-uniform branch)
      const T s1 = oct_bcast<1>(rs), c1 = oct_bcast<1>(rc), s2 = oct_bcast<2>(rs), c2 = oct_bcast<2>(rc);
      if (lane == 0) {
        bool done = false;
        T reward = T(0);
        if (rm == TDS_REWARD_LAIKAGO) {
          const T sp = rs, cp = rc, st = s1, ct = c1, ss = s2, cs2 = c2;
          const T qx = sp * ct * cs2 - cp * st * ss;
          const T qy = cp * st * cs2 + sp * ct * ss;
          const T qz = cp * ct * ss - sp * st * cs2;
          const T qw = cp * ct * cs2 + sp * st * ss;
          const T sq = T(2) / (qx * qx + qy * qy + qz * qz + qw * qw);
          const T up = T(1) - (qx * (qx * sq) + qy * (qy * sq));
          done = (up < T(0.6)) || (xr[2] < T(0.2));
          reward = done ? T(0) : xr[0];
        }
        xr[in_dim + 1] = done ? T(1) : T(0);
        xr[in_dim + 2] = reward;
      }
    }
    OCT_SYNC();
  };
  // the state part of a y record — q | qd | (visual poses: M1) | up.z | zero padding — from the LDS record
  auto y_state = [&](TR *y, int end) {
    {  // (the reads in one batch in front of the stores: as a loop of read -> store every value was an LDS round trip of its own)
      constexpr int NB = (nq + nd + 7) / 8;
      T sv[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) sv[k] = xr[lane + 8 * k < nq + nd ? lane + 8 * k : 0];
#pragma unroll
      for (int k = 0; k < NB; ++k)
        if (lane + 8 * k < nq + nd) OCT_ST((TR)sv[k], &y[lane + 8 * k]);
    }
    int tail = nq + nd;
    if ((int)CT[TB::SC + TB::PACK_VISUALS]) {
      tail += 7 * nv;
      if (lane == 0) OCT_ST((TR)(CT[TB::SC + TB::BASE_R8]), &y[tail]);  // up_dot_world_z (fixed base)
      tail += 1;
    }
    for (int i = tail + lane; i < end; i += 8) OCT_ST(TR(0), &y[i]);
  };
  auto main_pool = [&]() {
    OCT_MARK("main_pool");
    // ---- auto_reset_when_done through the reset pool (ctl.pool; ars_vectorized_environment.h:262-277): a done environment
    //      takes its next pre-settled state — y, reward and done describe the terminal step, the observation and the state
    //      the fresh environment: its y state goes out HERE, before the record is overwritten
    if (ctl.pool != nullptr && valid && (done_in_reg ? done_reg : xr[in_dim + 1] != T(0))) {
      TR *yo, *yo2;
      int yend, yend2;
      y_where(yo, yo2, yend, yend2);
      if (yo != nullptr) {
        y_state(yo, yend);
        if (yo2 != nullptr) y_state(yo2, yend2);
      }
      const unsigned c = ctl.reset_count[env];
      const TR *const src = oct_global((const TR *)ctl.pool) + ((size_t)(c % (unsigned)ctl.pool_depth) * ctl.pool_envs + env) * (nq + nd);
      T v0 = (T)src[lane], v1 = (T)src[lane + 8], v2 = (T)src[lane + 16], v3 = T(0);
      if (lane + 24 < nq + nd) v3 = (T)src[lane + 24];
      OCT_SYNC();
      xr[lane] = v0;
      xr[lane + 8] = v1;
      xr[lane + 16] = v2;
      if (lane + 24 < nq + nd) xr[lane + 24] = v3;
      if (lane == 0) {
        ctl.reset_count[env] = c + 1u;
        xr[in_dim + 3] = T(1);
      }
    }
    OCT_MARK("main_pool_end");
    OCT_STAMP(7, tid);
  };
  auto help_rec = [&]() {
    OCT_MARK("help_rec");
    // ================================ helper: the step's records ================================
    // (two-wavefront build: while the main wavefront starts the next step — it does not write the record before its own
    //  phase M, two barriers from here)
    // ---- y record: q | qd | (visual poses: M1) | up.z | zero padding
    TR *yo, *yo2;
    int yend, yend2;
    y_where(yo, yo2, yend, yend2);
    if (valid && yo != nullptr && xr[in_dim + 3] == T(0)) {
      y_state(yo, yend);
      if (yo2 != nullptr) y_state(yo2, yend2);
    }
    // ---- [obs | reward | done] record (obs[0] = obs[1] = 0, ars_vectorized_environment.h:283-288): the slot of an obs ring
    //      (every step of a step-loop launch; floats on the multi-GPU wire format) and / or the caller's record (last step)
    if constexpr (LOOP) {
      if (ctl.obs_ring != nullptr) {  // wave-uniform
        const int slot = o_slot;
        const int rf = ctl.ring_flags;
        const bool f32w = (rf & TDS_RING_OBS_F32) != 0 || sizeof(TR) == 4;
        const int np = ctl.peer_arrive != nullptr ? ctl.n_peers : 0;
        const bool rd_only = (rf & TDS_RING_PEER_REWARD_DONE) != 0;
        if (ctl.peer_arrive != nullptr && (rf & TDS_RING_WIDE) != 0 && __all(valid)) {
          // Peer-store exchange, the wavefront's eight records as one row of 8-byte units (240 scalars: 960 contiguous bytes
          // on a float wire): every lane takes 8 bytes of the row — read from the environments' LDS records, converted
          // once — and the row goes out with one 8-byte-per-lane store instruction per destination and pass: this rank's
          // own block (device scope, write-through), then every peer's (system scope, over xGMI), the table's pointers by
          // scalar loads (see tds_kernels.hip: put_obs_wide)
          const int wl = tid & 63;
          const size_t row0 = ((size_t)slot * ctl.obs_envs + (size_t)blockIdx.x * 8) * (size_t)w_obs;
          const int per_unit = f32w ? 2 : 1;
          const int n_units = (8 * w_obs) / per_unit;
          const unsigned long long *const __attribute__((address_space(4))) *tab =
              (const unsigned long long *const __attribute__((address_space(4))) *)(const __attribute__((address_space(4))) void *)ctl.peer_ring;
          for (int u0 = 0; u0 < n_units; u0 += 64) {
            const int uu = u0 + wl;
            const bool on = uu < n_units;
            unsigned lo = 0u, hi = 0u;
            bool tail = false;  // this unit holds a [reward | done] column
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              if (c < per_unit) {
                const int f = on ? uu * per_unit + c : 0;
                const int e = f / w_obs;
                const int i = f - e * w_obs;
                const int src = i < nq + nd ? i : (i == nq + nd ? in_dim + 2 : in_dim + 1);
                const T vv = i < 2 ? T(0) : sm[e * O.stride + src];
                tail = tail || i >= nq + nd;
                if (f32w) {
                  const unsigned b = (unsigned)__float_as_int((float)vv);
                  if (c == 0) lo = b; else hi = b;
                } else {
                  const double dv = (double)vv;
                  lo = (unsigned)__double2loint(dv);
                  hi = (unsigned)__double2hiint(dv);
                }
              }
            }
            const unsigned long long bits = ((unsigned long long)hi << 32) | (unsigned long long)lo;
            const size_t unit_at = row0 / per_unit + (size_t)uu;  // (row0 is a multiple of per_unit: TDS_RING_WIDE)
            if (on) __hip_atomic_store(oct_global((unsigned long long *)ctl.obs_ring) + unit_at, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (reward | done only: a unit travels if ANY of its columns is one of the two — on a float wire they may share a
            //  unit with an observation column)
            const bool to_peers = on && (!rd_only || tail);
            for (int p0 = 0; p0 < np; p0 += 4) {  // (the table is padded to a multiple of four entries)
              const unsigned long long *const b0 = oct_global(tab[p0]), *const b1 = oct_global(tab[p0 + 1]), *const b2 = oct_global(tab[p0 + 2]),
                                       *const b3 = oct_global(tab[p0 + 3]);
              const size_t po = (size_t)ctl.peer_off / 8 + unit_at;
              if (to_peers) {
                using G64 = __attribute__((address_space(1))) unsigned long long;
                __hip_atomic_store((G64 *)((unsigned long long *)b0 + po), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (p0 + 1 < np) __hip_atomic_store((G64 *)((unsigned long long *)b1 + po), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (p0 + 2 < np) __hip_atomic_store((G64 *)((unsigned long long *)b2 + po), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (p0 + 3 < np) __hip_atomic_store((G64 *)((unsigned long long *)b3 + po), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              }
            }
          }
        } else if (valid) {
          const size_t at = ((size_t)slot * ctl.obs_envs + env) * w_obs;
          constexpr int NB = (w_obs + 7) / 8;
          T ov[NB];  // (read in one batch: see y_state)
#pragma unroll
          for (int k = 0; k < NB; ++k) {
            const int i = lane + 8 * k;
            const T rv = xr[i >= w_obs ? 0 : (i < nq + nd ? i : (i == nq + nd ? in_dim + 2 : in_dim + 1))];
            ov[k] = i < 2 ? T(0) : rv;
          }
#pragma unroll
          for (int k = 0; k < NB; ++k) {
            const int i = lane + 8 * k;
            if (i >= w_obs) continue;
            const T vv = ov[k];
            // (TDS_RING_NOFENCE: device-scope write-through stores, visible to the exchange after a plain wait)
            if (rf & TDS_RING_OBS_F32) {
              float *const pp = oct_global((float *)ctl.obs_ring) + at + i;
              if (rf & TDS_RING_NOFENCE) __hip_atomic_store(pp, (float)vv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              else OCT_ST((float)vv, pp);
            } else {
              TR *const pp = oct_global((TR *)ctl.obs_ring) + at + i;
              if (rf & TDS_RING_NOFENCE) __hip_atomic_store(pp, (TR)vv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              else OCT_ST((TR)vv, pp);
            }
            if (np > 0 && (i >= nq + nd || !rd_only)) {
              for (int pr = 0; pr < np; ++pr) {
                char *const pb = (char *)oct_global(((void *const __attribute__((address_space(4))) *)(const __attribute__((address_space(4))) void *)ctl.peer_ring)[pr]) + ctl.peer_off;
                if (rf & TDS_RING_OBS_F32) __hip_atomic_store((float *)pb + (at + i), (float)vv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else __hip_atomic_store((TR *)pb + (at + i), (TR)vv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              }
            }
          }
        }
        // (peer-store exchange: EVERY step is counted in — the last one here, by the wavefront that has just stored it; kernel
        //  completion would tell this rank, not the peers)
        if (last && ctl.peer_arrive != nullptr) signal_slot(slot);
      }
    }
    if (valid && last) {
      for (int i = lane; i < nq + nd; i += 8) {
        const TR vv = (TR)xr[i];
        if (obs_out != nullptr) obs_out[(size_t)env * w_obs + i] = i < 2 ? TR(0) : vv;
        if (x_feedback != nullptr) x_feedback[(size_t)env * in_dim + i] = vv;
      }
      if (lane == 0 && obs_out != nullptr) {
        obs_out[(size_t)env * w_obs + nq + nd] = (TR)xr[in_dim + 2];
        obs_out[(size_t)env * w_obs + nq + nd + 1] = (TR)xr[in_dim + 1];
      }
    }
    OCT_MARK("help_rec_end");
    OCT_STAMP(11, tid);
  };

  // ================================ the step ================================
  // barriers of a two-wavefront workgroup: (1) the kinematics are in LDS; (1b) the leg blocks' factors and couplings; (2) the
  // root block's factors (main wavefront), the contact list and counts (helper); one per row window (window wi is in LDS: the
  // helper stays a window ahead of the sweep); (0) the
  // step's state, reward and done are in the LDS record — the helper stores the step's records while the main wavefront
  // starts the next step (it does not write the record before its own integration, two barriers on)
  if constexpr (W2) {
    if (wv == 0) {
      main_kin();
      OCT_BAR();  // (1)
      main_dyn();
      OCT_BAR();  // (1b)
      main_dyn2();
      OCT_BAR();  // (2)
      main_get_count();
      main_fd();
      const int nw = windows();
      // (the first window's root stage stays with the helper although the main wavefront holds the factors in registers: doing
      //  it here measured 1.6 k cycles on the chain against 1.1 k of waiting for the helper's)
      for (int wi = 0; wi < nw; ++wi) {
        OCT_BAR();
        if (wi == 0) OCT_STAMP(12, tid);
        main_sweep(wi);
        if (wi == 0) OCT_STAMP(13, u);
      }
      main_fin();
      main_pool();
      OCT_BAR();  // (0)
    } else {
      OCT_BAR();  // (1)
      help_np();
      if (NA > 0) rows_geom(0);
      OCT_BAR();  // (1b)
      if (NA > 0) rows_leg(0);
      help_poses();
      OCT_BAR();  // (2)
      const int nw = windows();
      if (nw > 0) {
        if constexpr (TDS_OCT_PRIO == 2) __builtin_amdgcn_s_setprio(3);  // (the stretch the main wavefront waits for)
        help_get_factors();
        rows_root(0);
        if constexpr (TDS_OCT_PRIO == 2) __builtin_amdgcn_s_setprio(0);
      }
      for (int wi = 0; wi < nw; ++wi) {
        if (wi > 0) {
          rows_geom(wi);
          rows_leg(wi);
          rows_root(wi);
        }
        OCT_BAR();
      }
      OCT_BAR();  // (0)
      help_rec();
    }
  } else {
    main_kin();
    OCT_SYNC();
    help_np();
    help_poses();
    main_dyn();
    OCT_SYNC();
    main_dyn2();
    main_fd();
    const int nw = windows();
    for (int wi = 0; wi < nw; ++wi) {
      rows_geom(wi);
      rows_leg(wi);
      rows_root(wi);
      OCT_SYNC();
      main_sweep(wi);
    }
    main_fin();
    main_pool();
    OCT_SYNC();
    help_rec();
    if constexpr (LOOP) OCT_SYNC();
  }
#ifdef TDS_OCT_PROF
  if (prof_on && (tid & 63) == 0) {
    // (main: stamps 0 .. 7 into buf[0 .. 7]; helper: stamps 8 .. 11 and the window stamp 10 into buf[8 .. 11]; one-wave: all)
#pragma unroll
    for (int k = 16; k < 24; ++k)
      if (is_main) tds_oct_prof_buf[k] = prof_t[k];
#pragma unroll
    for (int k = 0; k < 14; ++k)
      if (((k < 8 || k >= 12) && is_main) || (k >= 8 && k < 12 && is_help)) tds_oct_prof_buf[k] = prof_t[k];
    if (is_help) tds_oct_prof_buf[15] = (unsigned long long)NA;
  }
#endif
  if constexpr (LOOP) {
    act_blk = act_blk + 1 >= ctl_arg.act_blocks ? 0 : act_blk + 1;
    y_slot = y_slot + 1 >= ctl_arg.y_slots ? 0 : y_slot + 1;
    o_slot = o_slot + 1 >= ctl_arg.obs_slots ? 0 : o_slot + 1;
  }
  }  // ================================ end of the step loop ================================
#ifdef TDS_OCT_PROF
  if (threadIdx.x == 0 && blockIdx.x < 2048) {
    unsigned long long c_;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_)::"memory");
    tds_oct_prof_wg[3 * blockIdx.x + 2] = c_;
  }
#endif
}

template <typename T, typename TR, bool LOOP, int BUILD>
__global__ __launch_bounds__(BUILD >= 2 ? 128 : 64) __attribute__((amdgpu_waves_per_eu(BUILD == 2 ? 2 : 1, BUILD == 2 ? 2 : 1)))
void tds_oct_kernel(const DevModel<T> *__restrict__ mdl_arg, const TR *x_in, TR *__restrict__ y_out,
                    const TR *__restrict__ actions, TR *x_feedback, TR *__restrict__ obs_out, TdsStepCtl ctl_arg, int n_envs,
                    OctOff O) {
  oct_body<T, TR, LOOP, BUILD>(mdl_arg, x_in, y_out, actions, x_feedback, obs_out, ctl_arg, n_envs, O);
}
// (amdgpu_num_vgpr counts half of the unified register file of gfx90a and later: 120 = 240 registers.  Measured with 128 — the
//  254 registers of BUILD 2, no scratch —: a pass does not start beside a chunk of BUILD 3's 254, the rate falls back to 4.45e8)
template <typename T, typename TR, bool LOOP>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2), amdgpu_num_vgpr(120)))
void tds_oct_kernel_beside(const DevModel<T> *__restrict__ mdl_arg, const TR *x_in, TR *__restrict__ y_out,
                           const TR *__restrict__ actions, TR *x_feedback, TR *__restrict__ obs_out, TdsStepCtl ctl_arg,
                           int n_envs, OctOff O) {
  oct_body<T, TR, LOOP, 4>(mdl_arg, x_in, y_out, actions, x_feedback, obs_out, ctl_arg, n_envs, O);
}

}  // namespace

#ifdef TDS_OCT_PROF
extern "C" int tds_oct_prof_workgroups(unsigned long long *out, int n_wg) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(tds_oct_prof_wg), (size_t)3 * n_wg * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
extern "C" int tds_oct_prof_clocks(unsigned long long *out64) {
  return hipMemcpyFromSymbol(out64, HIP_SYMBOL(tds_oct_prof_clk), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
extern "C" int tds_oct_prof_read(unsigned long long *out32, int iter) {  // iter >= 0: which iteration the NEXT launches stamp
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(tds_oct_prof_buf), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (iter >= 0 && hipMemcpyToSymbol(HIP_SYMBOL(tds_oct_prof_iter), &iter, sizeof(int)) != hipSuccess) return -1;
  return 0;
}
#endif

// LDS bytes of one environment of the 8-lane kernel, and of one workgroup (eight environments + the constant table)
int tds_oct_lds_bytes(int input_dim) { return oct_layout(input_dim).stride * (int)sizeof(double); }
int tds_oct_workgroup_bytes(int input_dim) {
  return (oct_layout(input_dim).stride * 8 + TdsOctTab::TOTAL) * (int)sizeof(double);
}

// build: 1 one wavefront per workgroup, 2 / 3 the two-wavefront builds (the host grants them while every workgroup of the
// launch is resident with at most two wavefronts per SIMD / one: tds_api.hip), 4 the two-wavefront build of at most 240 registers
template <typename T, typename TR>
int tds_launch_oct(const DevModel<T> *d_model, const DevModel<T> &h_model, const TR *x_in, TR *y_out, const TR *actions,
                   TR *x_feedback, TR *obs_out, int n_envs, hipStream_t stream, const TdsStepCtl &ctl, int build) {
  const OctOff O = oct_layout(h_model.input_dim);
  const int blocks = (n_envs + 7) / 8;
  const size_t shmem = ((size_t)O.stride * 8 + TdsOctTab::TOTAL) * sizeof(T);
  // one plain step without rings: the straight-line form; K steps, record rings: the step-loop form
  const bool one_step = ctl.nsub == 1 && ctl.obs_ring == nullptr && ctl.y_ring == nullptr;
#define OCT_LAUNCH(LOOP_, B_)                                                                                                 \
  hipLaunchKernelGGL((tds_oct_kernel<T, TR, LOOP_, B_>), dim3(blocks), dim3(B_ >= 2 ? 128 : 64), shmem, stream, d_model, x_in, \
                     y_out, actions, x_feedback, obs_out, ctl, n_envs, O)
  if (build == 4) {
    if (one_step)
      hipLaunchKernelGGL((tds_oct_kernel_beside<T, TR, false>), dim3(blocks), dim3(128), shmem, stream, d_model, x_in, y_out, actions,
                         x_feedback, obs_out, ctl, n_envs, O);
    else
      hipLaunchKernelGGL((tds_oct_kernel_beside<T, TR, true>), dim3(blocks), dim3(128), shmem, stream, d_model, x_in, y_out, actions,
                         x_feedback, obs_out, ctl, n_envs, O);
  } else if (one_step) {
    if (build == 3) OCT_LAUNCH(false, 3);
    else if (build == 2) OCT_LAUNCH(false, 2);
    else OCT_LAUNCH(false, 1);
  } else {
    if (build == 3) OCT_LAUNCH(true, 3);
    else if (build == 2) OCT_LAUNCH(true, 2);
    else OCT_LAUNCH(true, 1);
  }
#undef OCT_LAUNCH
  return (int)hipGetLastError();
}
template int tds_launch_oct<double, double>(const DevModel<double> *, const DevModel<double> &, const double *, double *,
                                            const double *, double *, double *, int, hipStream_t, const TdsStepCtl &, int);
template int tds_launch_oct<double, float>(const DevModel<double> *, const DevModel<double> &, const float *, float *,
                                           const float *, float *, float *, int, hipStream_t, const TdsStepCtl &, int);
