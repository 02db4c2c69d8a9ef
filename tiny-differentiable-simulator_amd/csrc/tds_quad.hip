// tds_quad.hip — the step kernel of the LEGGED robots on 16 lanes per environment (round 5; BASELINE config 4: Laikago).
//
// The general kernel (tds_kernels.hip) gives an environment as many lanes as it has links and dofs: Laikago's 22 links /
// 18 dofs take a 32-lane group — two environments per wavefront — and its dense register LDL^T, its row solves and its
// forward-dynamics solve read all 153 entries of L although the robot is a STAR: a root body on the reference's six
// "virtual" links (prismatic x, y, z, revolute x, y, z: tds_device_model.h euler_root) and four legs that are serial
// chains of four links (hip, upper, lower, fixed toe: leg_len == 4) with nothing between them but the root.  Round 4's
// counts said 58 % of that kernel's VALU stream moves data and that it is issue-bound: the lever is instructions per
// environment.  This kernel is built for the star:
//
//   * 16 lanes per environment = the 16 leg links, one QUAD of lanes per leg (lane = 4 leg + position in the chain);
//     four environments per wavefront.  The root chain needs no lane: pose, motion axes, velocity and bias
//     acceleration of the root body in closed form from (q0..5, qd0..5) on every lane (the formulas of the general
//     kernel's phase C); its rigid inertia redundantly on every lane; the composite of the legs reaches it by two row
//     rotations.
//   * M in LEAVES-FIRST order [leg 0 | leg 1 | leg 2 | leg 3 | root]: four independent 3 x 3 leg blocks B_l, their
//     6-column couplings C_l to the root and the root's 6 x 6 block — 21 of the dense 171 entries per leg row instead of 18.
//     LDL^T: every lane of a quad factorises its leg's 3 x 3 block (values gathered by quad broadcasts: DPP moves, no LDS),
//     the coupling rows L_c = C D^-1 follow inside the quad, the root's Schur complement S = R - sum_l L_c D L_c^T is
//     formed lane-parallel (21 entries on 16 lanes, two passes over an LDS copy of L_c) and factorised redundantly on every
//     lane in registers.  Forward dynamics qdd = M^-1 (tau - C) and the final qd -= M^-1 J^T p are quad-local
//     substitutions plus six 16-lane sums.
//   * a toe's constraint row touches its own leg's three dofs and the root's six: z~ = D^-1/2 L^-1 J^T costs 9 + 18 + 15
//     multiply-adds per row (dense: 153), the rows are stored 9 wide, and the Gauss-Seidel sweep keeps the root part of
//     u~ on every lane.  Contact points are the toes' own lanes: the narrowphase reads no LDS at all.
//
// Same arithmetic contract as the general kernel — the reference's env step (locomotion_contact_simulation.h:151-304) to
// round-off, the same quirks (contacts from pre-step transforms with post-integration velocities, plane_space's k,
// unnormalised REVOLUTE_AXIS axes, visual poses that lag q by one step) — pinned by the same tests: a handle takes this
// kernel when its model is such a star (DevModel::quad, tds_device_model.h) and option quad is not 0; option quad = 0
// keeps the general kernel, and tests/test_quad.py holds the two against each other and against the reference.
//
// Two forms (template parameter LOOP): one step per launch, and K steps per launch with the state in LDS, a fresh action block
// per step, per-step record rings and the reset pool's "done environment takes its next pre-settled state" inside the loop —
// what tds_hip_step_many / _rings run for this model.  In-kernel reset + settle, substeps with one action, on-device
// rollouts and the exchange launches of the multi-GPU layer stay with the general kernel.  Reference files as in
// tds_kernels.hip.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "tds_device_model.h"
#include "tds_kernels.h"
#include "tds_lanes.h"

namespace {

// value of lane K of each QUAD of lanes, delivered to the four lanes of the quad (DPP quad_perm: a VALU move)
template <int K>
__device__ __forceinline__ double quad_bcast(double v) {
  constexpr int CTRL = K * 0x55;
  const int l = __double2loint(v), h = __double2hiint(v);
  const int lo = __builtin_amdgcn_update_dpp(0, l, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, h, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), K * 0x55, 0xF, 0xF, true));
}

// sin and cos of a joint angle: Cody-Waite reduction by pi / 2 in two fused steps and the fdlibm kernels on [-pi/4, pi/4]
// (< 1 ulp; ~35 instructions where the library routine takes ~90); lanes beyond 1e5 rad — or NaN — take the library routine
// without changing what the other lanes of the wavefront compute (see tds_oct.hip: oct_sincos)
__device__ __forceinline__ void quad_sincos(double x, double *sn, double *cs) {
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
  if (__builtin_expect(__any(big), 0)) {
    double s2, c2;
    sincos(x, &s2, &c2);
    s_ = big ? s2 : s_;
    c_ = big ? c2 : c_;
  }
  *sn = s_;
  *cs = c_;
}
__device__ __forceinline__ void quad_sincos(float x, float *sn, float *cs) { sincosf(x, sn, cs); }

// launder a model pointer WITHOUT losing its address space (laundered as a generic pointer every access behind it is a FLAT
// load: both counters, out of order with the DS instructions — see tds_kernels.hip)
#ifndef QUAD_MDL_AS
#define QUAD_MDL_AS 4  /* 4 constant address space (the model is read-only: scalar loads at uniform addresses), 1 global, 0 generic
                         (flat loads: round 5's first form) — laikago_soft x 8192: 23.2 / 24.5 / 24.8 us per step */
#endif
#define QUAD_LAUNDER_MODEL(dst, src)                                                                                    \
  const DevModel<T> *dst;                                                                                               \
  {                                                                                                                     \
    const __attribute__((address_space(QUAD_MDL_AS))) DevModel<T> *g_ = (const __attribute__((address_space(QUAD_MDL_AS))) DevModel<T> *)(src); \
    asm volatile("" : "+s"(g_));                                                                                        \
    dst = (const DevModel<T> *)g_;                                                                                      \
  }
#define QUAD_SYNC()                                            \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     \
    __builtin_amdgcn_wave_barrier();                           \
  } while (0)

// LDS per environment, in scalars of T (odd strides where lanes index rows)
struct QuadLds {
  static constexpr int LCW = 13;                  // [16 lanes][Lc(6) | W(6)] + pad
  static constexpr int ZW = 9;                    // a constraint row: 3 leg entries | 6 root entries
  // offsets are computed at run time from input_dim (see quad_layout)
};

struct QuadOff {
  int lcw, legf, swl, qdp, S, ax, Z, rws, xs, cp, stride;
  int in_dim, adim;  // the model's input_dim / action_dim (kernel arguments: the prologue's record loads do not wait for the model)
};
// the kernel's parameter list as a struct: the layout of its kernel-argument segment (see TdsKernArgs in tds_kernels.hip)
struct QuadKernArgs {
  const void *mdl, *x_in;
  void *y_out;
  const void *actions;
  void *x_feedback, *obs_out;
  TdsStepCtl ctl;
  int n_envs;
  QuadOff O;
};
// The model constants of a step-loop launch, in LDS: filled once per launch, behind the environments' regions.  Read from L2
// through a laundered model pointer — what keeps a step-loop body at its register budget — every iteration began with a
// round trip per group of constants, and the first of them also waited for the previous step's record stores (loads and
// stores share one in-order counter on gfx9): 35.2 us per laikago_soft x 8192 step against 27.2 for the chained graphs
// of the straight-line form (profiles/r05_quad_forms.txt).  The straight-line form reads the model directly.
// Phase stamps (a build of its own: -DTDS_QUAD_PROF, tools/quad_profile.sh): workgroup 3 writes the shader clock at the phase
// boundaries of a step (step-loop form: of iteration tds_quad_prof_iter) into tds_quad_prof_buf
#ifdef TDS_QUAD_PROF
__device__ unsigned long long tds_quad_prof_buf[32];
__device__ int tds_quad_prof_iter = 0;
#define QUAD_STAMP(k, pin)                                                                      \
  do {                                                                                          \
    unsigned long long t_;                                                                      \
    auto p_ = (pin); /* a value of the phase before: computed before the clock is read */       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(p_)::"memory");        \
    if (prof_on) prof_t[k] = t_;                                                                \
  } while (0)
#else
#define QUAD_STAMP(k, pin) do { } while (0)
#endif

template <typename T>
struct QuadTable {
  T S[6][16], X_T[12][16], mass[16], com[3][16], inertia[9][16], init_pose[16], stiffness[16], damping[16];
  int joint_type[16], act_index[16];
  T mass5, com5[3], inertia5[9];
  T cp_radius[4], cp_local[3][4];
  T vis_X[12][17];
  T dt, action_limit, base_t[3], grav[3], plane_n[3], plane_c, nb[3], t1[3], t2[3], cfm, erp_over_dt, restitution, friction, base_R8;
  int input_dim, action_dim, num_visuals, step_mode, reward_mode, pgs_iterations, pack_visuals, output_dim;
  int vis_flags;  // bit 0: every visual sits at its link's origin (no offset), bit 1: the root body's visual is not rotated against it
};
// constant `tab` of the table in a step-loop launch, `glob` of the model otherwise
#define QC(tab, glob) (LOOP ? (CT->tab) : (mdl->glob))

template <bool LOOP>
struct QuadCtlRef {
  using type = const TdsStepCtl &;
  static __device__ __forceinline__ type get(const TdsStepCtl &param, const __attribute__((address_space(4))) char *) { return param; }
};
template <>
struct QuadCtlRef<true> {
  using type = const __attribute__((address_space(4))) TdsStepCtl &;
  static __device__ __forceinline__ type get(const TdsStepCtl &, const __attribute__((address_space(4))) char *at) {
    return *(const __attribute__((address_space(4))) TdsStepCtl *)at;
  }
};
__host__ __device__ inline QuadOff quad_layout(int in_dim) {
  QuadOff o;
  int at = in_dim + 4;
  at = (at + 1) & ~1;
  o.lcw = at;  at += 12 * QuadLds::LCW;   // L_c and W = L_c D of the 12 leg-dof lanes (slot 3 leg + position)
  o.legf = at; at += 4 * 9 + 1;            // per leg: l10 l20 l21 | 1/d (3) | sqrt(1/d) (3)
  o.swl = at;  at += 12 * 7;               // world motion axis of the 12 leg-dof lanes (6, stride 7; slot 3 leg + position)
  o.qdp = at;  at += 16 + 6;               // velocities after integrate_euler_qdd: leg lanes | root
  o.S = at;    at += 21 + 1;               // Schur complement of the root block
  o.ax = at;   at += 6 * 7;                // the six root motion axes (stride 7)
  o.Z = at;    at += 12 * QuadLds::ZW;     // constraint rows z~
  o.rws = at;  at += 4 * 12;               // per row: b | 1 / (G + cfm) | G | leg of the row's contact
  o.xs = at;   at += 12;                   // impulses
  o.cp = at;   at += 5 * 4;                // contact list: point (3) | distance | leg, per slot
  o.stride = (at + 1) & ~1;
  o.in_dim = in_dim;
  o.adim = 0;  // (set by the launcher)
  return o;
}

// LOOP = false: one step per launch.  LOOP = true: ctl.nsub steps in ONE launch with the state in the LDS record between them
// (tds_hip_step_many / _rings for this model): a fresh action block per step (ctl.act_pool), every step's y and
// [obs | reward | done] records into ring slots (ctl.y_ring / obs_ring) — or the last step's only —, a done environment
// taking its next pre-settled state from the reset pool inside the loop (ctl.pool).  As in the general kernel's step-loop
// builds nothing but the loop state lives across an iteration: the model pointer and the kernel-argument segment are
// laundered per iteration, so the lane constants and the ctl fields are loaded where an iteration uses them.
// WAVES: wavefronts per workgroup.  1 everywhere but the WIDE step-loop launches (round 6): eight wavefronts — 32 environments
// — share ONE constant table, a workgroup takes a compute unit's whole LDS (8 x 19 456 + 6 600 of 163 840 B) and the 256
// workgroups of laikago_soft x 8192 are resident at once: two wavefronts per SIMD exactly as eight one-wavefront workgroups
// per compute unit would be, which the table's 6.6 KB per workgroup rules out.  The wavefronts of a workgroup never meet
// again after the table is filled (every wavefront writes all of it — the same values — and reads it behind its own
// wavefront barrier): no s_barrier anywhere.
template <typename T, typename TR, bool LOOP, int WAVES = 1>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2)))
void tds_quad_kernel(const DevModel<T> *__restrict__ mdl_arg, const TR *x_in, TR *__restrict__ y_out,
                     const TR *__restrict__ actions, TR *x_feedback /* may alias x_in */, TR *__restrict__ obs_out,
                     TdsStepCtl ctl_arg, int n_envs, QuadOff O) {
  extern __shared__ __align__(16) unsigned char tds_quad_smem[];
  T *const sm = reinterpret_cast<T *>(tds_quad_smem);
  constexpr int nq = 18, nd = 18;
  {
    // ---- A. x record -> LDS (coalesced), fresh actions over the action slice
    const int lane = threadIdx.x & 15, grp = threadIdx.x >> 4, env = blockIdx.x * (4 * WAVES) + grp;
    const bool valid = env < n_envs;
    T *const xr = sm + grp * O.stride;
    // (dimensions from the kernel arguments and all of a lane's record loads issued before the first is waited for: as a
    //  loop of load -> LDS store behind a load of the model's input_dim the prologue was five dependent round trips in
    //  front of EVERY single-step launch — tools/oct_clock_ramp.py measured the same pattern in the 8-lane kernel)
    const int in_dim = O.in_dim, adim = O.adim;
    constexpr int XN = 5;  // (records of up to 80 scalars: 18 + 18 + actions + 3)
    T xv[XN];
#pragma unroll
    for (int k = 0; k < XN; ++k) {
      const int i = lane + 16 * k;
      const bool act = actions != nullptr && i >= nq + nd && i < nq + nd + adim;
      xv[k] = (!valid || i >= in_dim) ? T(0) : act ? (T)actions[(size_t)env * adim + (i - nq - nd)] : (T)x_in[(size_t)env * in_dim + i];
    }
#pragma unroll
    for (int k = 0; k < XN; ++k) {
      const int i = lane + 16 * k;
      if (i < in_dim) xr[i] = xv[k];
    }
  }
  QuadTable<T> *const CT = reinterpret_cast<QuadTable<T> *>(sm + 4 * WAVES * O.stride);  // (step-loop launches only)
  if constexpr (LOOP) {
    const DevModel<T> *const md = mdl_arg;
    const int t = threadIdx.x & 63;
    if (t < 16) {
      const int l = 6 + t;
#pragma unroll
      for (int k = 0; k < 6; ++k) CT->S[k][t] = md->S[k][l];
#pragma unroll
      for (int k = 0; k < 12; ++k) CT->X_T[k][t] = md->X_T[k][l];
      CT->mass[t] = md->mass[l];
#pragma unroll
      for (int k = 0; k < 3; ++k) CT->com[k][t] = md->com[k][l];
#pragma unroll
      for (int k = 0; k < 9; ++k) CT->inertia[k][t] = md->inertia[k][l];
      CT->init_pose[t] = md->init_pose[l];
      CT->stiffness[t] = md->stiffness[l];
      CT->damping[t] = md->damping[l];
      CT->joint_type[t] = md->joint_type[l];
      CT->act_index[t] = md->act_index[l];
    } else if (t < 20) {
      const int k = t - 16;
      CT->cp_radius[k] = md->cp_radius[k];
#pragma unroll
      for (int c = 0; c < 3; ++c) CT->cp_local[c][k] = md->cp_local[c][k];
    } else if (t == 20) {
      CT->mass5 = md->mass[5];
#pragma unroll
      for (int k = 0; k < 3; ++k) CT->com5[k] = md->com[k][5];
#pragma unroll
      for (int k = 0; k < 9; ++k) CT->inertia5[k] = md->inertia[k][5];
    } else if (t == 21) {
      CT->dt = md->dt;
      CT->action_limit = md->action_limit;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        CT->base_t[k] = md->base_t[k];
        CT->grav[k] = md->grav[k];
        CT->plane_n[k] = md->plane_n[k];
        CT->nb[k] = md->nb[k];
        CT->t1[k] = md->t1[k];
        CT->t2[k] = md->t2[k];
      }
      CT->plane_c = md->plane_c;
      CT->cfm = md->cfm;
      CT->erp_over_dt = md->erp_over_dt;
      CT->restitution = md->restitution;
      CT->friction = md->friction;
      CT->base_R8 = md->base_R[8];
      CT->input_dim = md->input_dim;
      CT->action_dim = md->action_dim;
      CT->num_visuals = md->num_visuals;
      CT->step_mode = md->step_mode;
      CT->reward_mode = md->reward_mode;
      CT->pgs_iterations = md->pgs_iterations;
      CT->pack_visuals = md->pack_visuals;
      CT->output_dim = md->output_dim;
      // (what the visual poses can skip: Laikago's visuals have no offsets, its chassis visual no rotation)
      bool no_off = true, root_id = true;
      for (int k = 0; k < md->num_visuals && k < 17; ++k)
        no_off = no_off && md->vis_X[9][k] == T(0) && md->vis_X[10][k] == T(0) && md->vis_X[11][k] == T(0);
      for (int c = 0; c < 9; ++c) root_id = root_id && md->vis_X[c][0] == ((c & 3) == 0 ? T(1) : T(0));
      CT->vis_flags = (no_off ? 1 : 0) | (root_id ? 2 : 0);
    }
    for (int i = t; i < 12 * 17; i += 64) {
      const int c = i / 17, k = i - 17 * c;
      CT->vis_X[c][k] = k < md->num_visuals ? md->vis_X[c][k] : T(0);
    }
    QUAD_SYNC();
  }
  const int nsteps = LOOP ? ctl_arg.nsub : 1;
  // ring positions of a step-loop launch, carried as scalars from step to step: the action block the NEXT step takes, the y and
  // the obs slot of THIS step — computed per step they were three integer divisions (~25 instructions each) behind three
  // scalar-load round trips
  int pos_act = 0, pos_y = 0, pos_obs = 0, n_act = 1, n_y = 1, n_obs = 1;
  if constexpr (LOOP) {
    if (ctl_arg.act_pool != nullptr) {
      n_act = ctl_arg.act_blocks;
      pos_act = (ctl_arg.act_first + 1) % n_act;
    }
    if (ctl_arg.y_ring != nullptr) {
      n_y = ctl_arg.y_slots;
      pos_y = ctl_arg.y_first % n_y;
    }
    if (ctl_arg.obs_ring != nullptr) {
      n_obs = ctl_arg.obs_slots;
      pos_obs = ctl_arg.obs_first % n_obs;
    }
  }
  T next_act = T(0);  // (step-loop form: the action block of the NEXT step, requested a step ahead)
  for (int it = 0; it < nsteps; ++it) {  // ================================ step loop ================================
  // (nothing but `it` lives across an iteration: lane, model pointer and kernel-argument segment are laundered)
  const __attribute__((address_space(QUAD_MDL_AS))) DevModel<T> *mdl_g = (const __attribute__((address_space(QUAD_MDL_AS))) DevModel<T> *)mdl_arg;
  const __attribute__((address_space(4))) char *ka_seg = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
  int tid = threadIdx.x;
  if constexpr (LOOP) asm volatile("" : "+s"(mdl_g), "+s"(ka_seg), "+v"(tid));
  const DevModel<T> *const mdl = (const DevModel<T> *)mdl_g;
  const int lane = tid & 15;
  const int grp = tid >> 4;  // (environment of the workgroup: 4 per wavefront)
  const int env = blockIdx.x * (4 * WAVES) + grp;
  const bool valid = env < n_envs;
#ifdef TDS_QUAD_PROF
  unsigned long long prof_t[18];
  const bool prof_on = blockIdx.x == 3 && tid < 64 && it == (LOOP ? tds_quad_prof_iter : 0);
#endif
  T *const E = sm + grp * O.stride;
  T *const xr = E;
  const int leg = lane >> 2, pos = lane & 3;
  const int li = 6 + lane;            // my link
  const bool dofl = pos < 3;          // my link carries a dof (the toe's joint is fixed)
  const int dq = 6 + 3 * leg + pos;   // ... this one, in the q / qd records (nq == nd == 18)
  const int in_dim = QC(input_dim, input_dim), adim = QC(action_dim, action_dim);
  // (kernel-argument segment: mdl 0 | x_in 8 | y_out 16 | actions 24 | x_feedback 32 | obs_out 40 | ctl 48: see QuadKernArgs)
  typename QuadCtlRef<LOOP>::type ctl = QuadCtlRef<LOOP>::get(ctl_arg, ka_seg + __builtin_offsetof(QuadKernArgs, ctl));
  const T dt = QC(dt, dt);
  const bool last = it == nsteps - 1;
  if constexpr (LOOP) {
    // the action block of this step (step 0's came in with the record) was requested at the top of the step before and
    // moved into the record's action slots at the END of that step, in front of its record stores (see phase M); the
    // next step's — block (act_first + it + 1) % act_blocks of the pool — is requested now: no step waits for HBM
    if (ctl.act_pool != nullptr) {  // wave-uniform
      if (it + 1 < nsteps && valid && lane < adim) {
        const int blk = pos_act;
        next_act = (T)((const TR *)ctl.act_pool)[((size_t)blk * ctl.act_envs + env) * adim + lane];
      }
    }
  }
  // lane constants (issued under the latency of the record)
  const int jt = QC(joint_type[lane], joint_type[li]);
  const int act_i = QC(act_index[lane], act_index[li]);
  const T init_pose_l = QC(init_pose[lane], init_pose[li]), stiff_l = QC(stiffness[lane], stiffness[li]), damp_l = QC(damping[lane], damping[li]);
  T Sl[6], RT[9], tT[3];
#pragma unroll
  for (int k = 0; k < 6; ++k) Sl[k] = QC(S[k][lane], S[k][li]);
#pragma unroll
  for (int k = 0; k < 9; ++k) RT[k] = QC(X_T[k][lane], X_T[k][li]);
#pragma unroll
  for (int k = 0; k < 3; ++k) tT[k] = QC(X_T[9 + k][lane], X_T[9 + k][li]);
  const T act_lim = QC(action_limit, action_limit);
  QUAD_SYNC();
  const T q = dofl ? xr[dq] : T(0);
  const T qd = dofl ? xr[nq + dq] : T(0);

  QUAD_STAMP(0, q);
  // ---- PD controller (locomotion_contact_simulation.h:168-258) or direct torque; joint stiffness / damping
  T tau = T(0);
  if (QC(step_mode, step_mode) == TDS_STEP_LOCOMOTION) {
    if (act_i >= 0) {
      const int var = nq + nd + adim;
      const T kp = xr[var], kd = xr[var + 1], max_force = xr[var + 2];
      T a = xr[nq + nd + act_i];
      a = a < act_lim ? a : act_lim;
      a = a > -act_lim ? a : -act_lim;
      const T q_des = init_pose_l + a;
      T f = kp * (q_des - q) + kd * (T(0) - qd);
      f = f > -max_force ? f : -max_force;
      f = f < max_force ? f : max_force;
      tau = f;
    }
  } else if (dofl && dq < adim) {
    tau = xr[nq + nd + dq];
  }
  tau -= stiff_l * q + damp_l * qd;

  QUAD_STAMP(1, tau);
  // ---- B. jcalc (link.hpp:229-287); the toe lanes of legs 0..2 — fixed joints, no angle of their own — take the root's
  //         three angles: their sines and cosines reach every lane by one row broadcast each
  T Rp[9], tp[3], sn, cs;
  {
    const T ang = dofl ? (jt == TDS_JOINT_REVOLUTE_AXIS ? q * T(0.5) : q) : xr[3 + (leg < 3 ? leg : 0)];
    quad_sincos(ang, &sn, &cs);
    const bool rev = jt >= TDS_JOINT_REVOLUTE_X && jt <= TDS_JOINT_REVOLUTE_AXIS;
    const bool pris = jt >= TDS_JOINT_PRISMATIC_X && jt <= TDS_JOINT_PRISMATIC_AXIS;
    T RJ[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)};
    T tJ[3] = {T(0), T(0), T(0)};
    if (pris) {
      tJ[0] = Sl[3] * q;
      tJ[1] = Sl[4] * q;
      tJ[2] = Sl[5] * q;
    }
    if (rev) {
      if (jt == TDS_JOINT_REVOLUTE_X) {
        RJ[4] = cs; RJ[5] = -sn; RJ[7] = sn; RJ[8] = cs;
      } else if (jt == TDS_JOINT_REVOLUTE_Y) {
        RJ[0] = cs; RJ[2] = sn; RJ[6] = -sn; RJ[8] = cs;
      } else if (jt == TDS_JOINT_REVOLUTE_Z) {
        RJ[0] = cs; RJ[1] = -sn; RJ[3] = sn; RJ[4] = cs;
      } else {  // axis-angle quaternion with the UNNORMALISED axis (link.hpp:256-261)
        // (1 / |axis| by the reciprocal square root; the quaternion's squared norm n2 is 1 to rounding — sin^2 + cos^2 —, so
        //  2 / n2, the reference's quat_to_matrix scale, is 2 (2 - n2) to the last bit: one Newton step from 1)
        const T sh = sn * rsqrt_full<T>(Sl[0] * Sl[0] + Sl[1] * Sl[1] + Sl[2] * Sl[2]);
        const T qx = Sl[0] * sh, qy = Sl[1] * sh, qz = Sl[2] * sh, qw = cs;
        const T s2 = T(4) - T(2) * (qx * qx + qy * qy + qz * qz + qw * qw);
        const T xs = qx * s2, ys = qy * s2, zs = qz * s2;
        const T wx = qw * xs, wy = qw * ys, wz = qw * zs;
        const T xx = qx * xs, xy = qx * ys, xz = qx * zs;
        const T yy = qy * ys, yz = qy * zs, zz = qz * zs;
        RJ[0] = T(1) - (yy + zz); RJ[1] = xy - wz; RJ[2] = xz + wy;
        RJ[3] = xy + wz; RJ[4] = T(1) - (xx + zz); RJ[5] = yz - wx;
        RJ[6] = xz - wy; RJ[7] = yz + wx; RJ[8] = T(1) - (xx + yy);
      }
    }
    mat3_mul(RT, RJ, Rp);
    T r[3];
    mat3_mulv(RT, tJ, r);
    tp[0] = tT[0] + r[0];
    tp[1] = tT[1] + r[1];
    tp[2] = tT[2] + r[2];
  }

  QUAD_STAMP(2, tp[2]);
  // ---- C. the root chain in closed form (kinematics.hpp:64-97; see tds_kernels.hip phase C: same formulas), on every lane
  const T q0 = xr[0], q1 = xr[1], q2 = xr[2];
  const T sx = dpp_bcast<3>(sn), cx = dpp_bcast<3>(cs);
  const T sy = dpp_bcast<7>(sn), cy = dpp_bcast<7>(cs);
  const T sz = dpp_bcast<11>(sn), cz = dpp_bcast<11>(cs);
  T R5[9];  // the root body's rotation
  R5[0] = cy * cz;                 R5[1] = -cy * sz;                R5[2] = sy;
  R5[3] = sx * sy * cz + cx * sz;  R5[4] = cx * cz - sx * sy * sz;  R5[5] = -sx * cy;
  R5[6] = sx * sz - cx * sy * cz;  R5[7] = cx * sy * sz + sx * cz;  R5[8] = cx * cy;
  const T P[3] = {q0 + QC(base_t[0], base_t[0]), q1 + QC(base_t[1], base_t[1]), q2 + QC(base_t[2], base_t[2])};
  const T A3[3] = {T(1), T(0), T(0)}, A4[3] = {T(0), cx, sx}, A5[3] = {sy, -sx * cy, cx * cy};
  // the six root motion axes (angular | linear): prismatic e_x, e_y, e_z; revolute (A | P x A)
  T pA3[3], pA4[3], pA5[3];
  cross3(P, A3, pA3);
  cross3(P, A4, pA4);
  cross3(P, A5, pA5);
  T v5[6], a5[6];  // velocity and bias acceleration (a0) of the root body
  {
    const T d0 = xr[nq + 0], d1 = xr[nq + 1], d2 = xr[nq + 2], d3 = xr[nq + 3], d4 = xr[nq + 4], d5 = xr[nq + 5];
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
      a5[3 + k] = (l3[k] + l4[k] + l5[k]) - QC(grav[k], grav[k]);
    }
  }
  QUAD_STAMP(3, a5[5]);
  // ---- the legs: one segmented prefix scan along each quad (chain-local products of the joint transforms, then the
  //      root's pose in front; prefix sums of the joint velocities and of the velocity-product accelerations)
  T R[9], p[3], sw[6], vJ[6], v[6], a0[6];
  {
    T Rl[9], pl[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rl[k] = Rp[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) pl[k] = tp[k];
    static_for<0, 2>([&](auto dc) {
      constexpr int D = 1 << decltype(dc)::value;
      const bool take = pos >= D;
      T Rq[9], pq[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const T sh = dpp_shr<D>(Rl[k]);
        Rq[k] = take ? sh : ((k == 0 || k == 4 || k == 8) ? T(1) : T(0));
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const T sh = dpp_shr<D>(pl[k]);
        pq[k] = take ? sh : T(0);
      }
      T Rn[9], r[3];
      mat3_mul(Rq, Rl, Rn);
      mat3_mulv(Rq, pl, r);
#pragma unroll
      for (int k = 0; k < 9; ++k) Rl[k] = Rn[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) pl[k] = pq[k] + r[k];
    });
    T r[3];
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
#pragma unroll
    for (int k = 0; k < 6; ++k) vJ[k] = sw[k] * qd;
    T pre[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) pre[k] = vJ[k];
    static_for<0, 2>([&](auto dc) {
      constexpr int D = 1 << decltype(dc)::value;
      const bool take = pos >= D;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const T sh = dpp_shr<D>(pre[k]);
        pre[k] += take ? sh : T(0);
      }
    });
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = v5[k] + pre[k];
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
    for (int k = 0; k < 6; ++k) pre[k] = cb[k];
    static_for<0, 2>([&](auto dc) {
      constexpr int D = 1 << decltype(dc)::value;
      const bool take = pos >= D;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const T sh = dpp_shr<D>(pre[k]);
        pre[k] += take ? sh : T(0);
      }
    });
#pragma unroll
    for (int k = 0; k < 6; ++k) a0[k] = a5[k] + pre[k];
  }
  // my world motion axis, for the rows of the contacts (lane-dependent reads in phase J)
  {
    T *const swl = E + O.swl;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (dofl) swl[(3 * leg + pos) * 7 + k] = sw[k];
    }
  }

  QUAD_STAMP(4, sw[5]);
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
  QUAD_STAMP(5, tid);
  // ---- I. narrowphase: the contact points are the toes' own lanes (plane x sphere, contact_point.hpp:96-131)
  int na = 0;
  {
    T *const cpx = E + O.cp;
    const T rad = QC(cp_radius[leg], cp_radius[leg]);
    const T loc[3] = {QC(cp_local[0][leg], cp_local[0][leg]), QC(cp_local[1][leg], cp_local[1][leg]), QC(cp_local[2][leg], cp_local[2][leg])};
    T ctr[3];
    mat3_mulv(R, loc, ctr);
    ctr[0] += p[0];
    ctr[1] += p[1];
    ctr[2] += p[2];
    const T n[3] = {QC(plane_n[0], plane_n[0]), QC(plane_n[1], plane_n[1]), QC(plane_n[2], plane_n[2])};
    const T t = -((-dot3(ctr, n)) + QC(plane_c, plane_c));
    const T dist = t - rad;
    const bool act = valid && pos == 3 && dist < T(0);
    const unsigned long long bal = __ballot(act);
    const unsigned mine = (unsigned)((bal >> ((grp & 3) * 16)) & 0xFFFFull);
    const int pre = __popc(mine & ((1u << lane) - 1u));
    if (act) {
      cpx[0 * 4 + pre] = ctr[0] - rad * n[0];
      cpx[1 * 4 + pre] = ctr[1] - rad * n[1];
      cpx[2 * 4 + pre] = ctr[2] - rad * n[2];
      cpx[3 * 4 + pre] = dist;
      cpx[4 * 4 + pre] = (T)leg;
    }
    na = __popc(mine);
  }
  int NA = na;
#pragma unroll
  for (int msk = 16; msk < 64; msk <<= 1) {
    const int o = __shfl_xor(NA, msk, 64);
    NA = o > NA ? o : NA;
  }
  NA = __builtin_amdgcn_readfirstlane(NA);

  QUAD_STAMP(6, tid);
  // ---- M1. visual poses of y, from the PRE-step X_world (locomotion_contact_simulation.h:281-299): visual 1 + lane is
  //          my link's (DevModel::quad checks the order); visual 0 — the root body's — goes out on the toe lane of leg 3
  // where this step's y record goes: the slot of a y ring (every step of a step-loop launch), else the handle's y record
  // (the last step); the last step of a ring launch leaves its record in the handle's y record as well
  const int ystr = ctl.y_stride;
  const int out_dim = QC(output_dim, output_dim);
  TR *yo = nullptr, *yo2 = nullptr;
  int yend = ystr, yend2 = out_dim;
  if (LOOP && ctl.y_ring != nullptr) {
    yo = (TR *)ctl.y_ring + ((size_t)pos_y * ctl.ring_envs + env) * ystr;
    if (last && y_out != nullptr) yo2 = y_out + (size_t)env * out_dim;
  } else if (last && y_out != nullptr) {
    yo = y_out + (size_t)env * (LOOP ? out_dim : ystr);
    yend = LOOP ? out_dim : ystr;
  }
  const int nv = QC(num_visuals, num_visuals);
  if (valid && yo != nullptr && nv > 0) {
    QUAD_LAUNDER_MODEL(md3, mdl)  // (see the rigid inertia above: the visuals' constants are fetched where they are used)
    // (step-loop launches: the table says which products are with zeros and ones — wave-uniform branches)
    const int vflags = LOOP ? __builtin_amdgcn_readfirstlane(CT->vis_flags) : 0;
    auto pose_out = [&](const T *Rl, const T *pl, int k, auto rootc) {
      constexpr bool ROOT = decltype(rootc)::value;
      T Ro[9], po[3] = {T(0), T(0), T(0)}, qo[4];
      if (ROOT && (vflags & 2)) {
#pragma unroll
        for (int c = 0; c < 9; ++c) Ro[c] = Rl[c];
      } else {
        T Rv[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) Rv[c] = LOOP ? CT->vis_X[c][k] : md3->vis_X[c][k];
        mat3_mul(Rl, Rv, Ro);
      }
      if (!(vflags & 1)) {
        T pv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) pv[c] = LOOP ? CT->vis_X[9 + c][k] : md3->vis_X[9 + c][k];
        mat3_mulv(Rl, pv, po);
      }
      matrix_to_quat(Ro, qo);
      TR *o = yo + (nq + nd) + 7 * k;
      o[0] = (TR)(pl[0] + po[0]);
      o[1] = (TR)(pl[1] + po[1]);
      o[2] = (TR)(pl[2] + po[2]);
      o[3] = (TR)qo[0];
      o[4] = (TR)qo[1];
      o[5] = (TR)qo[2];
      o[6] = (TR)qo[3];
      if (yo2 != nullptr) {
        TR *o2 = yo2 + (nq + nd) + 7 * k;
        o2[0] = (TR)(pl[0] + po[0]);
        o2[1] = (TR)(pl[1] + po[1]);
        o2[2] = (TR)(pl[2] + po[2]);
        o2[3] = (TR)qo[0];
        o2[4] = (TR)qo[1];
        o2[5] = (TR)qo[2];
        o2[6] = (TR)qo[3];
      }
    };
    pose_out(R, p, 1 + lane, std::false_type{});
    if (lane == 15) pose_out(R5, P, 0, std::true_type{});
  }

  T Ic[10], fc[6];
  {
    // (my link's rigid inertia is fetched HERE, through a pointer laundered at this point: requested at the top of the step
    //  "under the latency of the record" the scheduler kept 13 values alive — or spilled — through the kinematics)
    QUAD_LAUNDER_MODEL(md2, mdl)
    T Il[9], com_l[3];
    const T mass_l = LOOP ? CT->mass[lane] : md2->mass[li];
#pragma unroll
    for (int k = 0; k < 3; ++k) com_l[k] = LOOP ? CT->com[k][lane] : md2->com[k][li];
#pragma unroll
    for (int k = 0; k < 9; ++k) Il[k] = LOOP ? CT->inertia[k][lane] : md2->inertia[k][li];
    rigid(R, p, mass_l, com_l, Il, v, a0, Ic, fc);
  }
  T It[10], ft[6];  // the root body's; below: + the legs' composites = the whole robot's
  {
    QUAD_LAUNDER_MODEL(md4, mdl)
    const T com5[3] = {LOOP ? CT->com5[0] : md4->com[0][5], LOOP ? CT->com5[1] : md4->com[1][5], LOOP ? CT->com5[2] : md4->com[2][5]};
    T I5[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) I5[k] = LOOP ? CT->inertia5[k] : md4->inertia[k][5];
    rigid(R5, P, LOOP ? CT->mass5 : md4->mass[5], com5, I5, v5, a5, It, ft);
  }

  QUAD_STAMP(7, tid);
  // ---- E. composite inertia / bias force (CRBA, mass_matrix.hpp:39-56): suffix sums along every quad; the chain heads'
  //         totals reach the root by two row rotations and come back to every lane by a quad broadcast
  static_for<0, 2>([&](auto dc) {
    constexpr int D = 1 << decltype(dc)::value;
    const T recv = (pos + D < 4) ? T(1) : T(0);
#pragma unroll
    for (int k = 0; k < 6; ++k) fc[k] += recv * dpp_shl<D>(fc[k]);
#pragma unroll
    for (int k = 0; k < 10; ++k) Ic[k] += recv * dpp_shl<D>(Ic[k]);
  });
  {
    const T head = pos == 0 ? T(1) : T(0);
    auto legs_total = [&](T x) -> T {
      // (rotation by 8 FIRST: legs l and l + 2 then hold the same pair sum, and the second round adds the same two pair
      //  sums on every leg — IEEE addition is commutative, not associative: with 4 before 8 the legs' totals differed in
      //  the last bit, and with them everything "redundant on every lane" behind this line: an environment's result then
      //  depended on which lane solved its constraint rows, i.e. on its wavefront-mates' contact counts)
      T s = head * x;
      s = dpp_add<0x128>(s);  // row_ror:8
      s = dpp_add<0x124>(s);  // row_ror:4   (lanes of position 0 now hold the sum over the four legs)
      return quad_bcast<0>(s);
    };
#pragma unroll
    for (int k = 0; k < 10; ++k) It[k] += legs_total(Ic[k]);
#pragma unroll
    for (int k = 0; k < 6; ++k) ft[k] += legs_total(fc[k]);
  }
  // F = Ic s, C = s . f of my dof
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
  T Fc[6];
  times_inertia(Ic, sw, Fc);
  const T Cb = dot6(sw, fc);
  // the six root axes, as (angular | linear)
  const T ax0[6] = {T(0), T(0), T(0), T(1), T(0), T(0)}, ax1[6] = {T(0), T(0), T(0), T(0), T(1), T(0)},
          ax2[6] = {T(0), T(0), T(0), T(0), T(0), T(1)};
  const T ax3[6] = {A3[0], A3[1], A3[2], pA3[0], pA3[1], pA3[2]}, ax4[6] = {A4[0], A4[1], A4[2], pA4[0], pA4[1], pA4[2]},
          ax5[6] = {A5[0], A5[1], A5[2], pA5[0], pA5[1], pA5[2]};
  const T *const axr[6] = {ax0, ax1, ax2, ax3, ax4, ax5};
  T Cr[6];  // bias forces of the root dofs
#pragma unroll
  for (int r = 0; r < 6; ++r) Cr[r] = dot6(axr[r], ft);

  QUAD_STAMP(8, Cr[5]);
  // ---- G. M in leaves-first order.  My row of my leg's 3 x 3 block (entries against the dofs in front of me in the
  //         chain: M[i][j] = F_i . s_j for j an ancestor of i, mass_matrix.hpp:87-109) and my coupling to the root dofs
  T Bm[3], Cc[6];
  {
    T s0[6], s1[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      s0[k] = quad_bcast<0>(sw[k]);
      s1[k] = quad_bcast<1>(sw[k]);
    }
    Bm[0] = dot6(Fc, s0);
    Bm[1] = dot6(Fc, s1);
    Bm[2] = dot6(Fc, sw);
#pragma unroll
    for (int r = 0; r < 6; ++r) Cc[r] = dofl ? dot6(Fc, axr[r]) : T(0);
  }
  QUAD_STAMP(9, tid);
  // ---- H. LDL^T.  The leg block: its six entries to every lane of the quad, factorised there
  T l10, l20, l21, id0, id1, id2, sq0, sq1, sq2;
  {
    const T b00 = quad_bcast<0>(Bm[0]);
    const T b10 = quad_bcast<1>(Bm[0]), b11 = quad_bcast<1>(Bm[1]);
    const T b20 = quad_bcast<2>(Bm[0]), b21 = quad_bcast<2>(Bm[1]), b22 = quad_bcast<2>(Bm[2]);
    // (1 / sqrt(d) by the hardware estimate + two Newton steps, 1 / d as its square: a division AND a square root per pivot —
    //  ~25 instructions — for 8; a lone wavefront pays per instruction: tools/ubench/lone_wave_latency.hip)
    sq0 = rsqrt_full<T>(b00);
    id0 = sq0 * sq0;
    l10 = b10 * id0;
    l20 = b20 * id0;
    const T d1 = b11 - l10 * b10;
    sq1 = rsqrt_full<T>(d1);
    id1 = sq1 * sq1;
    const T u21 = b21 - l20 * b10;  // = l21 d1
    l21 = u21 * id1;
    const T d2 = b22 - l20 * b20 - l21 * u21;
    sq2 = rsqrt_full<T>(d2);
    id2 = sq2 * sq2;
  }
  const T my_id = pos == 0 ? id0 : (pos == 1 ? id1 : id2);
  const T my_sq = pos == 0 ? sq0 : (pos == 1 ? sq1 : sq2);
  // the coupling rows: W_a = C_a - sum_{a' < a} L[a][a'] W_a', L_c = W / d
  T W[6], Lc[6];
  {
#pragma unroll
    for (int r = 0; r < 6; ++r) W[r] = Cc[r];
    const T m1 = pos == 1 ? l10 : (pos == 2 ? l20 : T(0));
#pragma unroll
    for (int r = 0; r < 6; ++r) W[r] -= m1 * quad_bcast<0>(W[r]);
    const T m2 = pos == 2 ? l21 : T(0);
#pragma unroll
    for (int r = 0; r < 6; ++r) W[r] -= m2 * quad_bcast<1>(W[r]);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      W[r] = dofl ? W[r] : T(0);
      Lc[r] = W[r] * my_id;
    }
  }
  {
    // (the toe lanes — fixed joints — have L_c = W = 0: no slot.  Twelve slots instead of sixteen, here and for the motion
    //  axes, are what brings a workgroup's four environments under 20 KB of LDS: EIGHT workgroups per compute unit, so
    //  that the 2048 workgroups of config 4 are resident at once — at seven per CU the last 256 ran behind the others,
    //  28.5 us per step against 24.8 for 7168 environments, tools/quad_occupancy_sweep.sh)
    T *const lcw = E + O.lcw + (3 * leg + pos) * QuadLds::LCW;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      if (dofl) {
        lcw[r] = Lc[r];
        lcw[6 + r] = W[r];
      }
    }
    if (pos == 0) {
      T *const lf = E + O.legf + leg * 9;
      lf[0] = l10; lf[1] = l20; lf[2] = l21;
      lf[3] = id0; lf[4] = id1; lf[5] = id2;
      lf[6] = sq0; lf[7] = sq1; lf[8] = sq2;
    }
    // the six root axes for the lane-parallel root block (every lane the same values to the same slots)
    T *const axl = E + O.ax;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int k = 0; k < 6; ++k) axl[r * 7 + k] = axr[r][k];
  }
  QUAD_SYNC();
  // the root's Schur complement S[r][r'] = s_r . (It s_r') - sum_lanes L_c[r] W[r'], entry e = r (r + 1) / 2 + r' on lane
  // e (two passes: 21 entries, 16 lanes)
  {
    const T *const lcw = E + O.lcw;
    const T *const axl = E + O.ax;
    T *const Sl_ = E + O.S;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      int e = lane + 16 * pass;
      e = e < 21 ? e : 20;
      int r = 0;
#pragma unroll
      for (int k = 1; k < 6; ++k) r += e >= (k * (k + 1)) / 2 ? 1 : 0;
      const int rp = e - (r * (r + 1)) / 2;
      T sr[6], srp[6], Fr[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        sr[k] = axl[r * 7 + k];
        srp[k] = axl[rp * 7 + k];
      }
      times_inertia(It, srp, Fr);
      T acc = dot6(sr, Fr);
#pragma unroll
      for (int l = 0; l < 12; ++l) acc -= lcw[l * QuadLds::LCW + r] * lcw[l * QuadLds::LCW + 6 + rp];
      Sl_[e] = acc;
    }
  }
  QUAD_SYNC();
  // ... factorised redundantly on every lane: Ls (strictly lower, row-major packed), 1 / D
  T Ls[15], ids[6], sq_ids[6];
  {
    const T *const Sl_ = E + O.S;
    T Sm[21];
#pragma unroll
    for (int e = 0; e < 21; ++e) Sm[e] = Sl_[e];
    // right-looking on the packed lower triangle: S(r, c) at r (r + 1) / 2 + c
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

  // the two substitutions with the factors, used three times (forward dynamics, contact impulse)
  //   leaves-first system  [B  C^T; C  R] = L D L^T,  unknown (x_leg on the dof lanes, x_root[6] on every lane)
  auto solve = [&](T b_leg, const T *b_root, T &x_leg, T *x_root) {
    // forward: legs, quad-local
    T y = b_leg;
    {
      const T m1 = pos == 1 ? l10 : (pos == 2 ? l20 : T(0));
      y -= m1 * quad_bcast<0>(y);
      const T m2 = pos == 2 ? l21 : T(0);
      y -= m2 * quad_bcast<1>(y);
      y = dofl ? y : T(0);
    }
    // root: y_r = b_r - sum_lanes L_c[r] y, then the root block's own forward substitution
    T yr[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) yr[r] = b_root[r] - group_sum<T, 16>(Lc[r] * y);
    static_for<1, 6>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      static_for<0, r>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        yr[r] -= Ls[(r * (r - 1)) / 2 + c] * yr[c];
      });
    });
    // diagonal, backward: root first
#pragma unroll
    for (int r = 0; r < 6; ++r) x_root[r] = yr[r] * ids[r];
    static_for<0, 5>([&](auto ic) {
      constexpr int r = 4 - decltype(ic)::value;
      static_for<r + 1, 6>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        x_root[r] -= Ls[(c * (c - 1)) / 2 + r] * x_root[c];
      });
    });
    // legs: x = y / d - L_c . x_root - sum_{a'' > a} L[a''][a] x_a''
    T x = y * my_id;
#pragma unroll
    for (int r = 0; r < 6; ++r) x -= Lc[r] * x_root[r];
    {
      const T x2 = quad_bcast<2>(x);
      x -= (pos == 1 ? l21 : (pos == 0 ? l20 : T(0))) * x2;
      const T x1 = quad_bcast<1>(x);
      x -= (pos == 0 ? l10 : T(0)) * x1;
    }
    x_leg = dofl ? x : T(0);
  };

  QUAD_STAMP(10, tid);
  // ---- F. forward dynamics qdd = M^-1 (tau - C), integrate_euler_qdd (integrator.hpp:169-181)
  T qd_new, qdr_new[6];
  {
    T br[6], xl, xrt[6];
    // (the root links are unactuated; their joint stiffness / damping is zero: DevModel::quad)
#pragma unroll
    for (int r = 0; r < 6; ++r) br[r] = -Cr[r];
    solve(dofl ? tau - Cb : T(0), br, xl, xrt);
    qd_new = qd + xl * dt;
#pragma unroll
    for (int r = 0; r < 6; ++r) qdr_new[r] = xr[nq + r] + xrt[r] * dt;
  }

  QUAD_STAMP(11, qd_new);
  // ---- J, K, L. contacts: rows of the penetrating toes (wave-uniform slots: NA = the largest count among the wavefront's
  //      environments; rows a: normals, NA + a: tangents 1, 2 NA + a: tangents 2, the reference's order under compaction)
  if (NA > 0) {
    T *const qdp = E + O.qdp;
    qdp[lane] = qd_new;
    QUAD_SYNC();
    const T *const cpx = E + O.cp;
    const T *const swl = E + O.swl;
    const T *const lcw = E + O.lcw;
    T *const Zs = E + O.Z;
    T *const rws = E + O.rws;  // [4][12]: b | 1 / (G + cfm) | G | leg
    T *const xs = E + O.xs;
    QUAD_LAUNDER_MODEL(md5, mdl)  // (the contact frame and the solver's scalars are fetched here, not at the top of the step)
    const T nb[3] = {LOOP ? CT->nb[0] : md5->nb[0], LOOP ? CT->nb[1] : md5->nb[1], LOOP ? CT->nb[2] : md5->nb[2]};
    const T t1v[3] = {LOOP ? CT->t1[0] : md5->t1[0], LOOP ? CT->t1[1] : md5->t1[1], LOOP ? CT->t1[2] : md5->t1[2]};
    const T t2v[3] = {LOOP ? CT->t2[0] : md5->t2[0], LOOP ? CT->t2[1] : md5->t2[1], LOOP ? CT->t2[2] : md5->t2[2]};
    const T cfm = LOOP ? CT->cfm : md5->cfm, erp_dt = LOOP ? CT->erp_over_dt : md5->erp_over_dt, rest = LOOP ? CT->restitution : md5->restitution, mu = LOOP ? CT->friction : md5->friction;
    {
      // lane == row, CONTACT-major: lane 3 a + t solves row t (normal, tangent 1, tangent 2) of contact slot a and stores it
      // at index 3 a + t — an assignment that does not depend on NA, i.e. on the wavefront-mates' contact counts (with the
      // kind-major index t NA + a of the sweep's order an environment's rows moved to other lanes when a mate had more
      // contacts: bit-level differences, caught by the permutation test of tests/test_hip_parity.py)
      const int a = (lane * 11) >> 5;  // lane / 3 for lane < 16
      const int t = lane - 3 * a;
      const bool real = a < na;        // (lanes 12 .. 15: a == 4, never real)
      const int ac = a < 4 ? a : 0;
      const T Pc[3] = {cpx[0 * 4 + ac], cpx[1 * 4 + ac], cpx[2 * 4 + ac]};
      const T dist = cpx[3 * 4 + ac];
      const int cl = real ? (int)cpx[4 * 4 + ac] : 0;  // the leg of the row's contact
      const T e[3] = {t == 0 ? nb[0] : (t == 1 ? t1v[0] : t2v[0]), t == 0 ? nb[1] : (t == 1 ? t1v[1] : t2v[1]),
                      t == 0 ? nb[2] : (t == 1 ? t1v[2] : t2v[2])};
      // column of the point Jacobian along e: e . s_lin + P . (e x s_ang)   (jacobian.hpp:56-72)
      auto jcol = [&](const T *s) -> T {
        T c[3];
        cross3(e, s, c);
        return dot3(e, s + 3) + dot3(Pc, c);
      };
      T z[3], zr[6];
      T vrow = T(0);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        T s[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) s[c] = swl[(3 * cl + k) * 7 + c];
        z[k] = jcol(s);
        vrow += z[k] * qdp[4 * cl + k];
      }
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) {
        zr[rr] = jcol(axr[rr]);
        vrow += zr[rr] * qdr_new[rr];
      }
      // rel_vel = vel_a - vel_b = -J qd:  b_n = -(1 + e) n.rel_vel - erp dist / dt,  b_t = -t.rel_vel
      const T brow = t == 0 ? (T(1) + rest) * vrow - erp_dt * dist : vrow;
      // forward substitution L z = J^T, leaves first: the contact's leg, then the root rows
      const T *const lf = E + O.legf + cl * 9;
      const T c10 = lf[0], c20 = lf[1], c21 = lf[2];
      z[1] -= c10 * z[0];
      z[2] -= c20 * z[0] + c21 * z[1];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) zr[rr] -= lcw[(3 * cl + k) * QuadLds::LCW + rr] * z[k];
      }
      static_for<1, 6>([&](auto rc) {
        constexpr int rr = decltype(rc)::value;
        static_for<0, rr>([&](auto cc) {
          constexpr int c = decltype(cc)::value;
          zr[rr] -= Ls[(rr * (rr - 1)) / 2 + c] * zr[c];
        });
      });
      // z~ = D^-1/2 z, G = z~ . z~
      T g = T(0);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        z[k] *= lf[6 + k];
        g += z[k] * z[k];
      }
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) {
        zr[rr] *= sq_ids[rr];
        g += zr[rr] * zr[rr];
      }
      const T ai = real ? rcp_full<T>(g + cfm) : T(0);
      if (lane < 12) {
#pragma unroll
        for (int k = 0; k < 3; ++k) Zs[lane * QuadLds::ZW + k] = real ? z[k] : T(0);
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) Zs[lane * QuadLds::ZW + 3 + rr] = real ? zr[rr] : T(0);
        rws[0 * 12 + lane] = real ? brow : T(0);
        rws[1 * 12 + lane] = ai;
        rws[2 * 12 + lane] = real ? g : T(0);
        rws[3 * 12 + lane] = (T)cl;
      }
    }
    QUAD_SYNC();
    // projected Gauss-Seidel (mb_constraint_solver.hpp:101-142) on u~ = sum_r z~_r x_r: the leg part on the dof lanes, the
    // root part on every lane
    T u = T(0), ur[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    const int iters = LOOP ? CT->pgs_iterations : md5->pgs_iterations;
    const T my_leg = (T)leg;
    // The reference's row order: normals, tangents 1, tangents 2, each by contact (rows live at 3 a + t).  A row's operands —
    // its z~ (my leg entry, six root entries), b, 1 / (G + cfm), G — are requested from LDS ONE ROW AHEAD, at the top of the
    // row before; the normal impulses the friction bounds depend on stay in registers (contact slots unrolled: four toes), and
    // with one PGS iteration (every shipped model) nothing of the sweep is written back to LDS: as a loop of read -> wait ->
    // compute -> write -> fence a row was three LDS round trips a lone wavefront sat out — 36 of them for four toes on the ground
    struct RowOps {
      T zl, zr[6], b, a, g;
    };
    const T my_leg_or_none = dofl ? my_leg : T(-1);  // (a toe lane matches no row's leg)
    auto load_row = [&](int tk, int ak) {
      const int r = 3 * ak + tk;
      RowOps o;
      // (my leg entry times a 0 / 1 mask, not a select: a select lets the compiler sink the read of the entry under the comparison
      //  of the row's leg — a second LDS round trip per row; a toe lane's `pos` picks the row's first root entry, masked out)
      o.zl = Zs[r * QuadLds::ZW + pos] * (rws[3 * 12 + r] == my_leg_or_none ? T(1) : T(0));
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) o.zr[rr] = Zs[r * QuadLds::ZW + 3 + rr];
      o.b = rws[0 * 12 + r];
      o.a = rws[1 * 12 + r];
      o.g = rws[2 * 12 + r];
      return o;
    };
    T xnn[4] = {T(0), T(0), T(0), T(0)};  // the normal rows' impulses of this sweep, by contact slot
    // one sweep; SINGLE: the only one (x starts from 0, nothing is written back — every shipped model), else sweep `it` of several
    auto sweep = [&](int it, auto singlec) {
      constexpr bool SINGLE = decltype(singlec)::value;
      RowOps nxt = load_row(0, 0);
      for (int tk = 0; tk < 3; ++tk) {
        const bool is_n = tk == 0;
        static_for<0, 4>([&](auto akc) {
          constexpr int ak = decltype(akc)::value;
          if (ak < NA) {  // (wave-uniform)
            const RowOps cur = nxt;
            {  // the row after this one: (tk, ak + 1), else (tk + 1, 0); behind the last row: the first again (unused)
              const bool wrap = ak + 1 >= NA;
              const int tkn = wrap ? (tk < 2 ? tk + 1 : 0) : tk;
              nxt = load_row(tkn, wrap ? 0 : ak + 1);
            }
            const int r = 3 * ak + tk;
            T x_old = T(0);
            if constexpr (!SINGLE) x_old = it > 0 ? xs[r] : T(0);
            const T sdep = is_n ? T(0) : xnn[ak];
            T jw = group_sum<T, 16>(cur.zl * u);
#pragma unroll
            for (int rr = 0; rr < 6; ++rr) jw += cur.zr[rr] * ur[rr];
            T xn;
            if constexpr (SINGLE) xn = (cur.b - jw) * cur.a;
            else xn = (cur.b - (jw - cur.g * x_old)) * cur.a;
            const T sc = sdep < T(0) ? T(0) : sdep;  // where_lt(s, 0, 0, s)
            const T lo = is_n ? T(0) : -mu * sc;
            const T hi = is_n ? T(100000) : mu * sc;
            xn = max_t<T>(xn, lo);
            xn = min_t<T>(xn, hi);
            T dx = xn;
            if constexpr (!SINGLE) dx = xn - x_old;
            u += cur.zl * dx;
#pragma unroll
            for (int rr = 0; rr < 6; ++rr) ur[rr] += cur.zr[rr] * dx;
            xnn[ak] = is_n ? xn : xnn[ak];
            if constexpr (!SINGLE) {  // (a further sweep reads the impulses back)
              xs[r] = xn;
              QUAD_SYNC();
            }
          }
        });
      }
    };
    if (iters == 1) {
      sweep(0, std::true_type{});
    } else {
      for (int it = 0; it < iters; ++it) sweep(it, std::false_type{});
    }
    // delta_qd = M^-1 J^T p = L^-T D^-1/2 u~   (mb_constraint_solver.hpp:476-496: qd_b -= delta_qd)
    {
      T w = dofl ? u * my_sq : T(0);
      T wr[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) wr[r] = ur[r] * sq_ids[r];
      static_for<0, 5>([&](auto ic) {
        constexpr int r = 4 - decltype(ic)::value;
        static_for<r + 1, 6>([&](auto cc) {
          constexpr int c = decltype(cc)::value;
          wr[r] -= Ls[(c * (c - 1)) / 2 + r] * wr[c];
        });
      });
#pragma unroll
      for (int r = 0; r < 6; ++r) w -= Lc[r] * wr[r];
      {
        const T w2 = quad_bcast<2>(w);
        w -= (pos == 1 ? l21 : (pos == 0 ? l20 : T(0))) * w2;
        const T w1 = quad_bcast<1>(w);
        w -= (pos == 0 ? l10 : T(0)) * w1;
      }
      qd_new -= dofl ? w : T(0);
#pragma unroll
      for (int r = 0; r < 6; ++r) qdr_new[r] -= wr[r];
    }
  }

  QUAD_STAMP(12, qd_new);
  // ---- M. integrate_euler: q += qd dt (integrator.hpp:126-131); the new state into the LDS record
  QUAD_SYNC();
  {
    const T q_old0 = xr[0];
    // root coordinates: lane r < 6 stores root value r
    T qr_sel = qdr_new[0];
#pragma unroll
    for (int r = 1; r < 6; ++r) qr_sel = lane == r ? qdr_new[r] : qr_sel;
    const T qn_root = xr[lane < 6 ? lane : 0] + qr_sel * dt;
    QUAD_SYNC();
    if (lane < 6) {
      xr[lane] = qn_root;
      xr[nq + lane] = qr_sel;
    }
    if (dofl) {
      xr[dq] = q + qd_new * dt;
      xr[nq + dq] = qd_new;
    }
    if (lane == 0) xr[in_dim] = q_old0;  // x_{t-1} (the Ant's reward reads it; kept for the record's sake)
  }
  if constexpr (LOOP) {
    // The next step's actions into the record's action slots (dead since the PD block) HERE, in front of this step's record
    // stores — not at the top of the next step: loads and stores return through one in-order counter on gfx9 and the
    // record stores sit in loops, so a wait for this load behind them is a wait for every one of them (vmcnt(0)): a full
    // HBM write latency per step — what kept the step-loop form 2.7 us per step behind the single-step graphs.
    if (ctl.act_pool != nullptr && !last && lane < adim) xr[nq + nd + lane] = next_act;
  }
  QUAD_SYNC();
  QUAD_STAMP(13, tid);
  // ---- y record: q | qd | (visual poses: M1) | up.z | zero padding
  if (valid && yo != nullptr) {
    // (every LDS / table read of the record in front of its first store: as a loop of read -> wait -> store the 36 state
    //  values were three LDS round trips, the two constants two more — a lone wavefront sits them out one after the other)
    static_assert(nq + nd == 36, "three values per lane: lanes 0..15 | 16..31 | 32..35");
    const T s0 = xr[lane], s1 = xr[16 + lane], s2 = xr[32 + (lane & 3)];
    const bool packs = QC(pack_visuals, pack_visuals) != 0;
    const T upz = QC(base_R8, base_R[8]);  // up_dot_world_z (fixed base)
    auto y_state = [&](TR *y, int end) {
      y[lane] = (TR)s0;
      y[16 + lane] = (TR)s1;
      if (lane < 4) y[32 + lane] = (TR)s2;
      int tail = nq + nd;
      if (packs) {
        tail += 7 * nv;
        if (lane == 0) y[tail] = (TR)upz;
        tail += 1;
      }
      for (int i = tail + lane; i < end; i += 16) y[i] = TR(0);
    };
    y_state(yo, yend);
    if (yo2 != nullptr) y_state(yo2, yend2);
  }
  QUAD_STAMP(14, tid);
  // ---- N. reward / done (laikago_environment2.h:130-171; ant_environment2.h:75-106)
  {
    // (Laikago's up . z: the reference goes rpy -> quaternion -> matrix entry (2, 2) = 1 - 2 (qx^2 + qy^2) / |q|^2, which IS
    //  cos(roll) cos(pitch) — two cosines and a product instead of three half-angle sincos and the quaternion's 30 products;
    //  to rounding the same number, and `done` compares it with 0.6)
    T rs = T(0), rc = T(1);
    quad_sincos(lane < 2 ? xr[3 + lane] : T(0), &rs, &rc);
    const T c1 = dpp_bcast<1>(rc);
    if (lane == 0) {
      bool done = false;
      T reward = T(0);
      const int rm = QC(reward_mode, reward_mode);
      if (rm == TDS_REWARD_ANT) {
        const T vel_x = (xr[0] - xr[in_dim]) / dt;
        done = xr[2] < T(0.26);
        reward = done ? T(0) : vel_x;
      } else if (rm == TDS_REWARD_LAIKAGO) {
        const T up = rc * c1;
        done = (up < T(0.6)) || (xr[2] < T(0.2));
        reward = done ? T(0) : xr[0];
      }
      xr[in_dim + 1] = done ? T(1) : T(0);
      xr[in_dim + 2] = reward;
    }
  }
  QUAD_SYNC();
  QUAD_STAMP(15, tid);
  // ---- auto_reset_when_done through the reset pool (ctl.pool; ars_vectorized_environment.h:262-277): a done environment
  //      takes its next pre-settled state — y, reward and done describe the terminal step, the observation and the state
  //      the fresh environment
  if (ctl.pool != nullptr && valid && xr[in_dim + 1] != T(0)) {
    const unsigned c = ctl.reset_count[env];
    const TR *const src = (const TR *)ctl.pool + ((size_t)(c % (unsigned)ctl.pool_depth) * ctl.pool_envs + env) * (nq + nd);
    T v0 = T(0), v1 = T(0), v2 = T(0);
    if (lane < nq + nd) v0 = (T)src[lane];
    if (lane + 16 < nq + nd) v1 = (T)src[lane + 16];
    if (lane + 32 < nq + nd) v2 = (T)src[lane + 32];
    QUAD_SYNC();
    if (lane < nq + nd) xr[lane] = v0;
    if (lane + 16 < nq + nd) xr[lane + 16] = v1;
    if (lane + 32 < nq + nd) xr[lane + 32] = v2;
    if (lane == 0) ctl.reset_count[env] = c + 1u;
  }
  QUAD_SYNC();
  QUAD_STAMP(16, tid);
  // ---- [obs | reward | done] record (obs[0] = obs[1] = 0, ars_vectorized_environment.h:283-288): the slot of an obs ring
  //      (every step of a step-loop launch; floats on the multi-GPU wire format) and / or the caller's record (last step)
  if (valid) {
    const int w = nq + nd + 2;
    // (the record's LDS reads in front of its stores, as in the y record: lanes 0..15 | 16..31 | 32..35 state, lane 4 / 5 of
    //  the third group reward / done)
    const T o0 = xr[lane], o1 = xr[16 + lane];
    const T o2 = xr[lane < 4 ? 32 + lane : (lane == 4 ? in_dim + 2 : in_dim + 1)];
    if (LOOP && ctl.obs_ring != nullptr) {
      const size_t at = ((size_t)pos_obs * ctl.obs_envs + env) * w;
      const T z0 = lane < 2 ? T(0) : o0;
      if (ctl.ring_flags & TDS_RING_OBS_F32) {
        float *const o = (float *)ctl.obs_ring + at;
        o[lane] = (float)z0;
        o[16 + lane] = (float)o1;
        if (lane < 6) o[32 + lane] = (float)o2;
      } else {
        TR *const o = (TR *)ctl.obs_ring + at;
        o[lane] = (TR)z0;
        o[16 + lane] = (TR)o1;
        if (lane < 6) o[32 + lane] = (TR)o2;
      }
    }
    if (last) {
      if (obs_out != nullptr) {
        TR *const o = obs_out + (size_t)env * w;
        o[lane] = lane < 2 ? TR(0) : (TR)o0;
        o[16 + lane] = (TR)o1;
        if (lane < 6) o[32 + lane] = (TR)o2;
      }
      if (x_feedback != nullptr) {
        TR *const f = x_feedback + (size_t)env * in_dim;
        f[lane] = (TR)o0;
        f[16 + lane] = (TR)o1;
        if (lane < 4) f[32 + lane] = (TR)o2;
      }
    }
  }
  QUAD_STAMP(17, tid);
#ifdef TDS_QUAD_PROF
  if (prof_on && (tid & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 18; ++k) tds_quad_prof_buf[k] = prof_t[k];
    tds_quad_prof_buf[18] = (unsigned long long)NA;
  }
#endif
  if constexpr (LOOP) {
    QUAD_SYNC();
    pos_act = pos_act + 1 == n_act ? 0 : pos_act + 1;
    pos_y = pos_y + 1 == n_y ? 0 : pos_y + 1;
    pos_obs = pos_obs + 1 == n_obs ? 0 : pos_obs + 1;
  }
  }  // ================================ end of the step loop ================================
}

}  // namespace

#ifdef TDS_QUAD_PROF
extern "C" int tds_quad_prof_read(unsigned long long *out32, int iter) {  // iter >= 0: which iteration the NEXT launches stamp
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(tds_quad_prof_buf), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (iter >= 0 && hipMemcpyToSymbol(HIP_SYMBOL(tds_quad_prof_iter), &iter, sizeof(int)) != hipSuccess) return -1;
  return 0;
}
#endif

// LDS bytes of one environment of the quadruped kernel
template <typename T>
int tds_quad_lds_bytes(int input_dim) {
  return quad_layout(input_dim).stride * (int)sizeof(T);
}
template int tds_quad_lds_bytes<double>(int);
// LDS bytes of one WORKGROUP (four environments) of a step-loop launch: the environments' regions + the constant table
int tds_quad_loop_workgroup_bytes(int input_dim, int waves) {
  return quad_layout(input_dim).stride * 4 * waves * (int)sizeof(double) + (int)sizeof(QuadTable<double>);
}
template int tds_quad_lds_bytes<float>(int);

template <typename T, typename TR>
int tds_launch_quad(const DevModel<T> *d_model, const DevModel<T> &h_model, const TR *x_in, TR *y_out, const TR *actions,
                    TR *x_feedback, TR *obs_out, int n_envs, hipStream_t stream, const TdsStepCtl &ctl, int wide) {
  QuadOff O = quad_layout(h_model.input_dim);
  O.adim = h_model.action_dim;
  if (O.in_dim > 80) return -1;  // (the prologue holds a record in five registers per lane)
  const int blocks = (n_envs + 3) / 4;
  // (+ the constant table of a step-loop launch behind the four environments' regions)
  const bool one_step = ctl.nsub == 1 && ctl.obs_ring == nullptr && ctl.y_ring == nullptr;
  if (!one_step && wide) {  // wide step-loop launch: TDS_QUAD_WIDE_WAVES wavefronts per workgroup around one table
    constexpr int W = TDS_QUAD_WIDE_WAVES;
    const size_t bytes = (size_t)O.stride * 4 * W * sizeof(T) + sizeof(QuadTable<T>);
    static bool attr_set[64] = {};  // (per instantiation and device: the attribute belongs to the device's copy of the function)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
      if (hipFuncSetAttribute((const void *)tds_quad_kernel<T, TR, true, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return (int)hipGetLastError();
      if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((tds_quad_kernel<T, TR, true, W>), dim3((n_envs + 4 * W - 1) / (4 * W)), dim3(64 * W), bytes, stream, d_model,
                       x_in, y_out, actions, x_feedback, obs_out, ctl, n_envs, O);
    return (int)hipGetLastError();
  }
  const size_t shmem = (size_t)O.stride * 4 * sizeof(T) + (one_step ? 0 : sizeof(QuadTable<T>));
  // one plain step without rings: the straight-line form; K steps, record rings: the step-loop form
  if (one_step)
    hipLaunchKernelGGL((tds_quad_kernel<T, TR, false>), dim3(blocks), dim3(64), shmem, stream, d_model, x_in, y_out, actions,
                       x_feedback, obs_out, ctl, n_envs, O);
  else
    hipLaunchKernelGGL((tds_quad_kernel<T, TR, true>), dim3(blocks), dim3(64), shmem, stream, d_model, x_in, y_out, actions,
                       x_feedback, obs_out, ctl, n_envs, O);
  return (int)hipGetLastError();
}
template int tds_launch_quad<double, double>(const DevModel<double> *, const DevModel<double> &, const double *, double *,
                                             const double *, double *, double *, int, hipStream_t, const TdsStepCtl &, int);
template int tds_launch_quad<double, float>(const DevModel<double> *, const DevModel<double> &, const float *, float *,
                                            const float *, float *, float *, int, hipStream_t, const TdsStepCtl &, int);
template int tds_launch_quad<float, float>(const DevModel<float> *, const DevModel<float> &, const float *, float *,
                                           const float *, float *, float *, int, hipStream_t, const TdsStepCtl &, int);
