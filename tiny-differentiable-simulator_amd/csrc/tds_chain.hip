// tds_chain.hip — the kernel of the fixed-base SERIAL CHAINS without contacts (BASELINE configs 1 and 2: cartpole, pendulum5):
//   ABA -> clear_forces -> integrate_euler, joint torques given directly (TDS_STEP_TAU without a plane;
//   /root/reference/examples/environments/cartpole_environment.h:88-94, src/dynamics/forward_dynamics.hpp:11-326,
//   src/dynamics/kinematics.hpp:18-148, src/link.hpp:229-336, src/dynamics/integrator.hpp:10-195), then the y record
//   [q | qd | visual poses of the PRE-step kinematics | up.z] (locomotion_contact_simulation.h:273-303).
//
// Why a kernel of its own.  At these batch sizes every SIMD holds ONE wavefront, and a lone wavefront issues one instruction
// per ~5 cycles whatever it depends on (tools/ubench/lone_wave_latency.hip): the step time IS the wavefront's instruction
// count.  The general kernel walks a chain level by level through LDS (5 levels x 2 sweeps for pendulum5) with 16 lanes per
// environment; here
//   * lane = link, EIGHT environments per wavefront, two environments INTERLEAVED in each 16-lane DPP row (lane = 16 row +
//     2 link + parity): a DPP row shift by 2 k lanes moves a value k links along the chain of BOTH environments and never
//     from one environment into the other — the lanes a shift has no source for keep the operation's identity (`old`
//     operand, bound_ctrl off): no select, no LDS;
//   * everything in WORLD coordinates about the world origin, where the chain's recursions are plain prefix sums: the world
//     transforms are a log-depth scan of transform products (3 rounds for 8 links), link velocities and bias accelerations
//     prefix sums of the world motion axes, composite inertias and bias forces suffix sums;
//   * q'' = M^-1 (tau - C) instead of the reference's articulated-body sweep (the same q'' to round-off, as in the general
//     kernel: DESIGN.md "Reformulation" 6): row i of M = s_j . (Ic_i s_i) from row shifts of the axes, the rows gathered through
//     LDS, and the n x n LDL^T + both substitutions redundantly on every lane (n <= 8: a few dozen FMAs).
// One template instantiation per chain length NL = 2 .. 8.  Results agree with the reference to round-off (tests/test_chain.py).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "tds_device_model.h"
#include "tds_kernels.h"
#include "tds_lanes.h"

namespace {

#define CH_SYNC()                                              \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     \
    __builtin_amdgcn_wave_barrier();                           \
  } while (0)

using TB = TdsChainTab;

// v of the lane K links DOWN the chain (towards the base) of my environment; lanes without one keep `old`
template <int K>
__device__ __forceinline__ double ch_from_parent(double old, double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), 0x110 + 2 * K, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), 0x110 + 2 * K, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// v of the lane K links UP the chain (towards the tip); lanes without one keep `old`
template <int K>
__device__ __forceinline__ double ch_from_child(double old, double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), 0x100 + 2 * K, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), 0x100 + 2 * K, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

template <typename P>
__device__ __forceinline__ P *ch_global(P *p) {  // a loaded pointer: not LDS, not scratch (global_ instead of flat_ accesses)
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_assume(!__builtin_amdgcn_is_shared((const void *)p) && !__builtin_amdgcn_is_private((const void *)p));
#endif
  return p;
}

// sin / cos of a joint angle (the 8-lane kernel's routine: Cody-Waite reduction + the fdlibm kernels, library routine beyond 1e5)
__device__ __forceinline__ void ch_sincos(double x, double *sn, double *cs) {
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
  if (__builtin_expect(__any(big), 0)) {  // (per lane: an environment's bits do not depend on its wavefront-mates)
    double s2, c2;
    sincos(x, &s2, &c2);
    s_ = big ? s2 : s_;
    c_ = big ? c2 : c_;
  }
  *sn = s_;
  *cs = c_;
}

template <bool LOOP>
struct ChainCtlRef {
  using type = const TdsStepCtl &;
  static __device__ __forceinline__ type get(const TdsStepCtl &param, const __attribute__((address_space(4))) char *) { return param; }
};
template <>
struct ChainCtlRef<true> {
  using type = const __attribute__((address_space(4))) TdsStepCtl &;
  static __device__ __forceinline__ type get(const TdsStepCtl &, const __attribute__((address_space(4))) char *at) {
    return *(const __attribute__((address_space(4))) TdsStepCtl *)at;
  }
};

// LDS per environment, in scalars: the x record [q | qd | tau] (+ 2: done, reward — always zero for these models, kept where
// the record code of the other kernels reads them), the rows of M (NL x NL) and the right-hand side
template <int NL>
struct ChainLds {
  // (+ two-wavefront build: the links' world transforms, 12 scalars each, in two buffers that alternate step by step — the
  //  recorder wavefront packs the poses of step k while the main wavefront is in step k + 1)
  static constexpr int IN = 3 * NL, DONE = IN, REWARD = IN + 1, M = ((IN + 2 + 1) & ~1), RHS = M + NL * NL,
                       KIN = ((RHS + NL + 1) & ~1), STRIDE = KIN + 2 * NL * 12 + 2;  // (+ 2: the eight regions start on different banks)
};

// NL: links of the chain.  LOOP: K steps per launch (state in LDS, per-step records into rings) / one step per launch.
// W2 (step-loop form, while every workgroup is resident with at most two wavefronts per SIMD): a second wavefront per
// workgroup is the RECORDER — it packs the visual poses and stores every record of step k (y, obs ring, the peers' rings, the
// exchange's counters) while the main wavefront runs step k + 1: per-step records cost the main wavefront twelve LDS stores and
// two barriers instead of a quarter of its instruction stream (pendulum5 x 4096: 4.31 -> see DESIGN 2e)
#define CH_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// CREG (with W2, launches of at most one wavefront per SIMD): my link's constants in registers for the whole launch — 276
// registers, so only where a SIMD holds one wavefront of this kernel anyway; otherwise they are read from the LDS table where a
// step uses them (140 registers: three wavefronts per SIMD)
template <typename T, typename TR, int NL, bool LOOP, bool W2 = false, bool CREG = false>
__global__ __launch_bounds__(W2 ? 128 : 64) void tds_chain_kernel(const DevModel<T> *__restrict__ mdl_arg, const TR *x_in, TR *__restrict__ y_out,
                                                       const TR *__restrict__ actions, TR *x_feedback /* may alias x_in */,
                                                       TR *__restrict__ obs_out, TdsStepCtl ctl_arg, int n_envs) {
  extern __shared__ __align__(16) unsigned char tds_chain_smem[];
  T *const sm = reinterpret_cast<T *>(tds_chain_smem);
  using LD = ChainLds<NL>;
  constexpr int nq = NL, nd = NL, adim = NL, in_dim = 3 * NL, w_obs = 2 * NL + 2;
  T *const CT = sm + 8 * LD::STRIDE;  // the constant table
  static_assert(!W2 || LOOP, "the recorder wavefront exists in the step-loop form only");
  static_assert(!CREG || W2, "constants in registers: the two-wavefront build at one wavefront per SIMD only");
  constexpr int NT = W2 ? 128 : 64;
  const int t_all = threadIdx.x;
  const int tid = threadIdx.x & 63;
  // lane = 16 row + 2 link + parity: environment 2 row + parity of the wavefront, link (tid & 15) >> 1
  const int link = (tid & 15) >> 1;
  const int grp = ((tid >> 4) << 1) | (tid & 1);
  const int env = blockIdx.x * 8 + grp;
  const bool valid = env < n_envs;
  const bool mine = link < NL;  // (lanes of links the chain does not have: they carry identity links)
  T *const E = sm + grp * LD::STRIDE;
  T *const xr = E;
  {
    // (every global load of the prologue issued before the first is waited for: as loops of load -> LDS store they were ten
    //  dependent round trips in front of every launch — tools/oct_clock_ramp.py)
    constexpr int TN = (TB::TOTAL + NT - 1) / NT, XN = (in_dim + 7) / 8;
    T tv[TN], xv[XN];
#pragma unroll
    for (int k = 0; k < TN; ++k) {
      const int i = t_all + NT * k;
      tv[k] = i < TB::TOTAL ? mdl_arg->oct_tab[i] : T(0);
    }
#pragma unroll
    for (int k = 0; k < XN; ++k) {
      const int i = link + 8 * k;
      const bool act = actions != nullptr && i >= nq + nd;
      xv[k] = (!valid || i >= in_dim) ? T(0) : act ? (T)actions[(size_t)env * adim + (i - nq - nd)] : (T)x_in[(size_t)env * in_dim + i];
    }
#pragma unroll
    for (int k = 0; k < TN; ++k) {
      const int i = t_all + NT * k;
      if (i < TB::TOTAL) CT[i] = tv[k];
    }
#pragma unroll
    for (int k = 0; k < XN; ++k) {
      const int i = link + 8 * k;
      if (i < in_dim) xr[i] = xv[k];
    }
    if (link == 0) {
      xr[LD::DONE] = T(0);
      xr[LD::REWARD] = T(0);
    }
    if constexpr (W2) __syncthreads();
    else CH_SYNC();
  }
  const T *const CL = CT + link * TB::LSTR;  // my link's constants
  const T dt = CT[TB::SC + TB::DT];
  const int nv = (int)CT[TB::SC + TB::NUM_VISUALS];
  const int out_dim = (int)CT[TB::SC + TB::OUTPUT_DIM];
  const bool pack_vis = CT[TB::SC + TB::PACK_VISUALS] != T(0);
  const bool xt_ident = CT[TB::SC + TB::XT_IDENT] != T(0);
  const bool vis_ident = CT[TB::SC + TB::VIS_IDENT] != T(0);
  // My link's constants IN REGISTERS for the whole launch (46 scalars: the kernel holds 140 registers without them, a SIMD has
  // room for 256 at the two wavefronts it ever sees of this kernel).  Read from the LDS table where a step uses them — the big
  // kernels' rule, they have no register to spare — they were seven groups of reads, each waited for at once: seven LDS round
  // trips on the main wavefront's path, ~1 k of a step's 8.8 k cycles
  T cS[6], cXT[12], cCOM[3], cINER[9], cNAX[3], cNN[6], cGRAV[3];
#pragma unroll
  for (int k = 0; k < 6; ++k) cS[k] = CL[TB::S + k];
#pragma unroll
  for (int k = 0; k < 12; ++k) cXT[k] = CL[TB::XT + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) cCOM[k] = CL[TB::COM + k];
#pragma unroll
  for (int k = 0; k < 9; ++k) cINER[k] = CL[TB::INER + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) cNAX[k] = CL[TB::NAX + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) cNN[k] = CL[TB::NN + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) cGRAV[k] = CT[TB::SC + TB::GRAV + k];
  const T cMASS = CL[TB::MASS], cSTIFF = CL[TB::STIFF], cDAMP = CL[TB::DAMP], cROTF = CL[TB::ROTF];
  const int nsteps = LOOP ? ctl_arg.nsub : 1;
  // my joint's coordinate, velocity and torque IN REGISTERS across the steps of a launch (the lane that integrates them is the
  // lane that reads them: the LDS record is written for the recorder, the main wavefront never reads it back — one LDS round
  // trip less at the head of every step)
  T q_c = mine ? xr[mine ? link : 0] : T(0), qd_c = mine ? xr[nq + (mine ? link : 0)] : T(0), tau_c = mine ? xr[nq + nd + (mine ? link : 0)] : T(0);
  T next_act = T(0);
  int act_blk = 0, y_slot = 0, o_slot = 0;
  if constexpr (LOOP) {
    if (ctl_arg.act_pool != nullptr && ctl_arg.act_blocks > 0) act_blk = (ctl_arg.act_first + 1) % ctl_arg.act_blocks;
    if (ctl_arg.y_ring != nullptr && ctl_arg.y_slots > 0) y_slot = ctl_arg.y_first % ctl_arg.y_slots;
    if (ctl_arg.obs_ring != nullptr && ctl_arg.obs_slots > 0) o_slot = ctl_arg.obs_first % ctl_arg.obs_slots;
  }

  for (int it = 0; it < nsteps; ++it) {  // ================================ step loop ================================
    // (step-loop form: nothing of the kernel's arguments lives across an iteration — the fields of ctl are scalar loads from
    //  the kernel-argument segment where an iteration uses them, through a pointer laundered per iteration; held in SGPRs
    //  across the loop they spilled: 47 spills in the 5-link instantiation)
    const __attribute__((address_space(4))) char *ka_seg = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    if constexpr (LOOP) asm volatile("" : "+s"(ka_seg));
    typename ChainCtlRef<LOOP>::type ctl = ChainCtlRef<LOOP>::get(ctl_arg, ka_seg + 48 /* six pointers in front of ctl */);
  // a step's records counted in for the multi-GPU layer (tds_shard.hip; see tds_kernels.hip: peer_signal)
    auto signal_slot = [&](int pslot) {
      if (ctl.peer_arrive != nullptr) {
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): every store of this wavefront acknowledged by the memory it went to
        const bool rel = (ctl.ring_flags & TDS_RING_PEER_RELEASE) != 0;
        if (rel) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        if (tid == 0) {
          constexpr unsigned SUB = TDS_PEER_SUB;
          const unsigned g = gridDim.x, j = blockIdx.x % SUB;
          const unsigned n1 = (g - j + SUB - 1u) / SUB;
          const unsigned n2 = g < SUB ? g : SUB;
          unsigned *const base = ch_global(ctl.peer_arrive) + (size_t)pslot * TDS_PEER_ARRIVE_STRIDE;
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
        if (ctl.ring_flags & TDS_RING_NOFENCE) __builtin_amdgcn_s_waitcnt(0x0f70);
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (tid == 0) __hip_atomic_fetch_add(ch_global(ctl.progress) + pslot, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    };

    const bool last = it == nsteps - 1;
    int wv_ = 0;
    if constexpr (W2) {
      int w_ = threadIdx.x >> 6;
      asm volatile("" : "+v"(w_));
      wv_ = __builtin_amdgcn_readfirstlane(w_);
    }
    const bool is_main = !W2 || wv_ == 0, is_rec = !W2 || wv_ == 1;  // wave-uniform
    if constexpr (LOOP) {
      // the records of step it - 1 are counted in here (by the wavefront that stored them): their stores have long been
      // acknowledged — or, in the recorder's case, it waits for the main wavefront's step anyway
      if (is_rec && it > 0 && ctl.obs_ring != nullptr) signal_slot((o_slot == 0 ? ctl.obs_slots : o_slot) - 1);
      // the NEXT step's torques are requested now (a different action block per step: tds_hip_step_many)
      if (is_main && ctl.act_pool != nullptr && it + 1 < nsteps && valid && mine)
        next_act = (T)ch_global((const TR *)ctl.act_pool)[((size_t)act_blk * ctl.act_envs + env) * adim + link];
    }
    // where this step's y record goes: the slot of a y ring (every step of a step-loop launch), else the handle's y record
    // (the last step); the last step of a ring launch leaves its record in the handle's y record as well
    const int ystr = ctl.y_stride;
    TR *yo = nullptr, *yo2 = nullptr;
    int yend = ystr, yend2 = out_dim;
    if (LOOP && ctl.y_ring != nullptr) {
      yo = ch_global((TR *)ctl.y_ring) + ((size_t)y_slot * ctl.ring_envs + env) * ystr;
      if (last && y_out != nullptr) yo2 = y_out + (size_t)env * out_dim;
    } else if (last && y_out != nullptr) {
      yo = y_out + (size_t)env * (LOOP ? out_dim : ystr);
      yend = LOOP ? out_dim : ystr;
    }

    const int me = mine ? link : 0;
    T R[9], p[3];  // my link's world transform (main wavefront: from the kinematics; recorder: from the hand-over in LDS)
    // ---- visual poses of y, from the PRE-step X_world (locomotion_contact_simulation.h:281-299): visual `link` is mine
    auto poses = [&]() {
      if (valid && yo != nullptr && link < nv) {
        T Ro[9], po[3], qo[4];
        if (vis_ident) {  // wave-uniform
#pragma unroll
          for (int k = 0; k < 9; ++k) Ro[k] = R[k];
        } else {
          mat3_mul(R, CL + TB::VIS, Ro);
        }
        mat3_mulv(R, CL + TB::VIS + 9, po);
        matrix_to_quat(Ro, qo);
        const T rec[7] = {p[0] + po[0], p[1] + po[1], p[2] + po[2], qo[0], qo[1], qo[2], qo[3]};
        TR *const o = yo + (nq + nd) + 7 * link;
#pragma unroll
        for (int k = 0; k < 7; ++k) o[k] = (TR)rec[k];
        if (yo2 != nullptr) {
          TR *const o2 = yo2 + (nq + nd) + 7 * link;
#pragma unroll
          for (int k = 0; k < 7; ++k) o2[k] = (TR)rec[k];
        }
      }
    };
    if (is_main) {  // ================================ main wavefront: the step ================================
    // ---- A. my joint: coordinate, velocity, torque (multi_body.hpp:557-570; joint stiffness / damping, forward_dynamics.hpp:122-123)
    const T q = q_c, qd = qd_c;
    T tau = tau_c;
    tau -= (CREG ? cSTIFF : CL[TB::STIFF]) * q + (CREG ? cDAMP : CL[TB::DAMP]) * qd;
    // ---- B. jcalc (link.hpp:229-287): R_J = cos I + sin [n]x + (1 - cos) n n^T about the unit axis (every revolute type; a
    //         prismatic joint's angle is multiplied by 0), t_J = S_linear q; X_parent = X_T X_J
    T S[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) S[k] = CREG ? cS[k] : CL[TB::S + k];
    {
      T sn, cs;
      ch_sincos(q * (CREG ? cROTF : CL[TB::ROTF]), &sn, &cs);
      const T c1 = T(1) - cs;
      const T nx = CREG ? cNAX[0] : CL[TB::NAX], ny = CREG ? cNAX[1] : CL[TB::NAX + 1], nz = CREG ? cNAX[2] : CL[TB::NAX + 2];
      T RJ[9];
      RJ[0] = cs + c1 * (CREG ? cNN[0] : CL[TB::NN + 0]);
      RJ[1] = c1 * (CREG ? cNN[1] : CL[TB::NN + 1]) - sn * nz;
      RJ[2] = c1 * (CREG ? cNN[2] : CL[TB::NN + 2]) + sn * ny;
      RJ[3] = c1 * (CREG ? cNN[1] : CL[TB::NN + 1]) + sn * nz;
      RJ[4] = cs + c1 * (CREG ? cNN[3] : CL[TB::NN + 3]);
      RJ[5] = c1 * (CREG ? cNN[4] : CL[TB::NN + 4]) - sn * nx;
      RJ[6] = c1 * (CREG ? cNN[2] : CL[TB::NN + 2]) - sn * ny;
      RJ[7] = c1 * (CREG ? cNN[4] : CL[TB::NN + 4]) + sn * nx;
      RJ[8] = cs + c1 * (CREG ? cNN[5] : CL[TB::NN + 5]);
      const T tJ[3] = {S[3] * q, S[4] * q, S[5] * q};
      if (xt_ident) {  // wave-uniform
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = RJ[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = (CREG ? cXT[9 + k] : CL[TB::XT + 9 + k]) + tJ[k];
      } else {
        T RT[9], r[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) RT[k] = CREG ? cXT[k] : CL[TB::XT + k];
        mat3_mul(RT, RJ, R);
        mat3_mulv(RT, tJ, r);
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = (CREG ? cXT[9 + k] : CL[TB::XT + 9 + k]) + r[k];
      }
    }
    // ---- C. forward kinematics (kinematics.hpp:64-97): X_world_i = X_world_(i-1) X_parent_i as an inclusive scan of transform
    //         products along the chain — round k: X_i <- X_(i-k) X_i, where X_(i-k) covers the k links below those X_i covers
    static_for<0, 3>([&](auto rc) {
      constexpr int K = 1 << decltype(rc)::value;
      if constexpr (K < NL) {
        T A[9], a[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) A[k] = ch_from_parent<K>((k == 0 || k == 4 || k == 8) ? T(1) : T(0), R[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) a[k] = ch_from_parent<K>(T(0), p[k]);
        T Rn[9], r[3];
        mat3_mul(A, R, Rn);
        mat3_mulv(A, p, r);
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = Rn[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = a[k] + r[k];
      }
    });
    if constexpr (W2) {  // hand-over to the recorder: buffer it & 1
      if (mine) {
        T *const kin = E + LD::KIN + ((it & 1) * NL + me) * 12;
#pragma unroll
        for (int k = 0; k < 9; ++k) kin[k] = R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) kin[9 + k] = p[k];
      }
    } else {
      poses();
    }
    // ---- D. the world motion axis of my joint about the world origin: s = [R S_a ; p x (R S_a) + R S_l]; link velocities
    //         v_i = sum_(j <= i) s_j qd_j, bias accelerations a0_i = a_base + sum_(j <= i) v_j x s_j qd_j with a_base =
    //         [0 ; -g] (forward_dynamics.hpp:242; kinematics.hpp:82-97 in world coordinates)
    T s[6];
    {
      T lin[3];
      mat3_mulv(R, S, s);
      mat3_mulv(R, S + 3, lin);
      cross3(p, s, s + 3);
#pragma unroll
      for (int k = 0; k < 3; ++k) s[3 + k] += lin[k];
    }
    T v[6], vj[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = vj[k] = s[k] * qd;
    static_for<0, 3>([&](auto rc) {
      constexpr int K = 1 << decltype(rc)::value;
      if constexpr (K < NL) {
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] += ch_from_parent<K>(T(0), v[k]);
      }
    });
    T a0[6];
    {
      // c = v x vJ (motion cross product; kinematics.hpp:96-97)
      T t1[3], t2[3];
      cross3(v, vj, a0);
      cross3(v, vj + 3, t1);
      cross3(v + 3, vj, t2);
#pragma unroll
      for (int k = 0; k < 3; ++k) a0[3 + k] = t1[k] + t2[k];
    }
    static_for<0, 3>([&](auto rc) {
      constexpr int K = 1 << decltype(rc)::value;
      if constexpr (K < NL) {
#pragma unroll
        for (int k = 0; k < 6; ++k) a0[k] += ch_from_parent<K>(T(0), a0[k]);
      }
    });
#pragma unroll
    for (int k = 0; k < 3; ++k) a0[3 + k] -= CREG ? cGRAV[k] : CT[TB::SC + TB::GRAV + k];
    // ---- E. world-frame rigid inertia and bias force of my link (kinematics.hpp:99-132, inertia.hpp:121-130):
    //         I = (Isym 6 | h 3 | m), f = I a0 + v x* I v
    T Ic[10], fc[6];
    {
      const T m = CREG ? cMASS : CL[TB::MASS];
      T cw[3];
      mat3_mulv(R, CREG ? (const T *)cCOM : CL + TB::COM, cw);
#pragma unroll
      for (int k = 0; k < 3; ++k) cw[k] += p[k];
      T RI[9], Iw[9];
      mat3_mul(R, CREG ? (const T *)cINER : CL + TB::INER, RI);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Iw[3 * r + c] = RI[3 * r] * R[3 * c] + RI[3 * r + 1] * R[3 * c + 1] + RI[3 * r + 2] * R[3 * c + 2];
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
      sym3_mulv(Ic, v, Iv);
      cross3(h, v + 3, t3);
#pragma unroll
      for (int k = 0; k < 3; ++k) Iv[k] += t3[k];
      cross3(h, v, t3);
#pragma unroll
      for (int k = 0; k < 3; ++k) Iv[3 + k] = m * v[3 + k] - t3[k];
      sym3_mulv(Ic, a0, Ia);
      cross3(h, a0 + 3, t3);
#pragma unroll
      for (int k = 0; k < 3; ++k) Ia[k] += t3[k];
      cross3(h, a0, t3);
#pragma unroll
      for (int k = 0; k < 3; ++k) Ia[3 + k] = m * a0[3 + k] - t3[k];
      T u3[3];
      cross3(v, Iv, fc);
      cross3(v + 3, Iv + 3, u3);
#pragma unroll
      for (int k = 0; k < 3; ++k) fc[k] += u3[k];
      cross3(v, Iv + 3, fc + 3);
#pragma unroll
      for (int k = 0; k < 6; ++k) fc[k] += Ia[k];
    }
    // ---- F. composite inertia and bias force of the sub-chain from my link to the tip (mass_matrix.hpp:39-56): suffix sums
    static_for<0, 3>([&](auto rc) {
      constexpr int K = 1 << decltype(rc)::value;
      if constexpr (K < NL) {
#pragma unroll
        for (int k = 0; k < 10; ++k) Ic[k] += ch_from_child<K>(T(0), Ic[k]);
#pragma unroll
        for (int k = 0; k < 6; ++k) fc[k] += ch_from_child<K>(T(0), fc[k]);
      }
    });
    // ---- G. row `link` of M and the right-hand side: M_ij = s_j . (Ic_i s_i) for j <= i (mass_matrix.hpp:58-127), C_i = s_i . fc_i
    {
      T F[6], t3[3];
      const T *const h = Ic + 6;
      sym3_mulv(Ic, s, F);
      cross3(h, s + 3, t3);
#pragma unroll
      for (int k = 0; k < 3; ++k) F[k] += t3[k];
      cross3(h, s, t3);
#pragma unroll
      for (int k = 0; k < 3; ++k) F[3 + k] = Ic[9] * s[3 + k] - t3[k];
      const T Cb = dot3(s, fc) + dot3(s + 3, fc + 3);
      T *const Mrow = E + LD::M + me * NL;
      if (mine) {
        Mrow[me] = dot3(s, F) + dot3(s + 3, F + 3);
        E[LD::RHS + me] = tau - Cb;
      }
      static_for<1, NL>([&](auto dc) {
        constexpr int D = decltype(dc)::value;
        T sj[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) sj[k] = ch_from_parent<D>(T(0), s[k]);
        const T mij = dot3(sj, F) + dot3(sj + 3, F + 3);
        if (mine && link >= D) Mrow[me - D] = mij;
      });
    }
    CH_SYNC();
    // ---- H. q'' = M^-1 (tau - C): LDL^T of the n x n matrix and both substitutions, redundantly on every lane of the
    //         environment (replaces forward_dynamics.hpp:111-302 as in the general kernel; same q'' to round-off)
    T qdd;
    {
      T L[NL * (NL + 1) / 2], x[NL];
      static_for<0, NL>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        static_for<0, i + 1>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          L[i * (i + 1) / 2 + j] = E[LD::M + i * NL + j];
        });
        x[i] = E[LD::RHS + i];
      });
      T dinv[NL];
      static_for<0, NL>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        // column j: d_j = a_jj - sum_k l_jk^2 d_k ; l_ij = (a_ij - sum_k l_ik l_jk d_k) / d_j   (w_ik = l_ik d_k kept in place of a_ik)
        T dj = L[j * (j + 1) / 2 + j];
        static_for<0, j>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          const T ljk = L[j * (j + 1) / 2 + k] * dinv[k];  // l_jk from w_jk
          dj -= L[j * (j + 1) / 2 + k] * ljk;
        });
        dinv[j] = rcp_full<T>(dj);  // (v_rcp_f64 + Newton steps: a third of the division's instructions)
        static_for<j + 1, NL>([&](auto icc) {
          constexpr int i = decltype(icc)::value;
          T w = L[i * (i + 1) / 2 + j];
          static_for<0, j>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            w -= L[i * (i + 1) / 2 + k] * (L[j * (j + 1) / 2 + k] * dinv[k]);
          });
          L[i * (i + 1) / 2 + j] = w;  // w_ij = l_ij d_j
        });
      });
      // forward: z_i = r_i - sum_(k<i) l_ik z_k ; scale ; backward: x_i = z_i / d_i - sum_(k>i) l_ki x_k
      static_for<0, NL>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        static_for<0, i>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          x[i] -= (L[i * (i + 1) / 2 + k] * dinv[k]) * x[k];
        });
      });
      static_for<0, NL>([&](auto ic) {
        constexpr int i = NL - 1 - decltype(ic)::value;
        x[i] *= dinv[i];
        static_for<i + 1, NL>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          x[i] -= (L[k * (k + 1) / 2 + i] * dinv[i]) * x[k];
        });
      });
      qdd = x[0];
      static_for<1, NL>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        qdd = link == i ? x[i] : qdd;
      });
    }
    // ---- I. integrate_euler (integrator.hpp:10-133): qd += q'' dt, q += qd dt; the new state into the LDS record, and the next
    //         step's torques into the record's action slots in front of this step's record stores
    if constexpr (W2) CH_BAR();  // (R) the recorder has stored the records of the step before: the LDS record may change
    {
      const T qd_new = qd + qdd * dt;
      const T q_new = q + qd_new * dt;
      if (mine) {
        xr[me] = q_new;
        xr[nq + me] = qd_new;
        q_c = q_new;
        qd_c = qd_new;
        if constexpr (LOOP) {
          if (ctl.act_pool != nullptr && !last) {
            xr[nq + nd + me] = next_act;
            tau_c = next_act;
          }
        }
      }
    }
    if constexpr (W2) CH_BAR();  // (S) the step's state is in the LDS record
    else CH_SYNC();
    }  // ================================ end of the main wavefront's step ================================
    if (is_rec) {  // ================================ the step's records (one-wave build: the same wavefront) ================================
    if constexpr (W2) {
      if (it == 0) CH_BAR();  // (R) of the first step: nothing stored yet
      CH_BAR();               // (S)
      if (mine) {
        const T *const kin = E + LD::KIN + ((it & 1) * NL + me) * 12;
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = kin[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = kin[9 + k];
      }
      poses();
    }
    // ---- J. the step's records.  y: q | qd | (visual poses: above) | up.z | zero padding
    auto y_state = [&](TR *y, int end) {
      for (int i = link; i < nq + nd; i += 8) y[i] = (TR)xr[i];
      int tail = nq + nd;
      if (pack_vis) {
        tail += 7 * nv;
        if (link == 0) y[tail] = (TR)(CT[TB::SC + TB::BASE_R8]);  // up_dot_world_z (fixed base)
        tail += 1;
      }
      for (int i = tail + link; i < end; i += 8) y[i] = TR(0);
    };
    if (valid && yo != nullptr) {
      y_state(yo, yend);
      if (yo2 != nullptr) y_state(yo2, yend2);
    }
    // [obs | reward | done] (obs[0] = obs[1] = 0, ars_vectorized_environment.h:283-288): the slot of an obs ring (every step of a
    // step-loop launch; floats on the multi-GPU wire format) and / or the caller's record (last step)
    if constexpr (LOOP) {
      if (ctl.obs_ring != nullptr) {  // wave-uniform
        const int slot = o_slot;
        const int rf = ctl.ring_flags;
        const bool f32w = (rf & TDS_RING_OBS_F32) != 0 || sizeof(TR) == 4;
        const int np = ctl.peer_arrive != nullptr ? ctl.n_peers : 0;
        const bool rd_only = (rf & TDS_RING_PEER_REWARD_DONE) != 0;
        if (ctl.peer_arrive != nullptr && (rf & TDS_RING_WIDE) != 0 && __all(valid)) {
          // peer-store exchange: the wavefront's eight records as one row of 8-byte units (see tds_oct.hip: help_rec)
          const int wl = tid;
          const size_t row0 = ((size_t)slot * ctl.obs_envs + (size_t)blockIdx.x * 8) * (size_t)w_obs;
          const int per_unit = f32w ? 2 : 1;
          const int n_units = (8 * w_obs) / per_unit;
          const unsigned long long *const __attribute__((address_space(4))) *tab =
              (const unsigned long long *const __attribute__((address_space(4))) *)(const __attribute__((address_space(4))) void *)ctl.peer_ring;
          for (int u0 = 0; u0 < n_units; u0 += 64) {
            const int uu = u0 + wl;
            const bool on = uu < n_units;
            unsigned lo = 0u, hi = 0u;
            bool tail = false;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              if (c < per_unit) {
                const int f = on ? uu * per_unit + c : 0;
                const int e = f / w_obs;
                const int i = f - e * w_obs;
                const int src = i < nq + nd ? i : (i == nq + nd ? LD::REWARD : LD::DONE);
                const T vv = i < 2 ? T(0) : sm[e * LD::STRIDE + src];
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
            const size_t unit_at = row0 / per_unit + (size_t)uu;
            if (on) __hip_atomic_store(ch_global((unsigned long long *)ctl.obs_ring) + unit_at, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool to_peers = on && (!rd_only || tail);
            for (int p0 = 0; p0 < np; p0 += 4) {  // (the table is padded to a multiple of four entries)
              const unsigned long long *const b0 = ch_global(tab[p0]), *const b1 = ch_global(tab[p0 + 1]), *const b2 = ch_global(tab[p0 + 2]),
                                       *const b3 = ch_global(tab[p0 + 3]);
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
          for (int i = link; i < w_obs; i += 8) {
            const T vv = i < 2 ? T(0) : xr[i < nq + nd ? i : (i == nq + nd ? LD::REWARD : LD::DONE)];
            if (rf & TDS_RING_OBS_F32) {
              float *const pp = ch_global((float *)ctl.obs_ring) + at + i;
              if (rf & TDS_RING_NOFENCE) __hip_atomic_store(pp, (float)vv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              else *pp = (float)vv;
            } else {
              TR *const pp = ch_global((TR *)ctl.obs_ring) + at + i;
              if (rf & TDS_RING_NOFENCE) __hip_atomic_store(pp, (TR)vv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              else *pp = (TR)vv;
            }
            if (np > 0 && (i >= nq + nd || !rd_only)) {
              for (int pr = 0; pr < np; ++pr) {
                char *const pb = (char *)ch_global(((void *const __attribute__((address_space(4))) *)(const __attribute__((address_space(4))) void *)ctl.peer_ring)[pr]) + ctl.peer_off;
                if (rf & TDS_RING_OBS_F32) __hip_atomic_store((float *)pb + (at + i), (float)vv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else __hip_atomic_store((TR *)pb + (at + i), (TR)vv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              }
            }
          }
        }
        if (last && ctl.peer_arrive != nullptr) signal_slot(slot);
      }
    }
    if (valid && last) {
      for (int i = link; i < nq + nd; i += 8) {
        const TR vv = (TR)xr[i];
        if (obs_out != nullptr) obs_out[(size_t)env * w_obs + i] = i < 2 ? TR(0) : vv;
        if (x_feedback != nullptr) x_feedback[(size_t)env * in_dim + i] = vv;
      }
      if (link == 0 && obs_out != nullptr) {
        obs_out[(size_t)env * w_obs + nq + nd] = TR(0);
        obs_out[(size_t)env * w_obs + nq + nd + 1] = TR(0);
      }
    }
    if constexpr (W2) {
      if (!last) CH_BAR();  // (R) of the next step
    }
    }  // ================================ end of the records ================================
    if constexpr (LOOP) {
      if constexpr (!W2) CH_SYNC();
      act_blk = act_blk + 1 >= ctl.act_blocks ? 0 : act_blk + 1;
      y_slot = y_slot + 1 >= ctl.y_slots ? 0 : y_slot + 1;
      o_slot = o_slot + 1 >= ctl.obs_slots ? 0 : o_slot + 1;
    }
  }  // ================================ end of the step loop ================================
}

template <int NL>
constexpr size_t chain_shmem() { return ((size_t)ChainLds<NL>::STRIDE * 8 + TdsChainTab::TOTAL) * sizeof(double); }

}  // namespace

// LDS bytes of one environment of the chain kernel
int tds_chain_lds_bytes(int num_links) {
  switch (num_links) {
    case 2: return ChainLds<2>::STRIDE * 8;
    case 3: return ChainLds<3>::STRIDE * 8;
    case 4: return ChainLds<4>::STRIDE * 8;
    case 5: return ChainLds<5>::STRIDE * 8;
    case 6: return ChainLds<6>::STRIDE * 8;
    case 7: return ChainLds<7>::STRIDE * 8;
    default: return ChainLds<8>::STRIDE * 8;
  }
}

template <typename T, typename TR>
int tds_launch_chain(const DevModel<T> *d_model, const DevModel<T> &h_model, const TR *x_in, TR *y_out, const TR *actions,
                     TR *x_feedback, TR *obs_out, int n_envs, hipStream_t stream, const TdsStepCtl &ctl, int w2_opt) {
  const int blocks = (n_envs + 7) / 8;
  // one plain step without rings: the straight-line form; K steps, record rings: the step-loop form
  const bool one_step = ctl.nsub == 1 && ctl.obs_ring == nullptr && ctl.y_ring == nullptr;
  // the recorder wavefront: step-loop launches that store per-step records, while the launch is resident with at most two
  // wavefronts per SIMD (1024 workgroups of two on 1024 SIMDs); option chain_w2 = 0 / 2: never / at any grid size
  const bool two_waves = !one_step && (ctl.obs_ring != nullptr || ctl.y_ring != nullptr) && w2_opt != 0 && (blocks <= 1024 || w2_opt == 2);
  // ... with my link's constants in registers while the launch puts at most ONE wavefront on a SIMD (that build holds 276
  // registers: a SIMD has room for one)
  static int n_simd = 0;
  if (n_simd == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      n_simd = 4 * cus;
    else
      n_simd = 1024;
  }
  const bool creg = two_waves && 2 * blocks <= n_simd;
#define CH_LAUNCH(NL_)                                                                                                           \
  case NL_:                                                                                                                      \
    if (one_step)                                                                                                                \
      hipLaunchKernelGGL((tds_chain_kernel<T, TR, NL_, false>), dim3(blocks), dim3(64), chain_shmem<NL_>(), stream, d_model, x_in, \
                         y_out, actions, x_feedback, obs_out, ctl, n_envs);                                                      \
    else if (creg)                                                                                                               \
      hipLaunchKernelGGL((tds_chain_kernel<T, TR, NL_, true, true, true>), dim3(blocks), dim3(128), chain_shmem<NL_>(), stream,    \
                         d_model, x_in, y_out, actions, x_feedback, obs_out, ctl, n_envs);                                       \
    else if (two_waves)                                                                                                          \
      hipLaunchKernelGGL((tds_chain_kernel<T, TR, NL_, true, true>), dim3(blocks), dim3(128), chain_shmem<NL_>(), stream, d_model, \
                         x_in, y_out, actions, x_feedback, obs_out, ctl, n_envs);                                                \
    else                                                                                                                         \
      hipLaunchKernelGGL((tds_chain_kernel<T, TR, NL_, true>), dim3(blocks), dim3(64), chain_shmem<NL_>(), stream, d_model, x_in,  \
                         y_out, actions, x_feedback, obs_out, ctl, n_envs);                                                      \
    break;
  switch (h_model.chain) {
    CH_LAUNCH(2) CH_LAUNCH(3) CH_LAUNCH(4) CH_LAUNCH(5) CH_LAUNCH(6) CH_LAUNCH(7) CH_LAUNCH(8)
    default: return -1;
  }
#undef CH_LAUNCH
  return (int)hipGetLastError();
}
template int tds_launch_chain<double, double>(const DevModel<double> *, const DevModel<double> &, const double *, double *,
                                              const double *, double *, double *, int, hipStream_t, const TdsStepCtl &, int);
template int tds_launch_chain<double, float>(const DevModel<double> *, const DevModel<double> &, const float *, float *,
                                             const float *, float *, float *, int, hipStream_t, const TdsStepCtl &, int);
