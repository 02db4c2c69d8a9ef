// tds_kernels.h — host-visible interface of tds_kernels.hip
#pragma once
#include <hip/hip_runtime.h>

#include "tds_device_model.h"

#define TDS_NUM_PHASE_STAMPS 14
// strides (in scalars, odd) of the per-link LDS records of the two sweep groups
#define TDS_S1 13  // X_world rot(9) trans(3), odd stride; (v | a0) only in the side records of branching links
#define TDS_S2 17  // f or F(6) | Ic(10), odd stride

// Per-environment LDS layout, offsets in units of the compute scalar T (see tds_make_lds_layout).
struct TdsLds {
  int stride;              // scalars per environment
  int NLp, NDP, NDs, NCPp; // links, padded dof (8/14/16/18/24/32), dof row stride (odd), contact-point stride
  int NPCp, pc;            // two-body worlds: stride / offset of the contact list between the bodies
  int zrows, ovrows;       // constraint rows held in LDS / surplus rows per env in the global slab
  int xrec, swd, cp, Lp, dinv, rows, xrow;  // persistent
  int Xw, v;               // phase group 1 (kinematics sweep)   } the three groups alias
  int IA, pA, F, Ic, a;    // phase group 2 (dynamics sweeps)     } each other
  int Z;                   // phase group 3 (constraint rows)     }
  int gram_ok;             // two-wavefront layout: the sweep groups hold the Gram buffer of tds_gram_solve
  int Lh;                  // two-wavefront layout: early copy of the first NDP/2 columns of L (tds_row_solve)
  int tau;                 // 18-dof kernels: per-link slot that keeps the generalised force from the PD block to phase F
  int in_dim, adim, nqnd;  // record dimensions as kernel arguments: the x record is requested from HBM before anything
                           // has been read from the model
  int cw;                  // step-loop launches of the plain kernels: rows ([G] scalars each) of the WORKGROUP's constant
                           // table behind the environments' regions (0: none; TDS_CW_LANE(bytes per scalar): the lane's
                           // small constants; + TDS_CW_XT: + X_T) — what every iteration of the step loop would otherwise
                           // re-fetch from L2
};
// init_pose | stiffness | damping | packed small integers (one 8-byte row, or two 4-byte rows)
#define TDS_CW_LANE(scalar_bytes) ((scalar_bytes) == 8 ? 4 : 5)
#define TDS_CW_XT 12  // X_T: rotation 9, translation 3

// what one launch does besides the physics (see the step loop in tds_kernels.hip)
#define TDS_RESET_NONE 0
#define TDS_RESET_AUTO 1    // re-initialise + settle the environments whose last step ends with done
#define TDS_RESET_FORCED 2  // re-initialise + settle the environments selected by mask (NULL = all)
#define TDS_CTL_RESET_CALL 8
struct TdsStepCtl {
  int nsub;                   // normal steps in this launch (0 for a pure reset launch)
  int reset_mode;             // TDS_RESET_*
  int settle_steps;           // zero-action steps after a re-initialisation
  unsigned long long seed;    // random stream of the reset distribution
  const unsigned char *mask;  // forced mode: per-env selection (device), NULL = all
  unsigned int *reset_count;  // [n_envs] per-env reset counter (device), position in the random stream
  // rollout mode (policy != NULL): every one of the nsub steps evaluates the environment's own linear
  // policy on its observation, steps, computes reward / done and accumulates the return on device
  const void *policy;         // [n_envs][action_dim * obs_dim + action_dim]: weights (row = action), then biases
  void *ret_sum;              // [n_envs] T: sum of (reward - shift) over the steps taken while not done
  int *ret_steps;             // [n_envs]   : number of those steps
  double shift;
  // straight-line launches only:
  const void *pool;           // != NULL: pre-settled reset states [pool_depth][pool_envs][dof_q + dof_qd] (record
  int pool_depth, pool_envs;  //          dtype), ring per environment indexed by its reset count; a done environment
                              //          takes its next state from here (auto-reset without the step loop)
  // step-loop launches only: a different action block per step (tds_hip_step_many as ONE launch)
  const void *act_pool;       // != NULL: [act_blocks][act_envs][action_dim] (record dtype); step k of the launch takes block
  int act_blocks, act_first;  //          (act_first + k) % act_blocks (step 0's block is also what `actions` points to)
  int act_envs;               //          environments per block (= the batch the blocks were laid out for)
  int flags;                  // bit 0: the first step observes the raw base x, y (state fresh from reset())
                              // TDS_CTL_RESET_CALL: the launch is tds_hip_reset (the environment's own reset():
                              // its observation keeps the base x, y where the model says so), not an auto-reset
  // step-loop launches only: per-step RECORD RINGS (tds_hip_step_many_rings).  With a ring set, EVERY step of the launch
  // does the whole output work of step_forward_original + VectorizedEnvironment::step — visual poses, y record,
  // reward / done, observation (locomotion_contact_simulation.h:273-303, ars_vectorized_environment.h:240-289) — and
  // stores it into the step's ring slot; step k of the launch owns slot (first + k) % slots.
  void *obs_ring;             // != NULL: [obs_slots][ring_envs][dof_q + dof_qd + 2]  obs | reward | done
  void *y_ring;               // != NULL: [y_slots][ring_envs][output_dim]
  int obs_slots, obs_first;
  int y_slots, y_first;
  int ring_envs;              // environments per ring slot (= the batch the rings were laid out for)
  int y_stride;               // scalars between consecutive records of the y ring (0: output_dim)
  int obs_envs;               // environments per slot of the obs ring (>= ring_envs: a slot laid out for all ranks)
  int ring_flags;             // TDS_RING_OBS_F32: the obs ring holds floats whatever the record dtype (wire format of the
                              // multi-GPU exchange: the slot is handed to ncclAllGather as it is)
  // != NULL: [obs_slots] counters, one per slot of the obs ring: every workgroup adds 1 to the counter of step k's slot
  // once its records of that step are visible device-wide — signalled from inside step k + 1 (the stores have long
  // drained by then: no wait on the step's own path), never for the last step of a launch (kernel completion covers
  // it).  What a communication stream polls to send ring slot k while the launch carries on.
  unsigned long long *progress;
  // PEER-STORE EXCHANGE (tds_shard.hip, round 5): the multi-GPU exchange without a collective.  The wavefront that stores a
  // step's [obs | reward | done] record into this rank's block of the gathered slot stores it into the SAME place of every
  // peer's gathered ring as well (peer memory mapped through hipIpcOpenMemHandle, system-scope write-through stores over
  // xGMI), and the workgroup that completes a slot — the last of the grid to count itself in on the slot's arrival counter —
  // raises the slot's flag on every rank (its own included) to the launch's sequence number.  No kernel of the exchange
  // ever needs a compute unit while the launch runs; the data movement of step k lies inside step k + 1.
  // peer_arrive != NULL selects it (then progress == NULL); EVERY step of the launch is signalled, the last one included.
  const void *const *peer_ring;     // [n_peers] device array: base of peer p's gathered ring as mapped in this process
  unsigned long long *const *peer_flags;  // [n_peers + 1]: peer p's flag array [slots][world]; the last entry is this rank's own
  unsigned int *peer_arrive;        // [obs_slots][TDS_PEER_ARRIVE_STRIDE] arrival counters of this launch's slots: TDS_PEER_SUB
                                    // first-level counters + one second-level counter per slot, a 128-byte line each
                                    // (all wrap at their own count: atomicInc, never reset) — see peer_signal
  long long peer_off;               // bytes from a ring's base to THIS rank's block of the launch's slot 0
  unsigned long long peer_epoch;    // the launch's sequence number (what a completed slot's flags are raised to)
  int n_peers;                      // ranks other than this one (0: one rank — the counters and the own flags only)
  int peer_flag_off;                // flag index of this rank in the launch's slot 0: slot0 * world + rank
  int peer_flag_stride;             // flags per slot (= world)
};
// arrival counting of the peer-store exchange: workgroup b counts itself in on first-level counter b mod TDS_PEER_SUB of
// the slot, the workgroup that completes a first-level counter on the slot's second-level counter.  One counter for the
// whole grid — round 5's first form — serialised 1024 returning atomics on one address at the end of every launch
// (device-scope atomics of eight XCDs meet at the memory side): + 35 us per launch whatever its length
// (tools/call_overhead_trace.sh).  The host allocates the full stride whatever TDS_PEER_SUB a kernel was built with.
#ifndef TDS_PEER_SUB
#define TDS_PEER_SUB 32
#endif
#define TDS_PEER_LINE 32  /* unsigned ints per counter: one 128-byte line each */
#define TDS_PEER_ARRIVE_STRIDE ((32 + 1) * TDS_PEER_LINE)
#define TDS_RING_OBS_F32 1
// the obs ring is written with device-scope write-through stores (sc1) and a step is signalled after a plain
// s_waitcnt vmcnt(0) — no release fence, whose buffer_wbl2 writes back every dirty line of the L2
#define TDS_RING_NOFENCE 2
// two-wavefront step-loop build: the helper wavefront counts the records of step k in at the top of its iteration k + 2,
// where it waits for the main wavefront's kinematics anyway (instead of in the middle of iteration k + 1, where its wait
// for the stores' acknowledgement sits in front of the workgroup barrier the main wavefront arrives at next)
#define TDS_RING_SIGNAL_LATE 4
// peer-store exchange: only [reward | done] of a record travel to the peers (option exchange_fields = 1: 8 of 120 bytes per
// Ant environment on a float wire — for runs whose policy lives on the device, SURVEY 8f N2: "removes the obs gather except
// for logging"); this rank's own block still receives the whole record
#define TDS_RING_PEER_REWARD_DONE 8
// peer-store exchange: every stride of the obs ring (record row of a wavefront, slot, rank block, peer offset) is a multiple
// of 8 bytes — a wavefront's records may leave as one row of 8-byte units (tds_kernels.hip: put_obs_wide); the peer table is
// padded to a multiple of four entries
#define TDS_RING_WIDE 16
// peer-store exchange, A/B switch for the first run on a real fabric (option shard_peer_release = 1): a SYSTEM-scope release
// fence in front of a workgroup's arrival count and in front of the flag stores, instead of relying on "s_waitcnt vmcnt(0) =
// acknowledged by the memory the store went to" + relaxed counters and flags
#define TDS_RING_PEER_RELEASE 32
// upper bound of the peers of a rank (ranks of one node - 1)
#define TDS_MAX_PEERS 15

// which build of the step kernel a launch takes (tds_launch_step's `form`)
#define TDS_FORM_W2 1         // L is the w2 layout: launch the two-wavefront form (plain kernels)
#define TDS_FORM_LOOP_OCC1 2  // step-loop build: the one-wavefront-per-SIMD compilation whatever the grid
#define TDS_FORM_LOOP_OCC2 4  // ... the two-wavefronts-per-SIMD compilation whatever the grid
#define TDS_FORM_OCT_W2 8     // the 8-lane kernel (tds_oct.hip): its two-wavefront build compiled for two wavefronts per SIMD
#define TDS_FORM_OCT_W2_OCC1 16  // ... compiled for one wavefront per SIMD (at most two workgroups per compute unit)
#define TDS_FORM_CHAIN_W1 32     // the serial-chain kernel (tds_chain.hip): no recorder wavefront (option chain_w2 = 0)
#define TDS_FORM_CHAIN_W2_ANY 64 // ... the recorder wavefront at any grid size (option chain_w2 = 2)
#define TDS_FORM_QUAD_WIDE 128   // the 16-lane kernel (tds_quad.hip): step-loop launch in workgroups of TDS_QUAD_WIDE_WAVES wavefronts
#define TDS_FORM_OCT_BESIDE 256  // the 8-lane kernel's two-wavefront build of at most 240 registers: fits on a SIMD beside a wavefront of the OCC1 build
#define TDS_QUAD_WIDE_WAVES 8    // ... around ONE constant table: a workgroup per compute unit, 32 environments each

// EXPERIMENT SLOTS (tools/build_alt.sh): the kernel sources compiled once more — other compiler flags, -DTDS_X_... source
// switches — as a small extra translation unit holding ONE (lanes, padded dof) instantiation of the f64 / KIND 0 kernels,
// linked into the SAME library under other names (-DTDS_ALT=k) and chosen per handle with the run-time option alt_build = k.
// Same-box A/B of a kernel experiment then costs one more object of ~0.5 MB instead of a second 55 MB library on the way
// to the GPU box.  The default build has no slot: the weak entry points of tds_api.hip are NULL and the option is refused.
#ifdef TDS_ALT
#define TDS_ALT_PASTE2(a, b) a##b
#define TDS_ALT_PASTE(a, b) TDS_ALT_PASTE2(a, b)
#define tds_launch_step_impl TDS_ALT_PASTE(tds_launch_step_impl_alt, TDS_ALT)
#define tds_kernel_max_dynamic_lds_impl TDS_ALT_PASTE(tds_kernel_max_dynamic_lds_impl_alt, TDS_ALT)
#endif
#define TDS_ALT_SLOTS 6
// what a slot's translation unit exports (tds_kernels.hip, bottom; extern "C", looked up weakly by tds_api.hip)
typedef int (*tds_alt_launch_fn)(const void *d_model, const void *h_model, const TdsLds *L, int lanes_per_env, const void *x_in,
                                 void *y_out, const void *actions, void *x_feedback, void *obs_out, void *ovf, int n_envs,
                                 hipStream_t stream, const TdsStepCtl *ctl, int form, int *lanes_ndp_key);

// na_cap: contacts whose rows stay in LDS (<= 0: all); w2: the layout of the two-wavefront workgroups (the LDS groups
// that alias each other in the one-wave layout laid out one after the other, + hand-over slots)
template <typename T>
TdsLds tds_make_lds_layout(const DevModel<T> &m, int na_cap, int lanes_per_env, bool w2 = false);

// Enqueue one step  y = f(x)  for n_envs environments on `stream`.
//   actions    (optional) [n_envs][action_dim] overrides the action slice of x
//   x_feedback (optional) [n_envs][input_dim]  receives the new q, qd (closed-loop stepping)
//   obs_out    (optional) [n_envs][dof_q+dof_qd+2]  observation | reward | done
//   ovf        [n_envs][ovrows][NDs+3] scratch slab for surplus constraint rows (NULL iff ovrows == 0)
// (KIND 0: plain fixed-base kernels, 1: floating base, 2: spherical joints, 3: worlds of several articulated bodies,
//  4: ... with floating bases among them; explicit instantiations live in the kernel translation units)
template <typename T, typename TR, int KIND>
int tds_launch_step_impl(const DevModel<T> *d_model, const DevModel<T> &h_model, const TdsLds &L, int lanes_per_env,
                         const TR *x_in, TR *y_out, const TR *actions, TR *x_feedback, TR *obs_out, T *ovf, int n_envs,
                         hipStream_t stream, const TdsStepCtl &ctl, long long *prof, int form);
// the 16-lane kernel of the star-shaped legged robots (tds_quad.hip; DevModel::quad): one straight-line step per launch
template <typename T, typename TR>
int tds_launch_quad(const DevModel<T> *d_model, const DevModel<T> &h_model, const TR *x_in, TR *y_out, const TR *actions,
                    TR *x_feedback, TR *obs_out, int n_envs, hipStream_t stream, const TdsStepCtl &ctl, int wide);
template <typename T>
int tds_quad_lds_bytes(int input_dim);
int tds_quad_loop_workgroup_bytes(int input_dim, int waves);  // LDS of one workgroup of its step-loop form: 4 * waves environments + the table
// what tds_launch_step hands to it: plain steps — one per launch, or K of them with action replay, record rings and
// reset-pool entries taken in the loop (tds_hip_step_many / _rings) —, no in-kernel reset, no policy, no exchange launch
// (progress counters / peer stores), no profile stamps
inline bool tds_quad_takes(int quad, const TdsStepCtl &ctl, const long long *prof) {
  return quad != 0 && prof == nullptr && ctl.nsub >= 1 && ctl.reset_mode == TDS_RESET_NONE && ctl.policy == nullptr &&
         ctl.progress == nullptr && ctl.peer_arrive == nullptr;
}

// the 8-lane kernel of the stars with two-link legs (tds_oct.hip; DevModel::oct: the Ant): the same launches as the 16-lane
// kernel, and the exchange launches of the multi-GPU layer (progress counters / peer stores) as well
template <typename T, typename TR>
int tds_launch_oct(const DevModel<T> *d_model, const DevModel<T> &h_model, const TR *x_in, TR *y_out, const TR *actions,
                   TR *x_feedback, TR *obs_out, int n_envs, hipStream_t stream, const TdsStepCtl &ctl, int build);
int tds_oct_lds_bytes(int input_dim);        // LDS of one environment
int tds_oct_workgroup_bytes(int input_dim);  // LDS of one workgroup: eight environments + the constant table
inline bool tds_oct_takes(int oct, const TdsStepCtl &ctl, const long long *prof) {
  return oct != 0 && prof == nullptr && ctl.nsub >= 1 && ctl.reset_mode == TDS_RESET_NONE && ctl.policy == nullptr;
}

// the serial-chain kernel (tds_chain.hip): plain steps and step loops incl. record rings and the exchange; no resets (its
// models have no reward rule: no environment is ever done), no policy, no phase stamps
template <typename T, typename TR>
int tds_launch_chain(const DevModel<T> *d_model, const DevModel<T> &h_model, const TR *x_in, TR *y_out, const TR *actions,
                     TR *x_feedback, TR *obs_out, int n_envs, hipStream_t stream, const TdsStepCtl &ctl, int w2_opt);
int tds_chain_lds_bytes(int num_links);
inline bool tds_chain_takes(int chain, const TdsStepCtl &ctl, const long long *prof) {
  return chain != 0 && prof == nullptr && ctl.nsub >= 1 && ctl.reset_mode == TDS_RESET_NONE && ctl.policy == nullptr;
}

// T: compute scalar, TR: record scalar (== T, or float under T = double: "f32 records / f64 arithmetic")
template <typename T, typename TR>
inline int tds_launch_step(const DevModel<T> *d_model, const DevModel<T> &h_model, const TdsLds &L, int lanes_per_env,
                           const TR *x_in, TR *y_out, const TR *actions, TR *x_feedback, TR *obs_out, T *ovf, int n_envs,
                           hipStream_t stream, const TdsStepCtl &ctl,
                           long long *prof = nullptr,  // prof: 14 phase stamps of workgroup 0 (diagnostic)
                           int form = 0) {   // TDS_FORM_*: which build of the kernel
#define TDS_ARGS d_model, h_model, L, lanes_per_env, x_in, y_out, actions, x_feedback, obs_out, ovf, n_envs, stream, ctl, prof, form
  if (tds_quad_takes(h_model.quad, ctl, prof))
    return tds_launch_quad<T, TR>(d_model, h_model, x_in, y_out, actions, x_feedback, obs_out, n_envs, stream, ctl,
                                  (form & TDS_FORM_QUAD_WIDE) != 0);
  if constexpr (sizeof(T) == 8) {
    if (tds_oct_takes(h_model.oct, ctl, prof))
      return tds_launch_oct<T, TR>(d_model, h_model, x_in, y_out, actions, x_feedback, obs_out, n_envs, stream, ctl,
                                   (form & TDS_FORM_OCT_BESIDE) ? 4 : (form & TDS_FORM_OCT_W2_OCC1) ? 3 : ((form & TDS_FORM_OCT_W2) ? 2 : 1));
    if (tds_chain_takes(h_model.chain, ctl, prof))
      return tds_launch_chain<T, TR>(d_model, h_model, x_in, y_out, actions, x_feedback, obs_out, n_envs, stream, ctl,
                                     (form & TDS_FORM_CHAIN_W1) ? 0 : ((form & TDS_FORM_CHAIN_W2_ANY) ? 2 : 1));
  }
  if (h_model.is_floating) return tds_launch_step_impl<T, TR, 1>(TDS_ARGS);
  // (pure float arithmetic — measured only, it misses the 1e-6 contract: tds_hip.h TDS_DTYPE_F32 — is built for the
  //  plain and the floating-base kernels; tds_hip_create refuses it for spherical joints and worlds of several bodies)
  if constexpr (sizeof(T) == 8) {
    if (h_model.num_spherical) return tds_launch_step_impl<T, TR, 2>(TDS_ARGS);
    if (h_model.num_bodies >= 2 && h_model.multi_floating) return tds_launch_step_impl<T, TR, 4>(TDS_ARGS);
    if (h_model.num_bodies >= 2) return tds_launch_step_impl<T, TR, 3>(TDS_ARGS);
  } else {
    if (h_model.num_spherical || h_model.num_bodies >= 2) return -4;
  }
  return tds_launch_step_impl<T, TR, 0>(TDS_ARGS);
#undef TDS_ARGS
}

template <typename T, typename TR, int KIND>
int tds_kernel_max_dynamic_lds_impl(int lanes_per_env, int ndp, int bytes);
template <typename T, typename TR>
inline int tds_kernel_max_dynamic_lds(int lanes_per_env, int ndp, int bytes, int kind) {
  if constexpr (sizeof(T) == 8) {
    if (kind == 2) return tds_kernel_max_dynamic_lds_impl<T, TR, 2>(lanes_per_env, ndp, bytes);
    if (kind == 3) return tds_kernel_max_dynamic_lds_impl<T, TR, 3>(lanes_per_env, ndp, bytes);
    if (kind == 4) return tds_kernel_max_dynamic_lds_impl<T, TR, 4>(lanes_per_env, ndp, bytes);
  } else {
    if (kind >= 2) return -4;
  }
  return kind == 1 ? tds_kernel_max_dynamic_lds_impl<T, TR, 1>(lanes_per_env, ndp, bytes)
                   : tds_kernel_max_dynamic_lds_impl<T, TR, 0>(lanes_per_env, ndp, bytes);
}

int tds_padded_dof(int nd, int lanes_per_env = 0);
