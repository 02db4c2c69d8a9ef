/*
 * tds_hip.h — C ABI of libtds_hip.so, the MI355X-native many-instance stepper.
 *
 * This is the drop-in boundary for ONE path of tiny-differentiable-simulator (TDS):
 * the per-environment simulation step
 *     PD -> forward_dynamics (ABA) -> integrate_euler_qdd -> World::step -> integrate_euler
 * (reference: examples/environments/locomotion_contact_simulation.h:151-304) replayed over
 * N independent environments.  Everything in this header is plain C: pointers, sizes,
 * ints.  No torch / STL types cross it.
 *
 * Two layers are exported:
 *
 *  (1) The handle API  tds_hip_*  — resident device state, explicit stream, status codes.
 *      It replaces  VectorizedEnvironment::CustomForwardDynamicsStepper::step
 *      (reference: examples/ars/ars_vectorized_environment.h:75-85) without the
 *      host<->device copy per step that the reference CUDA stepper performs
 *      (reference: examples/ars/ars_train_policy_cuda.cpp:246-308).
 *
 *  (2) The legacy  <model>_forward_zero{,_meta,_allocate,_deallocate}  symbols, exactly as the
 *      reference's generated CUDA libraries export them and as CudaModel<double> dlopen()s
 *      them (reference: examples/ars/ars_train_policy_cuda.cpp:220-229, 345-359; emitter
 *      src/utils/cuda_codegen.hpp:146-262).  They live in the thin shim libraries
 *      cuda_model_ant.so / cuda_model_laikago.so (csrc/legacy_shim.cpp) and forward to (1).
 *
 * The model description (tds_model_t) is a POD "flattened MultiBody + World": it is what
 * include/tds_hip_stepper.hpp::flatten_model() produces from an intact tds::MultiBody /
 * tds::World built by TDS's own URDF loader, so that every loader quirk is inherited.
 */
#ifndef TDS_HIP_H
#define TDS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDS_HIP_ABI_VERSION 5

#define TDS_MAX_LINKS 64   /* links of the model as the reference builds it; the kernels take <= 32 lanes: one per
                              moving link (fixed links are folded into their parents when lanes run short), six for
                              a floating base, three per spherical joint */
#define TDS_MAX_GEOMS 32
#define TDS_MAX_VISUALS 64
#define TDS_MAX_ACTIONS 32
#define TDS_MAX_DOF 32
/* contact points per environment: sphere 1, capsule 2, box 8 (contact_point.hpp:96-198) */
#define TDS_MAX_CONTACTS 64
/* articulated bodies of one world (World::multi_bodies_ without the plane; src/world.hpp:206-282 pairs them all) */
#define TDS_MAX_BODIES 4
/* contact points between the geometries of the articulated bodies of a world, summed over all body pairs
   (sphere-sphere 1, capsule-sphere 2: contact_point.hpp:43-94, 405-438), per environment */
#define TDS_MAX_PAIR_CONTACTS 160

/* status codes returned by every tds_hip_* function */
enum {
  TDS_OK = 0,
  TDS_ERR_INVALID_ARG = 1,
  TDS_ERR_UNSUPPORTED = 2, /* model uses a feature the HIP path does not implement */
  TDS_ERR_HIP = 3,         /* a HIP runtime call failed; see tds_hip_last_error() */
  TDS_ERR_NO_DEVICE = 4
};

/* joint types — numeric values identical to tds::JointType (src/link.hpp:9-21) */
enum {
  TDS_JOINT_FIXED = -1,
  TDS_JOINT_PRISMATIC_X = 0,
  TDS_JOINT_PRISMATIC_Y = 1,
  TDS_JOINT_PRISMATIC_Z = 2,
  TDS_JOINT_PRISMATIC_AXIS = 3,
  TDS_JOINT_REVOLUTE_X = 4,
  TDS_JOINT_REVOLUTE_Y = 5,
  TDS_JOINT_REVOLUTE_Z = 6,
  TDS_JOINT_REVOLUTE_AXIS = 7,
  TDS_JOINT_SPHERICAL = 8
};

/* geometry types — numeric values identical to tds::GeometryTypes (src/geometry.hpp:30-38) */
enum {
  TDS_GEOM_SPHERE = 0,
  TDS_GEOM_PLANE = 1,
  TDS_GEOM_CAPSULE = 2,
  TDS_GEOM_MESH = 3,
  TDS_GEOM_BOX = 4
};

/* which call sequence one "step" replays */
enum {
  /* PD -> ABA -> integrate_euler_qdd -> contacts + MLCP/PGS -> integrate_euler -> pack
     (locomotion_contact_simulation.h:151-304).  x = [q | qd | action | kp kd max_force] */
  TDS_STEP_LOCOMOTION = 0,
  /* joint torques given directly, no PD:              x = [q | qd | tau]
       has_plane == 0:  ABA -> clear_forces -> integrate_euler   (cartpole_environment.h:88-94)
       has_plane == 1:  ABA -> clear_forces -> integrate_euler_qdd -> World::step -> integrate_euler
                        (examples/soft_contact_example.cpp:107-115)                     */
  TDS_STEP_TAU = 1
};

/* reward / termination rule written into the observation record (tds_hip_step_obs) */
enum {
  TDS_REWARD_NONE = 0,
  /* reward = (x_t - x_{t-1})/dt, done = z < 0.26 (reward 0 when done)
     (examples/environments/ant_environment2.h:75-106) */
  TDS_REWARD_ANT = 1,
  /* reward = x, done = up_dot_world_z < 0.6 || z < 0.2, up from rpy = q[3..5]
     (examples/environments/laikago_environment2.h:130-171) */
  TDS_REWARD_LAIKAGO = 2,
  /* reward = x, done = up_dot_world_z < 0.6 || z < 0.8, up from the quaternion q[3..6] of the spherical root joint
     (examples/environments/humanoid_environment.h:155-197, fixed-base branch) */
  TDS_REWARD_HUMANOID = 3
};

/* scalar types: what the kernels compute in / what the records in HBM (x, y, actions, obs, policy) are stored in.
   F64        double / double — the reference's arithmetic; parity-gated at 1e-6 (measured <= 4e-11 per step)
   F32        float / float   — pure single precision; measured, NOT gated: the mass-matrix factorisation loses the
                                1e-6 contract in float (8e-6 without contacts, 1e-3 with), as does the reference's own
                                TinyAlgebra<float> instantiation (tests/test_f32.py)
   F64_REC32  double / float  — the reference's FLOAT record ABI (BASELINE config 2; half the bytes per env-step) with
                                the arithmetic kept in double registers: parity-gated at 1e-6 on float inputs */
enum { TDS_DTYPE_F64 = 0, TDS_DTYPE_F32 = 1, TDS_DTYPE_F64_REC32 = 2 };

/* All 3x3 matrices are row-major: m[3*r+c].  Transforms are TDS "right-associative":
   X.rot maps child-frame vectors into the parent frame, X.trans is the child origin in the
   parent frame (src/math/transform.hpp:123-137). */
typedef struct tds_link {
  int32_t joint_type; /* TDS_JOINT_* */
  int32_t parent;     /* parent link index, -1 = base */
  int32_t q_index;    /* index into q,  -2 for fixed joints (multi_body.hpp:324-349); a spherical joint owns 4 */
  int32_t qd_index;   /* index into qd, -2 for fixed joints; a spherical joint owns 3 */
  double X_T_rot[9];  /* parent link -> joint frame (link.hpp:39) */
  double X_T_trans[3];
  double S[6];        /* motion subspace [angular | linear], NOT normalised (link.hpp:125-193) */
  double mass;        /* RigidBodyInertia (src/math/inertia.hpp:9-37) */
  double com[3];
  double inertia[9];
  double stiffness;   /* link.hpp:88-89 (always 0 from the URDF loader) */
  double damping;
} tds_link_t;

typedef struct tds_geom {
  int32_t link; /* owning link (index into tds_model_t::links, i.e. global over all bodies of a multi-body world);
                   -1 - b = base of body b (-1 for a single body) */
  int32_t type; /* TDS_GEOM_SPHERE / CAPSULE / BOX */
  double radius;
  double length;     /* capsule */
  double extents[3]; /* box full extents */
  double X_rot[9];   /* X_collision: geometry frame in link frame (link.hpp:75) */
  double X_trans[3];
} tds_geom_t;

typedef struct tds_visual {
  int32_t link; /* owning link (>= 0; base visuals are not packed by the reference) */
  int32_t pad_;
  double X_rot[9]; /* X_visual (link.hpp:78-79) */
  double X_trans[3];
} tds_visual_t;

/* body b >= 1 of a world with several articulated bodies (tds_model_t::bodies) */
typedef struct tds_body {
  int32_t first_link;  /* index of the body's first link in tds_model_t::links */
  int32_t first_geom;  /* index of the body's first geometry in tds_model_t::geoms */
  int32_t is_floating; /* multi_body.hpp:66-78 */
  int32_t pad_;
  double base_X_world_rot[9];
  double base_X_world_trans[3];
  double base_mass; /* mb.base_rbi(), used only when is_floating */
  double base_com[3];
  double base_inertia[9];
} tds_body_t;

typedef struct tds_model {
  int32_t abi_version; /* TDS_HIP_ABI_VERSION */
  int32_t step_mode;   /* TDS_STEP_* */
  int32_t num_links;
  int32_t dof_q;       /* mb.dof()    */
  int32_t dof_qd;      /* mb.dof_qd() */
  int32_t is_floating; /* 1: floating base — q = [quat xyzw | pos | joints] (dof_q = 7 + n), qd = [omega | v | joints]
                          (multi_body.hpp:324-349, kinematics.hpp:35-62) */
  int32_t num_geoms;
  int32_t num_visuals;
  int32_t action_dim;    /* LOCOMOTION: #PD targets; TAU: dof_qd */
  int32_t pd_start_link; /* first link the PD loop visits (base_dof_=6, locomotion_contact_simulation.h:180) */
  int32_t has_plane;     /* 1: multi_bodies_[0] is the implicit plane (plane first!) */
  int32_t pgs_iterations;
  int32_t input_dim;  /* doubles per env in the reference record x */
  int32_t output_dim; /* doubles per env in the reference record y */
  int32_t pack_visuals; /* 1: y carries 7 doubles per visual + up_dot_z */
  int32_t reward_mode;  /* TDS_REWARD_*: which env's compute_reward_done the obs record mirrors */
  double dt;
  double gravity[3];
  double base_X_world_rot[9];
  double base_X_world_trans[3];
  double plane_normal[3]; /* unit */
  double plane_constant;
  double cfm;         /* mb_constraint_solver.hpp:64-67 */
  double erp;
  double friction;    /* World::default_friction   (world.hpp:68) */
  double restitution; /* World::default_restitution (world.hpp:69) */
  double action_limit; /* 0.4 (locomotion_contact_simulation.h:234) */
  double initial_poses[TDS_MAX_ACTIONS];
  /* environment reset (examples/environments/ant_environment2.h:109-165,
     laikago_environment2.h:63-116): q = reset_q + reset_noise * U(-1,1) per coordinate, qd = 0,
     followed by settle_steps steps with zero action.  The reference draws U from std::rand();
     the device uses a counter-based generator keyed by (seed, env, reset count, coordinate). */
  double reset_q[TDS_MAX_DOF];
  double reset_noise[TDS_MAX_DOF];
  int32_t settle_steps;
  /* 1: the observation tds_hip_reset hands out keeps the raw base x, y — what LaikagoContactSimulation::reset and
     HumanoidContactSimulation::reset return (laikago_environment2.h:63-116); 0: obs[0] = obs[1] = 0 as
     AntContactSimulation2::reset does (ant_environment2.h:162-163).  The STEP's observation always zeroes them
     (ars_vectorized_environment.h:283-288).  (Occupies the former padding slot: layout unchanged.) */
  int32_t reset_obs_raw_xy;
  /* rigid-body inertia of the base link, mb.base_rbi() (multi_body.hpp:78); used only when is_floating */
  double base_mass;
  double base_com[3];
  double base_inertia[9];
  /* Worlds with SEVERAL articulated bodies (World::step over multi_bodies_ = [plane,] body 0, 1, ...:
     src/world.hpp:206-282, 293-366).  num_bodies = B in 2..TDS_MAX_BODIES: body 0 is described by the fields above
     (base frame, is_floating, base inertia) and owns links [0, bodies[1].first_link) and geoms [0, bodies[1].first_geom);
     body b >= 1 is bodies[b] and owns links / geoms from its first_link / first_geom up to those of body b + 1 (or the
     end).  A link's parent is -1 (its own body's base) or a link of the same body; q / qd indices are dense over all
     bodies: q = [q_0 | q_1 | ...], qd = [qd_0 | qd_1 | ...] (a floating body's share is [quat xyzw | pos | joints] /
     [omega | v | joints]), in TAU mode x = [q | qd | tau_0 | tau_1 | ...] with dof_actuated entries per body; a geometry
     on the base of body b has link = -1 - b; geoms in the reference's order (base first, then link by link).  Each body is
     stepped by its own forward dynamics; contacts: every body against the plane (if any), then every pair of bodies
     i < j in the reference's order — sphere-sphere and capsule-sphere in either order, as the reference's dispatcher
     knows them — solved pair by pair (plane-0, plane-1, ..., 0-1, 0-2, ..., 1-2, ...) with both Jacobian blocks and both
     inverse mass matrices (mb_constraint_solver.hpp:191-498).  num_bodies 0 / 1: one body.  1-dof joints; step_mode TAU. */
  int32_t num_bodies;
  int32_t pad3_[3];
  tds_body_t bodies[TDS_MAX_BODIES]; /* entries 1 .. num_bodies - 1 (entry 0 is not read) */
  tds_link_t links[TDS_MAX_LINKS];
  tds_geom_t geoms[TDS_MAX_GEOMS];
  tds_visual_t visuals[TDS_MAX_VISUALS];
  char name[32];
} tds_model_t;

typedef struct tds_hip_sim tds_hip_sim_t; /* opaque */

/* Human-readable text of the last failure on the calling thread ("" if none). */
const char *tds_hip_last_error(void);

/* Library/ABI introspection. */
int tds_hip_abi_version(void);
int tds_hip_device_count(void);

/* Validate a model against what the HIP path implements (joint/geometry coverage,
   sizes).  TDS_OK or TDS_ERR_UNSUPPORTED / TDS_ERR_INVALID_ARG. */
int tds_hip_model_check(const tds_model_t *model);

/* Create a simulation of num_envs independent copies of `model` on HIP device `device`.
   dtype = TDS_DTYPE_*.  Device buffers owned by the handle: x [N][input_dim], y [N][output_dim] (both in the
   record dtype).  Every entry point selects the handle's device for the duration of the call and restores the
   caller's current device: one process may hold handles on several GPUs. */
int tds_hip_create(const tds_model_t *model, int num_envs, int device, int dtype,
                   tds_hip_sim_t **out);
int tds_hip_destroy(tds_hip_sim_t *sim);

/* All launches / copies of this handle are enqueued on `hip_stream` (a hipStream_t;
   NULL = the default stream).  The library never synchronises unless documented. */
int tds_hip_set_stream(tds_hip_sim_t *sim, void *hip_stream);

int tds_hip_num_envs(const tds_hip_sim_t *sim);
int tds_hip_input_dim(const tds_hip_sim_t *sim);
int tds_hip_output_dim(const tds_hip_sim_t *sim);
int tds_hip_dtype(const tds_hip_sim_t *sim);
int tds_hip_device(const tds_hip_sim_t *sim);
int tds_hip_record_bytes(const tds_hip_sim_t *sim); /* bytes per scalar of the records: 8 (F64) or 4 */
/* wait for everything enqueued on the handle's stream */
int tds_hip_sync(tds_hip_sim_t *sim);

/* Device pointers of the resident records, for zero-copy consumers (e.g. a torch tensor
   wrapping them).  Layout: env-major [N][dim], compute dtype. */
void *tds_hip_x_device(tds_hip_sim_t *sim);
void *tds_hip_y_device(tds_hip_sim_t *sim);

/* Host <-> resident record transfers (double on the host side, converted to the compute
   dtype).  These synchronise the handle's stream. */
int tds_hip_set_inputs(tds_hip_sim_t *sim, const double *x_host /* [N*input_dim] */);
int tds_hip_get_inputs(tds_hip_sim_t *sim, double *x_host /* [N*input_dim] */);
int tds_hip_get_outputs(tds_hip_sim_t *sim, double *y_host /* [N*output_dim] */);

/* y = f(x) on caller-provided DEVICE buffers in the compute dtype (pure function, the
   reference's forward_zero semantics).  Asynchronous on the handle's stream. */
int tds_hip_forward_zero_device(tds_hip_sim_t *sim, const void *x_dev, void *y_dev);

/* One closed-loop environment step on the resident records, asynchronous:
     - if actions_dev != NULL, x[:, dof_q+dof_qd : +action_dim] <- actions_dev [N][action_dim]
     - y = f(x)
     - x[:, 0 : dof_q+dof_qd] <- y[:, 0 : dof_q+dof_qd]     (what VectorizedEnvironment::step
       does on the host, ars_vectorized_environment.h:240-289, minus reward/reset)
   Repeated `substeps` times with the same action. */
int tds_hip_step(tds_hip_sim_t *sim, const void *actions_dev, int substeps);

/* tds_hip_step plus the per-env observation record the vectorised env hands to the policy
   (ars_vectorized_environment.h:250-289), written by the same kernel launch:
     obs_dev [N][obs_dim + 2] = [ q | qd  with obs[0] = obs[1] = 0 | reward | done ]   (compute dtype)
   obs_dim = dof_q + dof_qd.  reward/done follow model->reward_mode for the LAST substep (what
   repeated VectorizedEnvironment::step calls would report).  This record is what a multi-GPU
   job all-gathers. */
int tds_hip_step_obs(tds_hip_sim_t *sim, const void *actions_dev, int substeps, void *obs_dev);
int tds_hip_obs_dim(const tds_hip_sim_t *sim);

/* The same step for a caller whose actions and observations are HOST arrays of doubles — the signature the reference's
   VectorizedEnvironment::step works on (ars_vectorized_environment.h:213-291) — with the STATE resident on the device:
   per call only actions_host [N][action_dim] travel up and obs_host [N][obs_dim + 2] = [obs | reward | done]
   (and y_host [N][output_dim] = sim_states_with_graphics_, optional) travel down, through pinned staging memory owned
   by the handle.  Blocking.  With tds_hip_set_auto_reset on, done environments are reset inside the call.
   tds_hip_reset_host: tds_hip_reset with a host mask ([N] bytes, NULL = all) and the observations back on the host.
   tds_hip_set_states: overwrite [q | qd] of every environment from qqd_host [N][dof_q + dof_qd] (actions and the PD
   variables of the x records stay) — how a caller that resets with the REFERENCE's own reset() hands the states over.
   Used by tds_hip::VectorizedEnv (include/tds_hip_stepper.hpp). */
int tds_hip_step_host(tds_hip_sim_t *sim, const double *actions_host, int substeps, double *obs_host, double *y_host);
int tds_hip_reset_host(tds_hip_sim_t *sim, const unsigned char *mask_host, double *obs_host);
int tds_hip_set_states(tds_hip_sim_t *sim, const double *qqd_host);
/* Device memory on the handle's device for callers without HIP headers (action pools, record rings, policies handed to
   tds_hip_step_many_rings / tds_hip_rollout): alloc zero-fills; upload / download block on the handle's stream. */
int tds_hip_device_alloc(tds_hip_sim_t *sim, size_t bytes, void **out);
int tds_hip_device_free(tds_hip_sim_t *sim, void *ptr);
int tds_hip_device_upload(tds_hip_sim_t *sim, void *dst_dev, const void *src_host, size_t bytes);
int tds_hip_device_download(tds_hip_sim_t *sim, void *dst_host, const void *src_dev, size_t bytes);

/* n_steps closed-loop steps (tds_hip_step_obs with substeps = 1) per host call, replayed from a
   captured hipGraph: ONE graph launch instead of n_steps kernel launches, which removes the host-side launch gaps
   that cost a double-digit share of short runs at ~20 us per step.
     actions_dev  [action_blocks][N][action_dim] (record dtype) or NULL; step k uses block (first_block + k) % action_blocks
     obs_dev      optional [N][obs_dim + 2], overwritten by every step (holds the last step's record afterwards)
   The graph is cached on the handle (keyed by all arguments); _prepare builds it without running anything, so that a
   timed region holds the launch only.  n_steps <= 4096.

   Environment chains.  The environments are independent and the call holds n_steps steps of each, so the batch need
   not move in lockstep: the library splits it into C contiguous environment ranges ("chains"), captures one linear
   graph per chain and replays them concurrently on C streams forked from / joined to the handle's stream.  While one
   chain sits at a kernel boundary (launch latency, workgroup dispatch ramp, the wait for its slowest workgroup: ~3 us
   of a 20 us Ant x 4096 step) the other chains' workgroups have the compute units to themselves.  Records are bit
   for bit those of whole-batch launches.  C: tds_hip_set_graph_chains (0 = library default: 2 for models with
   contact points, 1 otherwise; environment TDS_HIP_GRAPH_CHAINS overrides), or measured on the spot by
   tds_hip_step_many_tune, which runs probe_steps steps (advancing the simulation) once untimed and once timed for
   C = 1, 2, 3, keeps the fastest and returns it in *chains. */
int tds_hip_step_many_prepare(tds_hip_sim_t *sim, const void *actions_dev, int action_blocks, int first_block,
                              int n_steps, void *obs_dev);
int tds_hip_step_many(tds_hip_sim_t *sim, const void *actions_dev, int action_blocks, int first_block, int n_steps,
                      void *obs_dev);
int tds_hip_set_graph_chains(tds_hip_sim_t *sim, int chains);

/* Run-time options.  Every switch of the library — kernel forms, launch forms, reset-pool geometry, exchange forms — is a
   named integer option (table: csrc/tds_options.h; tds_hip_option_count / _name enumerate it).  When a handle is
   CREATED each option takes, in this order: the value of tds_hip_default_option in this process, the environment
   variable TDS_HIP_<KEY> (upper case), or "unset" = the library's own rule.  The handle keeps that snapshot; afterwards
   only tds_hip_set_option changes it — the environment is never read while a handle is in use and nothing is latched
   per process, so one process can hold handles with different forms (how the tests select kernel builds).
   Create-time options (lanes_per_env, na_cap, w2, gram, no_chain, no_rootjoint, no_kinchain, no_eulerroot, no_legscan,
   fold_fixed) shape the device model / LDS layout: tds_hip_set_option refuses them, set them with
   tds_hip_default_option before tds_hip_create.  Setting an option that cached graphs or the reset pool depend on
   drops those (they are rebuilt by the next call).  The options of a shard are those of its sim handle
   (tds_hip_shard_sim).  Unknown key: TDS_ERR_INVALID_ARG. */
int tds_hip_default_option(const char *key, long long value);
int tds_hip_set_option(tds_hip_sim_t *sim, const char *key, long long value);
/* *is_set (optional) = 0: the option is unset (library rule), *value = 0 */
int tds_hip_get_option(const tds_hip_sim_t *sim, const char *key, long long *value, int *is_set);
int tds_hip_option_count(void);
const char *tds_hip_option_name(int index);
/* 1 if tds_hip_step_many(sim, ..., n_steps, ...) runs as ONE launch of the step-loop kernel instead of graphs: worlds
   without contact points (pendulums, the cartpole), fixed-base kernels up to 16 dof with contacts (the Ant) while
   the batch is at most three rounds of workgroups; the star-shaped legged robots of tds_hip_single_step_kernel take their
   own kernels' step-loop forms: the 16-lane kernel (Laikago) while every workgroup of the launch is resident at once —
   computed from the device's compute-unit count and LDS size, hipDeviceProp_t: in one-wavefront workgroups up to 6144
   environments on an MI355X, in workgroups of eight wavefronts around one constant table (a workgroup per compute unit:
   option quad_wide, default on) up to 8192 — and the chained graphs of its straight-line form beyond; the 8-lane kernel (the Ant) always — beyond one round of resident
   two-wavefront workgroups (8192 environments) as environment ranges of that size one after the other, each a launch of all the
   call's steps (the environments are independent; single steps and the exchange's launches stay whole).  No kernel boundaries; the state stays in
   LDS (in the compute scalar) for the n_steps steps, every step takes its own action block, y / obs / x are written
   once at the end — what the graph form leaves behind too, whose obs_dev is overwritten by every step.  With float
   records the state is rounded to float once per call instead of once per step.
   (TDS_HIP_STEP_MANY_LOOP=0 / 1 forbids / forces the form).

   With tds_hip_set_auto_reset on, every one of the n_steps steps resets the environments it ends with done
   (ars_vectorized_environment.h:262-277), through the reset pool: where _is_loop holds (then for the Ant at every
   batch size: the alternative is single steps, not graphs), as step-loop launches of up to
   128 steps in which a done environment copies its next pre-settled state from its ring into LDS and carries on, the
   rings being topped up between the launches on a side stream; elsewhere as n_steps single steps.  Same stream of
   random numbers and same records as n_steps calls of tds_hip_step_obs. */
int tds_hip_step_many_is_loop(const tds_hip_sim_t *sim, int n_steps);

/* tds_hip_step_many with PER-STEP RECORDS: every one of the n_steps steps produces what one call of the reference's
   VectorizedEnvironment::step produces (ars_vectorized_environment.h:240-289 on top of step_forward_original's output
   packing, locomotion_contact_simulation.h:273-303) — the [obs | reward | done] record and, if asked for, the whole y
   record (q, qd, visual poses, up.z, padding) — and stores it into the step's slot of a caller-owned ring in HBM:
     obs_ring  [obs_slots][N][obs_dim + 2]   step k of the call -> slot (obs_first + k) % obs_slots
     y_ring    [y_slots][N][y_stride]        step k of the call -> slot (y_first + k) % y_slots        (either may be NULL)
   in the record dtype; obs_f32 != 0: the obs ring holds FLOATS whatever the record dtype (the wire format of the
   multi-GPU exchange).  Where tds_hip_step_many_is_loop holds the n_steps steps are ONE launch of the step-loop kernel
   that packs and stores the records of every step (write-back stores; the state itself never leaves LDS between the
   steps); elsewhere they are the chained graphs of single-step launches with each launch pointed at its slots.  With
   auto-reset on, a step that ends with done leaves reward / done of the terminal step and the observation of the fresh
   environment in its slot, as the reference does.  Afterwards the handle's y record holds the last step's (a device
   copy of its slot).  This is the form bench.py times: all of step_forward_original's work, every step.
     progress  optional, step-loop form only, not with auto-reset (those calls are cut into several launches): an array
               of obs_slots device counters, one per slot of the obs ring.  Every workgroup adds 1 to the counter of
               step k's slot once its OBS-RING record of that step is visible device-wide (signalled while step k + 1
               runs; not for the last step of the call: stream order covers it): the slot's counter has grown by
               tds_hip_step_many_rings_blocks exactly when EVERY workgroup has stored that step — what
               tds_hip_shard_step_many polls to exchange slot k while the launch carries on (a single running total
               would be reached by the average workgroup while the slowest is steps behind).  It covers the obs ring
               only: the y ring is written with ordinary stores that become visible to other agents at the end of the
               launch (nothing exchanges y records; read them behind the launch in stream order).
               tds_hip_step_many_rings_blocks = increments of a slot's counter per use of the slot.
     y_stride  scalars between consecutive y records of the y ring (0: output_dim, i.e. packed).  A stride that is a
               multiple of the 128-byte line (Ant, f64: 160 instead of 155) lets every record start on a line boundary:
               the ring launch then writes whole lines only (measured: HBM write traffic per step down to the payload).
     obs_slot_envs  see the field.
   A slot is overwritten `slots` steps later: the ring's depth is the lag the consumer may have. */
typedef struct tds_hip_rings {
  void *obs_ring;
  int32_t obs_slots, obs_first, obs_f32, y_stride;
  void *y_ring;
  int32_t y_slots, y_first;
  unsigned long long *progress;
  int32_t obs_slot_envs; /* environments per SLOT of the obs ring (0: the handle's own N).  Larger than N: a slot laid out
                            for the environments of all ranks of a multi-GPU run — obs_ring then points at this rank's
                            block of slot 0, and the launch stores its records straight into the receive buffer of an
                            in-place all-gather (no copy of the local block on any rank) */
  int32_t pad1_;
} tds_hip_rings_t;
int tds_hip_step_many_rings(tds_hip_sim_t *sim, const void *actions_dev, int action_blocks, int first_block, int n_steps,
                            const tds_hip_rings_t *rings);
/* builds the graphs of the next tds_hip_step_many_rings with the same arguments (graph form; a no-op for the loop form) */
int tds_hip_step_many_rings_prepare(tds_hip_sim_t *sim, const void *actions_dev, int action_blocks, int first_block,
                                    int n_steps, const tds_hip_rings_t *rings);
int tds_hip_step_many_rings_blocks(const tds_hip_sim_t *sim);
int tds_hip_step_many_tune(tds_hip_sim_t *sim, const void *actions_dev, int action_blocks, int probe_steps,
                           void *obs_dev, int *chains);

/* On-device environment reset — replaces the host loop of VectorizedEnvironment::reset / the
   auto-reset branch of VectorizedEnvironment::step (ars_vectorized_environment.h:196-211, 262-277).
   tds_hip_reset: re-initialise the environments selected by mask_dev (uint8 [N], NULL = all) from
     the model's reset distribution, run model->settle_steps zero-action steps, write the new state
     into the resident x record and, if obs_dev != NULL, the observation part of their obs record
     (reward / done slots are left untouched).  One kernel launch, asynchronous.
   tds_hip_set_auto_reset(enable): tds_hip_step / tds_hip_step_obs then reset every environment whose
     step ends with done (model->reward_mode) inside the same launch: y, reward and done describe
     the terminal step, x and the observation describe the freshly reset + settled environment —
     exactly what the reference's auto_reset_when_done does.
   seed selects the random stream (counter-based: deterministic for a given seed, environment
   index and per-environment reset count, independent of launch geometry). */
int tds_hip_set_auto_reset(tds_hip_sim_t *sim, int enable, unsigned long long seed);
int tds_hip_reset(tds_hip_sim_t *sim, const unsigned char *mask_dev, void *obs_dev);

/* Policy rollout entirely on device: n_steps times { action = W obs + b with the
   environment's OWN parameters; step; reward / done; return bookkeeping } in ONE launch, on the
   resident records.  It is Worker::rollouts + VectorizedEnvironment::policy/step of the reference
   (examples/ars/ars_vectorized_worker.h:51-140, ars_vectorized_environment.h:213-300) for the linear
   policy those build (one linear layer obs_dim -> action_dim with bias, identity activation):
     policy_dev        [num_envs][action_dim*obs_dim + action_dim] in the compute dtype, NeuralNetwork
                       parameter order: weights row-major (row = action), then biases
                       (src/math/neural_network.hpp:406-415); obs_dim = dof_q + dof_qd
     obs               [q | qd] with obs[0] = obs[1] = 0 (ars_vectorized_environment.h:283-288);
                       flags bit 0: the FIRST step sees the raw base x, y, as it does right after
                       VectorizedEnvironment::reset (:196-211)
     return_sum_dev    [num_envs] compute dtype: sum of (reward - shift) over the steps taken while the
                       environment was not done; return_steps_dev [num_envs] int: their number
                       (total_rewards / vec_steps of Worker::rollouts); either may be NULL
     obs_dev           optional [num_envs][obs_dim + 2]: final observation | last reward | done
     flags             bit 0: see obs.  The rollout runs as ONE launch of the step-loop kernel; bit 1 forces one
                       launch of the straight-line step kernel per step with a small policy + bookkeeping kernel in
                       between (same results to round-off; slower at every batch size measured; no auto-reset).
   With tds_hip_set_auto_reset a done environment is re-initialised + settled and keeps collecting;
   without it, it stays done (the state keeps stepping, as under the reference's custom stepper). */
int tds_hip_rollout(tds_hip_sim_t *sim, const void *policy_dev, int n_steps, double shift, int flags,
                    void *return_sum_dev, int *return_steps_dev, void *obs_dev);

/* tds_hip_rollout plus the by-products Worker::rollouts produces alongside the returns
   (examples/ars/ars_vectorized_worker.h:88-135), each optional (NULL):
     stats_dev     [num_envs][obs_dim][3] record dtype, UPDATED in place (zero it for a fresh filter): RunningStat
                   (examples/ars/running_stat.h, Knuth's recurrence) of every observation component the policy was
                   evaluated on — (count, mean, S); variance = S / (count - 1).  Pushed every step, done or not.
     traj_dev      [num_envs][n_steps][output_dim] record dtype + traj_len_dev [num_envs] int: per environment the y
                   record (sim_states_with_graphics_) of every step taken while not done; a step that ends with done
                   repeats the previous entry if there is one — the trajectories vector of Worker::rollouts.
   Asking for either selects the per-step-launch form of the rollout (the bookkeeping kernel between the step
   launches produces them); auto-reset then goes through the reset pool. */
int tds_hip_rollout_ex(tds_hip_sim_t *sim, const void *policy_dev, int n_steps, double shift, int flags,
                       void *return_sum_dev, int *return_steps_dev, void *obs_dev, void *stats_dev, void *traj_dev,
                       int *traj_len_dev);

/* Policies with hidden layers (the reference's NeuralNetworkSpecification / NeuralNetwork::compute,
   src/math/neural_network.hpp:93-160, 223-300 — the vectorised ARS environment builds one linear layer and keeps its
   two ReLU layers commented out, examples/ars/ars_vectorized_environment.h:170-178).  num_layers entries, the input
   included: layer_sizes[0] = obs_dim = dof_q + dof_qd, layer_sizes[num_layers - 1] = action_dim, each
   <= TDS_NN_MAX_UNITS; activations[i - 1] (TDS_NN_ACT_*) is applied to layer i >= 1; use_bias[i] != 0 gives layer i a
   bias (use_bias[0]: a bias added to the observation, set_input_dim).  Afterwards the policy_dev argument of
   tds_hip_rollout(_ex) holds, per environment, tds_hip_policy_num_parameters() scalars in NeuralNetwork parameter order
   (set_parameters, :406-415): all weights layer by layer, each row-major [unit of layer i][unit of layer i - 1], then all
   biases layer by layer — and rollouts run as one straight-line step launch per step with the network evaluated by a
   kernel of its own in between (one wavefront per environment, activations in LDS).  num_layers = 0 restores the
   default (the linear policy inside the step-loop launch). */
enum {
  TDS_NN_ACT_IDENTITY = -1,
  TDS_NN_ACT_TANH = 0,
  TDS_NN_ACT_SIN = 1,
  TDS_NN_ACT_RELU = 2,
  TDS_NN_ACT_SOFT_RELU = 3,
  TDS_NN_ACT_ELU = 4,
  TDS_NN_ACT_SIGMOID = 5,
  TDS_NN_ACT_SOFTSIGN = 6
};
#define TDS_NN_MAX_LAYERS 8
#define TDS_NN_MAX_UNITS 256
int tds_hip_set_policy_network(tds_hip_sim_t *sim, int num_layers, const int *layer_sizes, const int *activations,
                               const int *use_bias);
int tds_hip_policy_num_parameters(const tds_hip_sim_t *sim);

/* Blocking convenience with HOST buffers in double, any N <= num_envs:
   H2D(x) -> kernel -> D2H(y), i.e. exactly what the reference's <model>_forward_zero does. */
int tds_hip_forward_zero_host(tds_hip_sim_t *sim, int n, const double *x_host, double *y_host);
/* The same call in two halves (F64 records only): _begin enqueues H2D -> kernel -> D2H on the handle's stream and
   returns, _end waits.  A host that drives several devices enqueues every device's share before it waits
   (include/tds_hip_stepper.hpp: HipStepper with a device list). */
int tds_hip_forward_zero_host_begin(tds_hip_sim_t *sim, int n, const double *x_host, double *y_host);
int tds_hip_forward_zero_host_end(tds_hip_sim_t *sim);

/* The same call split the way the reference's newer generated-library ABI splits it
   (src/utils/cuda/cuda_function.hpp:11-20, 78-99: <fn>_send_local, then <fn>):
   send_local uploads the first n input records; forward_zero_fetch runs the kernel on the
   first n resident records and downloads their outputs. Both block. */
int tds_hip_send_local(tds_hip_sim_t *sim, int n, const double *x_host);
int tds_hip_forward_zero_fetch(tds_hip_sim_t *sim, int n, double *y_host);

/* Duration of the most recent stepping CALL (all of its launches: one for a plain step, two for the split
   auto-reset step, 2 n + 1 for a per-step-launch rollout, the whole graph for tds_hip_step_many) measured with HIP
   events on the handle's stream, in milliseconds (enabled by tds_hip_set_timing(sim, 1); synchronises). */
int tds_hip_set_timing(tds_hip_sim_t *sim, int enable);
int tds_hip_last_kernel_ms(tds_hip_sim_t *sim, float *ms);

/* Diagnostic: run y = f(x) once on the resident records with the instrumented kernel build and
   return 14 shader-clock timestamps taken by workgroup 0 at the phase boundaries
   (A load + PD, B jcalc, C kinematics sweep, I narrowphase + visuals + D inertias, E composite
    inertia / bias force sweep, G mass matrix, H LDL^T, F forward-dynamics solve, (sync), J Jacobian rows,
    K row solves, L PGS, M pack, end).
   With n >= 28 and a grid the two-wavefront workgroups serve, THAT form is profiled: 0..13 are the main
   wavefront's stamps (3 -> 4, 7 -> 8 and 8 -> 9 contain the three workgroup barriers), 14..22 the helper's (start,
   constants loaded = before barrier 1, after barrier 1, narrowphase, visual poses + y tail, Jacobian rows = before
   barrier 2, after barrier 2, row solves = before barrier 3, after barrier 3).  Synchronises the stream. */
int tds_hip_profile_phases(tds_hip_sim_t *sim, long long *cycles_host, int n);

/* Host-side counterpart of the reference's SubmitProfileTiming hook (src/base.hpp:39; World::submit_profile_timing,
   world.hpp:82-86, called around "compute multi body contacts", "solve constraints", "integrate" ..., and
   MultiBodyConstraintSolver's "inverse_mass_matrix_a", "lcpA", "solve_pgs", mb_constraint_solver.hpp:225-439): runs ONE
   step of the resident state with the instrumented kernel build and calls fn(zone, microseconds, user) once per zone —
   the reference's zone names where a phase group of the kernel corresponds to one, "forward_dynamics/..." and
   "solve constraints/..." sub-zones for the kernel's own phases, "step" for the whole.  Durations are those of workgroup
   0's instruction stream (shader clock / the device's clock rate).  Synchronises the stream. */
typedef void (*tds_hip_profile_zone_fn)(const char *zone, double microseconds, void *user);
int tds_hip_profile_zones(tds_hip_sim_t *sim, tds_hip_profile_zone_fn fn, void *user);

/* Test aid: fills the LDS of every compute unit of the handle's device with a byte pattern (0xFF = NaN in every
   scalar type) by running workgroups that own a whole CU's LDS.  The step kernels never clear LDS, so a read of a slot
   nobody wrote normally sees benign leftovers; after this call it sees the pattern — a forgotten initialisation or a
   "0 x whatever" on an unwritten slot shows up as NaN deterministically instead of once in a blue moon
   (tests/test_hip_parity.py::test_no_step_reads_stale_lds).  Synchronises. */
int tds_hip_debug_poison_lds(tds_hip_sim_t *sim, int byte_pattern);

/* Static resource usage of the step kernel for this handle (for DESIGN.md / bench). */
int tds_hip_kernel_info(const tds_hip_sim_t *sim, int *lds_bytes_per_env, int *threads_per_env,
                        int *envs_per_block);

/* Which kernel a plain single step of this handle runs (tds_hip_step / _step_obs with substeps = 1, tds_hip_forward_zero_*,
   the launches of the tds_hip_step_many graphs): 0 the general kernel (csrc/tds_kernels.hip: lanes per environment as
   tds_hip_kernel_info reports), 1 the 16-lane kernel of the star-shaped legged robots with four-link legs (csrc/tds_quad.hip:
   a root body on the reference's six virtual links + four legs of three 1-dof joints and a fixed toe — Laikago, BASELINE
   config 4; create-time option quad = 0 keeps such a model on the general kernel), 2 the 8-lane kernel of the stars with
   two-link legs (csrc/tds_oct.hip: four legs of hip + ankle, a capsule on every leg link, a sphere on the root body — the gym
   Ant, BASELINE configs 3 and 5; create-time option oct = 0).  Both take models stepped with PD control on the leg joints
   only.  3 the kernel of the fixed-base serial chains without contacts (csrc/tds_chain.hip: 2 .. 8 links, link i the child of
   link i - 1, 1-dof joints, torques given directly — cartpole and pendulum5, BASELINE configs 1 and 2; create-time option
   chain = 0).  Optional outputs: that kernel's lanes and LDS bytes per environment. */
int tds_hip_single_step_kernel(const tds_hip_sim_t *sim, int *lanes_per_env, int *lds_bytes_per_env);

/* ======================================================================================
 * Multi-GPU: the global batch of environments is cut into equal contiguous shards, one per rank /
 * GPU (rank r owns environments [r N/G, (r+1) N/G)); each shard is an ordinary tds_hip_sim on its own device, the
 * model constants are replicated, and there is NO collective inside the step.  The one exchange is an all-gather of
 * the [obs | reward | done] records, once per policy step, over RCCL / xGMI — called from here (librccl is loaded
 * with dlopen at first use), on a private communication stream, overlapped with the following steps.  New
 * functionality: the reference has no multi-device path.
 *
 *   one process per GPU:   rank 0: tds_hip_shard_unique_id(id) -> ship the 128 bytes to the other ranks by any
 *                          means (MPI, a file, torch.distributed) -> every rank: tds_hip_shard_create(..., id, ...)
 *   one process, G GPUs:   tds_hip_shard_create_all(model, N, G, devices, ...) and tds_hip_shard_group_step
 *   then per policy step:  tds_hip_shard_step(shard, actions_of_my_shard, substeps)
 *   consumer:              tds_hip_shard_gathered(shard, consumer_stream, &records, &steps_in_block): the gathered
 *                          records of the most recently submitted exchange, [world][block][n_local][obs_dim + 2]
 *                          in the wire dtype (block = 1: [N][obs_dim + 2] in global environment order); the
 *                          consumer's stream is made to wait for the exchange, the host never blocks.
 * ====================================================================================== */
#define TDS_SHARD_ID_BYTES 128 /* sizeof(ncclUniqueId) */
typedef struct tds_hip_shard tds_hip_shard_t;

/* version of the RCCL library found at run time (ncclGetVersion), 0 if none can be loaded */
int tds_hip_shard_rccl_version(void);
int tds_hip_shard_unique_id(void *id_out /* TDS_SHARD_ID_BYTES */);
/* This rank's shard of `global_envs` environments (a multiple of `world`) on HIP device `device`.
   dtype: TDS_DTYPE_* of the shard's sim.  wire_dtype: TDS_DTYPE_F32 (4 bytes per scalar on the wire, as SURVEY 8e
   sizes the exchange; F64 records are converted on the communication stream) or TDS_DTYPE_F64 (as computed).
   unique_id NULL is allowed for world == 1 only: the "exchange" is then a device copy (no RCCL needed).
   Collective: every rank of the communicator must call it. */
int tds_hip_shard_create(const tds_model_t *model, int global_envs, int rank, int world, int device, int dtype,
                         const void *unique_id, int wire_dtype, tds_hip_shard_t **out);
/* One process driving n_devices GPUs (ncclCommInitAll): out[i] = shard of rank i on devices[i]. */
int tds_hip_shard_create_all(const tds_model_t *model, int global_envs, int n_devices, const int *devices, int dtype,
                             int wire_dtype, tds_hip_shard_t **out /* [n_devices] */);
int tds_hip_shard_destroy(tds_hip_shard_t *shard);
/* the shard's simulation: every tds_hip_* call (state upload, reset, auto-reset, streams ...) applies to it */
tds_hip_sim_t *tds_hip_shard_sim(tds_hip_shard_t *shard);
int tds_hip_shard_rank(const tds_hip_shard_t *shard);
int tds_hip_shard_world(const tds_hip_shard_t *shard);
int tds_hip_shard_local_envs(const tds_hip_shard_t *shard);
int tds_hip_shard_first_env(const tds_hip_shard_t *shard); /* global index of the shard's first environment */
int tds_hip_shard_wire_bytes(const tds_hip_shard_t *shard);
/* Records of `steps_per_exchange` consecutive steps travel in ONE all-gather (default 1 = SURVEY 8e's protocol:
   one exchange per policy step).  Larger blocks trade observation latency for fewer collectives.  Flushes. */
int tds_hip_shard_set_block(tds_hip_shard_t *shard, int steps_per_exchange);
/* One closed-loop step of the shard (tds_hip_step_obs into the ring's record block) and, when the block is complete,
   its all-gather on the communication stream.  Asynchronous. */
int tds_hip_shard_step(tds_hip_shard_t *shard, const void *actions_dev, int substeps);
/* The same for all shards of one process (tds_hip_shard_create_all): the steps are enqueued device by device, the
   all-gathers go out as one RCCL group. */
int tds_hip_shard_group_step(tds_hip_shard_t **shards, int n, const void *const *actions_dev, int substeps);
/* n_steps steps of the shard INCLUDING their exchanges as one hipGraph launch (the two-stream pattern of
   tds_hip_shard_step captured once, RCCL all-gathers as graph nodes): one host call per n_steps instead of ~8 per
   step — the eager form is host-bound at ~20 us per step.  actions_dev [action_blocks][n_local][action_dim], step k
   uses block (first_block + k) % action_blocks.  n_steps: a multiple of the exchange block, <= 4096.  Collective:
   every rank makes the same call.  Falls back to eager stepping if the capture is refused (TDS_HIP_SHARD_NO_GRAPH=1
   forces that), and steps eagerly when auto-reset is on (the refill passes of the reset pool are host-driven).

   RING EXCHANGE.  Where tds_hip_step_many_is_loop holds for the shard's simulation (and the exchange block is 1) the
   n_steps steps are step-loop launches of up to 256 steps (option shard_chunk) — the very launches of
   tds_hip_step_many_rings, storing every step's record straight into this rank's block of that step's gathered slot (in-
   place all-gather; the obs ring in the wire dtype) and a y ring, i.e. the same work per step as a single GPU does — with
   one ncclAllGather per policy step on the communication stream, no kernel boundary per step, nothing the step needs
   waiting for the exchange.  WHEN a slot travels depends on the build of the launch (option exchange_w2):
     1 (default)  the two-wavefront build N = 1 runs: it fills every SIMD's register file, so nothing of the exchange
                  could run beside it — the launch runs exactly as at N = 1 and the communication stream sends its slots
                  as ONE RCCL group as soon as it has completed (beside the NEXT launch; launch j + 2 waits for the
                  exchanges of launch j);
     0            the one-wave build: every workgroup counts itself in on the slot's own device counter once its records
                  of step k are visible device-wide, and the communication stream — one bounded one-lane wait kernel (or
                  hipStreamWaitValue64: option shard_wait) + one all-gather per step — sends slot k while the launch runs
                  step k + 1, as soon as the SLOWEST workgroup has stored it.
   PEER-STORE EXCHANGE (round 5; option shard_peer, default on where every rank can set it up).  The all-gather without a
   collective: at the first tds_hip_shard_step_many every rank maps the other ranks' gathered rings and flag arrays into its
   address space (hipIpcGetMemHandle / hipIpcOpenMemHandle, the handles exchanged once over the communicator; a token round
   trip over the mapped memory checks the mapping) and from then on the step-loop launch itself — the N = 1 two-wavefront
   build — stores every step's record into its own block of the slot on EVERY rank (system-scope write-through stores over
   xGMI, issued by the helper wavefront that stores the record anyway) and raises the slot's flag on every rank when its
   last workgroup has stored it.  The transfer of step k lies inside step k + 1 of the same launch; no kernel of the
   exchange needs a compute unit beside the launch, no host call per step.  Around a LAUNCH: a one-wave credit kernel in
   front of it (ranks drift apart by at most one launch: launch m waits until every peer has started launch m - 1) and a
   one-wave arrival check behind it on the communication stream (what tds_hip_shard_gathered / _flush wait for).  If IPC
   cannot be set up on ANY rank, every rank uses the RCCL forms above (shard_peer = 2: an error instead; = 0: never
   tried).  Option exchange_fields = 1 sends only [reward | done] to the peers (8 instead of 120 bytes per Ant
   environment and step on the float wire: for runs whose policy lives on the device, tds_hip_rollout).
   Option shard_peer_copy = 1 (round 6) is the STAGED form of the same exchange: the launch stores into this rank's block
   only and raises this rank's own flags; behind it the communication stream copies the launch's slots into every peer's ring
   (one strided device-to-device copy per peer: the runtime's copy engines) and a one-wave kernel raises the flags there —
   no store over the fabric from inside the kernel, full records only.  Option shard_peer_release = 1 puts system-scope
   release fences in front of the arrival counts and the flag stores of the in-kernel form (default: s_waitcnt vmcnt(0) +
   relaxed stores): the ordering assumption of the protocol can be A/B-ed on a fabric in one run.
   tds_hip_shard_exchange_form tells which form the most recent call ran.
   Option shard_graph = 1 replays each launch + its exchanges from one hipGraph instead (cached by arguments; slower on
   ROCm 7: a chain of dependent graph nodes pays a node-to-node latency a stream does not).  tds_hip_shard_gathered then
   returns the slot of the last step, [world][n_local][obs_dim + 2].  Option shard_ring = 0 forces the per-step-launch
   form. */
int tds_hip_shard_step_many(tds_hip_shard_t *shard, const void *actions_dev, int action_blocks, int first_block,
                            int n_steps);
/* which exchange the most recent tds_hip_shard_step / _step_many of the shard ran (0: none yet) */
enum {
  TDS_EXCHANGE_NONE = 0,
  TDS_EXCHANGE_RCCL_PER_STEP = 1,     /* one kernel launch + one ncclAllGather per step (eager or from one hipGraph) */
  TDS_EXCHANGE_RCCL_AFTER_LAUNCH = 2, /* ring exchange, two-wavefront build: the launch's slots as ONE RCCL group behind it */
  TDS_EXCHANGE_RCCL_PER_SLOT = 3,     /* ring exchange, one-wave build: wait kernel + ncclAllGather per slot beside the launch */
  TDS_EXCHANGE_PEER_STORES = 4,       /* ring exchange by peer stores: no collective, the launch writes every rank's ring */
  TDS_EXCHANGE_PEER_COPY = 5          /* the same rings, flags and credit protocol, STAGED (option shard_peer_copy = 1): the launch
                                         writes this rank's ring only; behind it the communication stream copies the launch's
                                         slots into every peer's ring (one strided device-to-device copy per peer: the runtime's
                                         copy engines) and raises the flags — no collective, no fabric store from the kernel */
};
int tds_hip_shard_exchange_form(const tds_hip_shard_t *shard);
/* ranks this shard stores its records to under the peer-store exchange (0 on one rank; -1: the exchange is not in use) */
int tds_hip_shard_peer_count(const tds_hip_shard_t *shard);
/* capture + instantiate the graph of the next tds_hip_shard_step_many with the same arguments; nothing executes */
int tds_hip_shard_step_many_prepare(tds_hip_shard_t *shard, const void *actions_dev, int action_blocks,
                                    int first_block, int n_steps);
/* Introspection of the ring exchange's host arithmetic (no device needed; csrc/tds_shard_plan.h): the step-loop launches
   ("chunks", here of up to 64 steps each; a shard takes its option shard_chunk, default 256) a tds_hip_shard_step_many call of n_steps steps is cut into after chunks_done earlier
   chunks, 6 ints per chunk in out [6 * cap]: ring half | steps | first step of the call | action block of its first
   step | ring slot of its first step | progress count the communication stream waits for before it sends that slot
   (n_blocks workgroups per step; 0 = "the launch's completion").  Returns the number of chunks, -1 on bad arguments.
   tds_hip_shard_gathered_offset: scalar offset of global environment e's record in a gathered block-1 slot. */
int tds_hip_shard_ring_plan(long long chunks_done, int n_steps, int act_first, int act_blocks, int n_blocks, int *out,
                            int cap);
long long tds_hip_shard_gathered_offset(int global_env, int n_local, int width);
/* Exchange a partially filled block, then wait (host) until every exchange in flight has completed. */
int tds_hip_shard_flush(tds_hip_shard_t *shard);
int tds_hip_shard_gathered(tds_hip_shard_t *shard, void *consumer_stream, void **records_dev, int *steps_in_block);
/* Ring exchange only: the gathered records [world][n_local][obs_dim + 2] of the step `steps_back` steps before the most
   recently submitted one (0: what tds_hip_shard_gathered returns), as long as it lies inside the most recently submitted
   step-loop launch (its slots are still in the ring); the consumer's stream waits for that launch's exchanges. */
int tds_hip_shard_gathered_step(tds_hip_shard_t *shard, int steps_back, void *consumer_stream, void **records_dev);

/* ======================================================================================
 * Free rigid bodies: World::step for worlds that hold tds::RigidBody
 * objects — apply gravity, pairwise narrowphase, RigidBodyConstraintSolver (sequential impulses
 * with Baumgarte stabilisation and Coulomb friction), integrate.
 * Reference: src/world.hpp:293-366 (step), :163-204 (pairs), src/rigid_body.hpp:26-123,
 * src/rb_constraint_solver.hpp:65-168, src/contact_point.hpp:43-198,405-506: every pair the
 * reference's CollisionDispatcher knows — sphere-sphere, plane-sphere, plane-capsule (2 end spheres),
 * plane-box (8 corner spheres), capsule-sphere, each also in the swapped order; any other pair
 * produces no contact, as in the reference.  N independent worlds of the same bodies, one launch.
 * ====================================================================================== */
#define TDS_RB_MAX_BODIES 16
#define TDS_RB_STATE 13 /* per body: position(3) | orientation quaternion x y z w (4) | linear velocity(3) | angular velocity(3) */

typedef struct tds_rb_body {
  double mass;          /* 0: static (inv_mass = 0, inv_inertia = 0; rigid_body.hpp:49-53), else inv_inertia = 1 */
  int32_t geom_type;    /* TDS_GEOM_SPHERE / PLANE / CAPSULE / BOX */
  int32_t pad_;
  double radius;        /* sphere, capsule (box: radius of the rounded corners, normally 0) */
  double length;        /* capsule: distance of the end-sphere centres along local z */
  double extents[3];    /* box: full edge lengths */
  double plane_normal[3];
  double plane_constant;
} tds_rb_body_t;

typedef struct tds_rb_model {
  int32_t abi_version;  /* TDS_HIP_ABI_VERSION */
  int32_t num_bodies;   /* <= TDS_RB_MAX_BODIES */
  int32_t solver_iterations; /* World::num_solver_iterations (default 1) */
  int32_t pad_;
  double dt;
  double gravity[3];    /* World::gravity_acceleration_ (default 0 0 -9.81) */
  double restitution;   /* World::default_restitution (0) */
  double friction;      /* World::default_friction (0.5) */
  double erp;           /* RigidBodyConstraintSolver::erp_ (0.1) */
  tds_rb_body_t bodies[TDS_RB_MAX_BODIES];
} tds_rb_model_t;

typedef struct tds_rb_sim tds_rb_sim_t;

const char *tds_rb_last_error(void);
int tds_rb_create(const tds_rb_model_t *model, int num_worlds, int device, int dtype, tds_rb_sim_t **out);
int tds_rb_destroy(tds_rb_sim_t *sim);
int tds_rb_set_stream(tds_rb_sim_t *sim, void *hip_stream);
/* resident state, world-major [num_worlds][num_bodies][TDS_RB_STATE] in the compute dtype */
void *tds_rb_state_device(tds_rb_sim_t *sim);
int tds_rb_set_state(tds_rb_sim_t *sim, const double *state_host);
int tds_rb_get_state(tds_rb_sim_t *sim, double *state_host);
/* `steps` World::step calls on every world, one launch (async on the stream) */
int tds_rb_step(tds_rb_sim_t *sim, int steps);

#ifdef __cplusplus
}
#endif
#endif /* TDS_HIP_H */
