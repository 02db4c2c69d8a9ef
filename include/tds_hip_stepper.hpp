// tds_hip_stepper.hpp — the reference-side binding: what a TDS maintainer includes to run
// VectorizedEnvironment on libtds_hip.so.  Header-only C++17, compiled against the UNMODIFIED
// TDS headers (it needs them on the include path; it is not part of libtds_hip.so itself).
//
//   * tds_hip::flatten_multibody / flatten_world / flatten_locomotion_env
//       walk an intact tds::MultiBody + tds::World (built by TDS's own URDF loader, so every
//       loader quirk is inherited) and fill the POD blob tds_model_t of include/tds_hip.h.
//   * tds_hip::HipStepper<Algebra, Sim>
//       implements VectorizedEnvironment<Algebra,Sim>::CustomForwardDynamicsStepper
//       (reference: examples/ars/ars_vectorized_environment.h:75-85) exactly like the
//       reference's own CudaStepper (examples/ars/ars_train_policy_cuda.cpp:476-499):
//           HipStepper<MyAlgebra, Sim> stepper(env.contact_sim, batch_size);
//           vec_env.default_stepper_ = &stepper;
//
// oracle/ref_harness.cpp includes this header, so it is compile- and run-tested against the
// real reference wherever /root/reference exists.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "tds_hip.h"

namespace tds_hip {

namespace detail {
template <typename Algebra, typename M3>
inline void copy_mat3(const M3 &m, double *out) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) out[3 * r + c] = Algebra::to_double(m(r, c));
}
template <typename Algebra, typename V3>
inline void copy_vec3(const V3 &v, double *out) {
  for (int k = 0; k < 3; ++k) out[k] = Algebra::to_double(v[k]);
}
template <typename Algebra>
inline int fill_geom(const tds::Geometry<Algebra> *geom, const tds::Transform<Algebra> &X, int link, tds_geom_t *G) {
  memset(G, 0, sizeof(*G));
  G->link = link;
  G->type = geom->get_type();
  switch (G->type) {
    case tds::TINY_SPHERE_TYPE:
      G->radius = Algebra::to_double(static_cast<const tds::Sphere<Algebra> *>(geom)->get_radius());
      break;
    case tds::TINY_CAPSULE_TYPE:
      G->radius = Algebra::to_double(static_cast<const tds::Capsule<Algebra> *>(geom)->get_radius());
      G->length = Algebra::to_double(static_cast<const tds::Capsule<Algebra> *>(geom)->get_length());
      break;
    case tds::TINY_BOX_TYPE:
      copy_vec3<Algebra>(static_cast<const tds::Box<Algebra> *>(geom)->get_extents(), G->extents);
      G->radius = Algebra::to_double(static_cast<const tds::Box<Algebra> *>(geom)->get_radius());
      break;
    default:
      break;  // planes / meshes: kept for completeness, the HIP path ignores them like the reference's dispatcher
  }
  copy_mat3<Algebra>(X.rotation, G->X_rot);
  copy_vec3<Algebra>(X.translation, G->X_trans);
  return 0;
}
}  // namespace detail

// Links, collision geometry, visuals and base transform of one articulated body.
// Returns 0, or a negative code if the body exceeds the blob's capacity.
template <typename Algebra>
int flatten_multibody(const tds::MultiBody<Algebra> &mb, tds_model_t *out) {
  using namespace detail;
  out->num_links = static_cast<int>(mb.num_links());
  if (out->num_links > TDS_MAX_LINKS) return -2;
  out->dof_q = mb.dof();
  out->dof_qd = mb.dof_qd();
  out->is_floating = mb.is_floating() ? 1 : 0;
  copy_mat3<Algebra>(mb.base_X_world().rotation, out->base_X_world_rot);
  copy_vec3<Algebra>(mb.base_X_world().translation, out->base_X_world_trans);
  out->base_mass = Algebra::to_double(mb.base_rbi().mass);
  copy_vec3<Algebra>(mb.base_rbi().com, out->base_com);
  copy_mat3<Algebra>(mb.base_rbi().inertia, out->base_inertia);
  int ng = 0, nv = 0;
  for (size_t g = 0; g < mb.collision_geometries(-1).size(); ++g) {
    if (ng >= TDS_MAX_GEOMS) return -3;
    fill_geom<Algebra>(mb.collision_geometries(-1)[g], mb.collision_transforms(-1)[g], -1, &out->geoms[ng++]);
  }
  for (int i = 0; i < out->num_links; ++i) {
    const tds::Link<Algebra> &l = mb[i];
    tds_link_t &L = out->links[i];
    memset(&L, 0, sizeof(L));
    L.joint_type = static_cast<int>(l.joint_type);
    L.parent = l.parent_index;
    L.q_index = l.q_index;
    L.qd_index = l.qd_index;
    copy_mat3<Algebra>(l.X_T.rotation, L.X_T_rot);
    copy_vec3<Algebra>(l.X_T.translation, L.X_T_trans);
    for (int k = 0; k < 6; ++k) L.S[k] = Algebra::to_double(l.S[k]);
    L.mass = Algebra::to_double(l.rbi.mass);
    copy_vec3<Algebra>(l.rbi.com, L.com);
    copy_mat3<Algebra>(l.rbi.inertia, L.inertia);
    L.stiffness = Algebra::to_double(l.stiffness);
    L.damping = Algebra::to_double(l.damping);
    for (size_t g = 0; g < l.collision_geometries.size(); ++g) {
      if (ng >= TDS_MAX_GEOMS) return -3;
      fill_geom<Algebra>(l.collision_geometries[g], l.X_collisions[g], i, &out->geoms[ng++]);
    }
    for (size_t v = 0; v < l.X_visuals.size(); ++v) {
      if (nv >= TDS_MAX_VISUALS) return -4;
      tds_visual_t &V = out->visuals[nv++];
      memset(&V, 0, sizeof(V));
      V.link = i;
      copy_mat3<Algebra>(l.X_visuals[v].rotation, V.X_rot);
      copy_vec3<Algebra>(l.X_visuals[v].translation, V.X_trans);
    }
  }
  out->num_geoms = ng;
  out->num_visuals = nv;
  return 0;
}

// The next articulated body of a world with several of them, appended behind the bodies already in `out` (the first
// one put there by flatten_multibody): links, collision geometry (base geoms: link -1 - b) and visuals with their
// indices shifted behind the earlier bodies'; q / qd indices dense over all bodies.  Returns 0, or a negative code if
// the blob's capacity is exceeded.
template <typename Algebra>
int append_multibody(const tds::MultiBody<Algebra> &mb, tds_model_t *out) {
  using namespace detail;
  const int l0 = out->num_links, q0 = out->dof_q, d0 = out->dof_qd;
  const int nl = static_cast<int>(mb.num_links());
  if (l0 + nl > TDS_MAX_LINKS) return -2;
  const int b = out->num_bodies < 2 ? 1 : out->num_bodies;
  if (b >= TDS_MAX_BODIES) return -8;
  out->num_bodies = b + 1;
  tds_body_t &B = out->bodies[b];
  memset(&B, 0, sizeof(B));
  B.first_link = l0;
  B.first_geom = out->num_geoms;
  B.is_floating = mb.is_floating() ? 1 : 0;
  copy_mat3<Algebra>(mb.base_X_world().rotation, B.base_X_world_rot);
  copy_vec3<Algebra>(mb.base_X_world().translation, B.base_X_world_trans);
  B.base_mass = Algebra::to_double(mb.base_rbi().mass);
  copy_vec3<Algebra>(mb.base_rbi().com, B.base_com);
  copy_mat3<Algebra>(mb.base_rbi().inertia, B.base_inertia);
  int ng = out->num_geoms, nv = out->num_visuals;
  for (size_t g = 0; g < mb.collision_geometries(-1).size(); ++g) {
    if (ng >= TDS_MAX_GEOMS) return -3;
    fill_geom<Algebra>(mb.collision_geometries(-1)[g], mb.collision_transforms(-1)[g], -1 - b, &out->geoms[ng++]);
  }
  for (int i = 0; i < nl; ++i) {
    const tds::Link<Algebra> &l = mb[i];
    tds_link_t &L = out->links[l0 + i];
    memset(&L, 0, sizeof(L));
    L.joint_type = static_cast<int>(l.joint_type);
    L.parent = l.parent_index >= 0 ? l0 + l.parent_index : -1;
    L.q_index = l.q_index >= 0 ? q0 + l.q_index : l.q_index;
    L.qd_index = l.qd_index >= 0 ? d0 + l.qd_index : l.qd_index;
    copy_mat3<Algebra>(l.X_T.rotation, L.X_T_rot);
    copy_vec3<Algebra>(l.X_T.translation, L.X_T_trans);
    for (int k = 0; k < 6; ++k) L.S[k] = Algebra::to_double(l.S[k]);
    L.mass = Algebra::to_double(l.rbi.mass);
    copy_vec3<Algebra>(l.rbi.com, L.com);
    copy_mat3<Algebra>(l.rbi.inertia, L.inertia);
    L.stiffness = Algebra::to_double(l.stiffness);
    L.damping = Algebra::to_double(l.damping);
    for (size_t g = 0; g < l.collision_geometries.size(); ++g) {
      if (ng >= TDS_MAX_GEOMS) return -3;
      fill_geom<Algebra>(l.collision_geometries[g], l.X_collisions[g], l0 + i, &out->geoms[ng++]);
    }
    for (size_t v = 0; v < l.X_visuals.size(); ++v) {
      if (nv >= TDS_MAX_VISUALS) return -4;
      tds_visual_t &V = out->visuals[nv++];
      memset(&V, 0, sizeof(V));
      V.link = l0 + i;
      copy_mat3<Algebra>(l.X_visuals[v].rotation, V.X_rot);
      copy_vec3<Algebra>(l.X_visuals[v].translation, V.X_trans);
    }
  }
  out->num_links = l0 + nl;
  out->dof_q = q0 + mb.dof();
  out->dof_qd = d0 + mb.dof_qd();
  out->num_geoms = ng;
  out->num_visuals = nv;
  return 0;
}

// Gravity, contact-solver parameters and default contact material of the World
// (reference: src/world.hpp:65-71, src/mb_constraint_solver.hpp:59-70).
template <typename Algebra>
void flatten_world(tds::World<Algebra> &world, tds_model_t *out) {
  detail::copy_vec3<Algebra>(world.get_gravity(), out->gravity);
  auto *solver = world.get_mb_constraint_solver();
  out->pgs_iterations = solver->pgs_iterations_;
  out->cfm = Algebra::to_double(solver->cfm_);
  out->erp = Algebra::to_double(solver->erp_);
  out->friction = Algebra::to_double(world.default_friction);
  out->restitution = Algebra::to_double(world.default_restitution);
}

// The implicit ground plane is multi_bodies_[0] of the env's World (it is loaded first,
// locomotion_contact_simulation.h:100-123) but World keeps its bodies private; reach it through
// the public contact list of one dry-run World::step at q = 0.  Returns 0 on success.
template <typename Algebra>
int flatten_plane(tds::World<Algebra> &world, tds::MultiBody<Algebra> &robot, double dt, tds_model_t *out) {
  robot.initialize();
  auto qd_save = robot.qd();
  tds::forward_kinematics(robot, robot.q(), robot.qd());
  world.step(Algebra::from_double(dt));
  robot.qd() = qd_save;
  if (world.mb_contacts_.empty() || world.mb_contacts_[0].empty()) return -5;
  const auto &cp = world.mb_contacts_[0][0];
  if (cp.multi_body_b != &robot) return -7;  // plane must be body a (dispatcher swap path)
  const auto &pg = cp.multi_body_a->collision_geometries(-1);
  if (pg.size() != 1 || pg[0]->get_type() != tds::TINY_PLANE_TYPE) return -6;
  const auto *plane = static_cast<const tds::Plane<Algebra> *>(pg[0]);
  detail::copy_vec3<Algebra>(plane->get_normal(), out->plane_normal);
  out->plane_constant = Algebra::to_double(plane->get_constant());
  out->has_plane = 1;
  robot.initialize();
  return 0;
}

// Everything for a LocomotionContactSimulation-derived environment (AntContactSimulation2,
// LaikagoContactSimulation, ...): x = [q | qd | action | kp kd max_force], PD step mode.
template <typename Algebra, typename Sim>
int flatten_locomotion_env(Sim &sim, tds_model_t *out, int reward_mode = TDS_REWARD_NONE) {
  memset(out, 0, sizeof(*out));
  out->abi_version = TDS_HIP_ABI_VERSION;
  out->step_mode = TDS_STEP_LOCOMOTION;
  int rc = flatten_multibody<Algebra>(*sim.mb_, out);
  if (rc) return rc;
  flatten_world<Algebra>(sim.world, out);
  out->dt = Algebra::to_double(sim.dt);
  out->action_dim = sim.action_dim();
  if (out->action_dim > TDS_MAX_ACTIONS) return -1;
  out->pd_start_link = sim.mb_->is_floating() ? 0 : sim.base_dof_;
  out->input_dim = sim.input_dim_with_action_and_variables();
  out->output_dim = sim.output_dim();
  out->pack_visuals = 1;
  out->reward_mode = reward_mode;
  // what the environment's own reset() hands out as observation: AntContactSimulation2 zeroes the base x, y
  // (ant_environment2.h:162-163), Laikago / Humanoid return the state as it is (laikago_environment2.h:63-116)
  out->reset_obs_raw_xy = (reward_mode == TDS_REWARD_LAIKAGO || reward_mode == TDS_REWARD_HUMANOID) ? 1 : 0;
  out->action_limit = 0.4;  // locomotion_contact_simulation.h:234
  for (size_t i = 0; i < sim.initial_poses_.size(); ++i) out->initial_poses[i] = Algebra::to_double(sim.initial_poses_[i]);
  // reset distribution of the fixed-base locomotion envs (ant_environment2.h:124-135,
  // laikago_environment2.h:78-90): base at m_start_base_position with zero rpy, joints at
  // initial_poses + 0.05 * U(-1,1), 10 settle steps
  if (!sim.mb_->is_floating() && out->dof_q <= TDS_MAX_DOF) {
    for (int k = 0; k < 3; ++k) out->reset_q[k] = Algebra::to_double(sim.m_start_base_position[k]);
    // xyz + xyz-rotation base: six scalars, rotation zero; xyz + spherical base (HumanoidEnv, base_dof_ = 7,
    // humanoid_environment.h:100-111): identity quaternion (0, 0, 0, 1), joints from coordinate 7
    const bool spherical_base = sim.mb_->num_links() > 3 && (*sim.mb_)[3].joint_type == tds::JOINT_SPHERICAL;
    if (spherical_base) out->reset_q[6] = 1.0;
    const int qoffset = spherical_base ? 7 : 6;
    for (size_t j = 0; j < sim.initial_poses_.size() && qoffset + (int)j < out->dof_q; ++j) {
      out->reset_q[qoffset + j] = Algebra::to_double(sim.initial_poses_[j]);
      out->reset_noise[qoffset + j] = 0.05;
    }
    out->settle_steps = 10;
  }
  // floating base (laikago_environment2.h:65-77): start orientation and position, joints at initial_poses
  // without noise, 10 settle steps
  if (sim.mb_->is_floating() && out->dof_q <= TDS_MAX_DOF) {
    for (int k = 0; k < 4; ++k) out->reset_q[k] = Algebra::to_double(sim.m_start_base_orientation[k]);
    for (int k = 0; k < 3; ++k) out->reset_q[4 + k] = Algebra::to_double(sim.m_start_base_position[k]);
    for (size_t j = 0; j < sim.initial_poses_.size() && 7 + (int)j < out->dof_q; ++j)
      out->reset_q[7 + j] = Algebra::to_double(sim.initial_poses_[j]);
    out->settle_steps = 10;
  }
  out->plane_normal[2] = 1.0;
  rc = flatten_plane<Algebra>(sim.world, *sim.mb_, out->dt, out);
  if (rc) return rc;
  snprintf(out->name, sizeof(out->name), "%s", sim.env_name().c_str());
  return 0;
}

#ifdef ARS_VECTORIZED_ENVIRONMENT_H
// Drop-in for the reference's CudaStepper.  Errors are reported the way the reference's generated
// library reports them for launch/allocation failures — fprintf(stderr) + exit — unless
// throw_on_error is set, in which case std::runtime_error is thrown instead.
template <typename Algebra, typename Sim>
struct HipStepper : public VectorizedEnvironment<Algebra, Sim>::CustomForwardDynamicsStepper {
  using Scalar = typename Algebra::Scalar;
  tds_model_t model_;
  // one simulation per device: device d steps the contiguous block [first_[d], first_[d + 1]) of the batch
  std::vector<tds_hip_sim_t *> sims_;
  std::vector<int> first_;
  int batch_size_;
  bool throw_on_error_;
  std::vector<double> in_, out_;

  HipStepper(Sim &contact_sim, int batch_size, int device = 0, bool throw_on_error = false,
             int reward_mode = TDS_REWARD_NONE)
      : HipStepper(contact_sim, batch_size, std::vector<int>(1, device), throw_on_error, reward_mode) {}

  // Several GPUs behind the same plug-in: the batch is cut into equal contiguous blocks, one per entry of `devices`
  // (an entry may repeat: two handles on one GPU), every call enqueues all devices' shares before it waits for any.
  HipStepper(Sim &contact_sim, int batch_size, const std::vector<int> &devices, bool throw_on_error = false,
             int reward_mode = TDS_REWARD_NONE)
      : batch_size_(batch_size), throw_on_error_(throw_on_error) {
    if (devices.empty() || batch_size < (int)devices.size()) fail("bad device list", TDS_ERR_INVALID_ARG);
    int rc = flatten_locomotion_env<Algebra>(contact_sim, &model_, reward_mode);
    if (rc) fail("flatten_locomotion_env failed", rc);
    const int nd = (int)devices.size();
    first_.resize(nd + 1);
    for (int d = 0; d <= nd; ++d) first_[d] = (int)((long long)batch_size * d / nd);
    sims_.assign(nd, nullptr);
    for (int d = 0; d < nd; ++d) {
      rc = tds_hip_create(&model_, first_[d + 1] - first_[d], devices[d], TDS_DTYPE_F64, &sims_[d]);
      if (rc != TDS_OK) {
        // (a constructor that throws is not followed by its destructor: the handles made so far go first)
        const std::string why = tds_hip_last_error();
        for (tds_hip_sim_t *&sp : sims_) {
          if (sp) tds_hip_destroy(sp);
          sp = nullptr;
        }
        fail(why.c_str(), rc);
      }
    }
    in_.resize((size_t)batch_size * model_.input_dim);
    out_.resize((size_t)batch_size * model_.output_dim);
  }
  virtual ~HipStepper() {
    for (tds_hip_sim_t *s : sims_) tds_hip_destroy(s);
  }

  void fail(const char *what, int rc) {
    std::string msg = std::string("tds_hip: ") + what + " (code " + std::to_string(rc) + ")";
    if (throw_on_error_) throw std::runtime_error(msg);
    fprintf(stderr, "%s\n", msg.c_str());
    exit(rc);
  }

  // Same contract as CudaStepper::step: every env is stepped (dones are ignored, as the
  // reference CUDA stepper does, ars_train_policy_cuda.cpp:492-499); blocking.
  void step(const std::vector<std::vector<Scalar>> &thread_inputs, std::vector<std::vector<Scalar>> &thread_outputs,
            std::vector<bool> &dones, int num_threads_per_block = 32,
            const std::vector<Scalar> &global_input = {}) override {
    (void)dones;
    (void)num_threads_per_block;
    (void)global_input;
    const int n = static_cast<int>(thread_inputs.size());
    if (n > batch_size_ || (int)thread_outputs.size() != n) fail("batch size mismatch", TDS_ERR_INVALID_ARG);
    const int in = model_.input_dim, od = model_.output_dim;
    for (int e = 0; e < n; ++e) {
      if ((int)thread_inputs[e].size() != in) fail("input record size mismatch", TDS_ERR_INVALID_ARG);
      for (int k = 0; k < in; ++k) in_[(size_t)e * in + k] = Algebra::to_double(thread_inputs[e][k]);
    }
    // enqueue every device's share (H2D -> kernel -> D2H on its own stream), then wait for all of them
    const int nd = (int)sims_.size();
    for (int d = 0; d < nd; ++d) {
      const int lo = first_[d] < n ? first_[d] : n, hi = first_[d + 1] < n ? first_[d + 1] : n;
      if (hi <= lo) continue;
      int rc = tds_hip_forward_zero_host_begin(sims_[d], hi - lo, in_.data() + (size_t)lo * in, out_.data() + (size_t)lo * od);
      if (rc != TDS_OK) fail(tds_hip_last_error(), rc);
    }
    for (int d = 0; d < nd; ++d) {
      if ((first_[d] < n ? first_[d] : n) >= (first_[d + 1] < n ? first_[d + 1] : n)) continue;
      int rc = tds_hip_forward_zero_host_end(sims_[d]);
      if (rc != TDS_OK) fail(tds_hip_last_error(), rc);
    }
    for (int e = 0; e < n; ++e) {
      if ((int)thread_outputs[e].size() != od) fail("output record size mismatch", TDS_ERR_INVALID_ARG);
      for (int k = 0; k < od; ++k) thread_outputs[e][k] = Algebra::from_double(out_[(size_t)e * od + k]);
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// tds_hip::VectorizedEnv<Algebra, Sim> — the reference's VectorizedEnvironment (examples/ars/ars_vectorized_environment.h:
// 141-300) with the environments RESIDENT on the GPU.  Same public surface — the members Worker<> touches
// (neural_networks_, observation_dim_, sim_states_with_graphics_), seed(), reset(config), step(actions, observations,
// rewards, dones, config), policy(), init_neural_network() — so that the reference's own rollout loop compiles against it
// unchanged:   Worker<tds_hip::VectorizedEnv<Alg, Sim>> worker(env, ...)   (examples/ars/ars_vectorized_worker.h:9-160).
//
// Where HipStepper (above) replaces the STEPPER — and therefore moves every state record to the device and back on every
// step (268 us per step at 4096 Ant environments) — this class replaces the ENVIRONMENT: state, PD control, physics,
// reward / done and auto-reset live on the device (tds_hip_step_obs); step() moves only what its signature names, the
// actions up and [obs | reward | done] down (+ the y records while fetch_graphics_ is set: Worker::rollouts reads
// sim_states_with_graphics_ every step).  No H2D / D2H of state.  Beyond the reference's surface, for callers that keep
// actions / policies on the device too (no HIP headers needed, the buffers come from tds_hip_device_alloc):
//   step_many_device   K steps per call with per-step record rings (tds_hip_step_many_rings) — the 2e8 env-steps/s path
//   rollouts_on_device Worker::rollouts in one call, the environments' own linear policies evaluated on the device
// Copies of the object share the device state (Worker<> keeps its environment BY VALUE).
//
// reset().  Default: on the device (tds_hip_reset / tds_hip_set_auto_reset: the reset distribution of the model blob and
// the settle steps, counter-based random stream — NOT std::rand's).  host_reset_ = true: through the reference's own
// contact_sim.reset() on the host, state uploaded afterwards — the std::rand stream and therefore the trajectories of
// the reference, environment for environment (what the parity test runs); auto_reset_when_done then resets done
// environments on the host as well.
// A done environment keeps stepping (as under the reference's CudaStepper, which ignores `dones`); without
// auto_reset_when_done its reward reads 0 and `done` stays set, as in VectorizedEnvironment::step.
// ---------------------------------------------------------------------------------------------------------------------
template <typename Algebra, typename Sim>
struct VectorizedEnv {
  using Scalar = typename Algebra::Scalar;
  struct Handle {
    tds_hip_sim_t *h = nullptr;
    ~Handle() {
      if (h) tds_hip_destroy(h);
    }
  };

  Sim &contact_sim;
  std::vector<std::vector<Scalar>> sim_states_;                           // [q | qd] mirror (refreshed with the y records)
  std::vector<std::vector<Scalar>> sim_states_with_action_and_variables;  // (surface parity; the device builds its own)
  std::vector<std::vector<Scalar>> sim_states_with_graphics_;             // y record of the last step per environment
  std::vector<tds::NeuralNetwork<Algebra>> neural_networks_;
  int observation_dim_{0};
  bool host_reset_ = false;     // resets through contact_sim.reset() on the host (the reference's std::rand stream)
  bool fetch_graphics_ = true;  // step() also brings the y records down (sim_states_with_graphics_, sim_states_)
  bool throw_on_error_ = true;

  VectorizedEnv(Sim &sim, int batch_size, int reward_mode, int device = 0, int dtype = TDS_DTYPE_F64)
      : contact_sim(sim), batch_size_(batch_size), handle_(std::make_shared<Handle>()) {
    int rc = flatten_locomotion_env<Algebra>(contact_sim, &model_, reward_mode);
    if (rc) fail("flatten_locomotion_env failed", rc);
    rc = tds_hip_create(&model_, batch_size, device, dtype, &handle_->h);
    if (rc != TDS_OK) fail(tds_hip_last_error(), rc);
    observation_dim_ = contact_sim.input_dim();
    od_ = observation_dim_;
    neural_networks_.resize(batch_size);
    sim_states_.resize(batch_size);
    sim_states_with_action_and_variables.resize(batch_size);
    sim_states_with_graphics_.resize(batch_size);
    for (int e = 0; e < batch_size; ++e) {  // the policy VectorizedEnvironment builds (:165-180)
      neural_networks_[e].set_input_dim(observation_dim_, false);
      neural_networks_[e].add_linear_layer(tds::NN_ACT_IDENTITY, sim.action_dim(), true);
    }
    act_.resize((size_t)batch_size * model_.action_dim);
    rec_.resize((size_t)batch_size * (od_ + 2));
    y_.resize((size_t)batch_size * model_.output_dim);
    // the x records carry kp, kd, max_force in their last slots (prepare_sim_state_with_action_and_variables,
    // locomotion_contact_simulation.h:138-148): filled once, resident from then on
    std::vector<double> x((size_t)batch_size * model_.input_dim, 0.0);
    std::vector<Scalar> v(contact_sim.input_dim_with_action_and_variables(), Scalar(0)), zero_act(sim.action_dim(), Scalar(0));
    contact_sim.prepare_sim_state_with_action_and_variables(v, zero_act);
    for (int e = 0; e < batch_size; ++e)
      for (int k = 0; k < model_.input_dim; ++k) x[(size_t)e * model_.input_dim + k] = Algebra::to_double(v[k]);
    rc = tds_hip_set_inputs(handle_->h, x.data());
    if (rc != TDS_OK) fail(tds_hip_last_error(), rc);
  }
  virtual ~VectorizedEnv() {}

  tds_hip_sim_t *handle() const { return handle_->h; }
  const tds_model_t &model() const { return model_; }

  void init_neural_network(int index, const std::vector<double> &x) { neural_networks_[index].set_parameters(x); }

  void seed(long long int s) {
    std::srand(s);  // (host_reset_: the stream contact_sim.reset() draws from, as in the reference)
    seed_ = (unsigned long long)s;
    auto_reset_set_ = -1;
  }

  std::vector<std::vector<double>> reset(const ARSConfig &config) {
    check_batch(config);
    std::vector<std::vector<double>> observations(batch_size_);
    if (host_reset_) {
      std::vector<double> qqd((size_t)batch_size_ * od_);
      for (int e = 0; e < batch_size_; ++e) {
        sim_states_[e].resize(0);
        sim_states_[e].resize(contact_sim.input_dim_with_action_and_variables(), Scalar(0));
        observations[e].resize(contact_sim.input_dim());
        contact_sim.reset(sim_states_[e], observations[e]);
        for (int k = 0; k < od_; ++k) qqd[(size_t)e * od_ + k] = Algebra::to_double(sim_states_[e][k]);
      }
      const int rc = tds_hip_set_states(handle_->h, qqd.data());
      if (rc != TDS_OK) fail(tds_hip_last_error(), rc);
      return observations;
    }
    apply_auto_reset(config);
    const int rc = tds_hip_reset_host(handle_->h, nullptr, rec_.data());
    if (rc != TDS_OK) fail(tds_hip_last_error(), rc);
    for (int e = 0; e < batch_size_; ++e) observations[e].assign(rec_.begin() + (size_t)e * (od_ + 2), rec_.begin() + (size_t)e * (od_ + 2) + od_);
    return observations;
  }

  void step(std::vector<std::vector<double>> &actions, std::vector<std::vector<double>> &observations,
            std::vector<double> &rewards, std::vector<bool> &dones, const ARSConfig &config) {
    check_batch(config);
    apply_auto_reset(config);
    const int adim = model_.action_dim, out = model_.output_dim;
    for (int e = 0; e < batch_size_; ++e) {
      if ((int)actions[e].size() < adim) fail("action vector too short", TDS_ERR_INVALID_ARG);
      for (int k = 0; k < adim; ++k) act_[(size_t)e * adim + k] = actions[e][k];
    }
    const bool want_y = fetch_graphics_ || (host_reset_ && config.auto_reset_when_done);
    const int rc = tds_hip_step_host(handle_->h, act_.data(), 1, rec_.data(), want_y ? y_.data() : nullptr);
    if (rc != TDS_OK) fail(tds_hip_last_error(), rc);
    bool any_host_reset = false;
    for (int e = 0; e < batch_size_; ++e) {
      const double *r = rec_.data() + (size_t)e * (od_ + 2);
      if (want_y) {
        sim_states_with_graphics_[e].assign(y_.begin() + (size_t)e * out, y_.begin() + (size_t)(e + 1) * out);
        sim_states_[e].assign(y_.begin() + (size_t)e * out, y_.begin() + (size_t)e * out + od_);
      }
      observations[e].assign(r, r + od_);  // (obs[0] = obs[1] = 0: the kernel's record, :283-288)
      if (!dones[e] || config.auto_reset_when_done) {
        rewards[e] = r[od_];
        const bool done = r[od_ + 1] != 0.0;
        if (done && config.auto_reset_when_done && host_reset_) {
          sim_states_[e].resize(0);
          sim_states_[e].resize(contact_sim.input_dim_with_action_and_variables(), Scalar(0));
          observations[e].resize(contact_sim.input_dim());
          contact_sim.reset(sim_states_[e], observations[e]);
          sim_states_[e].resize(od_);
          // (the reference then OVERWRITES the observation reset() wrote: observations = sim_states_ resized to input_dim
          //  with [0] = [1] = 0, ars_vectorized_environment.h:282-288 — a reset pose with non-zero base x, y shows the
          //  difference)
          observations[e].assign(sim_states_[e].begin(), sim_states_[e].begin() + od_);
          if (od_ > 1) observations[e][0] = observations[e][1] = 0.;
          any_host_reset = true;
        }
        dones[e] = done;
      } else {
        rewards[e] = 0;
      }
    }
    if (any_host_reset) {  // (parity mode: the fresh states of the environments the host has just reset go up)
      std::vector<double> qqd((size_t)batch_size_ * od_);
      for (int e = 0; e < batch_size_; ++e)
        for (int k = 0; k < od_; ++k) qqd[(size_t)e * od_ + k] = Algebra::to_double(sim_states_[e][k]);
      const int rc2 = tds_hip_set_states(handle_->h, qqd.data());
      if (rc2 != TDS_OK) fail(tds_hip_last_error(), rc2);
    }
  }

  inline const std::vector<double> policy(int index, const std::vector<double> &obs) {
    std::vector<double> action(neural_networks_[index].input_dim(), Scalar(0));
    neural_networks_[index].compute(obs, action);
    return action;
  }

  // ---- beyond the reference's surface: nothing but the call crosses PCIe --------------------------------------------
  // K closed-loop steps in ONE call, step k taking block (first_block + k) % action_blocks of an action pool in device
  // memory ([action_blocks][N][action_dim] doubles, tds_hip_device_alloc + tds_hip_device_upload), every step leaving its
  // [obs | reward | done] and y records in the rings (tds_hip_step_many_rings).  Asynchronous; tds_hip_sync(handle()).
  void step_many_device(const void *actions_dev, int action_blocks, int first_block, int n_steps, const tds_hip_rings_t &rings,
                        const ARSConfig &config) {
    apply_auto_reset(config);
    const int rc = tds_hip_step_many_rings(handle_->h, actions_dev, action_blocks, first_block, n_steps, &rings);
    if (rc != TDS_OK) fail(tds_hip_last_error(), rc);
  }
  // Worker::rollouts (ars_vectorized_worker.h:51-140) as ONE device call: reset, then rollout_length times { the
  // environment's own policy (neural_networks_[e]: the linear layer with bias), step, reward / done, return bookkeeping }.
  void rollouts_on_device(double shift, int rollout_length, std::vector<double> &total_rewards, std::vector<int> &vec_steps,
                          const ARSConfig &config) {
    check_batch(config);
    apply_auto_reset(config);
    const int np = neural_networks_[0].num_parameters();
    std::vector<double> params((size_t)batch_size_ * np);
    for (int e = 0; e < batch_size_; ++e) {
      // (NeuralNetwork parameter order, neural_network.hpp:406-415: all weights, then all biases)
      const auto &W = neural_networks_[e].weights;
      const auto &B = neural_networks_[e].biases;
      size_t i = 0;
      for (size_t k = 0; k < W.size(); ++k) params[(size_t)e * np + i++] = W[k];
      for (size_t k = 0; k < B.size(); ++k) params[(size_t)e * np + i++] = B[k];
    }
    // (the three device buffers are released on every way out, fail() throwing included)
    struct DeviceBuffer {
      tds_hip_sim_t *h;
      void *p = nullptr;
      explicit DeviceBuffer(tds_hip_sim_t *handle) : h(handle) {}
      ~DeviceBuffer() {
        if (p) tds_hip_device_free(h, p);
      }
      DeviceBuffer(const DeviceBuffer &) = delete;
      DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    } pol(handle_->h), ret(handle_->h), cnt(handle_->h);
    auto chk = [&](int rc) {
      if (rc != TDS_OK) fail(tds_hip_last_error(), rc);
    };
    chk(tds_hip_device_alloc(handle_->h, params.size() * 8, &pol.p));
    chk(tds_hip_device_alloc(handle_->h, (size_t)batch_size_ * 8, &ret.p));
    chk(tds_hip_device_alloc(handle_->h, (size_t)batch_size_ * sizeof(int), &cnt.p));
    chk(tds_hip_device_upload(handle_->h, pol.p, params.data(), params.size() * 8));
    chk(tds_hip_reset(handle_->h, nullptr, nullptr));
    chk(tds_hip_rollout(handle_->h, pol.p, rollout_length, shift, /*first step sees the raw base x, y*/ 1, ret.p, (int *)cnt.p, nullptr));
    total_rewards.resize(batch_size_);
    vec_steps.resize(batch_size_);
    chk(tds_hip_device_download(handle_->h, total_rewards.data(), ret.p, (size_t)batch_size_ * 8));
    chk(tds_hip_device_download(handle_->h, vec_steps.data(), cnt.p, (size_t)batch_size_ * sizeof(int)));
  }

 private:
  tds_model_t model_;
  int batch_size_ = 0, od_ = 0, auto_reset_set_ = -1;
  unsigned long long seed_ = 0x5DEECE66Dull;
  std::shared_ptr<Handle> handle_;
  std::vector<double> act_, rec_, y_;

  void check_batch(const ARSConfig &config) {
    if (config.batch_size != batch_size_) fail("config.batch_size differs from the environment's batch", TDS_ERR_INVALID_ARG);
  }
  // auto_reset_when_done travels in the config of every call (as in the reference): forwarded when it changes
  void apply_auto_reset(const ARSConfig &config) {
    const int want = (config.auto_reset_when_done && !host_reset_) ? 1 : 0;
    if (want == auto_reset_set_) return;
    const int rc = tds_hip_set_auto_reset(handle_->h, want, seed_);
    if (rc != TDS_OK) fail(tds_hip_last_error(), rc);
    auto_reset_set_ = want;
  }
  void fail(const char *what, int rc) {
    std::string msg = std::string("tds_hip::VectorizedEnv: ") + what + " (code " + std::to_string(rc) + ")";
    if (throw_on_error_) throw std::runtime_error(msg);
    fprintf(stderr, "%s\n", msg.c_str());
    exit(rc);
  }
};
#endif  // ARS_VECTORIZED_ENVIRONMENT_H

}  // namespace tds_hip
