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
#include <cstring>
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
#endif  // ARS_VECTORIZED_ENVIRONMENT_H

}  // namespace tds_hip
