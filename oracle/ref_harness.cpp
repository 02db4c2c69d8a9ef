// ref_harness.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Compiles the *unmodified* reference headers where they lie under /root/reference
// (recipe: oracle/Makefile, target _ref/libtds_ref.so) and exposes them over a tiny C ABI so
// that python tests / gen_golden.py can
//   (a) flatten the reference's own MultiBody + World into the tds_model_t blob
//       (same traversal a maintainer would do with include/tds_hip_stepper.hpp), and
//   (b) run the reference's own step as ground truth:
//         examples/environments/locomotion_contact_simulation.h:151-304 (Ant / Laikago)
//         examples/environments/cartpole_environment.h:88-94            (free ABA + Euler)
// Nothing here re-implements the algorithm; it only calls the reference.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <string>
#include <vector>

#include "math/tiny/tiny_algebra.hpp"
#include "math/tiny/tiny_double_utils.h"

using namespace TINY;
using namespace tds;
typedef TinyAlgebra<double, ::TINY::DoubleUtils> Alg;

#include "ant_environment2.h"
#include "laikago_environment2.h"
#include "humanoid_environment.h"
#include "dynamics/mass_matrix.hpp"
#include "dynamics/jacobian.hpp"
#include "../ars/ars_vectorized_environment.h"
#include "../ars/running_stat.h"
#include "../ars/shared_noise_table.h"
// what the reference's training programs define ahead of ars_vectorized_worker.h (ars_train_policy_cuda.cpp:58, 164-169)
struct PolicyParams {};
static void visualize_trajectories(std::vector<std::vector<std::vector<double>>> &, int, bool, int) {}
#include "../ars/ars_vectorized_worker.h"
#include <chrono>

#include "tds_hip.h"
#include "tds_hip_stepper.hpp"

namespace {

typedef LocomotionContactSimulation<Alg, 3> LocoSim;

static void copy_mat3(const Alg::Matrix3 &m, double *out) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) out[3 * r + c] = m(r, c);
}
static void copy_vec3(const Alg::Vector3 &v, double *out) {
  out[0] = v[0];
  out[1] = v[1];
  out[2] = v[2];
}

struct RefSim {
  std::string name;
  // exactly one of the two is used
  AntEnv2<Alg> *ant = nullptr;
  LaikagoEnv<Alg> *laikago = nullptr;
  // the env step on a FLOATING-base robot: LaikagoContactSimulation constructed with floating = true on
  // laikago/laikago_toes_zup.urdf (the constructor the reference offers, laikago_environment2.h:36-41)
  LaikagoContactSimulation<Alg> *lfloat = nullptr;
  // HumanoidEnv (humanoid_environment.h:212-232): xyz base + spherical joints, PD with the spherical branch
  HumanoidEnv<Alg> *humanoid = nullptr;
  // the PD loop's SPHERICAL branch (locomotion_contact_simulation.h:188-226): no environment of the reference
  // reaches it (HumanoidEnv skips its only spherical joint with base_dof_ = 7), so the env step is constructed
  // on URDFs of the reference's data directory whose spherical joints lie behind base_dof_
  HumanoidContactSimulation<Alg> *sphpd = nullptr;
  // generic model (URDF file from the reference data dir, optional plane)
  UrdfCache<Alg> cache;
  World<Alg> *gworld = nullptr;
  MultiBody<Alg> *gmb = nullptr;
  // all articulated bodies of a world with several of them, gmb first
  // ("two:<fileA>:<fileB>:<dx>:<dy>:<dz>[+capsA][+capsB][+plane]", "multi:<file>@x,y,z[/caps][/floating]:...[+plane]")
  std::vector<MultiBody<Alg> *> gbodies;
  bool g_plane = false;
  double g_dt = 1e-3;

  World<Alg> &world() {
    if (ant) return ant->contact_sim.world;
    if (laikago) return laikago->contact_sim.world;
    if (lfloat) return lfloat->world;
    if (humanoid) return humanoid->contact_sim.world;
    if (sphpd) return sphpd->world;
    return *gworld;
  }
  MultiBody<Alg> *mb() {
    if (ant) return ant->contact_sim.mb_;
    if (laikago) return laikago->contact_sim.mb_;
    if (lfloat) return lfloat->mb_;
    if (humanoid) return humanoid->contact_sim.mb_;
    if (sphpd) return sphpd->mb_;
    return gmb;
  }
  LocoSim *loco() {
    if (ant) return &ant->contact_sim;
    if (laikago) return &laikago->contact_sim;
    if (lfloat) return lfloat;
    if (humanoid) return &humanoid->contact_sim;
    if (sphpd) return sphpd;
    return nullptr;
  }
  bool has_plane() { return loco() ? true : g_plane; }
  double dt() { return loco() ? loco()->dt : g_dt; }

  int input_dim() {
    if (loco()) return loco()->input_dim_with_action_and_variables();
    if (!gbodies.empty()) {  // [q | qd | tau] over all bodies
      int n = 0;
      for (MultiBody<Alg> *m : gbodies) n += m->dof() + m->dof_qd() + m->dof_actuated();
      return n;
    }
    return gmb->dof() + gmb->dof_qd() + gmb->dof_actuated();  // == dof_qd for a fixed base, dof_qd - 6 floating
  }
  int num_visuals() {
    int n = 0;
    for (const auto &l : *mb()) n += (int)l.X_visuals.size();
    for (size_t b = 1; b < gbodies.size(); ++b)
      for (const auto &l : *gbodies[b]) n += (int)l.X_visuals.size();
    return n;
  }
  int output_dim() {
    // state_dim() counts num_links * num_visuals scalars for the visual poses (locomotion_contact_simulation.h:75-77)
    // but 7 are written per visual (:281-299): with fewer than 7 links (pendulum5_sph_pd) the reference would write
    // past its own output_dim.  For these constructions the record is as long as what the step writes.
    if (sphpd) return std::max(sphpd->output_dim(), sphpd->mb_->dof() + sphpd->mb_->dof_qd() + 7 * num_visuals() + 1);
    if (loco()) return loco()->output_dim();
    if (!gbodies.empty()) {
      int n = 7 * num_visuals() + 1;
      for (MultiBody<Alg> *m : gbodies) n += m->dof() + m->dof_qd();
      return n;
    }
    return gmb->dof() + gmb->dof_qd() + 7 * num_visuals() + 1;
  }

  // world of several articulated bodies: x = [q_0 q_1 .. | qd_0 qd_1 .. | tau_0 tau_1 ..]; the reference's call sequence
  // per body around ONE World::step (contacts plane-0, plane-1, .., 0-1, 0-2, .., 1-2, ..; resolve_collision pair by
  // pair, world.hpp:293-366):
  //   forward_dynamics(b) for all b; clear_forces; integrate_euler_qdd(b); world.step; integrate_euler(b).
  //   y = [q_0 q_1 .. | qd_0 qd_1 .. | visual poses body by body | base_0.R(2,2)]
  void multi_body_step(const double *x, double *y) {
    const std::vector<MultiBody<Alg> *> &bodies = gbodies;
    int nq = 0, nd = 0;
    for (MultiBody<Alg> *m : bodies) { nq += m->dof(); nd += m->dof_qd(); }
    int oq = 0, od = 0, ot = 0;
    for (MultiBody<Alg> *m : bodies) {
      m->initialize();
      for (int i = 0; i < m->dof(); ++i) m->q(i) = x[oq + i];
      for (int i = 0; i < m->dof_qd(); ++i) m->qd(i) = x[nq + od + i];
      for (int i = 0; i < m->dof_actuated(); ++i) m->tau(i) = x[nq + nd + ot + i];
      oq += m->dof();
      od += m->dof_qd();
      ot += m->dof_actuated();
    }
    for (MultiBody<Alg> *m : bodies) {
      forward_dynamics(*m, gworld->get_gravity());
      m->clear_forces();
    }
    for (MultiBody<Alg> *m : bodies) integrate_euler_qdd(*m, g_dt);
    gworld->step(g_dt);
    for (MultiBody<Alg> *m : bodies) integrate_euler(*m, g_dt);
    int j = 0;
    for (MultiBody<Alg> *m : bodies)
      for (int i = 0; i < m->dof(); ++i) y[j++] = m->q(i);
    for (MultiBody<Alg> *m : bodies)
      for (int i = 0; i < m->dof_qd(); ++i) y[j++] = m->qd(i);
    for (MultiBody<Alg> *m : bodies)
      for (const auto &link : *m)
        for (size_t v = 0; v < link.X_visuals.size(); ++v) {
          auto vx = link.X_world * link.X_visuals[v];
          y[j++] = vx.translation[0];
          y[j++] = vx.translation[1];
          y[j++] = vx.translation[2];
          auto orn = Alg::matrix_to_quat(vx.rotation);
          y[j++] = orn.x();
          y[j++] = orn.y();
          y[j++] = orn.z();
          y[j++] = orn.w();
        }
    y[j++] = gmb->get_world_transform(-1).rotation(2, 2);
  }

  // generic step: the reference call sequence with tau given directly.
  //   no plane : forward_dynamics; clear_forces; integrate_euler          (cartpole_environment.h:88-94)
  //   plane    : forward_dynamics; clear_forces; integrate_euler_qdd; world.step; integrate_euler
  //              (locomotion_contact_simulation.h:261-269, examples/soft_contact_example.cpp:107-115)
  void generic_step(const double *x, double *y) {
    MultiBody<Alg> &m = *gmb;
    m.initialize();
    for (int i = 0; i < m.dof(); ++i) m.q(i) = x[i];
    for (int i = 0; i < m.dof_qd(); ++i) m.qd(i) = x[m.dof() + i];
    for (int i = 0; i < m.dof_actuated(); ++i) m.tau(i) = x[m.dof() + m.dof_qd() + i];
    forward_dynamics(m, gworld->get_gravity());
    m.clear_forces();
    if (g_plane) {
      integrate_euler_qdd(m, g_dt);
      gworld->step(g_dt);
      integrate_euler(m, g_dt);
    } else {
      integrate_euler(m, g_dt);
    }
    int j = 0;
    for (int i = 0; i < m.dof(); ++i) y[j++] = m.q(i);
    for (int i = 0; i < m.dof_qd(); ++i) y[j++] = m.qd(i);
    for (const auto &link : m) {
      for (size_t v = 0; v < link.X_visuals.size(); ++v) {
        auto vx = link.X_world * link.X_visuals[v];
        y[j++] = vx.translation[0];
        y[j++] = vx.translation[1];
        y[j++] = vx.translation[2];
        auto orn = Alg::matrix_to_quat(vx.rotation);
        y[j++] = orn.x();
        y[j++] = orn.y();
        y[j++] = orn.z();
        y[j++] = orn.w();
      }
    }
    y[j++] = m.get_world_transform(-1).rotation(2, 2);
  }
};

}  // namespace

// replaces every sphere of a body by a capsule of the same radius (length 0.16, along the link's local y) so that
// capsule-sphere pairs occur in both argument orders of the dispatcher
static void sphere_to_capsules(RefSim *s, MultiBody<Alg> *mb) {
  for (auto &link : *mb)
    for (size_t g = 0; g < link.collision_geometries.size(); ++g)
      if (link.collision_geometries[g]->get_type() == TINY_SPHERE_TYPE) {
        const double r = ((const Sphere<Alg> *)link.collision_geometries[g])->get_radius();
        link.collision_geometries[g] = s->gworld->create_capsule(r, 0.16);
        // capsule axis = local z of the geometry frame: turn it onto the link's y
        link.X_collisions[g].rotation = Alg::rotation_x_matrix(1.5707963267948966);
      }
}

extern "C" {

// name: "ant" | "laikago" | "<file>.urdf" | "<file>.urdf+plane" | "<file>.urdf[+plane]+floating"   (file relative to <ref>/data)
void *tdsref_create(const char *name_c, const char *reference_root) {
  std::string name(name_c);
  RefSim *s = new RefSim;
  s->name = name;
  if (name == "ant") {
    s->ant = new AntEnv2<Alg>(false);
  } else if (name == "laikago") {
    s->laikago = new LaikagoEnv<Alg>(false);
  } else if (name == "humanoid") {
    s->humanoid = new HumanoidEnv<Alg>(false);
  } else if (name == "humanoid_sph_pd" || name == "pendulum5_sph_pd") {
    // humanoid_sph_pd : humanoid_partial_xyz_spherical.urdf (xyz prismatic, spherical root = link 3, three revolute
    //                   joints), base_dof_ = 3: the PD loop visits the spherical joint (pose_index += 4; its torque is
    //                   NOT stored, link index < 4, :215) and then the revolute joints with pose_index 4, 5, 6
    // pendulum5_sph_pd: pendulum5spherical.urdf (five spherical joints), base_dof_ = 0: every joint is visited,
    //                   only link 4 keeps its torque
    const bool hum = name == "humanoid_sph_pd";
    char cwd[4096];
    if (!getcwd(cwd, sizeof(cwd))) cwd[0] = 0;
    if (chdir(reference_root) != 0) return nullptr;
    std::vector<double> poses(hum ? 7 : 20, 0.0);
    if (hum) { poses[4] = 0.1; poses[5] = -0.2; poses[6] = 0.3; }
    s->sphpd = new HumanoidContactSimulation<Alg>(true, hum ? "humanoid_partial_xyz_spherical.urdf" : "pendulum5spherical.urdf",
                                                 "", poses, false);
    s->sphpd->base_dof_ = hum ? 3 : 0;
    if (cwd[0] && chdir(cwd) != 0) return nullptr;
  } else if (name == "laikago_floating_env") {
    // urdf_from_file = true: FileUtils::find_file looks under ./data of the working directory
    char cwd[4096];
    if (!getcwd(cwd, sizeof(cwd))) cwd[0] = 0;
    if (chdir(reference_root) != 0) return nullptr;
    s->lfloat = new LaikagoContactSimulation<Alg>(true, "laikago/laikago_toes_zup.urdf", "",
                                                  LaikagoContactSimulation<Alg>::get_initial_poses(), true);
    if (cwd[0] && chdir(cwd) != 0) return nullptr;
  } else if (name.rfind("multi:", 0) == 0) {
    // "multi:<file>@x,y,z[/caps][/floating]:<file>@x,y,z...[+plane]": up to TDS_MAX_BODIES articulated bodies from the
    // reference's data directory in ONE world, each base at (x, y, z); /caps: spheres -> capsules (see above);
    // /floating: MultiBody with a floating base (its pose then comes from q, the position given here is not used)
    std::string spec = name.substr(6);
    s->g_plane = spec.find("+plane") != std::string::npos;
    spec = spec.substr(0, spec.find('+'));
    std::string root(reference_root);
    s->gworld = new World<Alg>();
    if (s->g_plane) s->cache.construct(root + "/data/plane_implicit.urdf", *s->gworld, false, false);
    size_t a = 0;
    while (a <= spec.size()) {
      size_t b = spec.find(':', a);
      std::string item = spec.substr(a, b == std::string::npos ? b : b - a);
      const bool caps = item.find("/caps") != std::string::npos, floating = item.find("/floating") != std::string::npos;
      item = item.substr(0, item.find('/'));
      const size_t at = item.find('@');
      if (at == std::string::npos) return nullptr;
      double pos[3] = {0, 0, 0};
      if (sscanf(item.c_str() + at + 1, "%lf,%lf,%lf", &pos[0], &pos[1], &pos[2]) != 3) return nullptr;
      MultiBody<Alg> *mb = s->cache.construct(root + "/data/" + item.substr(0, at), *s->gworld, false, floating);
      mb->base_X_world().set_identity();
      mb->base_X_world().translation = Alg::Vector3(pos[0], pos[1], pos[2]);
      if (caps) sphere_to_capsules(s, mb);
      s->gbodies.push_back(mb);
      if (b == std::string::npos) break;
      a = b + 1;
    }
    if (s->gbodies.size() < 2 || s->gbodies.size() > TDS_MAX_BODIES) return nullptr;
    s->gmb = s->gbodies[0];
    s->gworld->default_friction = 1;
    s->gworld->get_mb_constraint_solver()->keep_all_points_ = true;
  } else if (name.rfind("two:", 0) == 0) {
    // "two:<fileA>:<fileB>:<dx>:<dy>:<dz>[+capsA][+capsB][+plane]": two articulated bodies from the reference's data
    // directory in ONE world, B's base shifted by (dx, dy, dz); +capsX replaces every sphere of body X by a capsule of
    // the same radius (length 0.16, along the link's local y) so that capsule-sphere pairs occur in both argument orders
    std::string spec = name.substr(4);
    const bool capsA = spec.find("+capsA") != std::string::npos, capsB = spec.find("+capsB") != std::string::npos;
    s->g_plane = spec.find("+plane") != std::string::npos;
    spec = spec.substr(0, spec.find('+'));
    std::vector<std::string> tok;
    size_t a = 0;
    while (true) {
      size_t b = spec.find(':', a);
      tok.push_back(spec.substr(a, b == std::string::npos ? b : b - a));
      if (b == std::string::npos) break;
      a = b + 1;
    }
    if (tok.size() != 5) return nullptr;
    std::string root(reference_root);
    s->gworld = new World<Alg>();
    if (s->g_plane) s->cache.construct(root + "/data/plane_implicit.urdf", *s->gworld, false, false);
    s->gmb = s->cache.construct(root + "/data/" + tok[0], *s->gworld, false, false);
    MultiBody<Alg> *gmb2 = s->cache.construct(root + "/data/" + tok[1], *s->gworld, false, false);
    s->gbodies = {s->gmb, gmb2};
    s->gmb->base_X_world().set_identity();
    gmb2->base_X_world().set_identity();
    gmb2->base_X_world().translation = Alg::Vector3(atof(tok[2].c_str()), atof(tok[3].c_str()), atof(tok[4].c_str()));
    if (capsA) sphere_to_capsules(s, s->gmb);
    if (capsB) sphere_to_capsules(s, gmb2);
    s->gworld->default_friction = 1;
    s->gworld->get_mb_constraint_solver()->keep_all_points_ = true;
  } else {
    std::string file = name;
    bool floating = false;
    size_t p = file.find("+floating");  // "<file>.urdf[+plane]+floating": MultiBody with a floating base
    if (p != std::string::npos) {
      floating = true;
      file = file.substr(0, p);
    }
    p = file.find("+plane");
    if (p != std::string::npos) {
      s->g_plane = true;
      file = file.substr(0, p);
    }
    std::string root(reference_root);
    s->gworld = new World<Alg>();
    if (s->g_plane) {
      // plane FIRST, so it is multi_bodies_[0] == mb_a (locomotion_contact_simulation.h:100-123)
      s->cache.construct(root + "/data/plane_implicit.urdf", *s->gworld, false, false);
    }
    s->gmb = s->cache.construct(root + "/data/" + file, *s->gworld, false, floating);
    s->gmb->base_X_world().set_identity();
    s->gworld->default_friction = 1;
    s->gworld->get_mb_constraint_solver()->keep_all_points_ = true;
  }
  return s;
}

void tdsref_destroy(void *h) {
  RefSim *s = (RefSim *)h;
  delete s->ant;
  delete s->laikago;
  delete s->lfloat;
  delete s->humanoid;
  delete s->sphpd;
  delete s->gworld;
  delete s;
}

int tdsref_input_dim(void *h) { return ((RefSim *)h)->input_dim(); }
int tdsref_output_dim(void *h) { return ((RefSim *)h)->output_dim(); }

void tdsref_set_dt(void *h, double dt) {
  RefSim *s = (RefSim *)h;
  if (s->loco())
    s->loco()->dt = dt;
  else
    s->g_dt = dt;
}
void tdsref_set_gravity(void *h, double gx, double gy, double gz) {
  ((RefSim *)h)->world().set_gravity(Alg::Vector3(gx, gy, gz));
}
void tdsref_set_solver(void *h, double cfm, double erp, int pgs_iterations, double friction,
                       double restitution) {
  RefSim *s = (RefSim *)h;
  auto *solver = s->world().get_mb_constraint_solver();
  solver->cfm_ = cfm;
  solver->erp_ = erp;
  solver->pgs_iterations_ = pgs_iterations;
  s->world().default_friction = friction;
  s->world().default_restitution = restitution;
}

// joint spring / damper of one link (Link::stiffness, Link::damping: link.hpp:88-89; the URDF loader leaves them
// at zero, forward_dynamics.hpp:62-76,118-123 apply them)
int tdsref_set_link_spring(void *h, int link, double stiffness, double damping) {
  RefSim *s = (RefSim *)h;
  if (link < 0 || link >= (int)s->mb()->num_links()) return -1;
  (*s->mb())[link].stiffness = stiffness;
  (*s->mb())[link].damping = damping;
  return 0;
}

// Flatten the reference's MultiBody + World into the C-ABI blob, through the SAME header a TDS
// maintainer would use (include/tds_hip_stepper.hpp) — so that header is exercised here.
int tdsref_flatten(void *h, tds_model_t *out) {
  RefSim *s = (RefSim *)h;
  int rc;
  if (s->ant) {
    rc = tds_hip::flatten_locomotion_env<Alg>(s->ant->contact_sim, out, TDS_REWARD_ANT);
  } else if (s->laikago) {
    rc = tds_hip::flatten_locomotion_env<Alg>(s->laikago->contact_sim, out, TDS_REWARD_LAIKAGO);
  } else if (s->lfloat) {
    rc = tds_hip::flatten_locomotion_env<Alg>(*s->lfloat, out, TDS_REWARD_NONE);
  } else if (s->humanoid) {
    rc = tds_hip::flatten_locomotion_env<Alg>(s->humanoid->contact_sim, out, TDS_REWARD_HUMANOID);
  } else if (s->sphpd) {
    rc = tds_hip::flatten_locomotion_env<Alg>(*s->sphpd, out, TDS_REWARD_NONE);
    out->settle_steps = 0;  // (no reset rule in the reference for these constructions)
    out->output_dim = s->output_dim();
  } else {
    memset(out, 0, sizeof(*out));
    out->abi_version = TDS_HIP_ABI_VERSION;
    out->step_mode = TDS_STEP_TAU;
    rc = tds_hip::flatten_multibody<Alg>(*s->gmb, out);
    if (rc) return rc;
    tds_hip::flatten_world<Alg>(*s->gworld, out);
    out->dt = s->g_dt;
    out->action_dim = s->gmb->dof_actuated();
    out->input_dim = s->input_dim();
    out->output_dim = s->output_dim();
    out->pack_visuals = 1;
    out->reward_mode = TDS_REWARD_NONE;
    out->action_limit = 0.4;
    out->plane_normal[2] = 1.0;
    for (size_t b = 1; b < s->gbodies.size(); ++b) {
      rc = tds_hip::append_multibody<Alg>(*s->gbodies[b], out);
      if (rc) return rc;
      out->action_dim += s->gbodies[b]->dof_actuated();
    }
    if (s->g_plane) rc = tds_hip::flatten_plane<Alg>(*s->gworld, *s->gmb, s->g_dt, out);
  }
  if (rc) return rc;
  snprintf(out->name, sizeof(out->name), "%s", s->name.c_str());
  return 0;
}

// Compile + behaviour check of the reference-side plug-in: build the reference's own
// VectorizedEnvironment for Ant, install tds_hip::HipStepper as default_stepper_ and run
// `steps` env steps of `batch` envs with zero actions; the same rollout through the reference's
// SerialForwardStepper must agree.  Returns 0 on success; on failure returns non-zero and writes
// the message (e.g. "no HIP device visible" on a machine without a GPU).
extern "C++" {
template <typename Sim, typename Env>
static int hipstepper_selftest(int batch, int steps, int reward_mode, double *obs0, char *msg, int msg_len,
                               const std::vector<int> &devices = std::vector<int>(1, 0)) {
  typedef VectorizedEnvironment<Alg, Sim> VecEnv;
  Env env(false);
  VecEnv vec_env(env.contact_sim, batch);
  ARSConfig config;
  config.batch_size = batch;
  config.auto_reset_when_done = false;
  try {
    tds_hip::HipStepper<Alg, Sim> stepper(env.contact_sim, batch, devices, /*throw_on_error=*/true, reward_mode);
    vec_env.seed(42);
    auto observations = vec_env.reset(config);
    vec_env.default_stepper_ = &stepper;
    std::vector<std::vector<double>> actions(batch, std::vector<double>(env.contact_sim.action_dim(), 0.0));
    std::vector<double> rewards(batch);
    std::vector<bool> dones(batch, false);
    for (int t = 0; t < steps; ++t) vec_env.step(actions, observations, rewards, dones, config);
    for (size_t k = 0; k < observations[0].size() && k < 64; ++k) obs0[k] = observations[0][k];
    VecEnv vec_ref(env.contact_sim, batch);
    vec_ref.seed(42);
    auto obs_ref = vec_ref.reset(config);
    vec_ref.default_stepper_ = &vec_ref.serial_stepper_;
    std::vector<bool> dones_ref(batch, false);
    for (int t = 0; t < steps; ++t) vec_ref.step(actions, obs_ref, rewards, dones_ref, config);
    double err = 0;
    for (int e = 0; e < batch; ++e)
      for (size_t k = 0; k < obs_ref[e].size(); ++k) {
        double d = std::fabs(obs_ref[e][k] - observations[e][k]) / std::max(std::fabs(obs_ref[e][k]), 1e-3);
        if (d > err) err = d;
      }
    snprintf(msg, msg_len, "ok max_rel_err_vs_SerialForwardStepper=%.3e", err);
    return err < 1e-6 ? 0 : -100;
  } catch (const std::exception &e) {
    snprintf(msg, msg_len, "%s", e.what());
    return 1000;
  }
}
}

int tdsref_hipstepper_selftest(int batch, int steps, double *obs0, char *msg, int msg_len) {
  return hipstepper_selftest<AntContactSimulation2<Alg>, AntEnv2<Alg>>(batch, steps, TDS_REWARD_ANT, obs0, msg, msg_len);
}

// the same for the other environments of the reference that sit on this path: "ant" | "laikago" | "humanoid"
int tdsref_hipstepper_selftest_env(const char *env, int batch, int steps, double *obs0, char *msg, int msg_len) {
  std::string e(env);
  if (e == "ant") return tdsref_hipstepper_selftest(batch, steps, obs0, msg, msg_len);
  if (e == "laikago")
    return hipstepper_selftest<LaikagoContactSimulation<Alg>, LaikagoEnv<Alg>>(batch, steps, TDS_REWARD_LAIKAGO, obs0, msg, msg_len);
  if (e == "humanoid")
    return hipstepper_selftest<HumanoidContactSimulation<Alg>, HumanoidEnv<Alg>>(batch, steps, TDS_REWARD_HUMANOID, obs0, msg, msg_len);
  snprintf(msg, msg_len, "unknown env %s", env);
  return -1;
}

// HipStepper with a device LIST (one handle per entry; an entry may repeat, so a 1-GPU box exercises the split):
// Ant, the batch cut into n_devices contiguous blocks
int tdsref_hipstepper_selftest_devices(int batch, int steps, int n_devices, const int *devices, double *obs0, char *msg,
                                       int msg_len) {
  return hipstepper_selftest<AntContactSimulation2<Alg>, AntEnv2<Alg>>(batch, steps, TDS_REWARD_ANT, obs0, msg, msg_len,
                                                                       std::vector<int>(devices, devices + n_devices));
}

// The reference's OWN rollout loop on its header-only CPU path: Worker::rollouts
// (examples/ars/ars_vectorized_worker.h:51-140) = per step { VectorizedEnvironment::policy (one linear
// layer per environment), VectorizedEnvironment::step with the SerialForwardStepper, return bookkeeping },
// started from given states instead of reset() (whose std::rand stream cannot be shared).  Pins the
// on-device rollout (tds_hip_rollout) against the real reference.
//   x0 [batch][dof_q + dof_qd], params [batch][action_dim*obs_dim + action_dim],
//   total_rewards [batch], vec_steps [batch], final_obs [batch][obs_dim]
extern "C++" {
// stats (optional) [batch][obs_dim][3] = (NumDataValues, Mean, S = Variance * (n - 1)) of the reference's RunningStat
// pushed exactly where Worker::rollouts pushes it (ars_vectorized_worker.h:88-110); traj (optional)
// [batch][steps][output_dim] + traj_len [batch]: the trajectories vector of Worker::rollouts (:118-135)
// nn (optional): {num_layers, units[num_layers], activations[num_layers - 1], use_bias[num_layers]} — every environment's
// network is rebuilt with these layers through the reference's own NeuralNetworkSpecification calls (the hidden ReLU
// layers the reference keeps commented out, ars_vectorized_environment.h:175-176); params then holds num_parameters() per env
template <typename Sim, typename Env>
static int ref_rollout(int batch, int steps, double shift, const double *x0, const double *params,
                       double *total_rewards, int *vec_steps, double *final_obs, double *stats = nullptr,
                       double *traj = nullptr, int *traj_len = nullptr, const int *nn = nullptr) {
  typedef VectorizedEnvironment<Alg, Sim> VecEnv;
  Env env(false);
  VecEnv vec_env(env.contact_sim, batch);
  vec_env.default_stepper_ = &vec_env.serial_stepper_;
  ARSConfig config;
  config.batch_size = batch;
  config.auto_reset_when_done = false;
  const int od = env.contact_sim.input_dim();
  int np = env.contact_sim.action_dim() * od + env.contact_sim.action_dim();
  if (nn) {
    const int nl = nn[0];
    const int *units = nn + 1, *acts = nn + 1 + nl, *bias = nn + 1 + nl + (nl - 1);
    if (units[0] != od || units[nl - 1] != env.contact_sim.action_dim()) return -2;
    for (int e = 0; e < batch; ++e) {
      tds::NeuralNetwork<Alg> net;
      net.set_input_dim(units[0], bias[0] != 0);
      for (int i = 1; i < nl; ++i) net.add_linear_layer((tds::NeuralNetworkActivation)acts[i - 1], units[i], bias[i] != 0);
      vec_env.neural_networks_[e] = net;
    }
    np = vec_env.neural_networks_[0].num_parameters();
  }
  std::vector<std::vector<double>> observations(batch);
  for (int e = 0; e < batch; ++e) {
    vec_env.sim_states_[e].assign(env.contact_sim.input_dim_with_action_and_variables(), 0.0);
    for (int k = 0; k < od; ++k) vec_env.sim_states_[e][k] = x0[(size_t)e * od + k];
    observations[e].assign(x0 + (size_t)e * od, x0 + (size_t)(e + 1) * od);  // as returned by reset(): raw x, y
    vec_env.init_neural_network(e, std::vector<double>(params + (size_t)e * np, params + (size_t)(e + 1) * np));
    total_rewards[e] = 0.0;
    vec_steps[e] = 0;
  }
  std::vector<double> rewards(batch);
  std::vector<bool> dones(batch, false);
  std::vector<std::vector<double>> actions(batch);
  std::vector<std::vector<RunningStat>> filters(batch, std::vector<RunningStat>(od));
  std::vector<std::vector<std::vector<double>>> trajectories(batch);
  const int out = env.contact_sim.output_dim();
  for (int r = 0; r < steps; ++r) {
    for (int e = 0; e < batch; ++e) actions[e] = vec_env.policy(e, observations[e]);
    for (int e = 0; e < batch; ++e)
      for (int o = 0; o < od; ++o) filters[e][o].Push(observations[e][o]);
    vec_env.step(actions, observations, rewards, dones, config);
    for (int e = 0; e < batch; ++e) {
      if (dones[e]) {
        const int sz = (int)trajectories[e].size();
        if (sz) {
          const auto prev = trajectories[e][sz - 1];
          trajectories[e].push_back(prev);
        }
      } else {
        trajectories[e].push_back(vec_env.sim_states_with_graphics_[e]);
        total_rewards[e] += rewards[e] - shift;
        vec_steps[e]++;
      }
    }
  }
  if (stats)
    for (int e = 0; e < batch; ++e)
      for (int o = 0; o < od; ++o) {
        double *st = stats + ((size_t)e * od + o) * 3;
        const int nn = filters[e][o].NumDataValues();
        st[0] = nn;
        st[1] = filters[e][o].Mean();
        st[2] = filters[e][o].Variance() * (nn > 1 ? nn - 1 : 0);
      }
  if (traj && traj_len)
    for (int e = 0; e < batch; ++e) {
      traj_len[e] = (int)trajectories[e].size();
      for (int t = 0; t < traj_len[e]; ++t)
        for (int k = 0; k < out; ++k) traj[((size_t)e * steps + t) * out + k] = trajectories[e][t][k];
    }
  for (int e = 0; e < batch; ++e)
    for (int k = 0; k < od; ++k) final_obs[(size_t)e * od + k] = observations[e][k];
  return 0;
}
}  // extern "C++"

int tdsref_rollout_ex(const char *name, int batch, int steps, double shift, const double *x0, const double *params,
                      double *total_rewards, int *vec_steps, double *final_obs, double *stats, double *traj,
                      int *traj_len) {
  const std::string n(name);
  if (n == "ant")
    return ref_rollout<AntContactSimulation2<Alg>, AntEnv2<Alg>>(batch, steps, shift, x0, params, total_rewards,
                                                                 vec_steps, final_obs, stats, traj, traj_len);
  if (n == "laikago")
    return ref_rollout<LaikagoContactSimulation<Alg>, LaikagoEnv<Alg>>(batch, steps, shift, x0, params, total_rewards,
                                                                       vec_steps, final_obs, stats, traj, traj_len);
  return -1;
}

// the rollout with a policy NETWORK per environment: nn = {num_layers, units.., activations.., use_bias..} (see ref_rollout)
int tdsref_rollout_nn(const char *name, int batch, int steps, double shift, const double *x0, const double *params,
                      const int *nn, double *total_rewards, int *vec_steps, double *final_obs) {
  const std::string n(name);
  if (n == "ant")
    return ref_rollout<AntContactSimulation2<Alg>, AntEnv2<Alg>>(batch, steps, shift, x0, params, total_rewards,
                                                                 vec_steps, final_obs, nullptr, nullptr, nullptr, nn);
  if (n == "laikago")
    return ref_rollout<LaikagoContactSimulation<Alg>, LaikagoEnv<Alg>>(batch, steps, shift, x0, params, total_rewards,
                                                                       vec_steps, final_obs, nullptr, nullptr, nullptr, nn);
  return -1;
}

int tdsref_rollout(const char *name, int batch, int steps, double shift, const double *x0, const double *params,
                   double *total_rewards, int *vec_steps, double *final_obs) {
  const std::string n(name);
  if (n == "ant")
    return ref_rollout<AntContactSimulation2<Alg>, AntEnv2<Alg>>(batch, steps, shift, x0, params, total_rewards,
                                                                 vec_steps, final_obs);
  if (n == "laikago")
    return ref_rollout<LaikagoContactSimulation<Alg>, LaikagoEnv<Alg>>(batch, steps, shift, x0, params,
                                                                       total_rewards, vec_steps, final_obs);
  if (n == "humanoid")
    return ref_rollout<HumanoidContactSimulation<Alg>, HumanoidEnv<Alg>>(batch, steps, shift, x0, params,
                                                                         total_rewards, vec_steps, final_obs);
  return -1;
}

// The reference's vectorised environment driven the way its Python binding drives it
// (python/pytinydiffsim_includes.h:58-141: VectorizedAntEnv::step / VectorizedLaikagoEnv::step — fresh obs / rewards /
// dones vectors per call, then VectorizedEnvironment::step, ars_vectorized_environment.h:213-291), serial stepper,
// auto_reset_when_done = false, from given states instead of reset().  Records what the binding returns per step:
//   obs [steps][batch][obs_dim], rewards [steps][batch], dones [steps][batch] (1.0 / 0.0),
//   vis [steps][batch][output_dim] = sim_states_with_graphics_ (visual_world_transforms)
// (doubles, before the binding's float casts).  Pins tds_amd.VectorizedAntEnv / VectorizedLaikagoEnv.
extern "C++" {
template <typename Sim, typename Env>
static int ref_vecenv_steps(int batch, int steps, const double *x0, const double *actions_in, double *obs_out,
                            double *rewards_out, double *dones_out, double *vis_out) {
  typedef VectorizedEnvironment<Alg, Sim> VecEnv;
  Env env(false);
  VecEnv vec_env(env.contact_sim, batch);
  vec_env.default_stepper_ = &vec_env.serial_stepper_;
  ARSConfig config;
  config.batch_size = batch;
  config.auto_reset_when_done = false;
  const int od = env.contact_sim.input_dim(), adim = env.contact_sim.action_dim(), out = env.contact_sim.output_dim();
  for (int e = 0; e < batch; ++e) {
    vec_env.sim_states_[e].assign(env.contact_sim.input_dim_with_action_and_variables(), 0.0);
    for (int k = 0; k < od; ++k) vec_env.sim_states_[e][k] = x0[(size_t)e * od + k];
  }
  std::vector<std::vector<double>> actions(batch, std::vector<double>(adim));
  for (int t = 0; t < steps; ++t) {
    std::vector<std::vector<double>> obs(batch, std::vector<double>(od));
    std::vector<double> rewards(batch);
    std::vector<bool> dones(batch);
    for (int e = 0; e < batch; ++e)
      for (int k = 0; k < adim; ++k) actions[e][k] = actions_in[((size_t)t * batch + e) * adim + k];
    vec_env.step(actions, obs, rewards, dones, config);
    for (int e = 0; e < batch; ++e) {
      for (int k = 0; k < od; ++k) obs_out[((size_t)t * batch + e) * od + k] = obs[e][k];
      rewards_out[(size_t)t * batch + e] = rewards[e];
      dones_out[(size_t)t * batch + e] = dones[e] ? 1.0 : 0.0;
      for (int k = 0; k < out; ++k) vis_out[((size_t)t * batch + e) * out + k] = vec_env.sim_states_with_graphics_[e][k];
    }
  }
  return 0;
}
}  // extern "C++"

int tdsref_vecenv_steps(const char *name, int batch, int steps, const double *x0, const double *actions,
                        double *obs, double *rewards, double *dones, double *vis) {
  const std::string n(name);
  if (n == "ant")
    return ref_vecenv_steps<AntContactSimulation2<Alg>, AntEnv2<Alg>>(batch, steps, x0, actions, obs, rewards, dones, vis);
  if (n == "laikago")
    return ref_vecenv_steps<LaikagoContactSimulation<Alg>, LaikagoEnv<Alg>>(batch, steps, x0, actions, obs, rewards,
                                                                            dones, vis);
  return -1;
}

// The reference's OWN rollout loop — Worker<Env>::rollouts, examples/ars/ars_vectorized_worker.h:51-140, compiled from
// the unmodified header — instantiated twice: on the reference's VectorizedEnvironment (serial CPU stepper) and on
// tds_hip::VectorizedEnv (include/tds_hip_stepper.hpp: the environments resident on the GPU).  Same seed, same policy
// parameters per environment; tds_hip::VectorizedEnv resets through the reference's contact_sim.reset() (host_reset_:
// the same std::rand stream), so both loops walk the same trajectories.  Returns both sets of results:
//   total_rewards / vec_steps [2][batch] (0: reference, 1: HIP), traj_last [2][batch][output_dim] = the last entry of
//   every environment's trajectory, traj_len [2][batch].  auto_reset: ARSConfig::auto_reset_when_done.
extern "C++" {
template <typename Sim, typename Env>
static int vecenv_hip_worker(int batch, int steps, double shift, int seed, int auto_reset, int reward_mode, const double *params,
                             double *total_rewards, int *vec_steps, double *traj_last, int *traj_len, char *msg, int msg_len) {
  typedef VectorizedEnvironment<Alg, Sim> RefEnv;
  typedef tds_hip::VectorizedEnv<Alg, Sim> HipEnv;
  ARSConfig config;
  config.batch_size = batch;
  config.auto_reset_when_done = auto_reset != 0;
  config.env_seed = seed;
  const std::vector<double> deltas(4096, 0.0);
  const PolicyParams pp;
  try {
    Env env(false);
    const int out = env.contact_sim.output_dim();
    const int np = env.contact_sim.action_dim() * env.contact_sim.input_dim() + env.contact_sim.action_dim();
    for (int which = 0; which < 2; ++which) {
      std::vector<double> tr(batch, 0.0);
      std::vector<int> vs(batch, 0);
      std::vector<std::vector<std::vector<double>>> trajectories(batch);
      if (which == 0) {
        RefEnv vec_env(env.contact_sim, batch);
        vec_env.default_stepper_ = &vec_env.serial_stepper_;
        for (int e = 0; e < batch; ++e)
          vec_env.init_neural_network(e, std::vector<double>(params + (size_t)e * np, params + (size_t)(e + 1) * np));
        Worker<RefEnv> worker(vec_env, np, pp, deltas, config);  // (seeds the environment: std::srand(env_seed))
        worker.rollouts(shift, steps, tr, vs, trajectories);
      } else {
        HipEnv vec_env(env.contact_sim, batch, reward_mode);
        vec_env.host_reset_ = true;
        for (int e = 0; e < batch; ++e)
          vec_env.init_neural_network(e, std::vector<double>(params + (size_t)e * np, params + (size_t)(e + 1) * np));
        Worker<HipEnv> worker(vec_env, np, pp, deltas, config);
        worker.rollouts(shift, steps, tr, vs, trajectories);
      }
      for (int e = 0; e < batch; ++e) {
        total_rewards[(size_t)which * batch + e] = tr[e];
        vec_steps[(size_t)which * batch + e] = vs[e];
        traj_len[(size_t)which * batch + e] = (int)trajectories[e].size();
        for (int k = 0; k < out; ++k)
          traj_last[((size_t)which * batch + e) * out + k] = trajectories[e].empty() ? 0.0 : trajectories[e].back()[k];
      }
    }
    snprintf(msg, msg_len, "ok");
    return 0;
  } catch (const std::exception &e) {
    snprintf(msg, msg_len, "%s", e.what());
    return 1000;
  }
}

// The two classes in LOCK STEP, re-synchronised after every step: both are reset from the same std::rand stream, then per
// step { actions = the environment's policy on the REFERENCE's observation; reference.step; hip.step (same std::rand
// seed in front of each, so host resets of auto_reset_when_done draw the same poses); compare observations, rewards, dones
// and the y records of EVERY environment; hip state := reference state (tds_hip_set_states) }.  Without the resync two
// chaotic contact trajectories drift apart and only aggregate agreement can be asserted (vecenv_hip_worker above); with it
// each step of each environment is a one-step comparison from identical inputs, so a per-environment bound holds.
//   worst [batch] = the largest |hip - ref| / max(1, |ref|) of an environment over all steps and all compared values,
//   counts[0] = done flags that differ, counts[1] = host resets the reference made, counts[2] = values compared.
template <typename Sim, typename Env>
static int vecenv_hip_lockstep(int batch, int steps, int seed, int auto_reset, int reward_mode, const double *params, double *worst,
                               long long *counts, char *msg, int msg_len) {
  typedef VectorizedEnvironment<Alg, Sim> RefEnv;
  typedef tds_hip::VectorizedEnv<Alg, Sim> HipEnv;
  ARSConfig config;
  config.batch_size = batch;
  config.auto_reset_when_done = auto_reset != 0;
  try {
    Env env(false);
    const int od = env.contact_sim.input_dim(), out = env.contact_sim.output_dim();
    const int np = env.contact_sim.action_dim() * od + env.contact_sim.action_dim();
    RefEnv ref(env.contact_sim, batch);
    ref.default_stepper_ = &ref.serial_stepper_;
    HipEnv hip(env.contact_sim, batch, reward_mode);
    hip.host_reset_ = true;
    for (int e = 0; e < batch; ++e) {
      const std::vector<double> w(params + (size_t)e * np, params + (size_t)(e + 1) * np);
      ref.init_neural_network(e, w);
      hip.init_neural_network(e, w);
    }
    std::srand(seed);
    auto obs_r = ref.reset(config);
    std::srand(seed);
    auto obs_h = hip.reset(config);
    std::vector<double> rew_r(batch, 0.0), rew_h(batch, 0.0);
    std::vector<bool> done_r(batch, false), done_h(batch, false);
    std::vector<std::vector<double>> actions(batch);
    std::vector<double> qqd((size_t)batch * od);
    counts[0] = counts[1] = counts[2] = 0;
    for (int e = 0; e < batch; ++e) worst[e] = 0.0;
    auto cmp = [&](int e, double a, double b) {
      const double d = std::fabs(a - b) / std::max(1.0, std::fabs(b));
      if (!(d <= worst[e])) worst[e] = std::isfinite(d) ? d : 1e300;
      ++counts[2];
    };
    for (int e = 0; e < batch; ++e)
      for (int k = 0; k < od; ++k) cmp(e, obs_h[e][k], obs_r[e][k]);
    for (int t = 0; t < steps; ++t) {
      for (int e = 0; e < batch; ++e) actions[e] = ref.policy(e, obs_r[e]);
      std::vector<std::vector<double>> act_h = actions;
      std::srand(seed + 7919 * (t + 1));
      ref.step(actions, obs_r, rew_r, done_r, config);
      std::srand(seed + 7919 * (t + 1));
      hip.step(act_h, obs_h, rew_h, done_h, config);
      for (int e = 0; e < batch; ++e) {
        for (int k = 0; k < od; ++k) cmp(e, obs_h[e][k], obs_r[e][k]);
        cmp(e, rew_h[e], rew_r[e]);
        if (done_h[e] != done_r[e]) ++counts[0];
        if (done_r[e] && config.auto_reset_when_done) ++counts[1];
        // (the y record of the step; after a host reset sim_states_ holds the fresh pose, also compared through it)
        for (int k = 0; k < out; ++k) cmp(e, hip.sim_states_with_graphics_[e][k], ref.sim_states_with_graphics_[e][k]);
        for (int k = 0; k < od; ++k) cmp(e, hip.sim_states_[e][k], ref.sim_states_[e][k]);
        done_h[e] = done_r[e];
        for (int k = 0; k < od; ++k) qqd[(size_t)e * od + k] = ref.sim_states_[e][k];
      }
      if (tds_hip_set_states(hip.handle(), qqd.data()) != TDS_OK) throw std::runtime_error(tds_hip_last_error());
    }
    snprintf(msg, msg_len, "ok");
    return 0;
  } catch (const std::exception &e) {
    snprintf(msg, msg_len, "%s", e.what());
    return 1000;
  }
}

// Throughput of tds_hip::VectorizedEnv driven from C++ (no Python, no torch): environment steps per second of
//   rates[0]  step(actions, observations, rewards, dones, config) — the reference's signature: actions up, records down
//   rates[1]  the same with fetch_graphics_ = false (no y records down)
//   rates[2]  step_many_device: K steps per call, per-step records into device rings, actions from a device pool
//   rates[3]  rollouts_on_device (reset + K policy steps, the linear policies evaluated on the device)
template <typename Sim, typename Env>
static int vecenv_hip_bench(int batch, int steps, int reward_mode, double amp, double *rates, char *msg, int msg_len) {
  typedef tds_hip::VectorizedEnv<Alg, Sim> HipEnv;
  ARSConfig config;
  config.batch_size = batch;
  config.auto_reset_when_done = false;
  try {
    Env env(false);
    HipEnv ve(env.contact_sim, batch, reward_mode);
    const int adim = env.contact_sim.action_dim(), od = env.contact_sim.input_dim(), out = env.contact_sim.output_dim();
    ve.seed(7);
    auto observations = ve.reset(config);
    std::vector<std::vector<double>> actions(batch, std::vector<double>(adim, 0.0));
    std::srand(11);
    for (int e = 0; e < batch; ++e)
      for (int k = 0; k < adim; ++k) actions[e][k] = amp * ((std::rand() * 2.0 / RAND_MAX) - 1.0);
    std::vector<double> rewards(batch);
    std::vector<bool> dones(batch, false);
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int mode = 0; mode < 2; ++mode) {
      ve.fetch_graphics_ = mode == 0;
      const int k_host = steps < 200 ? steps : 200;
      for (int t = 0; t < 5; ++t) ve.step(actions, observations, rewards, dones, config);
      const double t0 = now();
      for (int t = 0; t < k_host; ++t) ve.step(actions, observations, rewards, dones, config);
      rates[mode] = (double)batch * k_host / (now() - t0);
    }
    {
      const int pool = 16, slots = 64;
      std::vector<double> ap((size_t)pool * batch * adim);
      for (double &v : ap) v = amp * ((std::rand() * 2.0 / RAND_MAX) - 1.0);
      void *d_act = nullptr, *d_obs = nullptr, *d_y = nullptr;
      const int per_line = 16, ystr = (out + per_line - 1) / per_line * per_line;
      tds_hip_sim_t *h = ve.handle();
      if (tds_hip_device_alloc(h, ap.size() * 8, &d_act) || tds_hip_device_alloc(h, (size_t)slots * batch * (od + 2) * 8, &d_obs) ||
          tds_hip_device_alloc(h, (size_t)slots * batch * ystr * 8, &d_y) || tds_hip_device_upload(h, d_act, ap.data(), ap.size() * 8))
        throw std::runtime_error(tds_hip_last_error());
      tds_hip_rings_t r;
      memset(&r, 0, sizeof(r));
      r.obs_ring = d_obs;
      r.obs_slots = slots;
      r.y_ring = d_y;
      r.y_slots = slots;
      r.y_stride = ystr;
      ve.step_many_device(d_act, pool, 0, steps, r, config);
      tds_hip_sync(h);
      const double t0 = now();
      ve.step_many_device(d_act, pool, 0, steps, r, config);
      tds_hip_sync(h);
      rates[2] = (double)batch * steps / (now() - t0);
      tds_hip_device_free(h, d_act);
      tds_hip_device_free(h, d_obs);
      tds_hip_device_free(h, d_y);
    }
    {
      std::vector<double> tr;
      std::vector<int> vs;
      const int np = adim * od + adim;
      for (int e = 0; e < batch; ++e) {
        std::vector<double> w(np);
        for (double &v : w) v = 0.05 * ((std::rand() * 2.0 / RAND_MAX) - 1.0);
        ve.init_neural_network(e, w);
      }
      ve.rollouts_on_device(0.0, steps, tr, vs, config);
      const double t0 = now();
      ve.rollouts_on_device(0.0, steps, tr, vs, config);
      rates[3] = (double)batch * steps / (now() - t0);
    }
    snprintf(msg, msg_len, "ok");
    return 0;
  } catch (const std::exception &e) {
    snprintf(msg, msg_len, "%s", e.what());
    return 1000;
  }
}
}  // extern "C++"

int tdsref_vecenv_hip_worker(const char *name, int batch, int steps, double shift, int seed, int auto_reset, const double *params,
                             double *total_rewards, int *vec_steps, double *traj_last, int *traj_len, char *msg, int msg_len) {
  const std::string n(name);
  if (n == "ant")
    return vecenv_hip_worker<AntContactSimulation2<Alg>, AntEnv2<Alg>>(batch, steps, shift, seed, auto_reset, TDS_REWARD_ANT, params,
                                                                       total_rewards, vec_steps, traj_last, traj_len, msg, msg_len);
  if (n == "laikago")
    return vecenv_hip_worker<LaikagoContactSimulation<Alg>, LaikagoEnv<Alg>>(batch, steps, shift, seed, auto_reset, TDS_REWARD_LAIKAGO,
                                                                             params, total_rewards, vec_steps, traj_last, traj_len,
                                                                             msg, msg_len);
  snprintf(msg, msg_len, "unknown env %s", name);
  return -1;
}
int tdsref_vecenv_hip_lockstep(const char *name, int batch, int steps, int seed, int auto_reset, const double *params, double *worst,
                               long long *counts, char *msg, int msg_len) {
  const std::string n(name);
  if (n == "ant")
    return vecenv_hip_lockstep<AntContactSimulation2<Alg>, AntEnv2<Alg>>(batch, steps, seed, auto_reset, TDS_REWARD_ANT, params, worst,
                                                                         counts, msg, msg_len);
  if (n == "laikago")
    return vecenv_hip_lockstep<LaikagoContactSimulation<Alg>, LaikagoEnv<Alg>>(batch, steps, seed, auto_reset, TDS_REWARD_LAIKAGO,
                                                                               params, worst, counts, msg, msg_len);
  snprintf(msg, msg_len, "unknown env %s", name);
  return -1;
}
int tdsref_vecenv_hip_bench(const char *name, int batch, int steps, double *rates, char *msg, int msg_len) {
  const std::string n(name);
  if (n == "ant")
    return vecenv_hip_bench<AntContactSimulation2<Alg>, AntEnv2<Alg>>(batch, steps, TDS_REWARD_ANT, 0.4, rates, msg, msg_len);
  if (n == "laikago")
    return vecenv_hip_bench<LaikagoContactSimulation<Alg>, LaikagoEnv<Alg>>(batch, steps, TDS_REWARD_LAIKAGO, 0.1, rates, msg, msg_len);
  snprintf(msg, msg_len, "unknown env %s", name);
  return -1;
}

// Free rigid bodies: the reference's World::step on tds::RigidBody objects
// (world.hpp:293-366, rigid_body.hpp, rb_constraint_solver.hpp) for n worlds described by the same
// tds_rb_model_t; state[n][num_bodies][13] = position | quaternion xyzw | linear | angular velocity.
int tdsref_rb_step(const tds_rb_model_t *m, int n, int steps, double *state) {
  typedef RigidBody<Alg> RB;
  for (int e = 0; e < n; ++e) {
    World<Alg> world;
    world.set_gravity(Alg::Vector3(m->gravity[0], m->gravity[1], m->gravity[2]));
    world.num_solver_iterations = m->solver_iterations;
    world.default_friction = m->friction;
    world.default_restitution = m->restitution;
    world.get_rb_constraint_solver()->erp_ = m->erp;
    std::vector<RB *> bodies;
    double *S = state + (size_t)e * m->num_bodies * 13;
    for (int i = 0; i < m->num_bodies; ++i) {
      const tds_rb_body_t &b = m->bodies[i];
      const Geometry<Alg> *g;
      if (b.geom_type == TDS_GEOM_SPHERE) {
        g = world.create_sphere(b.radius);
      } else if (b.geom_type == TDS_GEOM_CAPSULE) {
        g = world.create_capsule(b.radius, b.length);
      } else if (b.geom_type == TDS_GEOM_BOX) {
        Box<Alg> *bx = world.create_box(Alg::Vector3(b.extents[0], b.extents[1], b.extents[2]));
        bx->set_radius(b.radius);
        g = bx;
      } else {
        Plane<Alg> *pl = world.create_plane();
        *pl = Plane<Alg>(Alg::Vector3(b.plane_normal[0], b.plane_normal[1], b.plane_normal[2]), b.plane_constant);
        g = pl;
      }
      RB *rb = world.create_rigid_body(b.mass, g);
      const double *B = S + i * 13;
      rb->world_pose_.position_ = Alg::Vector3(B[0], B[1], B[2]);
      rb->world_pose_.orientation_ = Alg::quat_from_xyzw(B[3], B[4], B[5], B[6]);
      rb->linear_velocity_ = Alg::Vector3(B[7], B[8], B[9]);
      rb->angular_velocity_ = Alg::Vector3(B[10], B[11], B[12]);
      bodies.push_back(rb);
    }
    for (int s = 0; s < steps; ++s) world.step(m->dt);
    for (int i = 0; i < m->num_bodies; ++i) {
      double *B = S + i * 13;
      const RB *rb = bodies[i];
      for (int k = 0; k < 3; ++k) {
        B[k] = rb->world_pose_.position_[k];
        B[7 + k] = rb->linear_velocity_[k];
        B[10 + k] = rb->angular_velocity_[k];
      }
      B[3] = rb->world_pose_.orientation_.x();
      B[4] = rb->world_pose_.orientation_.y();
      B[5] = rb->world_pose_.orientation_.z();
      B[6] = rb->world_pose_.orientation_.w();
    }
  }
  return 0;
}

// The reference's COMMITTED generated kernels (examples/environments/omp_model_{ant,laikago}_forward_zero.h,
// what its OpenMPForwardStepper runs: ars_vectorized_environment.h:127-135) — BASELINE.md's "B2, honest best
// CPU number".  Stateless, so one call per environment from any thread.  Note: the committed
// Ant artefact is stale w.r.t. the header-only path (1.9e-7 after one step), hence speed baseline only.
int tdsref_generated_step(const char *name, int n, const double *x, double *y) {
  const std::string nm(name);
  if (nm == "ant") {
    for (int e = 0; e < n; ++e) omp_model_ant_forward_zero_kernel<double>(1, y + (size_t)e * 155, x + (size_t)e * 39);
    return 0;
  }
  if (nm == "laikago") {
    for (int e = 0; e < n; ++e) omp_model_laikago_forward_zero_kernel<double>(1, y + (size_t)e * 411, x + (size_t)e * 51);
    return 0;
  }
  return -1;
}

// y[n][output_dim] = reference_step(x[n][input_dim]); y is zero-filled first (the reference's
// callers hand in zero-initialised vectors, ars_vectorized_environment.h:218-219).
void tdsref_step(void *h, int n, const double *x, double *y) {
  RefSim *s = (RefSim *)h;
  const int in = s->input_dim(), out = s->output_dim();
  LocoSim *loco = s->loco();
  std::vector<double> xi(in), yi(out);
  for (int e = 0; e < n; ++e) {
    xi.assign(x + (size_t)e * in, x + (size_t)(e + 1) * in);
    std::fill(yi.begin(), yi.end(), 0.0);
    if (loco)
      loco->step_forward_original(xi, yi);
    else if (!s->gbodies.empty())
      s->multi_body_step(xi.data(), yi.data());
    else
      s->generic_step(xi.data(), yi.data());
    memcpy(y + (size_t)e * out, yi.data(), sizeof(double) * out);
  }
}

// After a step of a generic world: per body pair of the reference's contact list (world.mb_contacts_, in its order:
// plane-A, plane-B, A-B for a two-body world with a plane) the number of contacts with distance < 0.  Returns the
// number of pairs written (<= max_pairs).
int tdsref_last_penetrating_contacts(void *h, int *counts, int max_pairs) {
  RefSim *s = (RefSim *)h;
  const auto &all = s->world().mb_contacts_;
  int n = 0;
  for (const auto &pair : all) {
    if (n >= max_pairs) break;
    int c = 0;
    for (const auto &cp : pair)
      if (cp.distance < 0) ++c;
    counts[n++] = c;
  }
  return n;
}

// Intermediates for localising a parity failure.  All computed by calling the reference's own
// public functions on the state x (q, qd taken from x; tau = 0 unless generic model).
//   qdd[dof_qd]           forward_dynamics (ABA)                 forward_dynamics.hpp:11
//   M[dof_qd*dof_qd]      mass_matrix (CRBA), row-major          mass_matrix.hpp:13
//   contacts[n_c*10]      (normal_on_b[3], point_on_b[3], point_on_a[3], distance) per contact,
//                         from World::compute_contacts_multi_body  world.hpp:206-282
//   jac[n_c*3*dof_qd]     point_jacobian2(robot, link_b, point_on_b) jacobian.hpp:85-90
//   links[n_links]        link_b per contact
// returns n_c
int tdsref_debug(void *h, const double *x, double *qdd, double *M, double *contacts, double *jac,
                 int *links, double *X_world /* n_links*12: rot(9) trans(3) */) {
  RefSim *s = (RefSim *)h;
  MultiBody<Alg> &m = *s->mb();
  m.initialize();
  for (int i = 0; i < m.dof(); ++i) m.q(i) = x[i];
  for (int i = 0; i < m.dof_qd(); ++i) m.qd(i) = x[m.dof() + i];
  if (!s->loco())
    for (int i = 0; i < m.dof_actuated(); ++i) m.tau(i) = x[m.dof() + m.dof_qd() + i];
  forward_dynamics(m, s->world().get_gravity());
  for (int i = 0; i < m.dof_qd(); ++i) qdd[i] = m.qdd(i);
  for (int i = 0; i < (int)m.num_links(); ++i) {
    copy_mat3(m[i].X_world.rotation, X_world + 12 * i);
    copy_vec3(m[i].X_world.translation, X_world + 12 * i + 9);
  }
  int n_c = 0;
  Alg::MatrixX Mm(m.dof_qd(), m.dof_qd());
  mass_matrix(m, &Mm);
  for (int r = 0; r < m.dof_qd(); ++r)
    for (int c = 0; c < m.dof_qd(); ++c) M[r * m.dof_qd() + c] = Mm(r, c);
  if (s->has_plane()) {
    // run one World::step on a copy of the velocities so that mb_contacts_ (public) is filled
    Alg::VectorX qd_save = m.qd();
    s->world().step(s->dt());
    m.qd() = qd_save;
    if (!s->world().mb_contacts_.empty()) {
      const auto &cps = s->world().mb_contacts_[0];
      n_c = (int)cps.size();
      for (int i = 0; i < n_c; ++i) {
        const auto &cp = cps[i];
        double *c = contacts + 10 * i;
        copy_vec3(cp.world_normal_on_b, c);
        copy_vec3(cp.world_point_on_b, c + 3);
        copy_vec3(cp.world_point_on_a, c + 6);
        c[9] = cp.distance;
        links[i] = cp.link_b;
        auto J = point_jacobian2(m, cp.link_b, cp.world_point_on_b, false);
        for (int r = 0; r < 3; ++r)
          for (int d = 0; d < m.dof_qd(); ++d) jac[(i * 3 + r) * m.dof_qd() + d] = J(r, d);
      }
    }
  }
  return n_c;
}

}  // extern "C"
