// ref_harness_f32.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// The reference's OWN float instantiation, TinyAlgebra<float, FloatUtils> (src/math/tiny/tiny_float_utils.h), of the
// same step the double harness (ref_harness.cpp) runs — compiled from the unmodified reference headers where they lie.
// BASELINE config 2 says "float": this is what "the reference in float" computes, so that the error of a float kernel
// is measured against the reference's float path and not only against its double path (tests/test_f32.py).
// Nothing here re-implements the algorithm.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "math/tiny/tiny_algebra.hpp"
#include "math/tiny/tiny_float_utils.h"

using namespace TINY;
using namespace tds;
typedef TinyAlgebra<float, ::TINY::FloatUtils> AlgF;

#include "ant_environment2.h"
#include "laikago_environment2.h"

namespace {

struct RefSimF {
  AntEnv2<AlgF> *ant = nullptr;
  LaikagoEnv<AlgF> *laikago = nullptr;
  UrdfCache<AlgF> cache;
  World<AlgF> *gworld = nullptr;
  MultiBody<AlgF> *gmb = nullptr;
  bool g_plane = false;
  float g_dt = 1e-3f;

  int num_visuals() {
    int n = 0;
    for (const auto &l : *gmb) n += (int)l.X_visuals.size();
    return n;
  }
  // same call sequence as RefSim::generic_step of ref_harness.cpp (cartpole_environment.h:88-94 /
  // locomotion_contact_simulation.h:261-269), in float
  void generic_step(const double *x, double *y) {
    MultiBody<AlgF> &m = *gmb;
    m.initialize();
    for (int i = 0; i < m.dof(); ++i) m.q(i) = (float)x[i];
    for (int i = 0; i < m.dof_qd(); ++i) m.qd(i) = (float)x[m.dof() + i];
    for (int i = 0; i < m.dof_actuated(); ++i) m.tau(i) = (float)x[m.dof() + m.dof_qd() + i];
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
        auto orn = AlgF::matrix_to_quat(vx.rotation);
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

extern "C" {

// name: "ant" | "laikago" | "<file>.urdf" | "<file>.urdf+plane" (file relative to <ref>/data)
void *tdsref_f32_create(const char *name_c, const char *reference_root, double dt) {
  std::string name(name_c);
  RefSimF *s = new RefSimF;
  if (name == "ant") {
    s->ant = new AntEnv2<AlgF>(false);
  } else if (name == "laikago") {
    s->laikago = new LaikagoEnv<AlgF>(false);
  } else {
    std::string file = name;
    size_t p = file.find("+plane");
    if (p != std::string::npos) {
      s->g_plane = true;
      file = file.substr(0, p);
    }
    std::string root(reference_root);
    s->gworld = new World<AlgF>();
    if (s->g_plane) s->cache.construct(root + "/data/plane_implicit.urdf", *s->gworld, false, false);
    s->gmb = s->cache.construct(root + "/data/" + file, *s->gworld, false, false);
    s->gmb->base_X_world().set_identity();
    s->gworld->default_friction = 1;
    s->gworld->get_mb_constraint_solver()->keep_all_points_ = true;
    s->g_dt = (float)dt;
  }
  return s;
}

void tdsref_f32_destroy(void *h) {
  RefSimF *s = (RefSimF *)h;
  delete s->ant;
  delete s->laikago;
  delete s->gworld;
  delete s;
}

// y = f(x) for n records, host doubles in / out (rounded to float on entry, widened on exit)
void tdsref_f32_step(void *h, int n, int in_dim, int out_dim, const double *x, double *y) {
  RefSimF *s = (RefSimF *)h;
  for (int e = 0; e < n; ++e) {
    const double *xe = x + (size_t)e * in_dim;
    double *ye = y + (size_t)e * out_dim;
    for (int k = 0; k < out_dim; ++k) ye[k] = 0.0;
    if (s->ant || s->laikago) {
      std::vector<float> in(in_dim), out(out_dim, 0.0f);  // (the reference's callers pre-size the output record)
      for (int k = 0; k < in_dim; ++k) in[k] = (float)xe[k];
      if (s->ant)
        s->ant->contact_sim.step_forward_original(in, out);
      else
        s->laikago->contact_sim.step_forward_original(in, out);
      for (int k = 0; k < out_dim && k < (int)out.size(); ++k) ye[k] = out[k];
    } else {
      s->generic_step(xe, ye);
    }
  }
}

}  // extern "C"
