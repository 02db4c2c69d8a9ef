// tds_rb.hip — free rigid bodies: World::step of worlds that hold tds::RigidBody
// objects, N independent worlds per launch.
//
// Reference: src/world.hpp:293-366 (step: gravity impulse, pairwise narrowphase, num_solver_iterations
// sweeps of RigidBodyConstraintSolver::resolve_collision over the contacts, integrate),
// src/rigid_body.hpp:26-123, src/rb_constraint_solver.hpp:112-165 (the non-CppAD branch),
// src/contact_point.hpp:43-125, 468-496 (sphere-sphere, plane-sphere, swapped order).
//
// Mapping: none of the five benchmark configurations creates a RigidBody, so this path is built for
// coverage, not speed: one LANE per world (the reference's own CUDA mapping), positions and velocities
// of a world's bodies in LDS as [component][lane] (bank-conflict free, dynamic body index without
// scratch), quaternions stay in HBM (touched once per step).  The contact list is not stored: body
// positions do not change during the solver sweeps, so each sweep re-derives the contact of a pair from
// the positions — the same numbers the reference keeps in rb_contacts_.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "tds_hip.h"

namespace {

thread_local char g_rb_err[256] = "";

int rb_fail(int code, const char *msg) {
  snprintf(g_rb_err, sizeof(g_rb_err), "%s", msg);
  return code;
}

template <typename T>
struct RbDev {
  int nb, iters;
  T dt, grav[3], restitution, friction, erp;
  T mass[TDS_RB_MAX_BODIES], inv_mass[TDS_RB_MAX_BODIES], inv_in[TDS_RB_MAX_BODIES], radius[TDS_RB_MAX_BODIES];
  T pn[TDS_RB_MAX_BODIES][3], pc[TDS_RB_MAX_BODIES];
  int type[TDS_RB_MAX_BODIES];
};

template <typename T>
__device__ __forceinline__ void cross3(const T *a, const T *b, T *o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
template <typename T>
__device__ __forceinline__ T dot3(const T *a, const T *b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

// LDS slot of (body b, component c) for this lane; c: 0..2 position, 3..5 linear, 6..8 angular velocity
#define RB_AT(b, c) sm[((b)*9 + (c)) * 64 + lane]

template <typename T>
__global__ __launch_bounds__(64) void tds_rb_kernel(RbDev<T> M, T *__restrict__ state, int n_worlds, int steps) {
  extern __shared__ __align__(16) unsigned char rb_smem_raw[];
  T *const sm = reinterpret_cast<T *>(rb_smem_raw);
  const int lane = threadIdx.x;
  const int world = blockIdx.x * 64 + lane;
  const bool valid = world < n_worlds;
  const int nb = M.nb;
  T *const S = state + (size_t)(valid ? world : 0) * nb * TDS_RB_STATE;
  for (int b = 0; b < nb; ++b) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      RB_AT(b, k) = valid ? S[b * TDS_RB_STATE + k] : T(0);
      RB_AT(b, 3 + k) = valid ? S[b * TDS_RB_STATE + 7 + k] : T(0);
      RB_AT(b, 6 + k) = valid ? S[b * TDS_RB_STATE + 10 + k] : T(0);
    }
  }
  const T dt = M.dt;
  for (int st = 0; st < steps; ++st) {
    // apply_gravity + apply_force_impulse + clear_forces (world.hpp:301-310, rigid_body.hpp:81-97)
    for (int b = 0; b < nb; ++b) {
#pragma unroll
      for (int k = 0; k < 3; ++k) RB_AT(b, 3 + k) += (M.mass[b] * M.grav[k]) * M.inv_mass[b] * dt;
    }
    // num_solver_iterations sweeps over the contacts in pair order i < j (world.hpp:163-191, 336-340)
    for (int it = 0; it < M.iters; ++it) {
      for (int i = 0; i < nb; ++i) {
        for (int j = i + 1; j < nb; ++j) {
          const int ti = M.type[i], tj = M.type[j];  // wave-uniform
          T nbv[3], pa[3], pb[3], dist;
          bool got = false;
          const T pi[3] = {RB_AT(i, 0), RB_AT(i, 1), RB_AT(i, 2)};
          const T pj[3] = {RB_AT(j, 0), RB_AT(j, 1), RB_AT(j, 2)};
          if (ti == TDS_GEOM_SPHERE && tj == TDS_GEOM_SPHERE) {  // contact_point.hpp:43-94
            const T diff[3] = {pi[0] - pj[0], pi[1] - pj[1], pi[2] - pj[2]};
            const T length = sqrt(dot3(diff, diff));
            dist = length - (M.radius[i] + M.radius[j]);
            got = length > T(1) / T(100000);
            const T il = T(1) / length;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              nbv[k] = il * diff[k];
              pa[k] = pi[k] - M.radius[i] * nbv[k];
              pb[k] = pa[k] - dist * nbv[k];
            }
          } else if (ti == TDS_GEOM_PLANE && tj == TDS_GEOM_SPHERE) {  // contact_point.hpp:96-125
            const T mn[3] = {-M.pn[i][0], -M.pn[i][1], -M.pn[i][2]};
            const T t = -(dot3(pj, mn) + M.pc[i]);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              pa[k] = pj[k] + t * mn[k];
              pb[k] = pj[k] - M.radius[j] * M.pn[i][k];
              nbv[k] = mn[k];
            }
            dist = t - M.radius[j];
            got = true;
          } else if (ti == TDS_GEOM_SPHERE && tj == TDS_GEOM_PLANE) {  // dispatcher swap, :478-493
            const T mn[3] = {-M.pn[j][0], -M.pn[j][1], -M.pn[j][2]};
            const T t = -(dot3(pi, mn) + M.pc[j]);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              pb[k] = pi[k] + t * mn[k];
              pa[k] = pi[k] - M.radius[i] * M.pn[j][k];
              nbv[k] = -mn[k];
            }
            dist = t - M.radius[i];
            got = true;
          } else {
            continue;
          }
          // RigidBodyConstraintSolver::resolve_collision (rb_constraint_solver.hpp:112-165)
          if (!(got && dist < T(0))) continue;
          T ra[3], rb[3], wa[3], wb[3], va[3], vb[3], rel[3], t3[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            ra[k] = pa[k] - pi[k];
            rb[k] = pb[k] - pj[k];
            wa[k] = RB_AT(i, 6 + k);
            wb[k] = RB_AT(j, 6 + k);
          }
          const T baumgarte = M.erp * dist / dt;
          cross3(wa, ra, t3);
#pragma unroll
          for (int k = 0; k < 3; ++k) va[k] = RB_AT(i, 3 + k) + t3[k];
          cross3(wb, rb, t3);
#pragma unroll
          for (int k = 0; k < 3; ++k) vb[k] = RB_AT(j, 3 + k) + t3[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) rel[k] = va[k] - vb[k];
          const T nrv = dot3(nbv, rel);
          if (!(nrv < T(0))) continue;
          T t1[3], t2[3], x1[3], x2[3], sum[3];
          cross3(ra, nbv, t1);
          cross3(rb, nbv, t2);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            t1[k] *= M.inv_in[i];
            t2[k] *= M.inv_in[j];
          }
          cross3(t1, ra, x1);
          cross3(t2, rb, x2);
#pragma unroll
          for (int k = 0; k < 3; ++k) sum[k] = x1[k] + x2[k];
          const T ang = dot3(nbv, sum);
          const T denom = M.inv_mass[i] + M.inv_mass[j] + ang;
          const T impulse = (-(T(1) + M.restitution) * nrv - baumgarte) / denom;
          if (!(impulse > T(0))) continue;
          T iv[3], miv[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            iv[k] = impulse * nbv[k];
            miv[k] = -iv[k];
          }
          // apply_impulse (rigid_body.hpp:103-108)
          cross3(ra, iv, t3);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            RB_AT(i, 3 + k) += M.inv_mass[i] * iv[k];
            RB_AT(i, 6 + k) += M.inv_in[i] * t3[k];
          }
          cross3(rb, miv, t3);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            RB_AT(j, 3 + k) += M.inv_mass[j] * miv[k];
            RB_AT(j, 6 + k) += M.inv_in[j] * t3[k];
          }
          // Coulomb friction from the PRE-impulse relative velocity
          T lat[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) lat[k] = rel[k] - nrv * nbv[k];
          const T latn = sqrt(dot3(lat, lat));
          const T trial = latn / denom;
          const T fimp = trial < M.friction * impulse ? trial : M.friction * impulse;
          if (latn > T(1) / T(10000)) {
            T fa[3], fb[3];
            const T il = T(1) / latn;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const T fd = lat[k] * il;
              fa[k] = -fimp * fd;
              fb[k] = fimp * fd;
            }
            cross3(ra, fa, t3);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              RB_AT(i, 3 + k) += M.inv_mass[i] * fa[k];
              RB_AT(i, 6 + k) += M.inv_in[i] * t3[k];
            }
            cross3(rb, fb, t3);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              RB_AT(j, 3 + k) += M.inv_mass[j] * fb[k];
              RB_AT(j, 6 + k) += M.inv_in[j] * t3[k];
            }
          }
        }
      }
    }
    // integrate (rigid_body.hpp:116-122, tiny_algebra.hpp:604-614): the quaternion lives in HBM
    for (int b = 0; b < nb; ++b) {
#pragma unroll
      for (int k = 0; k < 3; ++k) RB_AT(b, k) += RB_AT(b, 3 + k) * dt;
      if (valid) {
        T *const Q = S + b * TDS_RB_STATE + 3;
        const T qx = Q[0], qy = Q[1], qz = Q[2], qw = Q[3];
        const T w0 = RB_AT(b, 6), w1 = RB_AT(b, 7), w2 = RB_AT(b, 8);
        const T hd = T(0.5) * dt;
        const T ww = (-qx * w0 - qy * w1 - qz * w2) * hd;
        const T xx = (qw * w0 + qz * w1 - qy * w2) * hd;
        const T yy = (qw * w1 + qx * w2 - qz * w0) * hd;
        const T zz = (qw * w2 + qy * w0 - qx * w1) * hd;
        const T nx = qx + xx, ny = qy + yy, nz = qz + zz, nw = qw + ww;
        const T ql = sqrt(nx * nx + ny * ny + nz * nz + nw * nw);
        Q[0] = nx / ql;
        Q[1] = ny / ql;
        Q[2] = nz / ql;
        Q[3] = nw / ql;
      }
    }
  }
  if (valid) {
    for (int b = 0; b < nb; ++b) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        S[b * TDS_RB_STATE + k] = RB_AT(b, k);
        S[b * TDS_RB_STATE + 7 + k] = RB_AT(b, 3 + k);
        S[b * TDS_RB_STATE + 10 + k] = RB_AT(b, 6 + k);
      }
    }
  }
}

template <typename T>
void rb_build(const tds_rb_model_t *m, RbDev<T> *d) {
  memset(d, 0, sizeof(*d));
  d->nb = m->num_bodies;
  d->iters = m->solver_iterations;
  d->dt = (T)m->dt;
  for (int k = 0; k < 3; ++k) d->grav[k] = (T)m->gravity[k];
  d->restitution = (T)m->restitution;
  d->friction = (T)m->friction;
  d->erp = (T)m->erp;
  for (int i = 0; i < m->num_bodies; ++i) {
    const tds_rb_body_t &b = m->bodies[i];
    d->mass[i] = (T)b.mass;
    d->inv_mass[i] = b.mass == 0.0 ? T(0) : (T)(1.0 / b.mass);  // rigid_body.hpp:49-53
    d->inv_in[i] = b.mass == 0.0 ? T(0) : T(1);                  // zero33 / eye3
    d->radius[i] = (T)b.radius;
    d->type[i] = b.geom_type;
    // Plane's constructor normalises the normal (geometry.hpp:163-168)
    double nl = sqrt(b.plane_normal[0] * b.plane_normal[0] + b.plane_normal[1] * b.plane_normal[1] +
                     b.plane_normal[2] * b.plane_normal[2]);
    if (nl == 0.0) nl = 1.0;
    for (int k = 0; k < 3; ++k) d->pn[i][k] = (T)(b.plane_normal[k] / nl);
    d->pc[i] = (T)b.plane_constant;
  }
}

}  // namespace

struct tds_rb_sim {
  tds_rb_model_t model;
  int num_worlds = 0, device = 0, dtype = TDS_DTYPE_F64;
  size_t elem = 8;
  hipStream_t stream = nullptr;
  void *d_state = nullptr;
  RbDev<double> h64;
  RbDev<float> h32;
  std::vector<float> stage;
};

#define RB_TRY(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      snprintf(g_rb_err, sizeof(g_rb_err), "%s failed: %s", #expr, hipGetErrorString(e_)); \
      return TDS_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)

extern "C" {

const char *tds_rb_last_error(void) { return g_rb_err; }

int tds_rb_create(const tds_rb_model_t *model, int num_worlds, int device, int dtype, tds_rb_sim_t **out) {
  if (!model || !out) return rb_fail(TDS_ERR_INVALID_ARG, "NULL argument");
  *out = nullptr;
  if (model->abi_version != TDS_HIP_ABI_VERSION) return rb_fail(TDS_ERR_INVALID_ARG, "model abi_version mismatch");
  if (model->num_bodies < 1 || model->num_bodies > TDS_RB_MAX_BODIES) return rb_fail(TDS_ERR_INVALID_ARG, "num_bodies out of range");
  if (model->solver_iterations < 0 || !(model->dt > 0)) return rb_fail(TDS_ERR_INVALID_ARG, "bad solver_iterations / dt");
  for (int i = 0; i < model->num_bodies; ++i)
    if (model->bodies[i].geom_type != TDS_GEOM_SPHERE && model->bodies[i].geom_type != TDS_GEOM_PLANE)
      return rb_fail(TDS_ERR_UNSUPPORTED, "rigid bodies support sphere and plane geometries");
  if (num_worlds < 1) return rb_fail(TDS_ERR_INVALID_ARG, "num_worlds < 1");
  if (dtype != TDS_DTYPE_F64 && dtype != TDS_DTYPE_F32) return rb_fail(TDS_ERR_INVALID_ARG, "unknown dtype");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return rb_fail(TDS_ERR_NO_DEVICE, "no HIP device visible (there is no CPU fallback)");
  if (device < 0 || device >= ndev) return rb_fail(TDS_ERR_INVALID_ARG, "device index out of range");
  RB_TRY(hipSetDevice(device));
  tds_rb_sim *s = new tds_rb_sim;
  s->model = *model;
  s->num_worlds = num_worlds;
  s->device = device;
  s->dtype = dtype;
  s->elem = dtype == TDS_DTYPE_F64 ? 8 : 4;
  rb_build<double>(model, &s->h64);
  rb_build<float>(model, &s->h32);
  const size_t bytes = (size_t)num_worlds * model->num_bodies * TDS_RB_STATE * s->elem;
  hipError_t e = hipMalloc(&s->d_state, bytes);
  if (e != hipSuccess) {
    delete s;
    return rb_fail(TDS_ERR_HIP, "hipMalloc of the state failed");
  }
  (void)hipMemset(s->d_state, 0, bytes);
  const int lds = model->num_bodies * 9 * 64 * (int)s->elem;
  if (dtype == TDS_DTYPE_F64)
    (void)hipFuncSetAttribute((const void *)tds_rb_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  else
    (void)hipFuncSetAttribute((const void *)tds_rb_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  *out = s;
  return TDS_OK;
}

int tds_rb_destroy(tds_rb_sim_t *s) {
  if (!s) return TDS_OK;
  (void)hipSetDevice(s->device);
  (void)hipFree(s->d_state);
  delete s;
  return TDS_OK;
}

int tds_rb_set_stream(tds_rb_sim_t *s, void *stream) {
  if (!s) return rb_fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  s->stream = (hipStream_t)stream;
  return TDS_OK;
}

void *tds_rb_state_device(tds_rb_sim_t *s) { return s ? s->d_state : nullptr; }

int tds_rb_set_state(tds_rb_sim_t *s, const double *h) {
  if (!s || !h) return rb_fail(TDS_ERR_INVALID_ARG, "NULL argument");
  const size_t cnt = (size_t)s->num_worlds * s->model.num_bodies * TDS_RB_STATE;
  if (s->dtype == TDS_DTYPE_F64) {
    RB_TRY(hipMemcpyAsync(s->d_state, h, cnt * 8, hipMemcpyHostToDevice, s->stream));
  } else {
    s->stage.resize(cnt);
    for (size_t i = 0; i < cnt; ++i) s->stage[i] = (float)h[i];
    RB_TRY(hipMemcpyAsync(s->d_state, s->stage.data(), cnt * 4, hipMemcpyHostToDevice, s->stream));
  }
  RB_TRY(hipStreamSynchronize(s->stream));
  return TDS_OK;
}

int tds_rb_get_state(tds_rb_sim_t *s, double *h) {
  if (!s || !h) return rb_fail(TDS_ERR_INVALID_ARG, "NULL argument");
  const size_t cnt = (size_t)s->num_worlds * s->model.num_bodies * TDS_RB_STATE;
  if (s->dtype == TDS_DTYPE_F64) {
    RB_TRY(hipMemcpyAsync(h, s->d_state, cnt * 8, hipMemcpyDeviceToHost, s->stream));
    RB_TRY(hipStreamSynchronize(s->stream));
  } else {
    s->stage.resize(cnt);
    RB_TRY(hipMemcpyAsync(s->stage.data(), s->d_state, cnt * 4, hipMemcpyDeviceToHost, s->stream));
    RB_TRY(hipStreamSynchronize(s->stream));
    for (size_t i = 0; i < cnt; ++i) h[i] = s->stage[i];
  }
  return TDS_OK;
}

int tds_rb_step(tds_rb_sim_t *s, int steps) {
  if (!s) return rb_fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  if (steps < 1) return rb_fail(TDS_ERR_INVALID_ARG, "steps < 1");
  const int blocks = (s->num_worlds + 63) / 64;
  const size_t lds = (size_t)s->model.num_bodies * 9 * 64 * s->elem;
  if (s->dtype == TDS_DTYPE_F64)
    hipLaunchKernelGGL(tds_rb_kernel<double>, dim3(blocks), dim3(64), lds, s->stream, s->h64, (double *)s->d_state,
                       s->num_worlds, steps);
  else
    hipLaunchKernelGGL(tds_rb_kernel<float>, dim3(blocks), dim3(64), lds, s->stream, s->h32, (float *)s->d_state,
                       s->num_worlds, steps);
  RB_TRY(hipGetLastError());
  return TDS_OK;
}

}  // extern "C"
