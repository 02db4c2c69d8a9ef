// tds_rb.hip — free rigid bodies: World::step of worlds that hold tds::RigidBody
// objects, N independent worlds per launch.
//
// Reference: src/world.hpp:293-366 (step: gravity impulse, pairwise narrowphase, num_solver_iterations
// sweeps of RigidBodyConstraintSolver::resolve_collision over the contacts, integrate),
// src/rigid_body.hpp:26-123, src/rb_constraint_solver.hpp:112-165 (the non-CppAD branch),
// src/contact_point.hpp:43-198, 405-438, 468-496 (sphere-sphere, plane-sphere, plane-capsule, plane-box,
// capsule-sphere, and each in the swapped order).
//
// Mapping: none of the five benchmark configurations creates a RigidBody, so this path is built for
// coverage, not speed: one LANE per world (the reference's own CUDA mapping), positions and velocities
// of a world's bodies in LDS as [component][lane] (bank-conflict free, dynamic body index without
// scratch).  The contact list is not stored: body
// poses do not change during the solver sweeps, so each sweep re-derives the contacts of a pair from
// the poses — the same numbers the reference keeps in rb_contacts_.  Capsules and boxes collide as the
// reference's sets of spheres (2 end spheres / 8 corner spheres placed by Pose * offset).
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
  T mass[TDS_RB_MAX_BODIES], inv_mass[TDS_RB_MAX_BODIES], inv_in[TDS_RB_MAX_BODIES];
  T radius[TDS_RB_MAX_BODIES];  // of the body's collision spheres (box: max(1e-2, corner radius))
  T pn[TDS_RB_MAX_BODIES][3], pc[TDS_RB_MAX_BODIES];
  int type[TDS_RB_MAX_BODIES];
  int ns[TDS_RB_MAX_BODIES];    // collision spheres of the body: sphere 1, capsule 2, box 8 (plane 0)
  T off[TDS_RB_MAX_BODIES][8][3];  // their centres in body coordinates
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

// q v q^-1 for a unit quaternion (x, y, z, w) — tiny_quaternion.h:171-176
template <typename T>
__device__ __forceinline__ void quat_rotate(const T *q, const T *v, T *o) {
  const T t0 = q[3] * v[0] + q[1] * v[2] - q[2] * v[1];
  const T t1 = q[3] * v[1] + q[2] * v[0] - q[0] * v[2];
  const T t2 = q[3] * v[2] + q[0] * v[1] - q[1] * v[0];
  const T t3 = -q[0] * v[0] - q[1] * v[1] - q[2] * v[2];
  const T i0 = -q[0], i1 = -q[1], i2 = -q[2], i3 = q[3];
  o[0] = t3 * i0 + t0 * i3 + t1 * i2 - t2 * i1;
  o[1] = t3 * i1 + t1 * i3 + t2 * i0 - t0 * i2;
  o[2] = t3 * i2 + t2 * i3 + t0 * i1 - t1 * i0;
}

// LDS slot of (body b, component c) for this lane; c: 0..2 position, 3..5 linear, 6..8 angular velocity,
// 9..12 orientation quaternion (x, y, z, w)
#define RB_NC 13
#define RB_AT(b, c) sm[((b)*RB_NC + (c)) * 64 + lane]

template <typename T>
__global__ __launch_bounds__(64) void tds_rb_kernel(const RbDev<T> *__restrict__ Mp, T *__restrict__ state, int n_worlds,
                                                    int steps) {
  const RbDev<T> &M = *Mp;  // (too large for the kernel-argument segment)
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
#pragma unroll
    for (int k = 0; k < 4; ++k) RB_AT(b, 9 + k) = valid ? S[b * TDS_RB_STATE + 3 + k] : T(k == 3);
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
          // dispatcher (contact_point.hpp:444-496): P = the plane or the lone sphere, the other body is
          // expanded into its collision spheres; swapped = the reference ran the pair as (j, i)
          int pp, qq, kind;  // kind 0: plane(pp) vs spheres of qq;  1: spheres of pp (capsule / sphere) vs sphere qq
          bool swapped = false;
          const bool jball = tj == TDS_GEOM_SPHERE || tj == TDS_GEOM_CAPSULE || tj == TDS_GEOM_BOX;
          const bool iball = ti == TDS_GEOM_SPHERE || ti == TDS_GEOM_CAPSULE || ti == TDS_GEOM_BOX;
          if (ti == TDS_GEOM_PLANE && jball) {
            pp = i; qq = j; kind = 0;
          } else if (tj == TDS_GEOM_PLANE && iball) {
            pp = j; qq = i; kind = 0; swapped = true;
          } else if ((ti == TDS_GEOM_SPHERE || ti == TDS_GEOM_CAPSULE) && tj == TDS_GEOM_SPHERE) {
            pp = i; qq = j; kind = 1;
          } else if (ti == TDS_GEOM_SPHERE && tj == TDS_GEOM_CAPSULE) {
            pp = j; qq = i; kind = 1; swapped = true;
          } else {
            continue;
          }
          const int eb = kind == 0 ? qq : pp;  // the expanded body
          const T pi[3] = {RB_AT(i, 0), RB_AT(i, 1), RB_AT(i, 2)};
          const T pj[3] = {RB_AT(j, 0), RB_AT(j, 1), RB_AT(j, 2)};
          const T qe[4] = {RB_AT(eb, 9), RB_AT(eb, 10), RB_AT(eb, 11), RB_AT(eb, 12)};
          const T pe[3] = {RB_AT(eb, 0), RB_AT(eb, 1), RB_AT(eb, 2)};
          const T rad = M.radius[eb];
          for (int sx = 0; sx < M.ns[eb]; ++sx) {
          T ctr[3];
          if (M.type[eb] == TDS_GEOM_SPHERE) {
            ctr[0] = pe[0]; ctr[1] = pe[1]; ctr[2] = pe[2];
          } else {  // Pose * offset (pose.hpp:47-53)
            const T ov[3] = {M.off[eb][sx][0], M.off[eb][sx][1], M.off[eb][sx][2]};
            T r3[3];
            quat_rotate(qe, ov, r3);
            ctr[0] = pe[0] + r3[0]; ctr[1] = pe[1] + r3[1]; ctr[2] = pe[2] + r3[2];
          }
          T nbv[3], pa[3], pb[3], dist;
          bool got;
          if (kind == 0) {  // contact_plane_sphere (contact_point.hpp:96-125): A = plane, B = sphere at ctr
            const T mn[3] = {-M.pn[pp][0], -M.pn[pp][1], -M.pn[pp][2]};
            const T t = -(dot3(ctr, mn) + M.pc[pp]);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              pa[k] = ctr[k] + t * mn[k];
              pb[k] = ctr[k] - rad * M.pn[pp][k];
              nbv[k] = mn[k];
            }
            dist = t - rad;
            got = true;
          } else {          // contact_sphere_sphere (contact_point.hpp:43-94): A = sphere at ctr, B = body qq
            const T cq[3] = {RB_AT(qq, 0), RB_AT(qq, 1), RB_AT(qq, 2)};
            const T diff[3] = {ctr[0] - cq[0], ctr[1] - cq[1], ctr[2] - cq[2]};
            const T length = sqrt(dot3(diff, diff));
            dist = length - (rad + M.radius[qq]);
            got = length > T(1) / T(100000);
            const T il = T(1) / length;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              nbv[k] = il * diff[k];
              pa[k] = ctr[k] - rad * nbv[k];
              pb[k] = pa[k] - dist * nbv[k];
            }
          }
          if (swapped) {  // swap normal and points a, b (contact_point.hpp:484-491)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const T t = pa[k];
              pa[k] = pb[k];
              pb[k] = t;
              nbv[k] = -nbv[k];
            }
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
          }  // collision spheres of the expanded body
        }
      }
    }
    // integrate (rigid_body.hpp:116-122, tiny_algebra.hpp:604-614): the quaternion lives in HBM
    for (int b = 0; b < nb; ++b) {
#pragma unroll
      for (int k = 0; k < 3; ++k) RB_AT(b, k) += RB_AT(b, 3 + k) * dt;
      {
        const T qx = RB_AT(b, 9), qy = RB_AT(b, 10), qz = RB_AT(b, 11), qw = RB_AT(b, 12);
        const T w0 = RB_AT(b, 6), w1 = RB_AT(b, 7), w2 = RB_AT(b, 8);
        const T hd = T(0.5) * dt;
        const T ww = (-qx * w0 - qy * w1 - qz * w2) * hd;
        const T xx = (qw * w0 + qz * w1 - qy * w2) * hd;
        const T yy = (qw * w1 + qx * w2 - qz * w0) * hd;
        const T zz = (qw * w2 + qy * w0 - qx * w1) * hd;
        const T nx = qx + xx, ny = qy + yy, nz = qz + zz, nw = qw + ww;
        const T ql = sqrt(nx * nx + ny * ny + nz * nz + nw * nw);
        RB_AT(b, 9) = nx / ql;
        RB_AT(b, 10) = ny / ql;
        RB_AT(b, 11) = nz / ql;
        RB_AT(b, 12) = nw / ql;
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
#pragma unroll
      for (int k = 0; k < 4; ++k) S[b * TDS_RB_STATE + 3 + k] = RB_AT(b, 9 + k);
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
    d->type[i] = b.geom_type;
    double rad = b.radius;
    if (b.geom_type == TDS_GEOM_SPHERE) {
      d->ns[i] = 1;
    } else if (b.geom_type == TDS_GEOM_CAPSULE) {  // contact_point.hpp:143-158
      d->ns[i] = 2;
      d->off[i][0][2] = (T)(0.5 * b.length);
      d->off[i][1][2] = (T)(-0.5 * b.length);
    } else if (b.geom_type == TDS_GEOM_BOX) {      // contact_point.hpp:179-196, geometry.hpp:244-259
      d->ns[i] = 8;
      rad = b.radius > 1e-2 ? b.radius : 1e-2;
      const double dx = b.extents[0] * 0.5 - rad, dy = b.extents[1] * 0.5 - rad, dz = b.extents[2] * 0.5 - rad;
      for (int c = 0; c < 8; ++c) {
        d->off[i][c][0] = (T)((c & 4) ? -dx : dx);
        d->off[i][c][1] = (T)((c & 2) ? -dy : dy);
        d->off[i][c][2] = (T)((c & 1) ? -dz : dz);
      }
    }
    d->radius[i] = (T)rad;
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
  void *d_state = nullptr, *d_model = nullptr;
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
    if (model->bodies[i].geom_type != TDS_GEOM_SPHERE && model->bodies[i].geom_type != TDS_GEOM_PLANE &&
        model->bodies[i].geom_type != TDS_GEOM_CAPSULE && model->bodies[i].geom_type != TDS_GEOM_BOX)
      return rb_fail(TDS_ERR_UNSUPPORTED, "rigid bodies support sphere, plane, capsule and box geometries");
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
  {
    const size_t msz = dtype == TDS_DTYPE_F64 ? sizeof(RbDev<double>) : sizeof(RbDev<float>);
    const void *src = dtype == TDS_DTYPE_F64 ? (const void *)&s->h64 : (const void *)&s->h32;
    if (hipMalloc(&s->d_model, msz) != hipSuccess || hipMemcpy(s->d_model, src, msz, hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(s->d_state);
      delete s;
      return rb_fail(TDS_ERR_HIP, "upload of the rigid-body model failed");
    }
  }
  const int lds = model->num_bodies * RB_NC * 64 * (int)s->elem;
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
  (void)hipFree(s->d_model);
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
  const size_t lds = (size_t)s->model.num_bodies * RB_NC * 64 * s->elem;
  if (s->dtype == TDS_DTYPE_F64)
    hipLaunchKernelGGL(tds_rb_kernel<double>, dim3(blocks), dim3(64), lds, s->stream, (const RbDev<double> *)s->d_model,
                       (double *)s->d_state,
                       s->num_worlds, steps);
  else
    hipLaunchKernelGGL(tds_rb_kernel<float>, dim3(blocks), dim3(64), lds, s->stream, (const RbDev<float> *)s->d_model,
                       (float *)s->d_state,
                       s->num_worlds, steps);
  RB_TRY(hipGetLastError());
  return TDS_OK;
}

}  // extern "C"
