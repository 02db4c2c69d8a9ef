/* tds_oracle.c — TEST INFRASTRUCTURE ONLY (see tds_oracle.h for the pinning statement).
 *
 * Plain-C restatement of the reference's per-environment step.  "ref:" comments give the
 * file:line under /root/reference that each block follows.  Default (right-associative)
 * transform convention, TinyAlgebra<double> arithmetic.
 */
#include "tds_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NL TDS_MAX_LINKS
#define ND 32
#define NCMAX 64
#define NR (3 * NCMAX)

typedef struct { double r[9], t[3]; } xf_t;      /* Transform: rotation (row-major), translation */
typedef struct { double a[3], l[3]; } sv_t;      /* spatial vector: top (angular), bottom (linear) */
typedef struct { double I[9], H[9], M[9]; } abi_t; /* ArticulatedBodyInertia blocks */

/* ---------------------------------------------------------------- 3-vector / 3x3 helpers */
static void v3_cross(const double *a, const double *b, double *o) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static double v3_dot(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void m3_mulv(const double *m, const double *v, double *o) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void m3_tmulv(const double *m, const double *v, double *o) { /* m^T v */
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  double y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  double z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void m3_mul(const double *a, const double *b, double *o) {
  double t[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      t[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c];
  memcpy(o, t, sizeof(t));
}
static void m3_transpose(const double *a, double *o) {
  double t[9] = {a[0], a[3], a[6], a[1], a[4], a[7], a[2], a[5], a[8]};
  memcpy(o, t, sizeof(t));
}
static void m3_identity(double *m) { memset(m, 0, 9 * sizeof(double)); m[0] = m[4] = m[8] = 1.0; }
/* ref: src/math/tiny/tiny_matrix3x3.h:1015-1021 */
static void m3_cross_matrix(const double *v, double *m) {
  m[0] = 0; m[1] = -v[2]; m[2] = v[1];
  m[3] = v[2]; m[4] = 0; m[5] = -v[0];
  m[6] = -v[1]; m[7] = v[0]; m[8] = 0;
}

/* ---------------------------------------------------------------- quaternions (x,y,z,w) */
/* ref: src/math/tiny/tiny_matrix3x3.h:315-340 (setRotation, right-associative branch) */
static void quat_to_matrix(const double *q, double *m) {
  double d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (d == 0.0) { return; }
  double s = 2.0 / d;
  double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
  double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs;
  double xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs;
  double yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
  m[0] = 1.0 - (yy + zz); m[1] = xy - wz; m[2] = xz + wy;
  m[3] = xy + wz; m[4] = 1.0 - (xx + zz); m[5] = yz - wx;
  m[6] = xz - wy; m[7] = yz + wx; m[8] = 1.0 - (xx + yy);
}
/* ref: src/math/tiny/tiny_matrix3x3.h:432-465 (getRotation, non-CppAD, right-associative:
   off-diagonal differences transposed w.r.t. Bullet and w negated) */
static void matrix_to_quat(const double *m, double *q) {
  double trace = m[0] + m[4] + m[8];
  double temp[4];
  if (trace < 0.0) {
    int i = m[0] < m[4] ? (m[4] < m[8] ? 2 : 1) : (m[0] < m[8] ? 2 : 0);
    int j = (i + 1) % 3, k = (i + 2) % 3;
    double tmp = ((m[3 * i + i] - m[3 * j + j]) - m[3 * k + k]) + 1.0;
    double s = sqrt(tmp);
    temp[i] = s * 0.5;
    s = 0.5 / s;
    temp[3] = (m[3 * j + k] - m[3 * k + j]) * s;
    temp[j] = (m[3 * i + j] + m[3 * j + i]) * s;
    temp[k] = (m[3 * i + k] + m[3 * k + i]) * s;
  } else {
    double s = sqrt(trace + 1.0);
    temp[3] = s * 0.5;
    s = 0.5 / s;
    temp[0] = (m[5] - m[7]) * s;
    temp[1] = (m[6] - m[2]) * s;
    temp[2] = (m[1] - m[3]) * s;
  }
  q[0] = temp[0]; q[1] = temp[1]; q[2] = temp[2]; q[3] = -temp[3];
}
/* ref: src/math/tiny/tiny_algebra.hpp:219-222 */
static void quat_normalize(double *q) {
  double ql = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= ql; q[1] /= ql; q[2] /= ql; q[3] /= ql;
}
/* ref: src/math/tiny/tiny_quaternion.h:171-176, 306-345 (q * v) * q^-1 */
static void quat_rotate(const double *q, const double *v, double *o) {
  double t[4] = {q[3] * v[0] + q[1] * v[2] - q[2] * v[1], q[3] * v[1] + q[2] * v[0] - q[0] * v[2],
                 q[3] * v[2] + q[0] * v[1] - q[1] * v[0], -q[0] * v[0] - q[1] * v[1] - q[2] * v[2]};
  double i[4] = {-q[0], -q[1], -q[2], q[3]};
  /* t *= i  (tiny_quaternion.h:90-96) */
  o[0] = t[3] * i[0] + t[0] * i[3] + t[1] * i[2] - t[2] * i[1];
  o[1] = t[3] * i[1] + t[1] * i[3] + t[2] * i[0] - t[0] * i[2];
  o[2] = t[3] * i[2] + t[2] * i[3] + t[0] * i[1] - t[1] * i[0];
}

/* ---------------------------------------------------------------- Transform */
static void xf_identity(xf_t *x) { m3_identity(x->r); x->t[0] = x->t[1] = x->t[2] = 0; }
/* ref: src/math/transform.hpp:123-131  (A*B).t = A.t + A.R B.t ; R = A.R B.R */
static void xf_mul(const xf_t *a, const xf_t *b, xf_t *o) {
  xf_t r;
  double rt[3];
  m3_mulv(a->r, b->t, rt);
  for (int k = 0; k < 3; ++k) r.t[k] = a->t[k] + rt[k];
  m3_mul(a->r, b->r, r.r);
  *o = r;
}
/* ref: src/math/transform.hpp:210-226  X*V = (E w, E (v - r x w)),  E = R^T */
static void xf_apply_motion(const xf_t *x, const sv_t *in, sv_t *o) {
  double rxw[3], v_rxw[3];
  sv_t r;
  v3_cross(x->t, in->a, rxw);
  for (int k = 0; k < 3; ++k) v_rxw[k] = in->l[k] - rxw[k];
  m3_tmulv(x->r, in->a, r.a);
  m3_tmulv(x->r, v_rxw, r.l);
  *o = r;
}
/* ref: src/math/transform.hpp:232-243  inv(X)*V = (R w, R v + r x (R w)) */
static void xf_apply_inverse_motion(const xf_t *x, const sv_t *in, sv_t *o) {
  sv_t r;
  double c[3];
  m3_mulv(x->r, in->a, r.a);
  m3_mulv(x->r, in->l, r.l);
  v3_cross(x->t, r.a, c);
  for (int k = 0; k < 3; ++k) r.l[k] += c[k];
  *o = r;
}
/* ref: src/math/transform.hpp:249-262  X^T F = (R n + r x (R f), R f) */
static void xf_apply_force(const xf_t *x, const sv_t *in, sv_t *o) {
  sv_t r;
  double c[3];
  m3_mulv(x->r, in->l, r.l);
  m3_mulv(x->r, in->a, r.a);
  v3_cross(x->t, r.l, c);
  for (int k = 0; k < 3; ++k) r.a[k] += c[k];
  *o = r;
}
/* ref: src/math/transform.hpp:72-87  matrix(): [E 0; -E rx  E], E = R^T */
static void xf_matrix(const xf_t *x, double *m /*36*/) {
  double E[9], rx[9], mErx[9];
  m3_transpose(x->r, E);
  m3_cross_matrix(x->t, rx);
  m3_mul(E, rx, mErx);
  for (int k = 0; k < 9; ++k) mErx[k] = -mErx[k];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      m[6 * r + c] = E[3 * r + c];
      m[6 * r + 3 + c] = 0.0;
      m[6 * (r + 3) + c] = mErx[3 * r + c];
      m[6 * (r + 3) + 3 + c] = E[3 * r + c];
    }
}
/* ref: src/math/transform.hpp:89-104  matrix_transpose(): [Et (-E rx)^T; 0 Et] */
static void xf_matrix_transpose(const xf_t *x, double *m /*36*/) {
  double E[9], rx[9], mErx[9], mErxT[9];
  m3_transpose(x->r, E);
  m3_cross_matrix(x->t, rx);
  m3_mul(E, rx, mErx);
  for (int k = 0; k < 9; ++k) mErx[k] = -mErx[k];
  m3_transpose(mErx, mErxT);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      m[6 * r + c] = x->r[3 * r + c];
      m[6 * r + 3 + c] = mErxT[3 * r + c];
      m[6 * (r + 3) + c] = 0.0;
      m[6 * (r + 3) + 3 + c] = x->r[3 * r + c];
    }
}

/* ---------------------------------------------------------------- spatial algebra */
/* ref: src/math/tiny/tiny_algebra.hpp:101-105  V1 x V2 = (w1 x w2, w1 x v2 + v1 x w2) */
static void sv_cross_mm(const sv_t *a, const sv_t *b, sv_t *o) {
  sv_t r;
  double c1[3], c2[3];
  v3_cross(a->a, b->a, r.a);
  v3_cross(a->a, b->l, c1);
  v3_cross(a->l, b->a, c2);
  for (int k = 0; k < 3; ++k) r.l[k] = c1[k] + c2[k];
  *o = r;
}
/* ref: src/math/tiny/tiny_algebra.hpp:112-115  V x* F = (w x n + v x f, w x f) */
static void sv_cross_mf(const sv_t *a, const sv_t *b, sv_t *o) {
  sv_t r;
  double c1[3], c2[3];
  v3_cross(a->a, b->a, c1);
  v3_cross(a->l, b->l, c2);
  for (int k = 0; k < 3; ++k) r.a[k] = c1[k] + c2[k];
  v3_cross(a->a, b->l, r.l);
  *o = r;
}
static double sv_dot(const sv_t *a, const sv_t *b) { return v3_dot(a->a, b->a) + v3_dot(a->l, b->l); }

/* ref: src/math/inertia.hpp:121-130  ABI from RBI */
static void abi_from_rbi(double mass, const double *com, const double *inertia, abi_t *o) {
  double H[9], Ht[9], HHt[9];
  m3_cross_matrix(com, H);
  m3_transpose(H, Ht);
  m3_mul(H, Ht, HHt);
  for (int k = 0; k < 9; ++k) o->I[k] = inertia[k] + HHt[k] * mass;
  memset(o->M, 0, sizeof(o->M));
  o->M[0] = o->M[4] = o->M[8] = mass;
  for (int k = 0; k < 9; ++k) o->H[k] = H[k] * mass;
}
/* ref: src/math/inertia.hpp:205-210  IA*v = (I w + H v, M v + H^T w) */
static void abi_mul(const abi_t *A, const sv_t *v, sv_t *o) {
  sv_t r;
  double t1[3], t2[3];
  m3_mulv(A->I, v->a, t1);
  m3_mulv(A->H, v->l, t2);
  for (int k = 0; k < 3; ++k) r.a[k] = t1[k] + t2[k];
  m3_mulv(A->M, v->l, t1);
  m3_tmulv(A->H, v->a, t2);
  for (int k = 0; k < 3; ++k) r.l[k] = t1[k] + t2[k];
  *o = r;
}
/* ref: src/math/inertia.hpp:152-160 */
static void abi_matrix(const abi_t *A, double *m /*36*/) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      m[6 * r + c] = A->I[3 * r + c];
      m[6 * r + 3 + c] = A->H[3 * r + c];
      m[6 * (r + 3) + c] = A->H[3 * c + r];
      m[6 * (r + 3) + 3 + c] = A->M[3 * r + c];
    }
}
/* ref: src/math/inertia.hpp:138-143  only I, H (upper right) and M blocks are read */
static void abi_from_matrix(const double *m, abi_t *A) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      A->I[3 * r + c] = m[6 * r + c];
      A->H[3 * r + c] = m[6 * r + 3 + c];
      A->M[3 * r + c] = m[6 * (r + 3) + 3 + c];
    }
}
static void m6_mul(const double *a, const double *b, double *o) {
  double t[36];
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) {
      double s = 0.0;
      for (int k = 0; k < 6; ++k) s += a[6 * r + k] * b[6 * k + c];
      t[6 * r + c] = s;
    }
  memcpy(o, t, sizeof(t));
}
/* ref: forward_dynamics.hpp:187-189 / mass_matrix.hpp:45-46  X^T * Ia * X as dense 6x6 */
static void abi_congruence(const xf_t *X, const abi_t *Ia, abi_t *out) {
  double xt[36], im[36], xm[36], tmp[36], xix[36];
  xf_matrix_transpose(X, xt);
  abi_matrix(Ia, im);
  xf_matrix(X, xm);
  m6_mul(xt, im, tmp);
  m6_mul(tmp, xm, xix);
  abi_from_matrix(xix, out);
}

/* ---------------------------------------------------------------- per-env scratch */
/* ref: tiny_matrix3x3.h:539-559 (cofactor inverse) */
static void m3_inverse(const double *m, double *o) {
  double c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
  double s = 1.0 / (m[0] * c0 + m[1] * c1 + m[2] * c2);
  o[0] = c0 * s; o[1] = (m[2] * m[7] - m[1] * m[8]) * s; o[2] = (m[1] * m[5] - m[2] * m[4]) * s;
  o[3] = c1 * s; o[4] = (m[0] * m[8] - m[2] * m[6]) * s; o[5] = (m[2] * m[3] - m[0] * m[5]) * s;
  o[6] = c2 * s; o[7] = (m[1] * m[6] - m[0] * m[7]) * s; o[8] = (m[0] * m[4] - m[1] * m[3]) * s;
}

/* ref: src/math/inertia.hpp:302-329  ArticulatedBodyInertia::inverse() / inv_mul().  NOTE the reference takes
   C = -H for the lower-left block (exact only while H is skew, i.e. for a single rigid body); restated as is. */
static void abi_inv_mul(const abi_t *A, const sv_t *f, sv_t *o) {
  double Ainv[9], C[9], t1[9], t2[9], S[9], D[9], ABD[9], I2[9], H2[9], Ht[9];
  m3_inverse(A->I, Ainv);
  for (int k = 0; k < 9; ++k) C[k] = -A->H[k];
  m3_mul(C, Ainv, t1);
  m3_mul(t1, A->H, t2);
  for (int k = 0; k < 9; ++k) S[k] = A->M[k] - t2[k]; /* MCAinvB */
  m3_inverse(S, D);                                     /* DCAB */
  m3_mul(Ainv, A->H, t1);
  m3_mul(t1, D, ABD);                                   /* AinvBDCAB */
  m3_mul(ABD, C, t1);
  m3_mul(t1, Ainv, t2);
  for (int k = 0; k < 9; ++k) { I2[k] = Ainv[k] + t2[k]; H2[k] = -ABD[k]; }
  m3_transpose(H2, Ht);
  double a[3], b[3];
  m3_mulv(I2, f->a, a); m3_mulv(H2, f->l, b);
  for (int k = 0; k < 3; ++k) o->a[k] = a[k] + b[k];
  m3_mulv(D, f->l, a); m3_mulv(Ht, f->a, b);
  for (int k = 0; k < 3; ++k) o->l[k] = a[k] + b[k];
}

typedef struct {
  xf_t X_J, X_parent, X_world;
  sv_t S, vJ, v, c, a, pA, U;
  abi_t abi;
  double D, u;
  /* spherical joint (link.hpp:168-176: S_3d = [1 0]^T, three angular columns in the link frame) */
  sv_t U3[3];
  double invD3[9], u3[3];
} lstate_t;

typedef struct {
  double normal[3], point_a[3], point_b[3], distance;
  int link_b;
} contact_t;

typedef struct {
  lstate_t L[NL];
  xf_t base_X_world;
  /* floating base (multi_body.hpp:66-78): spatial velocity / acceleration, articulated inertia, bias force */
  sv_t base_v, base_a, base_bias;
  abi_t base_abi;
  double q[ND + 1 + NL], qd[ND], qdd[ND], tau[ND]; /* (+1 floating base, +1 per spherical joint) */
  contact_t cps[NCMAX];
  int n_c;
  double M[ND * ND], Minv[ND * ND];
  double J[NR * ND], JM[NR * ND], A[NR * NR], b[NR], p[NR], lo[NR], hi[NR];
  int dep[NR];
  double jac[3 * ND];
} scratch_t;

static void link_xf(const tds_link_t *l, xf_t *x) {
  memcpy(x->r, l->X_T_rot, sizeof(x->r));
  memcpy(x->t, l->X_T_trans, sizeof(x->t));
}

/* ref: src/link.hpp:229-287 (X_J, X_parent) and :289-329 (vJ) */
static void jcalc(const tds_link_t *l, const double *qp, const double *qdp, int have_qd, lstate_t *s) {
  xf_t XT;
  link_xf(l, &XT);
  xf_identity(&s->X_J);
  if (l->joint_type == TDS_JOINT_SPHERICAL) { /* link.hpp:262-266, :319-321 */
    quat_to_matrix(qp, s->X_J.r);
    xf_mul(&XT, &s->X_J, &s->X_parent);
    for (int k = 0; k < 3; ++k) { s->vJ.a[k] = have_qd ? qdp[k] : 0.0; s->vJ.l[k] = 0.0; }
    return;
  }
  const double q = qp ? qp[0] : 0.0;
  double qd = qdp ? qdp[0] : 0.0;
  double c = cos(q), sn = sin(q);
  switch (l->joint_type) {
    case TDS_JOINT_PRISMATIC_X: s->X_J.t[0] = q; break;
    case TDS_JOINT_PRISMATIC_Y: s->X_J.t[1] = q; break;
    case TDS_JOINT_PRISMATIC_Z: s->X_J.t[2] = q; break;
    case TDS_JOINT_PRISMATIC_AXIS:
      for (int k = 0; k < 3; ++k) s->X_J.t[k] = l->S[3 + k] * q;
      break;
    case TDS_JOINT_REVOLUTE_X: { /* ref: tiny_matrix3x3.h:218-234 */
      double m[9] = {1, 0, 0, 0, c, -sn, 0, sn, c};
      memcpy(s->X_J.r, m, sizeof(m));
      break;
    }
    case TDS_JOINT_REVOLUTE_Y: {
      double m[9] = {c, 0, sn, 0, 1, 0, -sn, 0, c};
      memcpy(s->X_J.r, m, sizeof(m));
      break;
    }
    case TDS_JOINT_REVOLUTE_Z: {
      double m[9] = {c, -sn, 0, sn, c, 0, 0, 0, 1};
      memcpy(s->X_J.r, m, sizeof(m));
      break;
    }
    case TDS_JOINT_REVOLUTE_AXIS: { /* ref: link.hpp:256-261, tiny_quaternion.h:178-183 */
      const double *axis = l->S;
      double d = sqrt(v3_dot(axis, axis));
      double sh = sin(q * 0.5) / d;
      double quat[4] = {axis[0] * sh, axis[1] * sh, axis[2] * sh, cos(q * 0.5)};
      quat_to_matrix(quat, s->X_J.r);
      break;
    }
    default: break; /* JOINT_FIXED: identity */
  }
  xf_mul(&XT, &s->X_J, &s->X_parent); /* ref: link.hpp:283 */
  /* vJ: every jcalc(qd) variant writes S-aligned components only == S*qd (S unit-axis for _X/_Y/_Z) */
  if (!have_qd) qd = 0.0;
  for (int k = 0; k < 3; ++k) {
    s->vJ.a[k] = l->S[k] * qd;
    s->vJ.l[k] = l->S[3 + k] * qd;
  }
}

/* ref: src/dynamics/kinematics.hpp:18-148 (fixed and floating base) */
static void forward_kinematics(const tds_model_t *m, scratch_t *s, int have_qd) {
  if (m->is_floating) { /* :35-62 */
    quat_to_matrix(s->q, s->base_X_world.r);
    for (int k = 0; k < 3; ++k) s->base_X_world.t[k] = s->q[4 + k];
    for (int k = 0; k < 3; ++k) {
      s->base_v.a[k] = have_qd ? s->qd[k] : 0.0;
      s->base_v.l[k] = have_qd ? s->qd[3 + k] : 0.0;
    }
    abi_from_rbi(m->base_mass, m->base_com, m->base_inertia, &s->base_abi); /* :50 */
    /* :52-59 gyroscopic force from the "world" inertia tensor R I R^T and the base angular velocity, stored
       as the top of the base-frame bias force (frames mixed exactly as the reference does) */
    double Rt[9], RI[9], Iw[9], Iwv[3];
    m3_transpose(s->base_X_world.r, Rt);
    m3_mul(s->base_X_world.r, m->base_inertia, RI);
    m3_mul(RI, Rt, Iw);
    m3_mulv(Iw, s->base_v.a, Iwv);
    v3_cross(s->base_v.a, Iwv, s->base_bias.a);
    s->base_bias.l[0] = s->base_bias.l[1] = s->base_bias.l[2] = 0.0; /* base_applied_force == 0 */
  }
  for (int i = 0; i < m->num_links; ++i) {
    const tds_link_t *l = &m->links[i];
    lstate_t *L = &s->L[i];
    const double *q = l->q_index >= 0 ? &s->q[l->q_index] : NULL;   /* multi_body.hpp:490-500 */
    const double *qd = l->qd_index >= 0 ? &s->qd[l->qd_index] : NULL;
    jcalc(l, q, qd, have_qd, L);
    if (l->parent >= 0) {
      xf_mul(&s->L[l->parent].X_world, &L->X_parent, &L->X_world); /* :82 */
      sv_t xv;
      xf_apply_motion(&L->X_parent, &s->L[l->parent].v, &xv); /* :86 */
      for (int k = 0; k < 3; ++k) { L->v.a[k] = xv.a[k] + L->vJ.a[k]; L->v.l[k] = xv.l[k] + L->vJ.l[k]; }
    } else if (m->is_floating) {
      xf_mul(&s->base_X_world, &L->X_parent, &L->X_world); /* :82 with parent_X_world = base_X_world */
      sv_t xv;
      xf_apply_motion(&L->X_parent, &s->base_v, &xv); /* :84-87 */
      for (int k = 0; k < 3; ++k) { L->v.a[k] = xv.a[k] + L->vJ.a[k]; L->v.l[k] = xv.l[k] + L->vJ.l[k]; }
    } else {
      xf_mul(&s->base_X_world, &L->X_parent, &L->X_world); /* :92 */
      L->v = L->vJ;
    }
    sv_cross_mm(&L->v, &L->vJ, &L->c); /* :96-97 (cJ == 0) */
    abi_from_rbi(l->mass, l->com, l->inertia, &L->abi); /* :99 */
    sv_t Iv;
    abi_mul(&L->abi, &L->v, &Iv);
    sv_cross_mf(&L->v, &Iv, &L->pA); /* :132, f_ext == 0 after clear_forces */
  }
}

/* ref: src/dynamics/forward_dynamics.hpp:11-326 (fixed base, 1-DoF + fixed joints) */
static void forward_dynamics(const tds_model_t *m, scratch_t *s) {
  forward_kinematics(m, s, 1);
  for (int i = m->num_links - 1; i >= 0; --i) {
    const tds_link_t *l = &m->links[i];
    lstate_t *L = &s->L[i];
    if (l->joint_type == TDS_JOINT_SPHERICAL) { /* :56-109 */
      double D3[9];
      for (int c = 0; c < 3; ++c) {
        sv_t e = {{0, 0, 0}, {0, 0, 0}};
        e.a[c] = 1.0;
        abi_mul(&L->abi, &e, &L->U3[c]);            /* U_3d = abi * S_3d */
      }
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) D3[3 * r + c] = L->U3[c].a[r]; /* D_3d = S_3d^T U_3d */
      /* tau - stiffness * axis_angle(quat) - damping * qd  (:62-76; the loader leaves both at zero) */
      for (int k = 0; k < 3; ++k)
        L->u3[k] = s->tau[l->qd_index + k] - l->damping * s->qd[l->qd_index + k] - L->pA.a[k]; /* :79 */
      if (l->stiffness != 0.0) {
        const double *qq = &s->q[l->q_index];
        double qn = sqrt(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2]);
        double theta = 2.0 * atan2(qn, qq[3]);       /* tiny_algebra.hpp:509-527 */
        double sc = qn < pow(2.220446049250313e-16, 0.25) ? 1.0 / (0.5 + theta * theta / 48.0) : theta / qn;
        for (int k = 0; k < 3; ++k) L->u3[k] -= l->stiffness * sc * qq[k];
      }
      m3_inverse(D3, L->invD3);                       /* :100 */
      /* u_dinv_ut = U (U invD)^T  (inertia.hpp:353-368),  UuD = U (invD u) */
      double w[3];
      m3_mulv(L->invD3, L->u3, w);
      abi_t Ia = L->abi;
      sv_t pa = L->pA;
      for (int c = 0; c < 3; ++c) {
        sv_t Ub = {{0, 0, 0}, {0, 0, 0}}; /* column c of U invD */
        for (int k = 0; k < 3; ++k)
          for (int j = 0; j < 3; ++j) { Ub.a[j] += L->U3[k].a[j] * L->invD3[3 * k + c]; Ub.l[j] += L->U3[k].l[j] * L->invD3[3 * k + c]; }
        for (int r = 0; r < 3; ++r)
          for (int j = 0; j < 3; ++j) {
            Ia.I[3 * r + j] -= L->U3[c].a[r] * Ub.a[j];
            Ia.H[3 * r + j] -= L->U3[c].a[r] * Ub.l[j];
            Ia.M[3 * r + j] -= L->U3[c].l[r] * Ub.l[j];
          }
        for (int j = 0; j < 3; ++j) { pa.a[j] += L->U3[c].a[j] * w[c]; pa.l[j] += L->U3[c].l[j] * w[c]; }
      }
      sv_t Ia_c;
      abi_mul(&Ia, &L->c, &Ia_c);
      for (int k = 0; k < 3; ++k) { pa.a[k] += Ia_c.a[k]; pa.l[k] += Ia_c.l[k]; }
      if (l->parent >= 0 || m->is_floating) {
        sv_t dpA;
        abi_t dI;
        xf_apply_force(&L->X_parent, &pa, &dpA);
        abi_congruence(&L->X_parent, &Ia, &dI);
        sv_t *PpA = l->parent >= 0 ? &s->L[l->parent].pA : &s->base_bias;
        abi_t *Pabi = l->parent >= 0 ? &s->L[l->parent].abi : &s->base_abi;
        for (int k = 0; k < 3; ++k) { PpA->a[k] += dpA.a[k]; PpA->l[k] += dpA.l[k]; }
        for (int k = 0; k < 9; ++k) { Pabi->I[k] += dI.I[k]; Pabi->H[k] += dI.H[k]; Pabi->M[k] += dI.M[k]; }
      }
      continue;
    }
    abi_mul(&L->abi, &L->S, &L->U);           /* :111 */
    L->D = sv_dot(&L->S, &L->U);              /* :115 */
    double tau_val = 0.0;                      /* multi_body.hpp:557-570 */
    if (l->joint_type != TDS_JOINT_FIXED) tau_val = s->tau[l->qd_index];
    double qv = l->q_index >= 0 ? s->q[l->q_index] : 0.0;
    double qdv = l->qd_index >= 0 ? s->qd[l->qd_index] : 0.0;
    tau_val -= l->stiffness * qv;              /* :122 */
    tau_val -= l->damping * qdv;               /* :123 */
    L->u = tau_val - sv_dot(&L->S, &L->pA);    /* :129 */
    double invD = l->joint_type == TDS_JOINT_FIXED ? 0.0 : 1.0 / L->D; /* :153 */
    /* u_dinv_ut = U (U invD)^T  (inertia.hpp:333-348) */
    abi_t Ia;
    sv_t Ub;
    for (int k = 0; k < 3; ++k) { Ub.a[k] = L->U.a[k] * invD; Ub.l[k] = L->U.l[k] * invD; }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        Ia.I[3 * r + c] = L->abi.I[3 * r + c] - L->U.a[r] * Ub.a[c];
        Ia.H[3 * r + c] = L->abi.H[3 * r + c] - L->U.a[r] * Ub.l[c];
        Ia.M[3 * r + c] = L->abi.M[3 * r + c] - L->U.l[r] * Ub.l[c];
      }                                         /* :168 */
    sv_t Ia_c, pa, UuD;
    abi_mul(&Ia, &L->c, &Ia_c);                /* :171 */
    double uinvD = L->u * invD;
    for (int k = 0; k < 3; ++k) { UuD.a[k] = L->U.a[k] * uinvD; UuD.l[k] = L->U.l[k] * uinvD; } /* :162 */
    for (int k = 0; k < 3; ++k) {
      pa.a[k] = L->pA.a[k] + Ia_c.a[k] + UuD.a[k];
      pa.l[k] = L->pA.l[k] + Ia_c.l[k] + UuD.l[k];
    }                                           /* :173 */
    if (l->parent >= 0 || m->is_floating) {
      sv_t dpA;
      abi_t dI;
      xf_apply_force(&L->X_parent, &pa, &dpA); /* :181 */
      abi_congruence(&L->X_parent, &Ia, &dI);  /* :187-189 */
      sv_t *PpA = l->parent >= 0 ? &s->L[l->parent].pA : &s->base_bias;   /* :201 / :206 */
      abi_t *Pabi = l->parent >= 0 ? &s->L[l->parent].abi : &s->base_abi; /* :202 / :207 */
      for (int k = 0; k < 3; ++k) { PpA->a[k] += dpA.a[k]; PpA->l[k] += dpA.l[k]; }
      for (int k = 0; k < 9; ++k) { Pabi->I[k] += dI.I[k]; Pabi->H[k] += dI.H[k]; Pabi->M[k] += dI.M[k]; }
    }
  }
  sv_t a_base;
  if (m->is_floating) { /* :232  base_acceleration = -base_abi.inv_mul(base_bias_force) */
    sv_t r;
    abi_inv_mul(&s->base_abi, &s->base_bias, &r);
    for (int k = 0; k < 3; ++k) { a_base.a[k] = -r.a[k]; a_base.l[k] = -r.l[k]; }
  } else { /* :242  base_acceleration = -spatial_gravity (not rotated: rbdl_convention=false) */
    for (int k = 0; k < 3; ++k) { a_base.a[k] = 0.0; a_base.l[k] = -m->gravity[k]; }
  }
  for (int i = 0; i < m->num_links; ++i) { /* :245-302 */
    const tds_link_t *l = &m->links[i];
    lstate_t *L = &s->L[i];
    const sv_t *ap = l->parent >= 0 ? &s->L[l->parent].a : &a_base;
    sv_t xa;
    xf_apply_motion(&L->X_parent, ap, &xa);
    for (int k = 0; k < 3; ++k) { L->a.a[k] = xa.a[k] + L->c.a[k]; L->a.l[k] = xa.l[k] + L->c.l[k]; }
    if (l->joint_type == TDS_JOINT_SPHERICAL) { /* :268-283 */
      double r3[3], qdd3[3];
      for (int c = 0; c < 3; ++c) r3[c] = L->u3[c] - sv_dot(&L->U3[c], &L->a);
      m3_mulv(L->invD3, r3, qdd3);
      for (int c = 0; c < 3; ++c) { s->qdd[l->qd_index + c] = qdd3[c]; L->a.a[c] += qdd3[c]; }
    } else if (l->qd_index >= 0) {
      double invD = l->joint_type == TDS_JOINT_FIXED ? 0.0 : 1.0 / L->D;
      double Ut_a = sv_dot(&L->U, &L->a);
      double qdd = invD * (L->u - Ut_a);
      s->qdd[l->qd_index] = qdd;
      for (int k = 0; k < 3; ++k) { L->a.a[k] += L->S.a[k] * qdd; L->a.l[k] += L->S.l[k] * qdd; }
    }
  }
  if (m->is_floating) { /* :315-319  gravity (WORLD components) is added to the base-frame acceleration */
    for (int k = 0; k < 3; ++k) {
      s->qdd[k] = a_base.a[k];
      s->qdd[3 + k] = a_base.l[k] + m->gravity[k];
    }
  }
}

/* ref: src/dynamics/mass_matrix.hpp:13-127 (fixed base, 1-DoF + fixed joints) */
static void mass_matrix(const tds_model_t *m, scratch_t *s) {
  const int n = m->num_links, nd = m->dof_qd;
  forward_kinematics(m, s, 0); /* :37  qd empty -> v = 0 */
  memset(s->M, 0, sizeof(double) * nd * nd);
  for (int i = n - 1; i >= 0; --i) {
    const tds_link_t *l = &m->links[i];
    lstate_t *L = &s->L[i];
    if (l->parent >= 0 || m->is_floating) {
      abi_t dI;
      abi_congruence(&L->X_parent, &L->abi, &dI); /* :45-46 */
      abi_t *P = l->parent >= 0 ? &s->L[l->parent].abi : &s->base_abi; /* :49-53 */
      for (int k = 0; k < 9; ++k) { P->I[k] += dI.I[k]; P->H[k] += dI.H[k]; P->M[k] += dI.M[k]; }
    }
    if (l->joint_type == TDS_JOINT_FIXED) continue; /* :56 */
    const int qd_i = l->qd_index;
    const int nci = l->joint_type == TDS_JOINT_SPHERICAL ? 3 : 1; /* :58-85 spherical: the same, column by column */
    for (int ci = 0; ci < nci; ++ci) {
      sv_t Si = L->S, Fi;
      if (nci == 3) { memset(&Si, 0, sizeof(Si)); Si.a[ci] = 1.0; }
      abi_mul(&L->abi, &Si, &Fi);                /* :87 / :59 */
      if (nci == 3) {
        for (int c2 = 0; c2 < 3; ++c2) s->M[(qd_i + ci) * nd + qd_i + c2] = Fi.a[c2]; /* S_3d^T Fi */
      } else {
        s->M[qd_i * nd + qd_i] = sv_dot(&L->S, &Fi); /* :89 */
      }
      int j = i;
      while (m->links[j].parent != -1) {         /* :92-109 / :63-79 */
        xf_apply_force(&s->L[j].X_parent, &Fi, &Fi);
        j = m->links[j].parent;
        if (m->links[j].joint_type == TDS_JOINT_FIXED) continue;
        int qd_j = m->links[j].qd_index;
        if (m->links[j].joint_type == TDS_JOINT_SPHERICAL) {
          for (int c2 = 0; c2 < 3; ++c2) {
            s->M[(qd_i + ci) * nd + qd_j + c2] = Fi.a[c2];
            s->M[(qd_j + c2) * nd + qd_i + ci] = Fi.a[c2];
          }
        } else {
          double h = sv_dot(&Fi, &s->L[j].S);
          s->M[(qd_i + ci) * nd + qd_j] = h;
          s->M[qd_j * nd + qd_i + ci] = h;
        }
      }
      if (m->is_floating) { /* :111-115  force carried into the base frame -> column / row of the base block */
        xf_apply_force(&s->L[j].X_parent, &Fi, &Fi);
        for (int k = 0; k < 3; ++k) {
          s->M[k * nd + qd_i + ci] = s->M[(qd_i + ci) * nd + k] = Fi.a[k];
          s->M[(3 + k) * nd + qd_i + ci] = s->M[(qd_i + ci) * nd + 3 + k] = Fi.l[k];
        }
      }
    }
  }
  if (m->is_floating) { /* :118-125  composite inertia of the base: [I H; H^T M] */
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        s->M[r * nd + c] = s->base_abi.I[3 * r + c];
        s->M[r * nd + 3 + c] = s->base_abi.H[3 * r + c];
        s->M[(3 + r) * nd + c] = s->base_abi.H[3 * c + r];
        s->M[(3 + r) * nd + 3 + c] = s->base_abi.M[3 * r + c];
      }
  }
}

/* ref: src/math/tiny/tiny_matrix_x.h:240-345  A^-1 = L^-T L^-1 through Cholesky.
   returns 0 if not positive definite */
static int symmetric_inverse(const double *A, double *a, int n) {
  double diag[ND];
  /* a[i][j] in the reference is column-major; the matrix is symmetric so we use row-major */
  for (int i = 0; i < n * n; ++i) a[i] = A[i];
  for (int i = 0; i < n; i++) {
    for (int j = i; j < n; j++) {
      double sum = a[i * n + j];
      for (int k = i - 1; k >= 0; k--) sum -= a[i * n + k] * a[j * n + k];
      if (i == j) {
        if (sum <= 0.0) return 0;
        diag[i] = sqrt(sum);
      } else {
        a[j * n + i] = sum / diag[i];
      }
    }
  }
  for (int i = 0; i < n; i++) {
    a[i * n + i] = 1.0 / diag[i];
    for (int j = i + 1; j < n; j++) {
      double sum = 0.0;
      for (int k = i; k < j; k++) sum -= a[j * n + k] * a[k * n + i];
      a[j * n + i] = sum / diag[j];
    }
  }
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++) a[i * n + j] = 0.0;
  for (int i = 0; i < n; i++) {
    a[i * n + i] = a[i * n + i] * a[i * n + i];
    for (int k = i + 1; k < n; k++) a[i * n + i] += a[k * n + i] * a[k * n + i];
    for (int j = i + 1; j < n; j++)
      for (int k = j; k < n; k++) a[i * n + j] += a[k * n + i] * a[k * n + j];
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++) a[i * n + j] = a[j * n + i];
  return 1;
}

/* ref: src/dynamics/jacobian.hpp:13-83 (fixed base, world point).  forward_kinematics_q
   (kinematics.hpp:168-236) recomputes the same X_world the links already hold for this q. */
static void point_jacobian(const tds_model_t *m, const scratch_t *s, int link_index,
                           const double *point, double *jac /* 3 x nd */) {
  const int nd = m->dof_qd;
  memset(jac, 0, sizeof(double) * 3 * nd);
  if (m->is_floating) { /* :39-56  base block [ -[r]x | 1 ] with r = point - base position (WORLD axes) */
    double r[3] = {point[0] - s->base_X_world.t[0], point[1] - s->base_X_world.t[1], point[2] - s->base_X_world.t[2]};
    double cr[9];
    m3_cross_matrix(r, cr);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) jac[a * nd + b] = cr[3 * b + a]; /* transpose(cross_matrix(r)) */
    jac[0 * nd + 3] = jac[1 * nd + 4] = jac[2 * nd + 5] = 1.0;
  }
  int i = link_index;
  while (i >= 0) {
    const tds_link_t *l = &m->links[i];
    if (l->joint_type == TDS_JOINT_SPHERICAL) { /* :65-69: the three columns of S_3d like three revolute axes */
      for (int c = 0; c < 3; ++c) {
        sv_t e = {{0, 0, 0}, {0, 0, 0}}, st;
        e.a[c] = 1.0;
        xf_apply_inverse_motion(&s->L[i].X_world, &e, &st);
        double rxw[3];
        v3_cross(point, st.a, rxw);
        for (int r = 0; r < 3; ++r) jac[r * nd + l->qd_index + c] = st.l[r] - rxw[r];
      }
    } else if (l->joint_type != TDS_JOINT_FIXED) {
      sv_t st;
      xf_apply_inverse_motion(&s->L[i].X_world, &s->L[i].S, &st); /* :74 */
      /* point_tf.apply(st): rotation = I, translation = point (transform.hpp:210-226) */
      double rxw[3];
      v3_cross(point, st.a, rxw);
      for (int r = 0; r < 3; ++r) jac[r * nd + l->qd_index] = st.l[r] - rxw[r]; /* :76 */
    }
    i = l->parent;
  }
}

/* ref: src/contact_point.hpp:96-125 */
static void contact_plane_sphere(const tds_model_t *m, const double *pos, double radius, int link,
                                 scratch_t *s) {
  const double *n = m->plane_normal;
  contact_t *c = &s->cps[s->n_c++];
  double mn[3] = {-n[0], -n[1], -n[2]};
  double t = -(v3_dot(pos, mn) + m->plane_constant);
  for (int k = 0; k < 3; ++k) {
    c->point_a[k] = pos[k] + t * mn[k];
    c->point_b[k] = pos[k] - radius * n[k];
    c->normal[k] = mn[k];
  }
  c->distance = t - radius;
  c->link_b = link;
}

/* ref: src/world.hpp:206-282 with mb_a = plane (one geom on its base), mb_b = robot;
   dispatch through contact_point.hpp:444-496 (plane-sphere / plane-capsule / plane-box) */
static void compute_contacts(const tds_model_t *m, scratch_t *s) {
  s->n_c = 0;
  if (!m->has_plane) return;
  for (int g = 0; g < m->num_geoms; ++g) {
    const tds_geom_t *G = &m->geoms[g];
    xf_t local, tr;
    memcpy(local.r, G->X_rot, sizeof(local.r));
    memcpy(local.t, G->X_trans, sizeof(local.t));
    const xf_t *Xw = G->link >= 0 ? &s->L[G->link].X_world : &s->base_X_world;
    xf_mul(Xw, &local, &tr);                 /* world.hpp:242 */
    double orn[4];
    matrix_to_quat(tr.r, orn);               /* world.hpp:244-245 */
    quat_normalize(orn);
    if (G->type == TDS_GEOM_SPHERE) {
      contact_plane_sphere(m, tr.t, G->radius, G->link, s);
    } else if (G->type == TDS_GEOM_CAPSULE) { /* contact_point.hpp:127-161 */
      for (int e = 0; e < 2; ++e) {
        double off[3] = {0.0, 0.0, (e == 0 ? 0.5 : -0.5) * G->length}, ro[3], p[3];
        quat_rotate(orn, off, ro);           /* pose.hpp:47-53 */
        for (int k = 0; k < 3; ++k) p[k] = tr.t[k] + ro[k];
        contact_plane_sphere(m, p, G->radius, G->link, s);
      }
    } else if (G->type == TDS_GEOM_BOX) {    /* contact_point.hpp:163-198, geometry.hpp:244-259 */
      double cr = G->radius > 1e-2 ? G->radius : 1e-2;
      double dx = G->extents[0] * 0.5 - cr, dy = G->extents[1] * 0.5 - cr, dz = G->extents[2] * 0.5 - cr;
      for (int c = 0; c < 8; ++c) {
        double off[3] = {(c & 4) ? -dx : dx, (c & 2) ? -dy : dy, (c & 1) ? -dz : dz}, ro[3], p[3];
        quat_rotate(orn, off, ro);
        for (int k = 0; k < 3; ++k) p[k] = tr.t[k] + ro[k];
        contact_plane_sphere(m, p, cr, G->link, s);
      }
    }
  }
}

/* ref: src/mb_constraint_solver.hpp:506-520 (branch-free form incl. its quirks) */
static void plane_space(const double *n, double *p, double *q) {
  double n_sqr = n[2] * n[2];
  int gt = n_sqr > 0.5;
  double a = n[1] * n[1] + (gt ? n_sqr : n[0] * n[0]);
  double k = sqrt(a);
  p[0] = gt ? 0.0 : -n[1] * k;
  p[1] = gt ? -n[2] * k : n[0] * k;
  p[2] = n[1] * k;
  q[0] = gt ? a * k : -n[2] * p[1];
  q[1] = gt ? -n[0] * p[2] : n[2] * p[0];
  q[2] = gt ? n[0] * p[1] : a * k;
}

/* ref: src/mb_constraint_solver.hpp:101-142 */
static void solve_pgs(const double *A, const double *b, double *x, int n, int iters, const double *lo,
                      const double *hi, const int *dep) {
  for (int k = 0; k < iters; ++k) {
    for (int i = 0; i < n; ++i) {
      double delta = 0.0;
      for (int j = 0; j < i; j++) delta += A[i * n + j] * x[j];
      for (int j = i + 1; j < n; j++) delta += A[i * n + j] * x[j];
      x[i] = (b[i] - delta) / A[i * n + i];
      double sc = 1.0;
      if (dep[i] >= 0) {
        sc = x[dep[i]];
        if (sc < 0.0) sc = 0.0;
      }
      if (x[i] < lo[i] * sc) x[i] = lo[i] * sc; /* Algebra::max */
      if (x[i] > hi[i] * sc) x[i] = hi[i] * sc; /* Algebra::min */
    }
  }
}

/* ref: src/mb_constraint_solver.hpp:191-498 with mb_a = plane (n_a = 0), mb_b = robot,
   keep_all_points_ = true (locomotion_contact_simulation.h:135) */
static int resolve_collision(const tds_model_t *m, scratch_t *s, tds_oracle_debug_t *dbg) {
  const int n_c = s->n_c, nd = m->dof_qd, nr = 3 * n_c;
  if (n_c == 0 || nd == 0) return 0;
  mass_matrix(m, s);                                   /* :232-233 */
  if (!symmetric_inverse(s->M, s->Minv, nd)) return -1; /* :245-246 */
  memset(s->J, 0, sizeof(double) * nr * nd);
  memset(s->b, 0, sizeof(double) * nr);
  for (int i = 0; i < n_c; ++i) {
    const contact_t *cp = &s->cps[i];
    double collision = cp->distance < 0.0 ? 1.0 : 0.0; /* :285 */
    point_jacobian(m, s, cp->link_b, cp->point_b, s->jac); /* :295 */
    if (dbg && dbg->jac) memcpy(dbg->jac + (size_t)i * 3 * nd, s->jac, sizeof(double) * 3 * nd);
    double nrm[3] = {cp->normal[0] * collision, cp->normal[1] * collision, cp->normal[2] * collision};
    for (int d = 0; d < nd; ++d) /* :300-307  jac_b^T (n*collision) */
      s->J[i * nd + d] = s->jac[d] * nrm[0] + s->jac[nd + d] * nrm[1] + s->jac[2 * nd + d] * nrm[2];
    double vel_b[3] = {0, 0, 0};                       /* :314 */
    for (int r = 0; r < 3; ++r)
      for (int d = 0; d < nd; ++d) vel_b[r] += s->jac[r * nd + d] * s->qd[d];
    double rel_vel[3] = {-vel_b[0], -vel_b[1], -vel_b[2]}; /* :315 (vel_a = 0) */
    double normal_rel_vel = v3_dot(cp->normal, rel_vel);
    double baumgarte = m->erp * cp->distance / m->dt;  /* :321 */
    s->b[i] = (-(1.0 + m->restitution) * normal_rel_vel - baumgarte) * collision; /* :323-325 */
    double f1[3], f2[3];
    plane_space(cp->normal, f1, f2);                   /* :361 */
    for (int k = 0; k < 3; ++k) { f1[k] *= collision; f2[k] *= collision; }
    s->b[n_c + i] = -v3_dot(f1, rel_vel);              /* :365-366 */
    s->b[2 * n_c + i] = -v3_dot(f2, rel_vel);          /* :369-370 */
    for (int d = 0; d < nd; ++d) {                     /* :378-384 */
      s->J[(n_c + i) * nd + d] = s->jac[d] * f1[0] + s->jac[nd + d] * f1[1] + s->jac[2 * nd + d] * f1[2];
      s->J[(2 * n_c + i) * nd + d] = s->jac[d] * f2[0] + s->jac[nd + d] * f2[1] + s->jac[2 * nd + d] * f2[2];
    }
  }
  /* :397  lcp_A = jac_con * mass_matrix_inv * jac_con_t  (tiny_matrix_x.h:127-141 i-j-k) */
  for (int i = 0; i < nr; ++i)
    for (int j = 0; j < nd; ++j) {
      double sum = 0.0;
      for (int k = 0; k < nd; ++k) sum += s->J[i * nd + k] * s->Minv[k * nd + j];
      s->JM[i * nd + j] = sum;
    }
  for (int i = 0; i < nr; ++i)
    for (int j = 0; j < nr; ++j) {
      double sum = 0.0;
      for (int k = 0; k < nd; ++k) sum += s->JM[i * nd + k] * s->J[j * nd + k];
      s->A[i * nr + j] = sum;
    }
  for (int i = 0; i < nr; ++i) s->A[i * nr + i] += m->cfm; /* :408-410 */
  for (int i = 0; i < n_c; ++i) {                          /* :424-436 */
    s->dep[i] = -1; s->lo[i] = 0.0; s->hi[i] = 100000.0;
    s->lo[n_c + i] = -m->friction; s->hi[n_c + i] = m->friction; s->dep[n_c + i] = i;
    s->lo[2 * n_c + i] = -m->friction; s->hi[2 * n_c + i] = m->friction; s->dep[2 * n_c + i] = i;
  }
  memset(s->p, 0, sizeof(double) * nr);
  solve_pgs(s->A, s->b, s->p, nr, m->pgs_iterations, s->lo, s->hi, s->dep); /* :440 */
  /* :476-496  qd_b -= M_b^-1 J^T p  (the three blocks summed) */
  double JtP[ND];
  for (int d = 0; d < nd; ++d) {
    double sum = 0.0;
    for (int i = 0; i < nr; ++i) sum += s->J[i * nd + d] * s->p[i];
    JtP[d] = sum;
  }
  for (int d = 0; d < nd; ++d) {
    double sum = 0.0;
    for (int k = 0; k < nd; ++k) sum += s->Minv[d * nd + k] * JtP[k];
    s->qd[d] -= sum;
  }
  if (dbg) {
    if (dbg->M) memcpy(dbg->M, s->M, sizeof(double) * nd * nd);
    if (dbg->Minv) memcpy(dbg->Minv, s->Minv, sizeof(double) * nd * nd);
    if (dbg->lcp_A) memcpy(dbg->lcp_A, s->A, sizeof(double) * nr * nr);
    if (dbg->lcp_b) memcpy(dbg->lcp_b, s->b, sizeof(double) * nr);
    if (dbg->lcp_p) memcpy(dbg->lcp_p, s->p, sizeof(double) * nr);
  }
  return 0;
}

/* integrate_euler_qdd: qd += qdd*dt (integrator.hpp:141-182); qdd = 0 (:194) */
static void integrate_euler_qdd(const tds_model_t *m, scratch_t *s) {
  if (m->is_floating) /* :152-166 */
    for (int k = 0; k < 6; ++k) s->qd[k] += s->qdd[k] * m->dt;
  for (int i = 0; i < m->num_links; ++i) {
    const tds_link_t *l = &m->links[i];
    const int nc = l->joint_type == TDS_JOINT_SPHERICAL ? 3 : 1; /* :171-174 */
    if (l->joint_type != TDS_JOINT_FIXED)
      for (int c = 0; c < nc; ++c) s->qd[l->qd_index + c] += s->qdd[l->qd_index + c] * m->dt;
  }
}

static void integrate_euler(const tds_model_t *m, scratch_t *s) {
  /* integrate_euler with qdd == 0: q += qd*dt (integrator.hpp:126-131) */
  if (m->is_floating) { /* :23-89: quaternion += quat_velocity(q, omega, dt) (tiny_algebra.hpp:604-614), normalise */
    const double *w = s->qd, h = 0.5 * m->dt;
    double *b = s->q;
    double ww = (-b[0] * w[0] - b[1] * w[1] - b[2] * w[2]) * h;
    double xx = (b[3] * w[0] + b[2] * w[1] - b[1] * w[2]) * h;
    double yy = (b[3] * w[1] + b[0] * w[2] - b[2] * w[0]) * h;
    double zz = (b[3] * w[2] + b[1] * w[0] - b[0] * w[1]) * h;
    b[0] += xx; b[1] += yy; b[2] += zz; b[3] += ww;
    quat_normalize(b);
    quat_to_matrix(b, s->base_X_world.r); /* :83 (the translation of base_X_world is NOT refreshed) */
    for (int k = 0; k < 3; ++k) b[4 + k] += s->qd[3 + k] * m->dt;
  }
  for (int i = 0; i < m->num_links; ++i) {
    const tds_link_t *l = &m->links[i];
    if (l->joint_type == TDS_JOINT_SPHERICAL) { /* :94-123 */
      double *w = &s->qd[l->qd_index], *b = &s->q[l->q_index];
      const double damping = pow(0.995, m->dt * 1000.0); /* MultiBody::joint_damping_ = 0.995 (multi_body.hpp:51) */
      for (int c = 0; c < 3; ++c) w[c] *= damping;
      const double h = 0.5 * m->dt; /* quat_velocity_spherical, tiny_algebra.hpp:618-629 */
      double ww = (-b[0] * w[0] - b[1] * w[1] - b[2] * w[2]) * h;
      double xx = (b[3] * w[0] + b[1] * w[2] - b[2] * w[1]) * h;
      double yy = (b[3] * w[1] + b[2] * w[0] - b[0] * w[2]) * h;
      double zz = (b[3] * w[2] + b[0] * w[1] - b[1] * w[0]) * h;
      b[0] += xx; b[1] += yy; b[2] += zz; b[3] += ww;
      quat_normalize(b);
    } else if (l->joint_type != TDS_JOINT_FIXED) {
      s->q[l->q_index] += s->qd[l->qd_index] * m->dt;
    }
  }
}

static int step_multi(const tds_model_t *m, const double *x, double *y);

/* ref: examples/environments/locomotion_contact_simulation.h:151-304 (LOCOMOTION) and
   examples/environments/cartpole_environment.h:71-117 (TAU) */
static int step_one(const tds_model_t *m, const double *x, double *y, scratch_t *s,
                    tds_oracle_debug_t *dbg) {
  if (m->num_bodies >= 2) return step_multi(m, x, y); /* worlds with several articulated bodies: below */
  const int nq = m->dof_q, nd = m->dof_qd;
  int nsph = 0;
  for (int i = 0; i < m->num_links && i < NL; ++i) nsph += m->links[i].joint_type == TDS_JOINT_SPHERICAL;
  if (m->num_links > NL || nd > ND || nq != nd + (m->is_floating ? 1 : 0) + nsph) return -2;
  /* floating base + spherical joints: the reference writes the base/joint block of M only one way round
     (mass_matrix.hpp:80-84) — not restated */
  if (m->is_floating && nsph) return -2;
  if (m->has_plane) {
    int nc = 0;
    for (int g = 0; g < m->num_geoms; ++g)
      nc += m->geoms[g].type == TDS_GEOM_SPHERE ? 1 : m->geoms[g].type == TDS_GEOM_CAPSULE ? 2
            : m->geoms[g].type == TDS_GEOM_BOX ? 8 : 0;
    if (nc > NCMAX) return -3;
  }
  /* mb_->initialize(): zero state (multi_body.hpp:324-378) */
  memset(s->q, 0, sizeof(s->q)); memset(s->qd, 0, sizeof(s->qd));
  memset(s->qdd, 0, sizeof(s->qdd)); memset(s->tau, 0, sizeof(s->tau));
  memcpy(s->base_X_world.r, m->base_X_world_rot, sizeof(s->base_X_world.r));
  memcpy(s->base_X_world.t, m->base_X_world_trans, sizeof(s->base_X_world.t));
  for (int i = 0; i < m->num_links; ++i)
    for (int k = 0; k < 3; ++k) { s->L[i].S.a[k] = m->links[i].S[k]; s->L[i].S.l[k] = m->links[i].S[3 + k]; }
  for (int i = 0; i < nq; ++i) s->q[i] = x[i];            /* :154-159 */
  for (int i = 0; i < nd; ++i) s->qd[i] = x[nq + i];
  if (m->step_mode == TDS_STEP_LOCOMOTION) {
    const int action_offset = nq + nd, var = nq + nd + m->action_dim;
    const double kp = x[var], kd = x[var + 1], max_force = x[var + 2]; /* :164-166 */
    int pose_index = 0;
    for (int i = m->pd_start_link; i < m->num_links; ++i) { /* :181-257 */
      const tds_link_t *l = &m->links[i];
      if (l->joint_type == TDS_JOINT_FIXED) continue;
      if (l->joint_type == TDS_JOINT_SPHERICAL) {          /* :188-226 */
        /* q_desired = identity, qd_desired = 0; position_error = get_axis_difference_quaternion(q_desired,
           q_actual) = matrix_to_euler_xyz(quat_to_matrix(inverse(q_desired) * q_actual))
           (matrix_utils.hpp:77-90; inverse = (-x,-y,-z,w), tiny_quaternion.h:80-82, so the product with
           the identity is q_actual itself) */
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pe[3];
        quat_to_matrix(s->q + l->q_index, R);
        /* matrix_to_euler_xyz (matrix_utils.hpp:18-50); get_matrix_elem(mat, k) = mat(k % 3, k / 3) (:11-16) */
        const double fi = R[6], e0 = R[0], e1 = R[3], e3 = R[1], e4 = R[4], e5 = R[7], e8 = R[8];
        const double half_pi = 3.14159265358979323846 / 2.; /* M_PI / 2., tiny_double_utils.h:40 */
        pe[0] = fi <= 1.0 ? (fi >= -1.0 ? atan2(-e5, e8) : -atan2(e3, e4)) : atan2(e3, e4);
        pe[1] = fi <= 1.0 ? (fi >= -1.0 ? asin(fi) : -half_pi) : half_pi;
        pe[2] = fi <= 1.0 ? (fi >= -1.0 ? atan2(-e1, e0) : 0.0) : 0.0;
        for (int k = 0; k < 3; ++k) {
          double f = kp * pe[k] + kd * (0.0 - s->qd[l->qd_index + k]); /* :207-208 */
          if (f < -max_force) f = -max_force;              /* :209-213: min(max(f, -max), max) */
          if (f > max_force) f = max_force;
          /* :215-221: the torque is only stored for link indices >= 4 (or on a floating base) */
          if (m->is_floating || i >= 4) s->tau[l->qd_index + k] = f;
        }
        pose_index += 4;                                   /* :223 */
        continue;
      }
      double a = x[action_offset + pose_index];
      if (a > m->action_limit) a = m->action_limit;        /* :235-236 */
      if (a < -m->action_limit) a = -m->action_limit;
      double q_des = m->initial_poses[pose_index++] + a;   /* :238 */
      double f = kp * (q_des - s->q[l->q_index]) + kd * (0.0 - s->qd[l->qd_index]); /* :242-245 */
      if (f < -max_force) f = -max_force;                  /* :247 */
      if (f > max_force) f = max_force;
      s->tau[l->qd_index] = f;
    }
  } else {
    /* tau has dof_actuated entries; get_tau_for_link reads tau[qd_index - 6] on a floating base
       (multi_body.hpp:557-570): keep it indexed by qd_index here */
    const int off = m->is_floating ? 6 : 0;
    for (int i = 0; i < nd - off; ++i) s->tau[off + i] = x[nq + nd + i];
  }
  forward_dynamics(m, s);                                  /* :261 */
  if (dbg && dbg->qdd) memcpy(dbg->qdd, s->qdd, sizeof(double) * nd);
  if (dbg && dbg->X_world)
    for (int i = 0; i < m->num_links; ++i) {
      memcpy(dbg->X_world + 12 * i, s->L[i].X_world.r, 9 * sizeof(double));
      memcpy(dbg->X_world + 12 * i + 9, s->L[i].X_world.t, 3 * sizeof(double));
    }
  integrate_euler_qdd(m, s);
  if (m->has_plane) {                                      /* world.step (world.hpp:293-366) */
    compute_contacts(m, s);
    if (dbg) {
      dbg->n_c = s->n_c;
      if (dbg->contacts)
        for (int i = 0; i < s->n_c; ++i) {
          double *c = dbg->contacts + 10 * i;
          memcpy(c, s->cps[i].normal, 24); memcpy(c + 3, s->cps[i].point_b, 24);
          memcpy(c + 6, s->cps[i].point_a, 24); c[9] = s->cps[i].distance;
        }
    }
    int rc = resolve_collision(m, s, dbg);
    if (rc) return rc;
  }
  integrate_euler(m, s);
  /* pack (:273-303); the rest of y is zero (caller's vector is zero-initialised) */
  int j = 0;
  for (int i = 0; i < m->output_dim; ++i) y[i] = 0.0;
  for (int i = 0; i < nq; ++i) y[j++] = s->q[i];
  for (int i = 0; i < nd; ++i) y[j++] = s->qd[i];
  if (m->pack_visuals) {
    for (int v = 0; v < m->num_visuals; ++v) {
      const tds_visual_t *V = &m->visuals[v];
      xf_t lv, vx;
      double orn[4];
      memcpy(lv.r, V->X_rot, sizeof(lv.r)); memcpy(lv.t, V->X_trans, sizeof(lv.t));
      xf_mul(&s->L[V->link].X_world, &lv, &vx);            /* :285 (X_world is pre-step) */
      y[j++] = vx.t[0]; y[j++] = vx.t[1]; y[j++] = vx.t[2];
      matrix_to_quat(vx.r, orn);                           /* :291 */
      y[j++] = orn[0]; y[j++] = orn[1]; y[j++] = orn[2]; y[j++] = orn[3];
    }
    y[j++] = s->base_X_world.r[8];                         /* :301-303 */
  }
  return 0;
}

/* ================================================================================================
 * Worlds with SEVERAL articulated bodies (tds_model_t::num_bodies in 2..TDS_MAX_BODIES).
 * ref: src/world.hpp:293-366 (World::step over multi_bodies_ = [plane,] body 0, 1, ...: contacts pair by pair, then
 * resolve_collision pair by pair in the same order), :206-282 (pair loop over all i < j), src/contact_point.hpp:43-94
 * (sphere-sphere), :405-438 (capsule-sphere), :478-495 (the dispatcher's swapped order),
 * src/mb_constraint_solver.hpp:191-498 (both Jacobian blocks, both inverse mass matrices).
 * Each body is handled by the single-body functions above on a sub-model of its own.
 * ================================================================================================ */
typedef struct {
  double normal[3], point_a[3], point_b[3], distance;
  int link_a, link_b;
} pair_contact_t;

/* body `which` of a multi-body blob as a single-body blob of its own */
static void sub_model(const tds_model_t *m, int which, tds_model_t *o) {
  const int B = m->num_bodies;
  const int f = which == 0 ? 0 : m->bodies[which].first_link;
  const int fe = which + 1 < B ? m->bodies[which + 1].first_link : m->num_links;
  const int g0 = which == 0 ? 0 : m->bodies[which].first_geom;
  const int ge = which + 1 < B ? m->bodies[which + 1].first_geom : m->num_geoms;
  /* first q / qd index of the body: everything the earlier bodies own (floating: 7 / 6 base coordinates,
     multi_body.hpp:324-349) */
  int q0 = 0, d0 = 0;
  for (int b = 0; b < which; ++b) {
    const int lb = b == 0 ? 0 : m->bodies[b].first_link, le = m->bodies[b + 1].first_link;
    if (b == 0 ? m->is_floating : m->bodies[b].is_floating) { q0 += 7; d0 += 6; }
    for (int i = lb; i < le; ++i)
      if (m->links[i].joint_type == TDS_JOINT_SPHERICAL) { q0 += 4; d0 += 3; }
      else if (m->links[i].joint_type != TDS_JOINT_FIXED) { ++q0; ++d0; }
  }
  *o = *m;
  o->num_bodies = 0;
  o->pack_visuals = 0;
  o->num_visuals = 0;
  if (which > 0) {
    const tds_body_t *bd = &m->bodies[which];
    o->is_floating = bd->is_floating;
    memcpy(o->base_X_world_rot, bd->base_X_world_rot, sizeof(o->base_X_world_rot));
    memcpy(o->base_X_world_trans, bd->base_X_world_trans, sizeof(o->base_X_world_trans));
    o->base_mass = bd->base_mass;
    memcpy(o->base_com, bd->base_com, sizeof(o->base_com));
    memcpy(o->base_inertia, bd->base_inertia, sizeof(o->base_inertia));
  }
  o->num_links = fe - f;
  int nq = o->is_floating ? 7 : 0, nd = o->is_floating ? 6 : 0;
  for (int i = 0; i < o->num_links; ++i) {
    o->links[i] = m->links[f + i];
    if (o->links[i].parent >= 0) o->links[i].parent -= f;
    if (o->links[i].q_index >= 0) o->links[i].q_index -= q0;
    if (o->links[i].qd_index >= 0) o->links[i].qd_index -= d0;
    if (o->links[i].joint_type == TDS_JOINT_SPHERICAL) { nq += 4; nd += 3; }
    else if (o->links[i].joint_type != TDS_JOINT_FIXED) { ++nq; ++nd; }
  }
  o->dof_q = nq;
  o->dof_qd = nd;
  o->num_geoms = ge - g0;
  for (int g = 0; g < o->num_geoms; ++g) {
    o->geoms[g] = m->geoms[g0 + g];
    o->geoms[g].link = o->geoms[g].link < 0 ? -1 : o->geoms[g].link - f;
  }
  o->action_dim = nd - (o->is_floating ? 6 : 0); /* dof_actuated */
}

/* ref: contact_point.hpp:43-94 (non-CppAD branch): at most one contact, emitted when the centres are apart */
static int pair_sphere_sphere(const double *pa, double ra, const double *pb, double rb, pair_contact_t *c) {
  double diff[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
  double length = sqrt(v3_dot(diff, diff));
  double distance = length - (ra + rb);
  if (!(length > 1e-5)) return 0; /* CONTACT_EPSILON */
  for (int k = 0; k < 3; ++k) {
    c->normal[k] = 1.0 / length * diff[k];
  }
  for (int k = 0; k < 3; ++k) {
    c->point_a[k] = pa[k] - ra * c->normal[k];
    c->point_b[k] = c->point_a[k] - distance * c->normal[k];
  }
  c->distance = distance;
  return 1;
}

/* world pose of a geometry: position + normalised quaternion (world.hpp:238-245) */
static void geom_pose(const tds_model_t *mb, const scratch_t *s, const tds_geom_t *G, double *pos, double *orn) {
  xf_t local, tr;
  memcpy(local.r, G->X_rot, sizeof(local.r));
  memcpy(local.t, G->X_trans, sizeof(local.t));
  const xf_t *Xw = G->link >= 0 ? &s->L[G->link].X_world : &s->base_X_world;
  (void)mb;
  xf_mul(Xw, &local, &tr);
  matrix_to_quat(tr.r, orn);
  quat_normalize(orn);
  memcpy(pos, tr.t, 3 * sizeof(double));
}

/* ref: world.hpp:206-282 for the pair (A, B) */
static int compute_pair_contacts(const tds_model_t *ma, const scratch_t *sa, const tds_model_t *mb, const scratch_t *sb,
                                 pair_contact_t *out, int cap) {
  int n = 0;
  for (int ga = 0; ga < ma->num_geoms; ++ga) {
    const tds_geom_t *A = &ma->geoms[ga];
    double pa[3], qa[4];
    geom_pose(ma, sa, A, pa, qa);
    for (int gb = 0; gb < mb->num_geoms; ++gb) {
      const tds_geom_t *B = &mb->geoms[gb];
      double pb[3], qb[4];
      geom_pose(mb, sb, B, pb, qb);
      pair_contact_t c[2];
      int nc = 0;
      if (A->type == TDS_GEOM_SPHERE && B->type == TDS_GEOM_SPHERE) {
        nc = pair_sphere_sphere(pa, A->radius, pb, B->radius, &c[0]);
      } else if (A->type == TDS_GEOM_CAPSULE && B->type == TDS_GEOM_SPHERE) { /* contact_point.hpp:405-438 */
        for (int e = 0; e < 2; ++e) {
          double off[3] = {0.0, 0.0, (e == 0 ? 0.5 : -0.5) * A->length}, ro[3], pe[3];
          quat_rotate(qa, off, ro);
          for (int k = 0; k < 3; ++k) pe[k] = pa[k] + ro[k];
          nc += pair_sphere_sphere(pe, A->radius, pb, B->radius, &c[nc]);
        }
      } else if (A->type == TDS_GEOM_SPHERE && B->type == TDS_GEOM_CAPSULE) { /* :478-495: run swapped, then swap back */
        for (int e = 0; e < 2; ++e) {
          double off[3] = {0.0, 0.0, (e == 0 ? 0.5 : -0.5) * B->length}, ro[3], pe[3];
          quat_rotate(qb, off, ro);
          for (int k = 0; k < 3; ++k) pe[k] = pb[k] + ro[k];
          pair_contact_t t;
          if (pair_sphere_sphere(pe, B->radius, pa, A->radius, &t)) {
            for (int k = 0; k < 3; ++k) {
              c[nc].point_a[k] = t.point_b[k];
              c[nc].point_b[k] = t.point_a[k];
              c[nc].normal[k] = -t.normal[k];
            }
            c[nc].distance = t.distance;
            ++nc;
          }
        }
      }
      for (int i = 0; i < nc; ++i) {
        if (n >= cap) return -1;
        c[i].link_a = A->link;
        c[i].link_b = B->link;
        out[n++] = c[i];
      }
    }
  }
  return n;
}

/* ref: mb_constraint_solver.hpp:191-498 with two articulated bodies: rows [J_a | J_b], A = J diag(M_a^-1, M_b^-1) J^T,
   right-hand side from rel_vel = J_a qd_a - J_b qd_b, qd_a += M_a^-1 J_a^T p, qd_b -= M_b^-1 J_b^T p */
static int resolve_collision_pair(const tds_model_t *m, const tds_model_t *ma, scratch_t *sa, const tds_model_t *mb,
                                  scratch_t *sb, const pair_contact_t *cps, int n_c) {
  const int na = ma->dof_qd, nb = mb->dof_qd, nab = na + nb, nr = 3 * n_c;
  if (n_c == 0 || nab == 0) return 0;
  if (nr > NR || nab > ND) return -3;
  mass_matrix(ma, sa);
  if (!symmetric_inverse(sa->M, sa->Minv, na)) return -1;
  mass_matrix(mb, sb);
  if (!symmetric_inverse(sb->M, sb->Minv, nb)) return -1;
  static _Thread_local double J[NR * ND], JM[NR * ND], A[NR * NR], b[NR], p[NR], lo[NR], hi[NR];
  static _Thread_local int dep[NR];
  memset(J, 0, sizeof(double) * nr * nab);
  memset(b, 0, sizeof(double) * nr);
  for (int i = 0; i < n_c; ++i) {
    const pair_contact_t *cp = &cps[i];
    const double collision = cp->distance < 0.0 ? 1.0 : 0.0;
    double ja[3 * ND], jb[3 * ND];
    point_jacobian(ma, sa, cp->link_a, cp->point_a, ja);
    point_jacobian(mb, sb, cp->link_b, cp->point_b, jb);
    double vel_a[3] = {0, 0, 0}, vel_b[3] = {0, 0, 0}, rel_vel[3];
    for (int r = 0; r < 3; ++r) {
      for (int d = 0; d < na; ++d) vel_a[r] += ja[r * na + d] * sa->qd[d];
      for (int d = 0; d < nb; ++d) vel_b[r] += jb[r * nb + d] * sb->qd[d];
      rel_vel[r] = vel_a[r] - vel_b[r];
    }
    double dir[3][3], f1[3], f2[3];
    plane_space(cp->normal, f1, f2);
    for (int k = 0; k < 3; ++k) {
      dir[0][k] = cp->normal[k] * collision;
      dir[1][k] = f1[k] * collision;
      dir[2][k] = f2[k] * collision;
    }
    const double normal_rel_vel = v3_dot(cp->normal, rel_vel);
    b[i] = (-(1.0 + m->restitution) * normal_rel_vel - m->erp * cp->distance / m->dt) * collision;
    b[n_c + i] = -v3_dot(dir[1], rel_vel);
    b[2 * n_c + i] = -v3_dot(dir[2], rel_vel);
    for (int e = 0; e < 3; ++e) {
      double *row = J + (size_t)(e * n_c + i) * nab;
      for (int d = 0; d < na; ++d) row[d] = ja[d] * dir[e][0] + ja[na + d] * dir[e][1] + ja[2 * na + d] * dir[e][2];
      for (int d = 0; d < nb; ++d) row[na + d] = jb[d] * dir[e][0] + jb[nb + d] * dir[e][1] + jb[2 * nb + d] * dir[e][2];
    }
  }
  /* lcp_A = jac_con * mass_matrix_inv * jac_con_t with the block-diagonal inverse */
  for (int i = 0; i < nr; ++i)
    for (int j = 0; j < nab; ++j) {
      double sum = 0.0;
      if (j < na)
        for (int k = 0; k < na; ++k) sum += J[i * nab + k] * sa->Minv[k * na + j];
      else
        for (int k = 0; k < nb; ++k) sum += J[i * nab + na + k] * sb->Minv[k * nb + (j - na)];
      JM[i * nab + j] = sum;
    }
  for (int i = 0; i < nr; ++i)
    for (int j = 0; j < nr; ++j) {
      double sum = 0.0;
      for (int k = 0; k < nab; ++k) sum += JM[i * nab + k] * J[j * nab + k];
      A[i * nr + j] = sum;
    }
  for (int i = 0; i < nr; ++i) A[i * nr + i] += m->cfm;
  for (int i = 0; i < n_c; ++i) {
    dep[i] = -1; lo[i] = 0.0; hi[i] = 100000.0;
    lo[n_c + i] = -m->friction; hi[n_c + i] = m->friction; dep[n_c + i] = i;
    lo[2 * n_c + i] = -m->friction; hi[2 * n_c + i] = m->friction; dep[2 * n_c + i] = i;
  }
  memset(p, 0, sizeof(double) * nr);
  solve_pgs(A, b, p, nr, m->pgs_iterations, lo, hi, dep);
  double JtP[ND];
  for (int d = 0; d < nab; ++d) {
    double sum = 0.0;
    for (int i = 0; i < nr; ++i) sum += J[i * nab + d] * p[i];
    JtP[d] = sum;
  }
  for (int d = 0; d < na; ++d) {
    double sum = 0.0;
    for (int k = 0; k < na; ++k) sum += sa->Minv[d * na + k] * JtP[k];
    sa->qd[d] += sum;
  }
  for (int d = 0; d < nb; ++d) {
    double sum = 0.0;
    for (int k = 0; k < nb; ++k) sum += sb->Minv[d * nb + k] * JtP[na + k];
    sb->qd[d] -= sum;
  }
  return 0;
}

static int step_multi(const tds_model_t *m, const double *x, double *y) {
  const int B = m->num_bodies;
  if (m->step_mode != TDS_STEP_TAU || B > TDS_MAX_BODIES) return -2;
  tds_model_t *sub = (tds_model_t *)malloc(B * sizeof(tds_model_t));
  scratch_t *sc = (scratch_t *)malloc(B * sizeof(scratch_t));
  if (!sub || !sc) { free(sub); free(sc); return -4; }
  for (int b = 0; b < B; ++b) sub_model(m, b, &sub[b]);
  const int nq = m->dof_q, nd = m->dof_qd;
  int oq = 0, od = 0, ot = 0, rc = 0;
  for (int b = 0; b < B && !rc; ++b) {
    const tds_model_t *mm = &sub[b];
    scratch_t *s = &sc[b];
    if (mm->num_links > NL || mm->dof_qd > ND) { rc = -2; break; }
    memset(s->q, 0, sizeof(s->q)); memset(s->qd, 0, sizeof(s->qd));
    memset(s->qdd, 0, sizeof(s->qdd)); memset(s->tau, 0, sizeof(s->tau));
    memcpy(s->base_X_world.r, mm->base_X_world_rot, sizeof(s->base_X_world.r));
    memcpy(s->base_X_world.t, mm->base_X_world_trans, sizeof(s->base_X_world.t));
    for (int i = 0; i < mm->num_links; ++i) {
      if (mm->links[i].joint_type == TDS_JOINT_SPHERICAL) rc = -2;
      for (int k = 0; k < 3; ++k) { s->L[i].S.a[k] = mm->links[i].S[k]; s->L[i].S.l[k] = mm->links[i].S[3 + k]; }
    }
    for (int i = 0; i < mm->dof_q; ++i) s->q[i] = x[oq + i];
    for (int i = 0; i < mm->dof_qd; ++i) s->qd[i] = x[nq + od + i];
    /* tau has dof_actuated entries per body; kept indexed by qd_index (multi_body.hpp:557-570) */
    const int off = mm->is_floating ? 6 : 0;
    for (int i = 0; i < mm->action_dim; ++i) s->tau[off + i] = x[nq + nd + ot + i];
    oq += mm->dof_q;
    od += mm->dof_qd;
    ot += mm->action_dim;
  }
  for (int b = 0; b < B && !rc; ++b) forward_dynamics(&sub[b], &sc[b]);
  for (int b = 0; b < B && !rc; ++b) integrate_euler_qdd(&sub[b], &sc[b]);
  /* World::step: contacts of every body pair first (world.hpp:321-333), then the pairs are resolved in order:
     multi_bodies_ = [plane,] body 0, 1, ...: plane-0, plane-1, ..., 0-1, 0-2, ..., 1-2, ... */
  static _Thread_local pair_contact_t pcs[TDS_MAX_PAIR_CONTACTS];
  int np[TDS_MAX_BODIES * TDS_MAX_BODIES], p0[TDS_MAX_BODIES * TDS_MAX_BODIES], npc = 0, npairs = 0;
  if (!rc) {
    if (m->has_plane)
      for (int b = 0; b < B; ++b) compute_contacts(&sub[b], &sc[b]);
    for (int i = 0; i < B && !rc; ++i)
      for (int j = i + 1; j < B && !rc; ++j) {
        const int n = compute_pair_contacts(&sub[i], &sc[i], &sub[j], &sc[j], pcs + npc, TDS_MAX_PAIR_CONTACTS - npc);
        if (n < 0) { rc = -3; break; }
        p0[npairs] = npc; np[npairs++] = n;
        npc += n;
      }
  }
  if (!rc && m->has_plane)
    for (int b = 0; b < B && !rc; ++b) rc = resolve_collision(&sub[b], &sc[b], NULL); /* plane-0, plane-1, ... */
  for (int i = 0, p = 0; i < B && !rc; ++i)
    for (int j = i + 1; j < B && !rc; ++j, ++p)
      rc = resolve_collision_pair(m, &sub[i], &sc[i], &sub[j], &sc[j], pcs + p0[p], np[p]);
  if (!rc) {
    for (int b = 0; b < B; ++b) integrate_euler(&sub[b], &sc[b]);
    int j = 0;
    for (int i = 0; i < m->output_dim; ++i) y[i] = 0.0;
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < sub[b].dof_q; ++i) y[j++] = sc[b].q[i];
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < sub[b].dof_qd; ++i) y[j++] = sc[b].qd[i];
    if (m->pack_visuals) {
      for (int v = 0; v < m->num_visuals; ++v) {
        const tds_visual_t *V = &m->visuals[v];
        int b = 0;
        while (b + 1 < B && V->link >= m->bodies[b + 1].first_link) ++b;
        const int li = V->link - (b ? m->bodies[b].first_link : 0);
        xf_t lv, vx;
        double orn[4];
        memcpy(lv.r, V->X_rot, sizeof(lv.r)); memcpy(lv.t, V->X_trans, sizeof(lv.t));
        xf_mul(&sc[b].L[li].X_world, &lv, &vx);
        y[j++] = vx.t[0]; y[j++] = vx.t[1]; y[j++] = vx.t[2];
        matrix_to_quat(vx.r, orn);
        y[j++] = orn[0]; y[j++] = orn[1]; y[j++] = orn[2]; y[j++] = orn[3];
      }
      y[j++] = sc[0].base_X_world.r[8];
    }
  }
  free(sub);
  free(sc);
  return rc;
}

int tds_oracle_step_debug(const tds_model_t *model, const double *x, double *y,
                          tds_oracle_debug_t *dbg) {
  scratch_t *s = (scratch_t *)calloc(1, sizeof(scratch_t));
  if (!s) return -10;
  int rc = step_one(model, x, y, s, dbg);
  free(s);
  return rc;
}

int tds_oracle_step(const tds_model_t *model, int n, const double *x, double *y) {
  scratch_t *s = (scratch_t *)calloc(1, sizeof(scratch_t));
  if (!s) return -10;
  int rc = 0;
  for (int e = 0; e < n && !rc; ++e)
    rc = step_one(model, x + (size_t)e * model->input_dim, y + (size_t)e * model->output_dim, s, NULL);
  free(s);
  return rc;
}

int tds_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int tds_oracle_step_omp(const tds_model_t *model, int n, const double *x, double *y, int num_threads) {
  int rc_all = 0;
#ifdef _OPENMP
  if (num_threads <= 0) num_threads = omp_get_max_threads();
#pragma omp parallel num_threads(num_threads)
  {
    scratch_t *s = (scratch_t *)calloc(1, sizeof(scratch_t));
#pragma omp for schedule(static)
    for (int e = 0; e < n; ++e) {
      int rc = step_one(model, x + (size_t)e * model->input_dim, y + (size_t)e * model->output_dim, s, NULL);
      if (rc) {
#pragma omp atomic write
        rc_all = rc;
      }
    }
    free(s);
  }
#else
  (void)num_threads;
  rc_all = tds_oracle_step(model, n, x, y);
#endif
  return rc_all;
}


/* ============================ free rigid bodies ============================ */
typedef struct { double nb[3], pa[3], pb[3], dist; int a, b; } rb_contact_t;

/* contact_point.hpp:43-94 (non-CppAD branch): emitted only when the centres are > 1e-5 apart */
static int rb_sphere_sphere(const double *pa, double ra, const double *pb, double rb, rb_contact_t *c) {
  double diff[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
  double length = sqrt(v3_dot(diff, diff));
  double distance = length - (ra + rb);
  if (!(length > 1.0 / 100000.0)) return 0;
  for (int k = 0; k < 3; ++k) {
    c->nb[k] = 1.0 / length * diff[k];
    c->pa[k] = pa[k] - ra * c->nb[k];
    c->pb[k] = c->pa[k] - distance * c->nb[k];
  }
  c->dist = distance;
  return 1;
}
/* contact_point.hpp:96-125: A = plane, B = sphere */
static int rb_plane_sphere(const double *n, double constant, const double *pb, double rb, rb_contact_t *c) {
  double mn[3] = {-n[0], -n[1], -n[2]};
  double t = -(v3_dot(pb, mn) + constant);
  for (int k = 0; k < 3; ++k) {
    c->pa[k] = pb[k] + t * mn[k];
    c->pb[k] = pb[k] - rb * n[k];
    c->nb[k] = mn[k];
  }
  c->dist = t - rb;
  return 1;
}

static void rb_world_step(const tds_rb_model_t *m, double *S /* [nb][13] */) {
  const int nb = m->num_bodies;
  const double dt = m->dt;
  double inv_mass[TDS_RB_MAX_BODIES], inv_in[TDS_RB_MAX_BODIES];
  for (int i = 0; i < nb; ++i) {
    inv_mass[i] = m->bodies[i].mass == 0.0 ? 0.0 : 1.0 / m->bodies[i].mass; /* rigid_body.hpp:49-53 */
    inv_in[i] = m->bodies[i].mass == 0.0 ? 0.0 : 1.0;                       /* eye3 or zero33 */
  }
  /* apply_gravity + apply_force_impulse + clear_forces (world.hpp:301-310, rigid_body.hpp:81-97) */
  for (int i = 0; i < nb; ++i)
    for (int k = 0; k < 3; ++k) S[i * 13 + 7 + k] += (m->bodies[i].mass * m->gravity[k]) * inv_mass[i] * dt;
  /* pairwise narrowphase, i < j (world.hpp:163-191), dispatcher incl. swap (contact_point.hpp:468-496).
     Capsules and boxes collide as sets of spheres placed by Pose * offset (contact_point.hpp:127-198,
     405-438): capsule = 2 end spheres at local (0,0,+-L/2), box = 8 corner spheres of radius
     max(1e-2, box radius) at the corners pulled in by that radius (geometry.hpp:244-262). */
  rb_contact_t cs[TDS_RB_MAX_BODIES * TDS_RB_MAX_BODIES * 4];
  int nc = 0;
  for (int i = 0; i < nb; ++i)
    for (int j = i + 1; j < nb; ++j) {
      const int ti = m->bodies[i].geom_type, tj = m->bodies[j].geom_type;
      /* P = the plane / the single sphere of the pair, Q = the body expanded into spheres */
      int p = -1, q = -1, kind = -1; /* kind 0: plane(p) vs spheres(q); 1: spheres(p = capsule or sphere) vs sphere(q) */
      int swap = 0;
      if (ti == TDS_GEOM_PLANE && (tj == TDS_GEOM_SPHERE || tj == TDS_GEOM_CAPSULE || tj == TDS_GEOM_BOX)) {
        p = i; q = j; kind = 0;
      } else if (tj == TDS_GEOM_PLANE && (ti == TDS_GEOM_SPHERE || ti == TDS_GEOM_CAPSULE || ti == TDS_GEOM_BOX)) {
        p = j; q = i; kind = 0; swap = 1;
      } else if ((ti == TDS_GEOM_SPHERE || ti == TDS_GEOM_CAPSULE) && tj == TDS_GEOM_SPHERE) {
        p = i; q = j; kind = 1;
      } else if (ti == TDS_GEOM_SPHERE && tj == TDS_GEOM_CAPSULE) {
        p = j; q = i; kind = 1; swap = 1;
      } else {
        continue;
      }
      /* the spheres of the expanded body: e = q (kind 0) or p (kind 1) */
      const int e = kind == 0 ? q : p;
      const tds_rb_body_t *E = &m->bodies[e];
      double off[8][3], rad;
      int ns;
      if (E->geom_type == TDS_GEOM_SPHERE) {
        ns = 1; rad = E->radius; off[0][0] = off[0][1] = off[0][2] = 0.0;
      } else if (E->geom_type == TDS_GEOM_CAPSULE) {
        ns = 2; rad = E->radius;
        off[0][0] = off[0][1] = 0.0; off[0][2] = 0.5 * E->length;
        off[1][0] = off[1][1] = 0.0; off[1][2] = -0.5 * E->length;
      } else {
        ns = 8; rad = E->radius > 1e-2 ? E->radius : 1e-2;
        const double dx = E->extents[0] * 0.5 - rad, dy = E->extents[1] * 0.5 - rad, dz = E->extents[2] * 0.5 - rad;
        for (int c = 0; c < 8; ++c) {
          off[c][0] = (c & 4) ? -dx : dx;
          off[c][1] = (c & 2) ? -dy : dy;
          off[c][2] = (c & 1) ? -dz : dz;
        }
      }
      for (int sidx = 0; sidx < ns; ++sidx) {
        double ctr[3], r3[3];
        if (E->geom_type == TDS_GEOM_SPHERE) {
          for (int k = 0; k < 3; ++k) ctr[k] = S[e * 13 + k];
        } else {
          quat_rotate(S + e * 13 + 3, off[sidx], r3); /* Pose::operator*, pose.hpp:47-53 */
          for (int k = 0; k < 3; ++k) ctr[k] = S[e * 13 + k] + r3[k];
        }
        rb_contact_t c;
        int got;
        if (kind == 0) {
          /* Plane's constructor normalises the normal (geometry.hpp:163-168) */
          const double *pn = m->bodies[p].plane_normal;
          const double nl = sqrt(v3_dot(pn, pn));
          const double nn[3] = {pn[0] / nl, pn[1] / nl, pn[2] / nl};
          got = rb_plane_sphere(nn, m->bodies[p].plane_constant, ctr, rad, &c);
        }
        else got = rb_sphere_sphere(ctr, rad, S + q * 13, m->bodies[q].radius, &c);
        if (!got) continue;
        if (swap) { /* swap normal and points a, b */
          for (int k = 0; k < 3; ++k) {
            double t = c.pa[k];
            c.pa[k] = c.pb[k];
            c.pb[k] = t;
            c.nb[k] = -c.nb[k];
          }
        }
        c.a = i;
        c.b = j;
        cs[nc++] = c;
      }
    }
  /* sequential impulses (world.hpp:336-340, rb_constraint_solver.hpp:112-165) */
  for (int it = 0; it < m->solver_iterations; ++it)
    for (int ci = 0; ci < nc; ++ci) {
      const rb_contact_t *c = &cs[ci];
      if (!(c->dist < 0.0)) continue;
      double *Sa = S + c->a * 13, *Sb = S + c->b * 13;
      double ra[3], rb[3], va[3], vb[3], rel[3], t3[3];
      for (int k = 0; k < 3; ++k) {
        ra[k] = c->pa[k] - Sa[k];
        rb[k] = c->pb[k] - Sb[k];
      }
      const double baumgarte = m->erp * c->dist / dt;
      v3_cross(Sa + 10, ra, t3);
      for (int k = 0; k < 3; ++k) va[k] = Sa[7 + k] + t3[k];
      v3_cross(Sb + 10, rb, t3);
      for (int k = 0; k < 3; ++k) vb[k] = Sb[7 + k] + t3[k];
      for (int k = 0; k < 3; ++k) rel[k] = va[k] - vb[k];
      const double nrv = v3_dot(c->nb, rel);
      if (!(nrv < 0.0)) continue;
      double t1[3], t2[3], x1[3], x2[3], sum[3];
      v3_cross(ra, c->nb, t1);
      v3_cross(rb, c->nb, t2);
      for (int k = 0; k < 3; ++k) {
        t1[k] *= inv_in[c->a];
        t2[k] *= inv_in[c->b];
      }
      v3_cross(t1, ra, x1);
      v3_cross(t2, rb, x2);
      for (int k = 0; k < 3; ++k) sum[k] = x1[k] + x2[k];
      const double ang = v3_dot(c->nb, sum);
      const double denom = inv_mass[c->a] + inv_mass[c->b] + ang;
      const double impulse = (-(1.0 + m->restitution) * nrv - baumgarte) / denom;
      if (!(impulse > 0.0)) continue;
      double iv[3];
      for (int k = 0; k < 3; ++k) iv[k] = impulse * c->nb[k];
      /* apply_impulse(iv, ra) on a, (-iv, rb) on b: rigid_body.hpp:103-108 */
      for (int k = 0; k < 3; ++k) Sa[7 + k] += inv_mass[c->a] * iv[k];
      v3_cross(ra, iv, t3);
      for (int k = 0; k < 3; ++k) Sa[10 + k] += inv_in[c->a] * t3[k];
      for (int k = 0; k < 3; ++k) Sb[7 + k] += inv_mass[c->b] * -iv[k];
      double miv[3] = {-iv[0], -iv[1], -iv[2]};
      v3_cross(rb, miv, t3);
      for (int k = 0; k < 3; ++k) Sb[10 + k] += inv_in[c->b] * t3[k];
      /* friction uses the PRE-impulse relative velocity */
      double lat[3];
      for (int k = 0; k < 3; ++k) lat[k] = rel[k] - nrv * c->nb[k];
      const double latn = sqrt(v3_dot(lat, lat));
      const double trial = latn / denom;
      const double fimp = trial < m->friction * impulse ? trial : m->friction * impulse;
      if (latn > 1.0 / 10000.0) {
        double fd[3], fa[3], fb[3];
        for (int k = 0; k < 3; ++k) {
          fd[k] = lat[k] * (1.0 / latn);
          fa[k] = -fimp * fd[k];
          fb[k] = fimp * fd[k];
        }
        for (int k = 0; k < 3; ++k) Sa[7 + k] += inv_mass[c->a] * fa[k];
        v3_cross(ra, fa, t3);
        for (int k = 0; k < 3; ++k) Sa[10 + k] += inv_in[c->a] * t3[k];
        for (int k = 0; k < 3; ++k) Sb[7 + k] += inv_mass[c->b] * fb[k];
        v3_cross(rb, fb, t3);
        for (int k = 0; k < 3; ++k) Sb[10 + k] += inv_in[c->b] * t3[k];
      }
    }
  /* integrate (rigid_body.hpp:116-122, tiny_algebra.hpp:604-614) */
  for (int i = 0; i < nb; ++i) {
    double *B = S + i * 13;
    for (int k = 0; k < 3; ++k) B[k] += B[7 + k] * dt;
    const double qx = B[3], qy = B[4], qz = B[5], qw = B[6];
    const double *w = B + 10;
    const double ww = (-qx * w[0] - qy * w[1] - qz * w[2]) * (0.5 * dt);
    const double xx = (qw * w[0] + qz * w[1] - qy * w[2]) * (0.5 * dt);
    const double yy = (qw * w[1] + qx * w[2] - qz * w[0]) * (0.5 * dt);
    const double zz = (qw * w[2] + qy * w[0] - qx * w[1]) * (0.5 * dt);
    double q[4] = {qx + xx, qy + yy, qz + zz, qw + ww};
    quat_normalize(q);
    for (int k = 0; k < 4; ++k) B[3 + k] = q[k];
  }
}

int tds_oracle_rb_step(const tds_rb_model_t *model, int n, int steps, double *state) {
  if (!model || model->num_bodies < 1 || model->num_bodies > TDS_RB_MAX_BODIES) return -1;
  for (int e = 0; e < n; ++e)
    for (int s = 0; s < steps; ++s) rb_world_step(model, state + (size_t)e * model->num_bodies * 13);
  return 0;
}
