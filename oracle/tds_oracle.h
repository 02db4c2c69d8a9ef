/* tds_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C (C99) CPU restatement of the reference algorithm for the hot path, operating on the
 * same flattened model blob (include/tds_hip.h::tds_model_t) as the HIP path.  It follows the
 * reference's formulation step by step (link-local frames, dense 6x6 congruence, explicit
 * M^-1, dense A = J M^-1 J^T, row-wise PGS) so that it is an independent check of the
 * re-formulated HIP kernels.  Every function cites the reference file:line it restates.
 *
 * PARITY PINNING: this restatement is checked against (a) the reference itself, compiled from
 * /root/reference into oracle/_ref/libtds_ref.so (tests/test_oracle_vs_reference.py, runs where
 * the reference is present), (b) the golden vectors generated from that library and committed
 * under tests/golden/ (tests/test_oracle_golden.py, runs anywhere), and (c) the known-answer
 * values of SURVEY.md Appendix B.  The reference's own tests hold no golden vectors for the
 * contact/LCP part of the path, so (a)/(b) are the pin.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 */
#ifndef TDS_ORACLE_H
#define TDS_ORACLE_H
#include "tds_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* optional per-env intermediates (any pointer may be NULL) */
typedef struct tds_oracle_debug {
  double *qdd;      /* [dof_qd]            after forward_dynamics                      */
  double *M;        /* [dof_qd*dof_qd]     joint-space inertia (row-major)             */
  double *Minv;     /* [dof_qd*dof_qd]                                                  */
  double *contacts; /* [n_c*10] normal_on_b(3) point_on_b(3) point_on_a(3) distance     */
  double *jac;      /* [n_c*3*dof_qd]      point Jacobians of the robot                */
  double *lcp_A;    /* [(3n_c)^2]                                                       */
  double *lcp_b;    /* [3n_c]                                                           */
  double *lcp_p;    /* [3n_c]                                                           */
  double *X_world;  /* [n_links*12]        rot(9) trans(3)                              */
  int n_c;
} tds_oracle_debug_t;

/* y[n][output_dim] = step(x[n][input_dim]); returns 0 on success.  Serial. */
int tds_oracle_step(const tds_model_t *model, int n, const double *x, double *y);
/* same, OpenMP over environments (one scratch object per thread) — bench.py cpu_baseline */
int tds_oracle_step_omp(const tds_model_t *model, int n, const double *x, double *y,
                        int num_threads);
/* one environment, with intermediates */
int tds_oracle_step_debug(const tds_model_t *model, const double *x, double *y,
                          tds_oracle_debug_t *dbg);
int tds_oracle_max_threads(void);

/* free rigid bodies (row a20): state[n][num_bodies][13] advanced by `steps` World::step calls */
int tds_oracle_rb_step(const tds_rb_model_t *model, int n, int steps, double *state);

#ifdef __cplusplus
}
#endif
#endif
