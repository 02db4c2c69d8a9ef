// tds_kernels.h — host-visible interface of tds_kernels.hip
#pragma once
#include <hip/hip_runtime.h>

#include "tds_device_model.h"

#define TDS_NUM_PHASE_STAMPS 14

// Per-environment LDS layout, offsets in units of the compute scalar T (see tds_make_lds_layout).
struct TdsLds {
  int stride;              // scalars per environment
  int NLp, NDP, NDs, NCPp; // links, padded dof (8/16/24/32), dof row stride (odd), contact-point stride
  int xrec, Xw, swd, Lp, dinv, cp, rowb, rowai, rowx;
  int v, IA, pA, Ic, F;    // sweep arrays            } these two groups alias each other:
  int Z;                   // constraint rows         } rows are built after the sweeps are done
};

template <typename T>
TdsLds tds_make_lds_layout(const DevModel<T> &m);

// Enqueue one step  y = f(x)  for n_envs environments on `stream`.
//   actions    (optional) [n_envs][action_dim] overrides the action slice of x
//   x_feedback (optional) [n_envs][input_dim]  receives the new q, qd (closed-loop stepping)
//   obs_out    (optional) [n_envs][dof_q+dof_qd+2]  observation | reward | done
template <typename T>
int tds_launch_step(const DevModel<T> *d_model, const DevModel<T> &h_model, const TdsLds &L, int lanes_per_env,
                    const T *x_in, T *y_out, const T *actions, T *x_feedback, T *obs_out, int n_envs,
                    hipStream_t stream, long long *prof = nullptr);  // prof: 14 phase stamps of workgroup 0 (diagnostic)

template <typename T>
int tds_kernel_max_dynamic_lds(int lanes_per_env, int ndp, int bytes);

int tds_padded_dof(int nd);
