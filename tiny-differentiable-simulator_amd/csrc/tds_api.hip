// tds_api.hip — implementation of the C ABI declared in include/tds_hip.h.
//
// Replaces, on the reference side:
//   * VectorizedEnvironment::CustomForwardDynamicsStepper::step
//       (examples/ars/ars_vectorized_environment.h:75-85)
//   * the generated <model>_forward_zero{,_meta,_allocate,_deallocate} library
//       (examples/ars/ars_train_policy_cuda.cpp:220-308; src/utils/cuda_codegen.hpp:146-262)
// There is NO CPU fallback in this library: without a HIP device every entry point that needs
// one returns TDS_ERR_NO_DEVICE / TDS_ERR_HIP.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "tds_device_model.h"
#include "tds_hip.h"
#include "tds_kernels.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *detail = "") {
  snprintf(g_err, sizeof(g_err), fmt, detail);
  return code;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      snprintf(g_err, sizeof(g_err), "%s failed: %s", #expr, hipGetErrorString(e_));       \
      return TDS_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)

// lanes per environment: the smallest wave-group that holds every link and every padded dof
int default_lanes_per_env(int num_links, int dof) {
  const int need = num_links > tds_padded_dof(dof) ? num_links : tds_padded_dof(dof);
  const char *env = getenv("TDS_HIP_LANES_PER_ENV");
  int g = env ? atoi(env) : 0;
  // 64 lanes per environment is only instantiated for systems of <= 16 dof: the <G=64, NDP>=24>
  // build was miscompiled by hipcc 7.2 under its register pressure (caught by the golden tests),
  // and it is never the fast choice anyway.
  if (g == 16 || g == 32 || (g == 64 && tds_padded_dof(dof) <= 16)) {
    if (g >= need) return g;
  }
  return need <= 16 ? 16 : (need <= 32 ? 32 : 64);
}

}  // namespace

struct tds_hip_sim {
  tds_model_t model;
  int num_envs = 0, device = 0, dtype = TDS_DTYPE_F64, lanes = 64;
  size_t elem = 8;
  hipStream_t stream = nullptr;
  void *d_model = nullptr;  // DevModel<T>
  DevModel<double> h64;
  DevModel<float> h32;
  TdsLds lds;
  void *d_x = nullptr, *d_y = nullptr, *d_ovf = nullptr;
  unsigned int *d_reset_count = nullptr;
  void *d_split = nullptr;  // records + done mask of the two-launch auto-reset step
  void *d_ro = nullptr;  // scratch of the per-step-launch rollout (actions | records | returns | counts | latches)
  bool auto_reset = false;
  unsigned long long seed = 0x5DEECE66Dull;
  std::vector<double> stage;
  bool timing = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float last_ms = 0.f;
  bool have_ms = false;

  int input_dim() const { return model.input_dim; }
  int output_dim() const { return model.output_dim; }
};

namespace {

struct Rollout {
  const void *policy;
  void *ret_sum;
  int *ret_steps;
  double shift;
  int flags;
};

int launch(tds_hip_sim *s, const void *x, void *y, const void *actions, void *fb, void *obs, int n, int nsub,
           int reset_mode, const unsigned char *mask, const Rollout *ro = nullptr) {
  if (s->timing) (void)hipEventRecord(s->ev0, s->stream);
  TdsStepCtl ctl;
  memset(&ctl, 0, sizeof(ctl));
  if (ro) {
    ctl.policy = ro->policy;
    ctl.ret_sum = ro->ret_sum;
    ctl.ret_steps = ro->ret_steps;
    ctl.shift = ro->shift;
    ctl.flags = ro->flags;
  }
  ctl.nsub = nsub;
  ctl.reset_mode = reset_mode;
  ctl.settle_steps = s->model.settle_steps < 0 ? 0 : s->model.settle_steps;
  ctl.seed = s->seed;
  ctl.mask = mask;
  ctl.reset_count = s->d_reset_count;
  int rc;
  if (s->dtype == TDS_DTYPE_F64)
    rc = tds_launch_step<double>((const DevModel<double> *)s->d_model, s->h64, s->lds, s->lanes, (const double *)x,
                                 (double *)y, (const double *)actions, (double *)fb, (double *)obs,
                                 (double *)s->d_ovf, n, s->stream, ctl);
  else
    rc = tds_launch_step<float>((const DevModel<float> *)s->d_model, s->h32, s->lds, s->lanes, (const float *)x,
                                (float *)y, (const float *)actions, (float *)fb, (float *)obs, (float *)s->d_ovf, n,
                                s->stream, ctl);
  if (rc != 0) {
    snprintf(g_err, sizeof(g_err), "kernel launch failed: %s", rc > 0 ? hipGetErrorString((hipError_t)rc) : "bad lanes_per_env");
    return TDS_ERR_HIP;
  }
  if (s->timing) {
    (void)hipEventRecord(s->ev1, s->stream);
    s->have_ms = true;
  }
  return TDS_OK;
}

// host double <-> device compute dtype
int upload(tds_hip_sim *s, void *dst, const double *src, size_t count) {
  if (s->dtype == TDS_DTYPE_F64) {
    HIP_TRY(hipMemcpyAsync(dst, src, count * 8, hipMemcpyHostToDevice, s->stream));
  } else {
    std::vector<float> tmp(count);
    for (size_t i = 0; i < count; ++i) tmp[i] = (float)src[i];
    HIP_TRY(hipMemcpyAsync(dst, tmp.data(), count * 4, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  return TDS_OK;
}
int download(tds_hip_sim *s, double *dst, const void *src, size_t count) {
  if (s->dtype == TDS_DTYPE_F64) {
    HIP_TRY(hipMemcpyAsync(dst, src, count * 8, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  } else {
    std::vector<float> tmp(count);
    HIP_TRY(hipMemcpyAsync(tmp.data(), src, count * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (size_t i = 0; i < count; ++i) dst[i] = (double)tmp[i];
  }
  return TDS_OK;
}

}  // namespace

extern "C" {

const char *tds_hip_last_error(void) { return g_err; }
int tds_hip_abi_version(void) { return TDS_HIP_ABI_VERSION; }

int tds_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int tds_hip_model_check(const tds_model_t *model) {
  if (!model) return fail(TDS_ERR_INVALID_ARG, "model is NULL");
  DevModel<double> *d = new (std::nothrow) DevModel<double>;
  if (!d) return fail(TDS_ERR_INVALID_ARG, "out of host memory");
  char why[128];
  int rc = tds_build_dev_model<double>(model, d, why);
  delete d;
  if (rc != TDS_OK) return fail(rc, "%s", why);
  if (model->num_links > 64) return fail(TDS_ERR_UNSUPPORTED, "more than 64 links");
  return TDS_OK;
}

int tds_hip_create(const tds_model_t *model, int num_envs, int device, int dtype, tds_hip_sim_t **out) {
  if (!out) return fail(TDS_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  if (num_envs <= 0) return fail(TDS_ERR_INVALID_ARG, "num_envs must be positive");
  if (dtype != TDS_DTYPE_F64 && dtype != TDS_DTYPE_F32) return fail(TDS_ERR_INVALID_ARG, "unknown dtype");
  int rc = tds_hip_model_check(model);
  if (rc != TDS_OK) return rc;
  int ndev = tds_hip_device_count();
  if (ndev <= 0) return fail(TDS_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(TDS_ERR_INVALID_ARG, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  tds_hip_sim *s = new (std::nothrow) tds_hip_sim;
  if (!s) return fail(TDS_ERR_INVALID_ARG, "out of host memory");
  s->model = *model;
  s->num_envs = num_envs;
  s->device = device;
  s->dtype = dtype;
  s->elem = dtype == TDS_DTYPE_F64 ? 8 : 4;
  // (a floating base takes six more lanes: its pseudo links, tds_device_model.h)
  char why[128];
  size_t msize;
  const void *hsrc;
  if (dtype == TDS_DTYPE_F64) {
    tds_build_dev_model<double>(model, &s->h64, why);
    msize = sizeof(DevModel<double>);
    hsrc = &s->h64;
  } else {
    tds_build_dev_model<float>(model, &s->h32, why);
    msize = sizeof(DevModel<float>);
    hsrc = &s->h32;
  }
  // lanes per environment: the device model's link count (pseudo links of a floating base / of spherical joints
  // included, folded fixed links excluded) and the padded dof count
  s->lanes = default_lanes_per_env(dtype == TDS_DTYPE_F64 ? s->h64.num_links : s->h32.num_links, model->dof_qd);
  const int epw = 64 / s->lanes;
  auto layout = [&](int cap) {
    return dtype == TDS_DTYPE_F64 ? tds_make_lds_layout<double>(s->h64, cap, s->lanes)
                                  : tds_make_lds_layout<float>(s->h32, cap, s->lanes);
  };
  // Contacts whose constraint rows stay in LDS (the surplus goes to a global slab: exact, slower).
  // Default: up to 8, lowered (not below 5) if that is what lets EIGHT workgroups share a CU's 160 KiB,
  // i.e. two wavefronts per SIMD, which hides most of the instruction-stream latency once the batch
  // provides them (Ant f64: 6 -> 19.8 KiB per workgroup).  TDS_HIP_NA_CAP overrides.
  int na_cap = 8;
  if (const char *e = getenv("TDS_HIP_NA_CAP")) {
    na_cap = atoi(e);
  } else {
    const size_t budget = (160 * 1024) / 8;
    for (int cap = 8; cap >= 5; --cap)
      if ((size_t)layout(cap).stride * epw * s->elem <= budget) {
        na_cap = cap;
        break;
      }
  }
  s->lds = layout(na_cap);
  const int lds_bytes = (int)((size_t)s->lds.stride * epw * s->elem);
  if (lds_bytes > 160 * 1024) {
    delete s;
    return fail(TDS_ERR_UNSUPPORTED, "model needs more than 160 KiB of LDS per workgroup");
  }
  if (lds_bytes > 64 * 1024) {
    const int kind = s->h64.is_floating || s->h32.is_floating ? 1 : (s->h64.num_spherical || s->h32.num_spherical ? 2 : 0);
    int e = dtype == TDS_DTYPE_F64 ? tds_kernel_max_dynamic_lds<double>(s->lanes, s->lds.NDP, lds_bytes, kind)
                                   : tds_kernel_max_dynamic_lds<float>(s->lanes, s->lds.NDP, lds_bytes, kind);
    if (e != 0) {
      delete s;
      return fail(TDS_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    }
  }
#define CREATE_TRY(expr)                                                                   \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      snprintf(g_err, sizeof(g_err), "%s failed: %s", #expr, hipGetErrorString(e_));       \
      tds_hip_destroy(s);                                                                  \
      return TDS_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)
  CREATE_TRY(hipMalloc(&s->d_model, msize));
  CREATE_TRY(hipMemcpy(s->d_model, hsrc, msize, hipMemcpyHostToDevice));
  CREATE_TRY(hipMalloc(&s->d_x, (size_t)num_envs * model->input_dim * s->elem));
  CREATE_TRY(hipMalloc(&s->d_y, (size_t)num_envs * model->output_dim * s->elem));
  CREATE_TRY(hipMemset(s->d_x, 0, (size_t)num_envs * model->input_dim * s->elem));
  CREATE_TRY(hipMemset(s->d_y, 0, (size_t)num_envs * model->output_dim * s->elem));
  if (s->lds.ovrows > 0)
    CREATE_TRY(hipMalloc(&s->d_ovf, (size_t)num_envs * s->lds.ovrows * (s->lds.NDs + 3) * s->elem));
  CREATE_TRY(hipMalloc((void **)&s->d_reset_count, (size_t)num_envs * sizeof(unsigned int)));
  CREATE_TRY(hipMemset(s->d_reset_count, 0, (size_t)num_envs * sizeof(unsigned int)));
  CREATE_TRY(hipEventCreate(&s->ev0));
  CREATE_TRY(hipEventCreate(&s->ev1));
#undef CREATE_TRY
  *out = s;
  return TDS_OK;
}

int tds_hip_destroy(tds_hip_sim_t *s) {
  if (!s) return TDS_OK;
  if (s->d_model) (void)hipFree(s->d_model);
  if (s->d_x) (void)hipFree(s->d_x);
  if (s->d_y) (void)hipFree(s->d_y);
  if (s->d_ovf) (void)hipFree(s->d_ovf);
  if (s->d_reset_count) (void)hipFree(s->d_reset_count);
  if (s->d_ro) (void)hipFree(s->d_ro);
  if (s->d_split) (void)hipFree(s->d_split);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  delete s;
  return TDS_OK;
}

int tds_hip_set_stream(tds_hip_sim_t *s, void *hip_stream) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  s->stream = (hipStream_t)hip_stream;
  return TDS_OK;
}

int tds_hip_num_envs(const tds_hip_sim_t *s) { return s ? s->num_envs : 0; }
int tds_hip_input_dim(const tds_hip_sim_t *s) { return s ? s->model.input_dim : 0; }
int tds_hip_output_dim(const tds_hip_sim_t *s) { return s ? s->model.output_dim : 0; }
int tds_hip_dtype(const tds_hip_sim_t *s) { return s ? s->dtype : -1; }
void *tds_hip_x_device(tds_hip_sim_t *s) { return s ? s->d_x : nullptr; }
void *tds_hip_y_device(tds_hip_sim_t *s) { return s ? s->d_y : nullptr; }

int tds_hip_set_inputs(tds_hip_sim_t *s, const double *x_host) {
  if (!s || !x_host) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  int rc = upload(s, s->d_x, x_host, (size_t)s->num_envs * s->model.input_dim);
  if (rc != TDS_OK) return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  return TDS_OK;
}
int tds_hip_get_inputs(tds_hip_sim_t *s, double *x_host) {
  if (!s || !x_host) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  return download(s, x_host, s->d_x, (size_t)s->num_envs * s->model.input_dim);
}
int tds_hip_get_outputs(tds_hip_sim_t *s, double *y_host) {
  if (!s || !y_host) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  return download(s, y_host, s->d_y, (size_t)s->num_envs * s->model.output_dim);
}

int tds_hip_forward_zero_device(tds_hip_sim_t *s, const void *x_dev, void *y_dev) {
  if (!s || !x_dev || !y_dev) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  return launch(s, x_dev, y_dev, nullptr, nullptr, nullptr, s->num_envs, 1, TDS_RESET_NONE, nullptr);
}

extern "C++" {
namespace {
// done column of the [obs | reward | done] records -> byte mask of tds_hip_reset
template <typename T>
__global__ void tds_done_mask_kernel(const T *__restrict__ rec, int width, unsigned char *__restrict__ mask, int n) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env < n) mask[env] = rec[(size_t)env * width + width - 1] != T(0) ? 1 : 0;
}
}  // namespace
}  // extern "C++"

int tds_hip_step_obs(tds_hip_sim_t *s, const void *actions_dev, int substeps, void *obs_dev) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  if (substeps < 1) return fail(TDS_ERR_INVALID_ARG, "substeps must be >= 1");
  // Auto-reset at large batches: the in-kernel reset needs the step-loop build (one wavefront per SIMD whatever
  // the batch).  From two wavefronts per SIMD on, a single step is cheaper as the straight-line launch followed by
  // a forced-reset launch masked with the done flags (idle lane groups leave at once); same random stream
  // (seed, environment, reset counter), same records.  TDS_HIP_AUTO_RESET_SPLIT=1 / 0 forces / forbids it.
  if (s->auto_reset && substeps == 1) {
    const char *e = getenv("TDS_HIP_AUTO_RESET_SPLIT");
    const long waves = ((long)s->num_envs * s->lanes + 63) / 64;
    if (e ? e[0] == '1' : waves >= 2048) {
      const int n = s->num_envs, w = s->model.dof_q + s->model.dof_qd + 2;
      const size_t b_rec = ((size_t)n * w * s->elem + 255) & ~(size_t)255;
      if (!s->d_split && hipMalloc(&s->d_split, b_rec + n) != hipSuccess)
        return fail(TDS_ERR_HIP, "hipMalloc (auto-reset scratch)");
      void *rec = obs_dev ? obs_dev : s->d_split;
      unsigned char *mask = (unsigned char *)s->d_split + b_rec;
      int rc = launch(s, s->d_x, s->d_y, actions_dev, s->d_x, rec, n, 1, TDS_RESET_NONE, nullptr);
      if (rc != TDS_OK) return rc;
      if (s->dtype == TDS_DTYPE_F64)
        hipLaunchKernelGGL(tds_done_mask_kernel<double>, dim3((n + 255) / 256), dim3(256), 0, s->stream,
                           (const double *)rec, w, mask, n);
      else
        hipLaunchKernelGGL(tds_done_mask_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, s->stream,
                           (const float *)rec, w, mask, n);
      if (hipGetLastError() != hipSuccess) return fail(TDS_ERR_HIP, "mask kernel launch");
      return launch(s, s->d_x, s->d_y, nullptr, s->d_x, obs_dev, n, 0, TDS_RESET_FORCED, mask);
    }
  }
  // ONE launch: the kernel loops over the substeps (same action) with the state kept in LDS, writes
  // y / reward / done of the last substep and, with auto-reset on, re-initialises + settles the
  // environments that ended with done before it writes their observation and resident state
  return launch(s, s->d_x, s->d_y, actions_dev, s->d_x, obs_dev, s->num_envs, substeps,
                s->auto_reset ? TDS_RESET_AUTO : TDS_RESET_NONE, nullptr);
}

int tds_hip_set_auto_reset(tds_hip_sim_t *s, int enable, unsigned long long seed) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  if (enable && s->model.reward_mode == TDS_REWARD_NONE)
    return fail(TDS_ERR_INVALID_ARG, "auto-reset needs a model with a termination rule (reward_mode)");
  s->auto_reset = enable != 0;
  s->seed = seed;
  return TDS_OK;
}

int tds_hip_reset(tds_hip_sim_t *s, const unsigned char *mask_dev, void *obs_dev) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  return launch(s, s->d_x, s->d_y, nullptr, s->d_x, obs_dev, s->num_envs, 0, TDS_RESET_FORCED, mask_dev);
}

int tds_hip_step(tds_hip_sim_t *s, const void *actions_dev, int substeps) {
  return tds_hip_step_obs(s, actions_dev, substeps, nullptr);
}

int tds_hip_obs_dim(const tds_hip_sim_t *s) { return s ? s->model.dof_q + s->model.dof_qd : 0; }

int tds_hip_forward_zero_host(tds_hip_sim_t *s, int n, const double *x_host, double *y_host) {
  if (!s || !x_host || !y_host) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  if (n < 1 || n > s->num_envs) return fail(TDS_ERR_INVALID_ARG, "n out of range");
  int rc = upload(s, s->d_x, x_host, (size_t)n * s->model.input_dim);
  if (rc != TDS_OK) return rc;
  rc = launch(s, s->d_x, s->d_y, nullptr, nullptr, nullptr, n, 1, TDS_RESET_NONE, nullptr);
  if (rc != TDS_OK) return rc;
  return download(s, y_host, s->d_y, (size_t)n * s->model.output_dim);
}

extern "C++" {
namespace {

// Between two step launches of the per-step-launch rollout: return bookkeeping of the step just taken
// (Worker::rollouts: a done environment is not counted and, without auto-reset, stays done) and the
// environment's own linear policy on the new state (VectorizedEnvironment::policy -> NeuralNetwork::compute:
// one linear layer with bias, identity; obs = [q | qd] with obs[0] = obs[1] = 0,
// ars_vectorized_environment.h:165-180,283-300).
template <typename T>
__global__ void tds_policy_book_kernel(const T *__restrict__ x, int in_dim, int od, int adim,
                                       const T *__restrict__ policy, T *__restrict__ actions,
                                       T *__restrict__ rec, T *__restrict__ ret, int *__restrict__ cnt,
                                       unsigned char *__restrict__ frozen, T shift, int do_book, int do_policy,
                                       int raw_xy, int n) {
  // one wavefront per environment; L = 2^k lanes per action, consecutive lanes on consecutive weights
  const int env = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const int lane = threadIdx.x & 63;
  if (env >= n) return;
  if (do_book && lane == 0) {
    bool fr = frozen[env] != 0;
    if (!fr) {
      const T reward = rec[(size_t)env * (od + 2) + od];
      if (rec[(size_t)env * (od + 2) + od + 1] != T(0)) {
        frozen[env] = 1;
        fr = true;
      } else {
        ret[env] += reward - shift;
        cnt[env] += 1;
      }
    }
    // after the last step the record's done column is the latch ("was done at some step"), as the one-launch
    // rollout leaves it
    if (!do_policy) rec[(size_t)env * (od + 2) + od + 1] = fr ? T(1) : T(0);
  }
  if (do_policy) {
    int L = 64;
    while (L * adim > 64) L >>= 1;  // adim <= TDS_MAX_ACTIONS = 32: L >= 2
    const int a = lane / L, sub = lane - a * L;
    const T *const W = policy + (size_t)env * (adim * od + adim);
    const T *const xe = x + (size_t)env * in_dim;
    T acc = T(0);
    if (a < adim)
      for (int o = sub; o < od; o += L) {
        const T ob = (o < 2 && !raw_xy) ? T(0) : xe[o];
        acc += ob * W[a * od + o];
      }
    for (int d = 1; d < L; d <<= 1) acc += __shfl_xor(acc, d, 64);
    if (a < adim && sub == 0) actions[(size_t)env * adim + a] = acc + W[adim * od + a];
  }
}

template <typename T>
int rollout_per_step(tds_hip_sim *s, const void *policy_dev, int n_steps, double shift, int flags,
                     void *return_sum_dev, int *return_steps_dev, void *obs_dev) {
  const int n = s->num_envs, adim = s->model.action_dim, od = s->model.dof_q + s->model.dof_qd;
  const size_t b_act = ((size_t)n * adim * sizeof(T) + 255) & ~(size_t)255;
  const size_t b_rec = ((size_t)n * (od + 2) * sizeof(T) + 255) & ~(size_t)255;
  const size_t b_ret = ((size_t)n * sizeof(T) + 255) & ~(size_t)255;
  const size_t b_cnt = ((size_t)n * sizeof(int) + 255) & ~(size_t)255;
  const size_t b_frz = ((size_t)n + 255) & ~(size_t)255;
  if (!s->d_ro && hipMalloc(&s->d_ro, b_act + b_rec + b_ret + b_cnt + b_frz) != hipSuccess)
    return fail(TDS_ERR_HIP, "hipMalloc (rollout scratch)");
  char *base = (char *)s->d_ro;
  T *actions = (T *)base;
  T *rec = obs_dev ? (T *)obs_dev : (T *)(base + b_act);
  T *ret = return_sum_dev ? (T *)return_sum_dev : (T *)(base + b_act + b_rec);
  int *cnt = return_steps_dev ? return_steps_dev : (int *)(base + b_act + b_rec + b_ret);
  unsigned char *frozen = (unsigned char *)(base + b_act + b_rec + b_ret + b_cnt);
  if (hipMemsetAsync(ret, 0, (size_t)n * sizeof(T), s->stream) != hipSuccess ||
      hipMemsetAsync(cnt, 0, (size_t)n * sizeof(int), s->stream) != hipSuccess ||
      hipMemsetAsync(frozen, 0, (size_t)n, s->stream) != hipSuccess)
    return fail(TDS_ERR_HIP, "hipMemsetAsync (rollout scratch)");
  const int threads = 256, blocks = (n + threads / 64 - 1) / (threads / 64);
  for (int t = 0; t <= n_steps; ++t) {
    hipLaunchKernelGGL(tds_policy_book_kernel<T>, dim3(blocks), dim3(threads), 0, s->stream, (const T *)s->d_x,
                       s->model.input_dim, od, adim, (const T *)policy_dev, actions, rec, ret, cnt, frozen,
                       (T)shift, t > 0 ? 1 : 0, t < n_steps ? 1 : 0, ((flags & 1) && t == 0) ? 1 : 0, n);
    if (hipGetLastError() != hipSuccess) return fail(TDS_ERR_HIP, "policy kernel launch");
    if (t == n_steps) break;
    const int rc = launch(s, s->d_x, s->d_y, actions, s->d_x, rec, n, 1, TDS_RESET_NONE, nullptr);
    if (rc != TDS_OK) return rc;
  }
  return TDS_OK;
}

}  // namespace
}  // extern "C++"

int tds_hip_rollout(tds_hip_sim_t *s, const void *policy_dev, int n_steps, double shift, int flags,
                    void *return_sum_dev, int *return_steps_dev, void *obs_dev) {
  if (!s || !policy_dev) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  if (n_steps < 1) return fail(TDS_ERR_INVALID_ARG, "n_steps < 1");
  if (s->model.action_dim < 1) return fail(TDS_ERR_INVALID_ARG, "model has no actions");
  // One launch for the whole rollout (the step-loop build: 256 VGPR + AGPR copies, one wavefront per SIMD) or one
  // launch per step of the straight-line build with the policy + bookkeeping kernel in between: from two
  // wavefronts per SIMD on (8192 Ant environments) the straight-line build overlaps them and wins.
  // flags bit 1: force the per-step launches, bit 2: force the single launch.  Auto-reset lives in the step loop.
  const long waves = ((long)s->num_envs * s->lanes + 63) / 64;
  const bool per_step = !s->auto_reset && !(flags & 4) && ((flags & 2) || waves >= 2048);
  if (per_step)
    return s->dtype == TDS_DTYPE_F64
               ? rollout_per_step<double>(s, policy_dev, n_steps, shift, flags, return_sum_dev, return_steps_dev, obs_dev)
               : rollout_per_step<float>(s, policy_dev, n_steps, shift, flags, return_sum_dev, return_steps_dev, obs_dev);
  Rollout ro;
  ro.policy = policy_dev;
  ro.ret_sum = return_sum_dev;
  ro.ret_steps = return_steps_dev;
  ro.shift = shift;
  ro.flags = flags;
  return launch(s, s->d_x, s->d_y, nullptr, s->d_x, obs_dev, s->num_envs, n_steps,
                s->auto_reset ? TDS_RESET_AUTO : TDS_RESET_NONE, nullptr, &ro);
}

int tds_hip_send_local(tds_hip_sim_t *s, int n, const double *x_host) {
  if (!s || !x_host) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  if (n < 1 || n > s->num_envs) return fail(TDS_ERR_INVALID_ARG, "n out of range");
  return upload(s, s->d_x, x_host, (size_t)n * s->model.input_dim);
}

int tds_hip_forward_zero_fetch(tds_hip_sim_t *s, int n, double *y_host) {
  if (!s || !y_host) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  if (n < 1 || n > s->num_envs) return fail(TDS_ERR_INVALID_ARG, "n out of range");
  int rc = launch(s, s->d_x, s->d_y, nullptr, nullptr, nullptr, n, 1, TDS_RESET_NONE, nullptr);
  if (rc != TDS_OK) return rc;
  return download(s, y_host, s->d_y, (size_t)n * s->model.output_dim);
}

int tds_hip_set_timing(tds_hip_sim_t *s, int enable) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  s->timing = enable != 0;
  s->have_ms = false;
  return TDS_OK;
}
int tds_hip_last_kernel_ms(tds_hip_sim_t *s, float *ms) {
  if (!s || !ms) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  if (!s->have_ms) return fail(TDS_ERR_INVALID_ARG, "no timed launch recorded");
  HIP_TRY(hipEventSynchronize(s->ev1));
  HIP_TRY(hipEventElapsedTime(ms, s->ev0, s->ev1));
  return TDS_OK;
}

int tds_hip_profile_phases(tds_hip_sim_t *s, long long *cycles_host, int n) {
  if (!s || !cycles_host) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  if (n < TDS_NUM_PHASE_STAMPS) return fail(TDS_ERR_INVALID_ARG, "need room for 14 stamps");
  long long *d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(long long) * TDS_NUM_PHASE_STAMPS));
  HIP_TRY(hipMemset(d, 0, sizeof(long long) * TDS_NUM_PHASE_STAMPS));
  TdsStepCtl ctl;
  memset(&ctl, 0, sizeof(ctl));
  ctl.nsub = 1;
  int rc;
  if (s->dtype == TDS_DTYPE_F64)
    rc = tds_launch_step<double>((const DevModel<double> *)s->d_model, s->h64, s->lds, s->lanes, (const double *)s->d_x,
                                 (double *)s->d_y, nullptr, nullptr, nullptr, (double *)s->d_ovf, s->num_envs, s->stream, ctl, d);
  else
    rc = tds_launch_step<float>((const DevModel<float> *)s->d_model, s->h32, s->lds, s->lanes, (const float *)s->d_x,
                                (float *)s->d_y, nullptr, nullptr, nullptr, (float *)s->d_ovf, s->num_envs, s->stream, ctl, d);
  if (rc != 0) {
    (void)hipFree(d);
    return fail(TDS_ERR_HIP, "profiling launch failed");
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipMemcpy(cycles_host, d, sizeof(long long) * TDS_NUM_PHASE_STAMPS, hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return TDS_OK;
}

int tds_hip_kernel_info(const tds_hip_sim_t *s, int *lds_bytes_per_env, int *threads_per_env, int *envs_per_block) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  if (lds_bytes_per_env) *lds_bytes_per_env = (int)(s->lds.stride * s->elem);
  if (threads_per_env) *threads_per_env = s->lanes;
  if (envs_per_block) *envs_per_block = 64 / s->lanes;
  return TDS_OK;
}

}  // extern "C"
