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

#include "tds_api_internal.h"

namespace tds_internal {
thread_local char g_err[512] = "";
}
using namespace tds_internal;

#define HIP_TRY TDS_HIP_TRY

namespace {

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

// API-level timing (tds_hip_set_timing): ev0 when the entry point starts enqueueing, ev1 after its LAST launch, so
// that multi-launch forms (auto-reset split, per-step rollout, step_many) report the whole sequence
struct TimedCall {
  tds_hip_sim *s;
  explicit TimedCall(tds_hip_sim *sim) : s(sim) {
    if (s->timing) (void)hipEventRecord(s->ev0, s->stream);
  }
  ~TimedCall() {
    if (s->timing) {
      (void)hipEventRecord(s->ev1, s->stream);
      s->have_ms = true;
    }
  }
};

// host double <-> device record dtype
int upload(tds_hip_sim *s, void *dst, const double *src, size_t count) {
  if (s->records_f64()) {
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
  if (s->records_f64()) {
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

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

namespace tds_internal {

int launch(tds_hip_sim *s, const void *x, void *y, const void *actions, void *fb, void *obs, int n, int nsub,
           int reset_mode, const unsigned char *mask, const Rollout *ro, int ctl_flags) {
  TdsStepCtl ctl;
  memset(&ctl, 0, sizeof(ctl));
  if (ro) {
    ctl.policy = ro->policy;
    ctl.ret_sum = ro->ret_sum;
    ctl.ret_steps = ro->ret_steps;
    ctl.shift = ro->shift;
    ctl.flags = ro->flags;
  }
  ctl.flags |= ctl_flags;
  ctl.nsub = nsub;
  ctl.reset_mode = reset_mode;
  ctl.settle_steps = s->model.settle_steps < 0 ? 0 : s->model.settle_steps;
  ctl.seed = s->seed;
  ctl.mask = mask;
  ctl.reset_count = s->d_reset_count;
  int rc;
  if (s->dtype == TDS_DTYPE_F64)
    rc = tds_launch_step<double, double>((const DevModel<double> *)s->d_model, s->h64, s->lds, s->lanes,
                                         (const double *)x, (double *)y, (const double *)actions, (double *)fb,
                                         (double *)obs, (double *)s->d_ovf, n, s->stream, ctl);
  else if (s->dtype == TDS_DTYPE_F64_REC32)
    rc = tds_launch_step<double, float>((const DevModel<double> *)s->d_model, s->h64, s->lds, s->lanes,
                                        (const float *)x, (float *)y, (const float *)actions, (float *)fb, (float *)obs,
                                        (double *)s->d_ovf, n, s->stream, ctl);
  else
    rc = tds_launch_step<float, float>((const DevModel<float> *)s->d_model, s->h32, s->lds, s->lanes, (const float *)x,
                                       (float *)y, (const float *)actions, (float *)fb, (float *)obs,
                                       (float *)s->d_ovf, n, s->stream, ctl);
  if (rc != 0) {
    snprintf(g_err, sizeof(g_err), "kernel launch failed: %s", rc > 0 ? hipGetErrorString((hipError_t)rc) : "bad lanes_per_env");
    return TDS_ERR_HIP;
  }
  return TDS_OK;
}

}  // namespace tds_internal

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
  if (dtype != TDS_DTYPE_F64 && dtype != TDS_DTYPE_F32 && dtype != TDS_DTYPE_F64_REC32)
    return fail(TDS_ERR_INVALID_ARG, "unknown dtype");
  int rc = tds_hip_model_check(model);
  if (rc != TDS_OK) return rc;
  int ndev = tds_hip_device_count();
  if (ndev <= 0) return fail(TDS_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(TDS_ERR_INVALID_ARG, "device index out of range");
  DeviceGuard guard(device);  // (the caller's current device is restored on return)
  tds_hip_sim *s = new (std::nothrow) tds_hip_sim();  // value-initialised: both host models start zeroed
  if (!s) return fail(TDS_ERR_INVALID_ARG, "out of host memory");
  s->model = *model;
  s->num_envs = num_envs;
  s->device = device;
  s->dtype = dtype;
  s->elem = dtype == TDS_DTYPE_F64 ? 8 : 4;
  const bool c64 = dtype != TDS_DTYPE_F32;  // compute scalar (DevModel / LDS / slab): double unless the pure f32 build
  const size_t celem = c64 ? 8 : 4;
  // (a floating base takes six more lanes: its pseudo links, tds_device_model.h)
  char why[128];
  size_t msize;
  const void *hsrc;
  if (c64) {
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
  s->lanes = default_lanes_per_env(c64 ? s->h64.num_links : s->h32.num_links, model->dof_qd);
  const int epw = 64 / s->lanes;
  auto layout = [&](int cap) {
    return c64 ? tds_make_lds_layout<double>(s->h64, cap, s->lanes) : tds_make_lds_layout<float>(s->h32, cap, s->lanes);
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
      if ((size_t)layout(cap).stride * epw * celem <= budget) {
        na_cap = cap;
        break;
      }
  }
  s->lds = layout(na_cap);
  const int lds_bytes = (int)((size_t)s->lds.stride * epw * celem);
  if (lds_bytes > 160 * 1024) {
    delete s;
    return fail(TDS_ERR_UNSUPPORTED, "model needs more than 160 KiB of LDS per workgroup");
  }
  if (lds_bytes > 64 * 1024) {
    const bool is_fl = c64 ? s->h64.is_floating : s->h32.is_floating;
    const bool is_sph = c64 ? s->h64.num_spherical != 0 : s->h32.num_spherical != 0;
    const int kind = is_fl ? 1 : (is_sph ? 2 : 0);
    int e = dtype == TDS_DTYPE_F64         ? tds_kernel_max_dynamic_lds<double, double>(s->lanes, s->lds.NDP, lds_bytes, kind)
            : dtype == TDS_DTYPE_F64_REC32 ? tds_kernel_max_dynamic_lds<double, float>(s->lanes, s->lds.NDP, lds_bytes, kind)
                                           : tds_kernel_max_dynamic_lds<float, float>(s->lanes, s->lds.NDP, lds_bytes, kind);
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
    CREATE_TRY(hipMalloc(&s->d_ovf, (size_t)num_envs * s->lds.ovrows * (s->lds.NDs + 3) * celem));
  CREATE_TRY(hipMalloc((void **)&s->d_reset_count, (size_t)num_envs * sizeof(unsigned int)));
  CREATE_TRY(hipMemset(s->d_reset_count, 0, (size_t)num_envs * sizeof(unsigned int)));
  {
    // scratch of the multi-launch forms, allocated here (sizes are known) so that no step call ever allocates:
    // d_split = [obs | reward | done] records + done mask of the two-launch auto-reset step;
    // d_ro = actions | records | returns | counts | latches of the per-step-launch rollout
    const size_t n = (size_t)num_envs, w = (size_t)(model->dof_q + model->dof_qd + 2);
    const size_t adim = (size_t)(model->action_dim > 0 ? model->action_dim : 1);
    CREATE_TRY(hipMalloc(&s->d_split, align256(n * w * s->elem) + n));
    CREATE_TRY(hipMalloc(&s->d_ro, align256(n * adim * s->elem) + align256(n * w * s->elem) + align256(n * s->elem) +
                                       align256(n * sizeof(int)) + align256(n)));
  }
  CREATE_TRY(hipEventCreate(&s->ev0));
  CREATE_TRY(hipEventCreate(&s->ev1));
#undef CREATE_TRY
  *out = s;
  return TDS_OK;
}

int tds_hip_destroy(tds_hip_sim_t *s) {
  if (!s) return TDS_OK;
  DeviceGuard guard(s->device);
  if (s->graph_exec) (void)hipGraphExecDestroy(s->graph_exec);
  if (s->graph_stream) (void)hipStreamDestroy(s->graph_stream);
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
int tds_hip_device(const tds_hip_sim_t *s) { return s ? s->device : -1; }
int tds_hip_record_bytes(const tds_hip_sim_t *s) { return s ? (int)s->elem : 0; }
void *tds_hip_x_device(tds_hip_sim_t *s) { return s ? s->d_x : nullptr; }
void *tds_hip_y_device(tds_hip_sim_t *s) { return s ? s->d_y : nullptr; }

int tds_hip_set_inputs(tds_hip_sim_t *s, const double *x_host) {
  if (!s || !x_host) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard guard(s->device);
  int rc = upload(s, s->d_x, x_host, (size_t)s->num_envs * s->model.input_dim);
  if (rc != TDS_OK) return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  return TDS_OK;
}
int tds_hip_get_inputs(tds_hip_sim_t *s, double *x_host) {
  if (!s || !x_host) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard guard(s->device);
  return download(s, x_host, s->d_x, (size_t)s->num_envs * s->model.input_dim);
}
int tds_hip_get_outputs(tds_hip_sim_t *s, double *y_host) {
  if (!s || !y_host) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard guard(s->device);
  return download(s, y_host, s->d_y, (size_t)s->num_envs * s->model.output_dim);
}
int tds_hip_sync(tds_hip_sim_t *s) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  DeviceGuard guard(s->device);
  HIP_TRY(hipStreamSynchronize(s->stream));
  return TDS_OK;
}

int tds_hip_forward_zero_device(tds_hip_sim_t *s, const void *x_dev, void *y_dev) {
  if (!s || !x_dev || !y_dev) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard guard(s->device);
  TimedCall timed(s);
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

// one closed-loop step incl. the auto-reset forms (device selected, timing handled by the caller)
int step_obs_impl(tds_hip_sim *s, const void *actions_dev, int substeps, void *obs_dev) {
  // Auto-reset at large batches: the in-kernel reset needs the step-loop build (one wavefront per SIMD whatever
  // the batch).  From two wavefronts per SIMD on, a single step is cheaper as the straight-line launch followed by
  // a forced-reset launch masked with the done flags (idle lane groups leave at once); same random stream
  // (seed, environment, reset counter), same records.  TDS_HIP_AUTO_RESET_SPLIT=1 / 0 forces / forbids it.
  if (s->auto_reset && substeps == 1) {
    const char *e = getenv("TDS_HIP_AUTO_RESET_SPLIT");
    const long waves = ((long)s->num_envs * s->lanes + 63) / 64;
    if (e ? e[0] == '1' : waves >= 2048) {
      const int n = s->num_envs, w = s->obs_width();
      const size_t b_rec = align256((size_t)n * w * s->elem);
      void *rec = obs_dev ? obs_dev : s->d_split;
      unsigned char *mask = (unsigned char *)s->d_split + b_rec;
      int rc = launch(s, s->d_x, s->d_y, actions_dev, s->d_x, rec, n, 1, TDS_RESET_NONE, nullptr);
      if (rc != TDS_OK) return rc;
      if (s->records_f64())
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
}  // namespace
}  // extern "C++"

int tds_hip_step_obs(tds_hip_sim_t *s, const void *actions_dev, int substeps, void *obs_dev) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  if (substeps < 1) return fail(TDS_ERR_INVALID_ARG, "substeps must be >= 1");
  DeviceGuard guard(s->device);
  TimedCall timed(s);
  return step_obs_impl(s, actions_dev, substeps, obs_dev);
}

// K closed-loop steps per host call, replayed from a captured hipGraph (one graph launch instead of K kernel
// launches: at ~20 us per step the host-side launch gaps are otherwise a double-digit share of short runs).
extern "C++" {
namespace {
int graph_matches(const tds_hip_sim *s, const void *actions, int pool, int first, int n_steps, void *obs) {
  return s->graph_exec && s->graph_actions == actions && s->graph_pool == pool && s->graph_first == first &&
         s->graph_steps == n_steps && s->graph_obs == obs;
}
int build_graph(tds_hip_sim *s, const void *actions, int pool, int first, int n_steps, void *obs) {
  if (s->graph_exec) {
    (void)hipGraphExecDestroy(s->graph_exec);
    s->graph_exec = nullptr;
  }
  if (!s->graph_stream) HIP_TRY(hipStreamCreateWithFlags(&s->graph_stream, hipStreamNonBlocking));
  // capture on a private stream (the handle's stream may be the NULL stream, which cannot be captured);
  // the launches are recorded, not executed
  hipStream_t user = s->stream;
  s->stream = s->graph_stream;
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamBeginCapture(s->graph_stream, hipStreamCaptureModeThreadLocal);
  int rc = TDS_OK;
  if (e == hipSuccess) {
    const size_t blk = (size_t)s->num_envs * s->model.action_dim * s->elem;
    for (int k = 0; k < n_steps && rc == TDS_OK; ++k) {
      const void *a = actions ? (const char *)actions + (size_t)((first + k) % pool) * blk : nullptr;
      rc = launch(s, s->d_x, s->d_y, a, s->d_x, obs, s->num_envs, 1, TDS_RESET_NONE, nullptr);
    }
    e = hipStreamEndCapture(s->graph_stream, &graph);
  }
  s->stream = user;
  if (e != hipSuccess || rc != TDS_OK || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    if (rc == TDS_OK) snprintf(g_err, sizeof(g_err), "graph capture failed: %s", hipGetErrorString(e));
    return TDS_ERR_HIP;
  }
  e = hipGraphInstantiate(&s->graph_exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    s->graph_exec = nullptr;
    snprintf(g_err, sizeof(g_err), "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return TDS_ERR_HIP;
  }
  s->graph_actions = actions;
  s->graph_pool = pool;
  s->graph_first = first;
  s->graph_steps = n_steps;
  s->graph_obs = obs;
  return TDS_OK;
}
}  // namespace
}  // extern "C++"

int tds_hip_step_many_prepare(tds_hip_sim_t *s, const void *actions_dev, int action_blocks, int first_block,
                              int n_steps, void *obs_dev) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  if (n_steps < 1 || n_steps > 4096) return fail(TDS_ERR_INVALID_ARG, "n_steps must be in 1..4096");
  if (actions_dev && action_blocks < 1) return fail(TDS_ERR_INVALID_ARG, "action_blocks must be >= 1");
  if (s->auto_reset) return fail(TDS_ERR_INVALID_ARG, "step_many replays plain closed-loop steps (auto-reset is off the graph)");
  DeviceGuard guard(s->device);
  const int pool = actions_dev ? action_blocks : 1;
  const int first = actions_dev ? ((first_block % pool) + pool) % pool : 0;
  if (graph_matches(s, actions_dev, pool, first, n_steps, obs_dev)) return TDS_OK;
  return build_graph(s, actions_dev, pool, first, n_steps, obs_dev);
}

int tds_hip_step_many(tds_hip_sim_t *s, const void *actions_dev, int action_blocks, int first_block, int n_steps,
                      void *obs_dev) {
  int rc = tds_hip_step_many_prepare(s, actions_dev, action_blocks, first_block, n_steps, obs_dev);
  if (rc != TDS_OK) return rc;
  DeviceGuard guard(s->device);
  TimedCall timed(s);
  HIP_TRY(hipGraphLaunch(s->graph_exec, s->stream));
  return TDS_OK;
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
  DeviceGuard guard(s->device);
  TimedCall timed(s);
  return launch(s, s->d_x, s->d_y, nullptr, s->d_x, obs_dev, s->num_envs, 0, TDS_RESET_FORCED, mask_dev, nullptr,
                TDS_CTL_RESET_CALL);
}

int tds_hip_step(tds_hip_sim_t *s, const void *actions_dev, int substeps) {
  return tds_hip_step_obs(s, actions_dev, substeps, nullptr);
}

int tds_hip_obs_dim(const tds_hip_sim_t *s) { return s ? s->model.dof_q + s->model.dof_qd : 0; }

int tds_hip_forward_zero_host(tds_hip_sim_t *s, int n, const double *x_host, double *y_host) {
  if (!s || !x_host || !y_host) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  if (n < 1 || n > s->num_envs) return fail(TDS_ERR_INVALID_ARG, "n out of range");
  DeviceGuard guard(s->device);
  int rc = upload(s, s->d_x, x_host, (size_t)n * s->model.input_dim);
  if (rc != TDS_OK) return rc;
  {
    TimedCall timed(s);
    rc = launch(s, s->d_x, s->d_y, nullptr, nullptr, nullptr, n, 1, TDS_RESET_NONE, nullptr);
  }
  if (rc != TDS_OK) return rc;
  return download(s, y_host, s->d_y, (size_t)n * s->model.output_dim);
}

// the same call in two halves, so that a host that drives several devices (HipStepper with a device list) can
// enqueue every device's share before it waits for any of them
int tds_hip_forward_zero_host_begin(tds_hip_sim_t *s, int n, const double *x_host, double *y_host) {
  if (!s || !x_host || !y_host) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  if (n < 1 || n > s->num_envs) return fail(TDS_ERR_INVALID_ARG, "n out of range");
  if (!s->records_f64()) return fail(TDS_ERR_INVALID_ARG, "begin/end needs f64 records (no host-side conversion)");
  DeviceGuard guard(s->device);
  HIP_TRY(hipMemcpyAsync(s->d_x, x_host, (size_t)n * s->model.input_dim * 8, hipMemcpyHostToDevice, s->stream));
  int rc = launch(s, s->d_x, s->d_y, nullptr, nullptr, nullptr, n, 1, TDS_RESET_NONE, nullptr);
  if (rc != TDS_OK) return rc;
  HIP_TRY(hipMemcpyAsync(y_host, s->d_y, (size_t)n * s->model.output_dim * 8, hipMemcpyDeviceToHost, s->stream));
  return TDS_OK;
}
int tds_hip_forward_zero_host_end(tds_hip_sim_t *s) { return tds_hip_sync(s); }

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
    while (L * adim > 64) L >>= 1;  // 1 <= adim <= TDS_MAX_ACTIONS = 32 (tds_hip_model_check): L >= 2
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
  const size_t b_act = align256((size_t)n * adim * sizeof(T));
  const size_t b_rec = align256((size_t)n * (od + 2) * sizeof(T));
  const size_t b_ret = align256((size_t)n * sizeof(T));
  const size_t b_cnt = align256((size_t)n * sizeof(int));
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
  DeviceGuard guard(s->device);
  TimedCall timed(s);
  // One launch for the whole rollout (the step-loop build: 256 VGPR + AGPR copies, one wavefront per SIMD) or one
  // launch per step of the straight-line build with the policy + bookkeeping kernel in between: from two
  // wavefronts per SIMD on (8192 Ant environments) the straight-line build overlaps them and wins.
  // flags bit 1: force the per-step launches, bit 2: force the single launch.  Auto-reset lives in the step loop.
  const long waves = ((long)s->num_envs * s->lanes + 63) / 64;
  const bool per_step = !s->auto_reset && !(flags & 4) && ((flags & 2) || waves >= 2048);
  if (per_step)
    return s->records_f64()
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
  DeviceGuard guard(s->device);
  return upload(s, s->d_x, x_host, (size_t)n * s->model.input_dim);
}

int tds_hip_forward_zero_fetch(tds_hip_sim_t *s, int n, double *y_host) {
  if (!s || !y_host) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  if (n < 1 || n > s->num_envs) return fail(TDS_ERR_INVALID_ARG, "n out of range");
  DeviceGuard guard(s->device);
  int rc;
  {
    TimedCall timed(s);
    rc = launch(s, s->d_x, s->d_y, nullptr, nullptr, nullptr, n, 1, TDS_RESET_NONE, nullptr);
  }
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
  DeviceGuard guard(s->device);
  HIP_TRY(hipEventSynchronize(s->ev1));
  HIP_TRY(hipEventElapsedTime(ms, s->ev0, s->ev1));
  return TDS_OK;
}

int tds_hip_profile_phases(tds_hip_sim_t *s, long long *cycles_host, int n) {
  if (!s || !cycles_host) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  if (n < TDS_NUM_PHASE_STAMPS) return fail(TDS_ERR_INVALID_ARG, "need room for 14 stamps");
  DeviceGuard guard(s->device);
  long long *d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(long long) * TDS_NUM_PHASE_STAMPS));
  HIP_TRY(hipMemset(d, 0, sizeof(long long) * TDS_NUM_PHASE_STAMPS));
  TdsStepCtl ctl;
  memset(&ctl, 0, sizeof(ctl));
  ctl.nsub = 1;
  int rc;
  if (s->dtype == TDS_DTYPE_F64)
    rc = tds_launch_step<double, double>((const DevModel<double> *)s->d_model, s->h64, s->lds, s->lanes,
                                         (const double *)s->d_x, (double *)s->d_y, nullptr, nullptr, nullptr,
                                         (double *)s->d_ovf, s->num_envs, s->stream, ctl, d);
  else if (s->dtype == TDS_DTYPE_F64_REC32)
    rc = tds_launch_step<double, float>((const DevModel<double> *)s->d_model, s->h64, s->lds, s->lanes,
                                        (const float *)s->d_x, (float *)s->d_y, nullptr, nullptr, nullptr,
                                        (double *)s->d_ovf, s->num_envs, s->stream, ctl, d);
  else
    rc = tds_launch_step<float, float>((const DevModel<float> *)s->d_model, s->h32, s->lds, s->lanes,
                                       (const float *)s->d_x, (float *)s->d_y, nullptr, nullptr, nullptr,
                                       (float *)s->d_ovf, s->num_envs, s->stream, ctl, d);
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
  if (lds_bytes_per_env) *lds_bytes_per_env = (int)(s->lds.stride * (s->compute_f64() ? 8 : 4));
  if (threads_per_env) *threads_per_env = s->lanes;
  if (envs_per_block) *envs_per_block = 64 / s->lanes;
  return TDS_OK;
}

}  // extern "C"
