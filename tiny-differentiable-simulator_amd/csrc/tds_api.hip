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
  const long long g_opt = tds_opt_now(TDS_OPT_LANES_PER_ENV);
  const int g = g_opt == TDS_OPT_UNSET ? 0 : (int)g_opt;
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

void pool_free(tds_hip_sim *s);  // (reset pool, defined with the rest of it below)
void pool_reset(tds_hip_sim *s);  // ... and back to "no pool yet" (an option that shapes the pool has changed)
void drop_graphs(tds_hip_sim *s);

}  // namespace

// experiment slots (tds_kernels.h): weak — NULL unless tools/build_alt.sh linked a slot's translation unit in
extern "C" {
#define TDS_ALT_DECL(k)                                                                                                  \
  __attribute__((weak)) int tds_alt_launch_##k(const void *, const void *, const TdsLds *, int, const void *, void *,    \
                                               const void *, void *, void *, void *, int, hipStream_t,                   \
                                               const TdsStepCtl *, int, int *);
TDS_ALT_DECL(1) TDS_ALT_DECL(2) TDS_ALT_DECL(3) TDS_ALT_DECL(4) TDS_ALT_DECL(5) TDS_ALT_DECL(6)
#undef TDS_ALT_DECL
}
// The 16-lane kernel's step-loop form (tds_quad.hip): how its workgroups are shaped for a launch over n_envs environments —
// 1: one wavefront per workgroup (resident up to six workgroups per compute unit: the constant table costs LDS), W =
// TDS_QUAD_WIDE_WAVES: W wavefronts around one table, a workgroup per compute unit (resident up to 32 environments per
// compute unit: laikago_soft x 8192), 0: neither form has every workgroup resident (the caller takes the chained graphs)
static int quad_loop_waves(const tds_hip_sim *s, int n_envs) {
  const int in_dim = s->model.input_dim;
  const int per_cu = (int)(s->lds_per_cu / (size_t)tds_quad_loop_workgroup_bytes(in_dim, 1));
  const long long wide = s->opt.get(TDS_OPT_QUAD_WIDE, 1);  // 0: never, 1: where the narrow form is not resident, 2: always
  const bool wide_fits = (size_t)tds_quad_loop_workgroup_bytes(in_dim, TDS_QUAD_WIDE_WAVES) <= s->lds_per_cu &&
                         (n_envs + 4 * TDS_QUAD_WIDE_WAVES - 1) / (4 * TDS_QUAD_WIDE_WAVES) <= s->num_cus;
  if (wide == 2 && wide_fits) return TDS_QUAD_WIDE_WAVES;
  if ((n_envs + 3) / 4 <= s->num_cus * (per_cu < 8 ? per_cu : 8)) return 1;
  return (wide != 0 && wide_fits) ? TDS_QUAD_WIDE_WAVES : 0;
}
static tds_alt_launch_fn tds_alt_slot(int k) {
  switch (k) {
    case 1: return tds_alt_launch_1;
    case 2: return tds_alt_launch_2;
    case 3: return tds_alt_launch_3;
    case 4: return tds_alt_launch_4;
    case 5: return tds_alt_launch_5;
    case 6: return tds_alt_launch_6;
    default: return nullptr;
  }
}

namespace tds_internal {

int launch(tds_hip_sim *s, const void *x, void *y, const void *actions, void *fb, void *obs, int n, int nsub,
           int reset_mode, const unsigned char *mask, const Rollout *ro, int ctl_flags, const LaunchOpts *opts) {
  TdsStepCtl ctl;
  memset(&ctl, 0, sizeof(ctl));
  bool y_stride_set = false;
  if (opts && opts->extra) {  // work-list / reset-pool fields of a straight-line launch
    ctl.pool = opts->extra->pool;
    ctl.pool_depth = opts->extra->pool_depth;
    ctl.pool_envs = opts->extra->pool_envs;
  }
  const hipStream_t stream = (opts && opts->other_stream) ? opts->stream : s->stream;
  // two-wavefront workgroups: plain straight-line launches whose whole grid is resident at once (the helper wavefront
  // then fills issue slots that would otherwise idle; beyond that the one-wave form with more workgroups per CU wins)
  const int n_resident = (opts && opts->env_total > 0) ? opts->env_total : n;
  const int n_blocks = (n_resident + (64 / s->lanes) - 1) / (64 / s->lanes);
  // ... straight-line launches; and step-loop launches of plain steps (no policy, no reset, no reset pool): there the
  // helper wavefront loops along and is also the RECORDER of per-step rings (option loop_w2 = 0: the one-wave loop build)
  const bool loop_w2 = s->opt.get(TDS_OPT_LOOP_W2, 1) != 0;
  const bool w2_fits = s->w2_max_blocks > 0 && n_blocks <= s->w2_max_blocks && reset_mode == TDS_RESET_NONE && !ro &&
                       !(opts && opts->lds);
  const bool is_loop_launch = nsub > 1 || (opts && opts->rings);
  // (loop_w2 = 2: not for launches that take reset states from the pool)
  const bool loop_w2_pool = s->opt.get(TDS_OPT_LOOP_W2, 1) != 2;
  // A launch whose ring slots are exchanged while it runs (rings->progress) takes the SAME two-wavefront build an N = 1
  // launch takes (round 4: every rank of an N > 1 run executes the N = 1 kernel).  Round 3 dropped such launches to the
  // one-wave loop build — two wavefronts of 256 registers per SIMD leave no register for anybody else, and the exchange's
  // kernels (the one-lane wait, RCCL's all-gather) get onto a compute unit only when a workgroup of the launch retires
  // (profiles/r03_ring_exchange_forms.txt: the first wait of a 64-step launch returned after 86 % of it) — at the price of
  // 13 % of the step rate before a byte travelled.  Measured on one rank (profiles/r04_same_box_ab_and_exchange_forms.txt):
  // two-wavefront build + 256-step launches 14.9 us per step (0.935 of the N = 1 rate), one-wave build 18.6.
  // Option exchange_w2 = 0 brings the one-wave build back (bench.py times both forms in its warm-up at N > 1 and keeps the
  // faster one on every rank); the tests pin BOTH builds on the reference.
  const bool exchanged = opts && opts->rings && opts->rings->progress && s->opt.get(TDS_OPT_EXCHANGE_W2, 1) == 0;
  const bool two_waves = w2_fits && (is_loop_launch ? (loop_w2 && !exchanged && (loop_w2_pool || !(opts && opts->extra)) &&
                                                       s->lds_w2.NDP <= 16)
                                                    : !(opts && opts->rings));
  const TdsLds &lds = (opts && opts->lds) ? *opts->lds : (two_waves ? s->lds_w2 : s->lds);
  const long long occ = s->opt.get(TDS_OPT_LOOP_OCC, 0);
  // (the one-wavefront-per-SIMD compilation of the step loop does not exist below 24 padded dof: built without
  //  MachineLICM its <double, double, 16, 8> instantiation never terminated — profiles/r04_diag_loop_hang.txt — and the
  //  two-wavefront compilation holds no scratch there; at 14 - 18 dof it paid for the Ant and Laikago at small batches,
  //  which run in kernels of their own since rounds 5 / 6; asked for by option, the launch is refused instead of
  //  falling back silently)
  if (occ == 1 && lds.NDP < 24 && !two_waves && (nsub != 1 || reset_mode != TDS_RESET_NONE || ro || (opts && opts->rings)))
    return fail(TDS_ERR_UNSUPPORTED, "option loop_occ = 1: no one-wavefront-per-SIMD step-loop build below 24 padded dof");
  // the 8-lane kernel (tds_oct.hip) takes the launch: its two-wavefront build while every workgroup of the launch is resident
  // with at most two wavefronts per SIMD — four workgroups per compute unit, LDS permitting (Ant: up to 8192 environments)
  int oct_form = 0;
  if (s->compute_f64() && s->h64.oct != 0 && nsub >= 1 && reset_mode == TDS_RESET_NONE && !ro) {
    const long long o2 = s->opt.get(TDS_OPT_OCT_W2, 1);
    const int per_cu = (int)(s->lds_per_cu / (size_t)tds_oct_workgroup_bytes(s->model.input_dim));
    const int blocks = (n_resident + 7) / 8;
    // (two workgroups per compute unit = one wavefront per SIMD: the build compiled for that — no register limit to spill at)
    // (option oct_w2 = 3: the two-wavefronts-per-SIMD compilation at any grid size — 256 registers, so that OTHER launches fit
    //  beside it on a SIMD: the reset pool's refill passes, see pool_step_many)
    if (o2 != 0 && o2 != 3 && per_cu >= 2 && blocks <= 2 * s->num_cus) oct_form = TDS_FORM_OCT_W2_OCC1;
    else if (o2 == 2 || o2 == 3 || (o2 != 0 && blocks <= s->num_cus * (per_cu < 4 ? per_cu : 4))) oct_form = TDS_FORM_OCT_W2;
    // a refill pass of the reset pool (pool stream) while the handle's own chunks are of the one-wavefront-per-SIMD build: the
    // 240-register build, so that the pass runs BESIDE the chunk it was issued next to instead of in its tail
    if (opts && opts->other_stream && opts->lds == &s->pool_lds && o2 != 0 && o2 != 3 && per_cu >= 4 &&
        (s->num_envs + 7) / 8 <= 2 * s->num_cus && s->opt.get(TDS_OPT_POOL_BESIDE, 1) != 0)
      oct_form = TDS_FORM_OCT_BESIDE;
  }
  const int quad_form = (s->compute_f64() && s->h64.quad && quad_loop_waves(s, n_resident) > 1) ? TDS_FORM_QUAD_WIDE : 0;
  const long long cw2 = s->opt.get(TDS_OPT_CHAIN_W2, 1);
  const int form = (two_waves ? TDS_FORM_W2 : 0) | (occ == 1 ? TDS_FORM_LOOP_OCC1 : (occ == 2 ? TDS_FORM_LOOP_OCC2 : 0)) |
                   oct_form | quad_form | (cw2 == 0 ? TDS_FORM_CHAIN_W1 : (cw2 == 2 ? TDS_FORM_CHAIN_W2_ANY : 0));
  void *ovf = (opts && opts->ovf) ? opts->ovf : ((opts && opts->lds) ? nullptr : s->d_ovf);
  if (opts && opts->env_first > 0) {  // a sub-range of the environments: every per-environment array moves along
    const size_t e0 = (size_t)opts->env_first, el = s->elem;
    auto at = [&](const void *p, size_t per_env, size_t bytes) -> void * {
      return p ? (void *)((const char *)p + e0 * per_env * bytes) : nullptr;
    };
    x = at(x, s->model.input_dim, el);
    y = at(y, (opts->y_stride > 0 ? opts->y_stride : s->model.output_dim), el);
    actions = at(actions, s->model.action_dim, el);
    fb = at(fb, s->model.input_dim, el);
    obs = at(obs, s->obs_width(), el);
    ovf = at(ovf, (size_t)s->lds.ovrows * (s->lds.NDs + 3), s->compute_f64() ? 8 : 4);
    if (mask || ro) return fail(TDS_ERR_INVALID_ARG, "environment sub-ranges: plain steps only");
    // (the reset pool's rings are [slot][environment][q | qd]: the kernel indexes them with its LOCAL environment number)
    if (ctl.pool) ctl.pool = at(ctl.pool, (size_t)(s->model.dof_q + s->model.dof_qd), el);
  }
  if (ro) {
    ctl.policy = ro->policy;
    ctl.ret_sum = ro->ret_sum;
    ctl.ret_steps = ro->ret_steps;
    ctl.shift = ro->shift;
    ctl.flags = ro->flags;
  }
  if (opts && opts->act_pool) {
    ctl.act_pool = (const char *)opts->act_pool + (size_t)opts->env_first * s->model.action_dim * s->elem;
    ctl.act_blocks = opts->act_blocks;
    ctl.act_first = opts->act_first;
    ctl.act_envs = s->num_envs;
  }
  if (opts && opts->rings) {  // (step-loop launches: the caller made sure of that)
    const tds_hip_rings_t &r = *opts->rings;
    const size_t e0 = (size_t)opts->env_first;
    if (r.obs_ring) {
      const size_t ob = r.obs_f32 ? 4 : s->elem;
      ctl.obs_ring = (char *)r.obs_ring + e0 * s->obs_width() * ob;
      ctl.obs_slots = r.obs_slots;
      ctl.obs_envs = r.obs_slot_envs > 0 ? r.obs_slot_envs : s->num_envs;
      ctl.obs_first = (r.obs_first + opts->ring_step0) % r.obs_slots;
      if (r.obs_f32) ctl.ring_flags |= TDS_RING_OBS_F32;
      // (how a step's records are made visible to the exchange before its progress count: write-through stores + a
      //  plain wait, or streaming stores + a release fence — TDS_HIP_RING_NOFENCE=0 / 1)
      // Default: write-through.  The release fence's buffer_wbl2 writes back every dirty line of the L2 on every step of
      // every workgroup: + 9 us per 4096-environment step (profiles/r03_ring_exchange_forms.txt).
      const bool nofence = s->opt.get(TDS_OPT_RING_NOFENCE, 1) == 1;
      if (r.progress && nofence) ctl.ring_flags |= TDS_RING_NOFENCE;
      if (r.progress && s->opt.get(TDS_OPT_RING_SIGNAL_LATE, 0) == 1) ctl.ring_flags |= TDS_RING_SIGNAL_LATE;
    }
    if (r.y_ring) {
      ctl.y_stride = r.y_stride > 0 ? r.y_stride : s->model.output_dim;
      ctl.y_ring = (char *)r.y_ring + e0 * ctl.y_stride * s->elem;
      y_stride_set = true;
      ctl.y_slots = r.y_slots;
      ctl.y_first = (r.y_first + opts->ring_step0) % r.y_slots;
    }
    ctl.ring_envs = s->num_envs;
    ctl.progress = r.progress;
    if (s->peer_launch && r.obs_ring) {  // peer-store exchange (tds_shard.hip): the launch counts EVERY step in on arrival counters
      const TdsPeerLaunch &pl = *s->peer_launch;
      ctl.progress = nullptr;
      ctl.peer_ring = pl.rings;
      ctl.peer_flags = pl.flags;
      ctl.peer_arrive = pl.arrive;
      ctl.peer_off = pl.ring_off + (long long)(e0 * s->obs_width() * (r.obs_f32 ? 4 : s->elem));
      ctl.peer_epoch = pl.epoch;
      ctl.n_peers = pl.n_peers;
      ctl.peer_flag_off = pl.flag_off;
      ctl.peer_flag_stride = pl.flag_stride;
      if (pl.reward_done_only) ctl.ring_flags |= TDS_RING_PEER_REWARD_DONE;
      if (s->opt.get(TDS_OPT_SHARD_PEER_RELEASE, 0) == 1) ctl.ring_flags |= TDS_RING_PEER_RELEASE;
      {  // a wavefront's records as one row of 8-byte units (put_obs_wide): every stride a multiple of 8 bytes
        // (environments per wavefront: eight where the 8-lane kernel takes the launch — tds_oct_takes)
        const bool oct_launch = s->compute_f64() && (s->h64.oct != 0 || s->h64.chain != 0) && nsub >= 1 && reset_mode == TDS_RESET_NONE && !ro;
        const size_t wb = r.obs_f32 ? 4 : s->elem, w = (size_t)s->obs_width(), epw = oct_launch ? 8 : (size_t)(64 / s->lanes);
        if ((epw * w * wb) % 8 == 0 && ((size_t)ctl.obs_envs * w * wb) % 8 == 0 && ((size_t)(uintptr_t)ctl.obs_ring) % 8 == 0 &&
            (size_t)ctl.peer_off % 8 == 0 && (size_t)n % epw == 0 && pl.wide_ok)
          ctl.ring_flags |= TDS_RING_WIDE;
      }
      // this rank's own block: write-through stores, visible device-wide when the slot's flag is raised (a consumer may read a
      // slot before the launch has completed); counted in at the top of the helper wavefront's NEXT iteration, where it
      // waits for the kinematics anyway — the acknowledgements of stores that crossed xGMI have had a whole step by then
      ctl.ring_flags |= TDS_RING_NOFENCE;
      if (s->opt.get(TDS_OPT_RING_SIGNAL_LATE, 1) == 1) ctl.ring_flags |= TDS_RING_SIGNAL_LATE;
    }
  }
  if (opts && !opts->rings && opts->y_stride > 0) ctl.y_stride = opts->y_stride;
  else if (!y_stride_set) ctl.y_stride = s->model.output_dim;  // (the kernels read the stride as it is: never 0)
  ctl.flags |= ctl_flags;
  ctl.nsub = nsub;
  ctl.reset_mode = reset_mode;
  ctl.settle_steps = s->model.settle_steps < 0 ? 0 : s->model.settle_steps;
  ctl.seed = s->seed;
  ctl.mask = mask;
  ctl.reset_count = s->d_reset_count ? s->d_reset_count + ((opts && opts->env_first > 0) ? opts->env_first : 0) : nullptr;
  int rc;
  const long long alt = s->opt.get(TDS_OPT_ALT_BUILD, 0);
  if (alt != 0) {  // an experiment slot (tds_kernels.h): f64 plain kernels of the one instantiation the slot was built for
    tds_alt_launch_fn fn = alt >= 1 && alt <= TDS_ALT_SLOTS ? tds_alt_slot((int)alt) : nullptr;
    int key = 0;
    if (fn) (void)fn(nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, &key);
    const bool plain = !s->h64.is_floating && !s->h64.num_spherical && s->h64.num_bodies < 2;
    if (!fn || s->dtype != TDS_DTYPE_F64 || !plain || key != s->lanes * 100 + lds.NDP)
      return fail(TDS_ERR_UNSUPPORTED, "option alt_build: no such experiment slot in this library for this handle's kernels");
    rc = fn(s->d_model, &s->h64, &lds, s->lanes, x, y, actions, fb, obs, ovf, n, stream, &ctl, form, nullptr);
  } else if (s->dtype == TDS_DTYPE_F64)
    rc = tds_launch_step<double, double>((const DevModel<double> *)s->d_model, s->h64, lds, s->lanes,
                                         (const double *)x, (double *)y, (const double *)actions, (double *)fb,
                                         (double *)obs, (double *)ovf, n, stream, ctl, nullptr, form);
  else if (s->dtype == TDS_DTYPE_F64_REC32)
    rc = tds_launch_step<double, float>((const DevModel<double> *)s->d_model, s->h64, lds, s->lanes,
                                        (const float *)x, (float *)y, (const float *)actions, (float *)fb, (float *)obs,
                                        (double *)ovf, n, stream, ctl, nullptr, form);
  else
    rc = tds_launch_step<float, float>((const DevModel<float> *)s->d_model, s->h32, lds, s->lanes, (const float *)x,
                                       (float *)y, (const float *)actions, (float *)fb, (float *)obs,
                                       (float *)ovf, n, stream, ctl, nullptr, form);
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

int tds_hip_default_option(const char *key, long long value) {
  const int k = tds_opt_find(key);
  if (k < 0) return fail(TDS_ERR_INVALID_ARG, "unknown option '%s'", key ? key : "(null)");
  tds_opt_overrides().v[k] = value;
  return TDS_OK;
}
int tds_hip_set_option(tds_hip_sim_t *s, const char *key, long long value) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  const int k = tds_opt_find(key);
  if (k < 0) return fail(TDS_ERR_INVALID_ARG, "unknown option '%s'", key ? key : "(null)");
  if (tds_opt_rows()[k].create_time)
    return fail(TDS_ERR_INVALID_ARG, "option '%s' is fixed when a handle is created: tds_hip_default_option before tds_hip_create", key);
  if (s->opt.v[k] == value) return TDS_OK;
  if (k == TDS_OPT_LOOP_OCC && value == 1 && s->lds.NDP < 24)
    return fail(TDS_ERR_UNSUPPORTED, "option loop_occ = 1: no one-wavefront-per-SIMD step-loop build below 24 padded dof");
  // options that shape the shard layer's ring are read ONCE, when the ring is first used: afterwards a new value would be
  // ignored silently — refused instead (set them before the first tds_hip_shard_step_many, or with tds_hip_default_option)
  if (s->shard_ring_shaped && (k == TDS_OPT_SHARD_CHUNK || k == TDS_OPT_SHARD_INPLACE || k == TDS_OPT_SHARD_WAIT ||
                               k == TDS_OPT_SHARD_REGISTER || k == TDS_OPT_Y_STRIDE || k == TDS_OPT_SHARD_PEER ||
                               k == TDS_OPT_EXCHANGE_FIELDS))
    return fail(TDS_ERR_INVALID_ARG, "option '%s' shapes the shard's ring, which exists already: set it before the first "
                                     "tds_hip_shard_step_many", key);
  // (cached graphs / a pool laid out for the old value must not outlive it; alt_build: the cached step_many graphs
  //  replay the kernel of the build that was current when they were captured)
  if (k == TDS_OPT_GRAPH_CHAINS || k == TDS_OPT_STEP_MANY_LOOP || k == TDS_OPT_LOOP_W2 || k == TDS_OPT_LOOP_OCC || k == TDS_OPT_CHAIN_W2 ||
      k == TDS_OPT_NO_GRAPH_UPLOAD || k == TDS_OPT_ALT_BUILD) {
    DeviceGuard guard(s->device);
    (void)hipStreamSynchronize(s->stream);
    drop_graphs(s);
  }
  if (k == TDS_OPT_POOL_EVERY || k == TDS_OPT_POOL_HOST_LAG || k == TDS_OPT_POOL_LAG || k == TDS_OPT_POOL_CHUNK ||
      k == TDS_OPT_POOL_CAP || k == TDS_OPT_POOL_SLAB) {
    DeviceGuard guard(s->device);
    (void)hipStreamSynchronize(s->stream);
    pool_reset(s);
  }
  s->opt.v[k] = value;
  return TDS_OK;
}
int tds_hip_get_option(const tds_hip_sim_t *s, const char *key, long long *value, int *is_set) {
  if (!s || !value) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  const int k = tds_opt_find(key);
  if (k < 0) return fail(TDS_ERR_INVALID_ARG, "unknown option '%s'", key ? key : "(null)");
  *value = s->opt.v[k] == TDS_OPT_UNSET ? 0 : s->opt.v[k];
  if (is_set) *is_set = s->opt.v[k] != TDS_OPT_UNSET;
  return TDS_OK;
}
int tds_hip_option_count(void) { return TDS_OPT_COUNT; }
const char *tds_hip_option_name(int index) { return (index >= 0 && index < TDS_OPT_COUNT) ? tds_opt_rows()[index].key : nullptr; }

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
  if (dtype == TDS_DTYPE_F32) {
    // pure float arithmetic is a measured-only variant (it misses the 1e-6 contract, tds_hip.h): built for the plain and
    // the floating-base kernels, not for spherical joints or worlds of several bodies
    bool sph = false;
    for (int i = 0; i < model->num_links; ++i) sph = sph || model->links[i].joint_type == TDS_JOINT_SPHERICAL;
    if (sph || model->num_bodies >= 2)
      return fail(TDS_ERR_UNSUPPORTED, "TDS_DTYPE_F32 (pure float arithmetic) is not built for spherical joints or worlds of "
                                       "several bodies: use TDS_DTYPE_F64_REC32 (float records, double arithmetic)");
  }
  int ndev = tds_hip_device_count();
  if (ndev <= 0) return fail(TDS_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(TDS_ERR_INVALID_ARG, "device index out of range");
  DeviceGuard guard(device);  // (the caller's current device is restored on return)
  tds_hip_sim *s = new (std::nothrow) tds_hip_sim();  // value-initialised: both host models start zeroed
  if (!s) return fail(TDS_ERR_INVALID_ARG, "out of host memory");
  s->opt = tds_opt_snapshot();  // (override > environment > library default; the environment is not read again)
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
      if (prop.multiProcessorCount > 0) s->num_cus = prop.multiProcessorCount;
      // (gfx950: 160 KiB per compute unit, of which one workgroup may take 64 KiB without opting in)
      if (prop.maxSharedMemoryPerMultiProcessor >= 64 * 1024) s->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
    }
  }
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
  if (s->opt.is_set(TDS_OPT_NA_CAP)) {
    na_cap = (int)s->opt.v[TDS_OPT_NA_CAP];
  } else {
    const size_t budget = (160 * 1024) / 8;
    for (int cap = 8; cap >= 5; --cap)
      if ((size_t)layout(cap).stride * epw * celem <= budget) {
        na_cap = cap;
        break;
      }
  }
  s->lds = layout(na_cap);
  {
    // two-wavefront workgroups: plain kernels up to 18 padded dof; same row cap (the scratch slab is shared)
    const bool is_fl = c64 ? s->h64.is_floating : s->h32.is_floating;
    const bool is_sph = c64 ? s->h64.num_spherical != 0 : s->h32.num_spherical != 0;
    // (option w2: 0 never, 2 at any grid size, any other value: also for worlds without contact points)
    const bool w2_set = s->opt.is_set(TDS_OPT_W2);
    const long long w2_opt = s->opt.get(TDS_OPT_W2, 1);
    const bool is_two_w = c64 ? s->h64.num_bodies >= 2 : s->h32.num_bodies >= 2;
    // (a world without contact points leaves the helper wavefront only the visual poses: the one-wave form is then the
    //  faster one — pendulum5 x 4096: 10.2 vs 10.9 us — unless TDS_HIP_W2=1/2 insists)
    const bool has_cp = model->has_plane && (c64 ? s->h64.num_cp : s->h32.num_cp) > 0;
    // (round 6: two-wavefront builds of the general kernel up to 14 padded dof only — at 16 / 18 dof they carried 130 - 180 B of
    //  scratch per lane, profiles/r06_kernel_resources_f64_k0.txt; Laikago, the model they were for, has its own kernel)
    if (!is_fl && !is_sph && !is_two_w && s->lds.NDP <= 14 && w2_opt != 0 && (has_cp || w2_set)) {
      s->lds_w2 = c64 ? tds_make_lds_layout<double>(s->h64, na_cap, s->lanes, true)
                      : tds_make_lds_layout<float>(s->h32, na_cap, s->lanes, true);
      const size_t b2 = (size_t)s->lds_w2.stride * epw * celem;
      const int per_cu = b2 > 0 ? (int)((160 * 1024) / b2) : 0;
      // 256 CUs; two wavefronts per workgroup, four SIMDs per CU: the form pays while every wavefront is resident
      // with at most two per SIMD, i.e. up to four workgroups per CU
      const int wg_per_cu = per_cu < 4 ? per_cu : 4;
      if (b2 <= 64 * 1024 && wg_per_cu >= 1) s->w2_max_blocks = w2_opt == 2 ? (1 << 30) : wg_per_cu * 256;
      // the workgroup's constant table of the step-loop launches (TdsLds::cw): as many rows as the LDS left over by
      // wg_per_cu workgroups holds — never at the price of a workgroup per CU (LDS is granted in 512-byte units)
      for (const int rows : {TDS_CW_LANE(celem) + TDS_CW_XT, TDS_CW_LANE(celem)}) {
        const size_t with = ((b2 + (size_t)rows * s->lanes * celem + 511) / 512) * 512;
        if (wg_per_cu >= 1 && with <= 64 * 1024 && (size_t)wg_per_cu * with <= 160 * 1024) {
          s->lds_w2.cw = rows;
          break;
        }
      }
    }
  }
  // ... and of the one-wave step-loop launches: out of what the workgroups of the form's occupancy leave over (two
  // wavefronts per SIMD = eight one-wave workgroups per CU below 24 padded dof, four from there on)
  {
    const bool plain = c64 ? (!s->h64.is_floating && !s->h64.num_spherical && s->h64.num_bodies < 2)
                           : (!s->h32.is_floating && !s->h32.num_spherical && s->h32.num_bodies < 2);
    const size_t b1 = (size_t)s->lds.stride * epw * celem;
    const size_t per_cu = s->lds.NDP < 24 ? 8 : 4;
    for (const int rows : {TDS_CW_LANE(celem) + TDS_CW_XT, TDS_CW_LANE(celem)}) {
      const size_t with = ((b1 + (size_t)rows * s->lanes * celem + 511) / 512) * 512;
      if (plain && with <= 64 * 1024 && per_cu * with <= 160 * 1024) {
        s->lds.cw = rows;
        break;
      }
    }
  }
  const int lds_bytes = (int)((size_t)s->lds.stride * epw * celem);
  if (lds_bytes > 160 * 1024) {
    delete s;
    return fail(TDS_ERR_UNSUPPORTED, "model needs more than 160 KiB of LDS per workgroup");
  }
  if (lds_bytes > 64 * 1024) {
    const bool is_fl = c64 ? s->h64.is_floating : s->h32.is_floating;
    const bool is_sph = c64 ? s->h64.num_spherical != 0 : s->h32.num_spherical != 0;
    const bool is_two = c64 ? s->h64.num_bodies >= 2 : s->h32.num_bodies >= 2;
    const bool is_mfl = c64 ? s->h64.multi_floating != 0 : s->h32.multi_floating != 0;
    const int kind = is_fl ? 1 : (is_sph ? 2 : (is_two ? (is_mfl ? 4 : 3) : 0));
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
  pool_free(s);
  for (int c = 0; c < tds_hip_sim::kMaxChains; ++c)
    if (s->graph_exec[c]) (void)hipGraphExecDestroy(s->graph_exec[c]);
  if (s->graph_stream) (void)hipStreamDestroy(s->graph_stream);
  for (int c = 0; c < tds_hip_sim::kMaxChains - 1; ++c) {
    if (s->graph_chain[c]) (void)hipStreamDestroy(s->graph_chain[c]);
    if (s->graph_join[c]) (void)hipEventDestroy(s->graph_join[c]);
  }
  if (s->graph_fork) (void)hipEventDestroy(s->graph_fork);
  if (s->d_model) (void)hipFree(s->d_model);
  if (s->d_x) (void)hipFree(s->d_x);
  if (s->d_y) (void)hipFree(s->d_y);
  if (s->d_ovf) (void)hipFree(s->d_ovf);
  if (s->d_reset_count) (void)hipFree(s->d_reset_count);
  if (s->d_ro) (void)hipFree(s->d_ro);
  if (s->d_split) (void)hipFree(s->d_split);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->h_stage) (void)hipHostFree(s->h_stage);
  if (s->d_stage_act) (void)hipFree(s->d_stage_act);
  if (s->d_stage_obs) (void)hipFree(s->d_stage_obs);
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
  // the pre-settled reset states were settled with the gains (kp, kd, max_force) the records held when they were
  // staged: new records, new pool
  s->pool_ready = false;
  s->pool_discard = true;
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

// ======================================================================================================
// Reset pool — auto_reset_when_done (ars_vectorized_environment.h:262-277) at straight-line speed.
//
// The state an environment restarts from is a pure function of (seed, environment, reset count): the reset
// distribution drawn from the counter-based stream, then settle_steps zero-action steps
// (ant_environment2.h:109-165).  So it can be computed BEFORE it is needed.  Every environment owns a ring of D
// pre-settled states in HBM (entry k lives in slot k mod D; valid entries are [count, filled)); the straight-line step
// kernel turns "done" into a copy of the next entry (tds_kernels.hip, ctl.pool), and consumed entries are replaced in
// the background on a side stream: every R steps a PASS
//     plan     one thread per environment: claim the missing entries [filled, count + D) in a work list
//     stage    one thread per record component: the reset states of the work list into staging records
//     settle   settle_steps launches of the straight-line step kernel over the staging records (they co-reside with
//              the main launches: same kernel, two wavefronts per SIMD)
//     scatter  staging records -> ring slots
// The grids are exact: the host reads the size of a planned work list H steps after planning it (hipEventQuery,
// blocking only if the GPU is more than H steps behind the host — it never idles the GPU), then enqueues settle +
// scatter.  Before step t the step stream waits for pass floor((t - 1 - W) / R).  With D >= R + W an environment can
// never run out of entries — at most one is consumed per step — so the results are those of resetting inside the step
// launch (same stream of random numbers, same settle steps: bit for bit with f64 records; with float records the
// settle launches round the state to float between the settle steps, the in-kernel reset keeps it in double), whatever
// the rate of resets; a burst of resets only makes the step stream wait.
// The entries embed the gains the x records held at staging time: tds_hip_set_inputs discards the pool; a caller that
// rewrites the gains through the zero-copy x pointer must call tds_hip_set_auto_reset again.
// ======================================================================================================
extern "C++" {
namespace {

struct PoolDist {
  double q[TDS_MAX_DOF], noise[TDS_MAX_DOF];
};

__device__ __forceinline__ double pool_uniform01(unsigned long long seed, unsigned env, unsigned count, unsigned j) {
  // == tds_uniform01 of tds_kernels.hip (splitmix64 finaliser of (seed, env, reset count, coordinate))
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (((unsigned long long)env << 32) | (unsigned long long)(count * 64u + j + 1u));
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// one thread per environment: claim the missing entries of its ring in the work list (environment, ring slot, entry)
__global__ void tds_pool_plan_kernel(const unsigned *__restrict__ count, unsigned *__restrict__ filled, int depth,
                                     int n, int cap, int *__restrict__ items) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const unsigned c = count[e];
  unsigned f = filled[e];
  if (f < c) f = c;  // entries consumed behind the pool's back (forced reset, rollout with auto-reset)
  const int need = (int)(c + (unsigned)depth - f);
  if (need <= 0) {
    filled[e] = f;
    return;
  }
  const int base = atomicAdd(&items[0], need);
  int take = cap - base;
  take = take < 0 ? 0 : (take < need ? take : need);
  for (int j = 0; j < take; ++j) {
    const unsigned cc = f + (unsigned)j;
    const int it = base + j;
    items[1 + it] = e;
    items[1 + cap + it] = (int)(cc % (unsigned)depth);
    items[1 + 2 * cap + it] = (int)cc;
  }
  filled[e] = f + (unsigned)take;
}

// one thread per (work item, record component): the reset state of (seed, environment, entry) into the staging record
// T: compute scalar of the handle (the reset state is formed in it, as the step-loop kernel does), TR: record scalar
template <typename T, typename TR>
__global__ void tds_pool_stage_kernel(const int *__restrict__ items, int n_items, int cap, TR *__restrict__ stage,
                                      const TR *__restrict__ x, int in_dim, int nq, int nd, int adim, PoolDist dist,
                                      unsigned long long seed) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int it = (int)(idx / (size_t)in_dim), i = (int)(idx % (size_t)in_dim);
  if (it >= n_items) return;
  const int e = items[1 + it];
  const unsigned cc = (unsigned)items[1 + 2 * cap + it];
  TR v;
  if (i < nq) {
    const T u01 = (T)pool_uniform01(seed, (unsigned)e, cc, (unsigned)i);
    v = (TR)((T)dist.q[i] + (T)dist.noise[i] * ((u01 - T(0.5)) * T(2)));
  } else if (i < nq + nd + adim) {
    v = TR(0);  // qd = 0, zero action while settling
  } else {
    v = x[(size_t)e * in_dim + i];  // kp, kd, max_force of the environment
  }
  stage[(size_t)it * in_dim + i] = v;
}

template <typename TR>
__global__ void tds_pool_scatter_kernel(const int *__restrict__ items, int n_items, int cap,
                                        const TR *__restrict__ stage, int in_dim, int w, TR *__restrict__ pool, int n) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int it = (int)(idx / (size_t)w), i = (int)(idx % (size_t)w);
  if (it >= n_items) return;
  const int e = items[1 + it], slot = items[1 + cap + it];
  pool[((size_t)slot * n + e) * w + i] = stage[(size_t)it * in_dim + i];
}

int pool_param(const tds_hip_sim *s, int key, int dflt) {
  const int v = (int)s->opt.get(key, 0);
  return v > 0 ? v : dflt;
}

// ring depth D by form: single steps D >= R + W (+ slack); pool_step_many D >= 2 x chunk (+ slack)
void pool_params(tds_hip_sim *s) {
  if (s->pool_every > 0) return;
  s->pool_every = pool_param(s, TDS_OPT_POOL_EVERY, 16);                     // R
  s->pool_host_lag = pool_param(s, TDS_OPT_POOL_HOST_LAG, s->pool_every / 2);  // H
  if (s->pool_host_lag >= s->pool_every) s->pool_host_lag = s->pool_every - 1;
  if (s->pool_host_lag < 1) s->pool_host_lag = 1;
  const int settle = s->model.settle_steps > 0 ? s->model.settle_steps : 0;
  s->pool_lag = pool_param(s, TDS_OPT_POOL_LAG, s->pool_host_lag + settle + 6);  // W
  // (W > H: before step t the step stream waits for the event of pass floor((t - 1 - W) / R), which the host records at
  //  most H steps after planning that pass — with W <= H it would wait on an event this pass has not recorded yet, i.e.
  //  not at all, and a done environment could copy a ring slot that has not been refilled)
  if (s->pool_lag < s->pool_host_lag + 1) s->pool_lag = s->pool_host_lag + 1;
  s->pool_chunk = pool_param(s, TDS_OPT_POOL_CHUNK, 128);                      // steps per launch of pool_step_many
}
int pool_depth_for(const tds_hip_sim *s, bool many) {
  return many ? 2 * s->pool_chunk + 4 : s->pool_every + s->pool_lag + 4;
}

// (first use, or the form now in use needs deeper rings than the one the pool was set up for: the rings start empty)
int pool_alloc(tds_hip_sim *s) {
  pool_params(s);
  const size_t n = (size_t)s->num_envs, w = (size_t)(s->model.dof_q + s->model.dof_qd);
  const int need = pool_depth_for(s, s->pool_many);
  if (s->d_pool && s->pool_depth >= need) return TDS_OK;
  if (s->d_pool) {
    TDS_HIP_TRY(hipStreamSynchronize(s->stream));
    TDS_HIP_TRY(hipStreamSynchronize(s->pool_stream));
    TDS_HIP_TRY(hipFree(s->d_pool));
    s->d_pool = nullptr;
    s->pool_planned = 0;  // (its work list names slots of the old rings)
  }
  s->pool_depth = need;
  s->pool_discard = true;
  TDS_HIP_TRY(hipMalloc(&s->d_pool, (size_t)s->pool_depth * n * w * s->elem));
  if (s->d_pool_filled) return TDS_OK;
  // work list of a pass: what R + 4 single steps can consume; pool_step_many, whose two launches could consume more,
  // carries on with further passes when a list was cut short (more than 24 resets per environment on average)
  s->pool_cap = pool_param(s, TDS_OPT_POOL_CAP, (int)(n * (size_t)(s->pool_every + 4 > 24 ? s->pool_every + 4 : 24)));
  TDS_HIP_TRY(hipMalloc((void **)&s->d_pool_filled, n * sizeof(unsigned)));
  TDS_HIP_TRY(hipMalloc((void **)&s->d_pool_items, (1 + 3 * (size_t)s->pool_cap) * sizeof(int)));
  TDS_HIP_TRY(hipHostMalloc((void **)&s->h_pool_nitems, sizeof(int), 0));
  TDS_HIP_TRY(hipMalloc(&s->d_stage_x, (size_t)s->pool_cap * s->model.input_dim * s->elem));
  TDS_HIP_TRY(hipStreamCreateWithFlags(&s->pool_stream, hipStreamNonBlocking));
  for (int i = 0; i < tds_hip_sim::kPoolEvents; ++i)
    TDS_HIP_TRY(hipEventCreateWithFlags(&s->pool_ev[i], hipEventDisableTiming));
  TDS_HIP_TRY(hipEventCreateWithFlags(&s->pool_step_ev, hipEventDisableTiming));
  TDS_HIP_TRY(hipEventCreateWithFlags(&s->pool_plan_ev, hipEventDisableTiming));
  TDS_HIP_TRY(hipEventCreateWithFlags(&s->pool_sync_ev, hipEventDisableTiming));
  // LDS layout of the refill launches: every constraint row in LDS, so that they need no scratch slab
  const int epw = 64 / s->lanes;
  s->pool_lds = s->compute_f64() ? tds_make_lds_layout<double>(s->h64, 0, s->lanes)
                                 : tds_make_lds_layout<float>(s->h32, 0, s->lanes);
  // The refill launches run with the handle's own layout — 19.8 KB per Ant workgroup, two wavefronts per SIMD — and a
  // surplus-row slab of their own, sized for a whole work list, where that slab stays under 4 GiB (Ant: 4.6 KB per
  // entry, 0.46 GB at 4096 environments); otherwise (or with TDS_HIP_POOL_SLAB=0) with every constraint row in LDS
  // (37.7 KB, one wavefront per SIMD).  Ant x 8192 at 5 % resets per step: single steps 2.21e8 -> 2.73e8, step_many
  // 2.74e8 -> 3.01e8; x 4096 unchanged (the refill fits beside the steps either way).
  {
    const size_t per_env = (size_t)s->lds.ovrows * (s->lds.NDs + 3) * (s->compute_f64() ? 8 : 4);
    if (s->opt.get(TDS_OPT_POOL_SLAB, 1) != 0 && per_env > 0 && per_env * (size_t)s->pool_cap <= ((size_t)4 << 30)) {
      TDS_HIP_TRY(hipMalloc(&s->d_pool_ovf, per_env * (size_t)s->pool_cap));
      s->pool_lds = s->lds;
    }
  }
  const int lds_bytes = (int)((size_t)s->pool_lds.stride * epw * (s->compute_f64() ? 8 : 4));
  if (lds_bytes > 160 * 1024) return fail(TDS_ERR_UNSUPPORTED, "reset pool: the refill launches need more than 160 KiB of LDS");
  if (lds_bytes > 64 * 1024) {
    const bool is_fl = s->compute_f64() ? s->h64.is_floating : s->h32.is_floating;
    const bool is_sph = s->compute_f64() ? s->h64.num_spherical != 0 : s->h32.num_spherical != 0;
    const bool is_two = s->compute_f64() ? s->h64.num_bodies >= 2 : s->h32.num_bodies >= 2;
    const bool is_mfl = s->compute_f64() ? s->h64.multi_floating != 0 : s->h32.multi_floating != 0;
    const int kind = is_fl ? 1 : (is_sph ? 2 : (is_two ? (is_mfl ? 4 : 3) : 0));
    const int e = s->dtype == TDS_DTYPE_F64         ? tds_kernel_max_dynamic_lds<double, double>(s->lanes, s->pool_lds.NDP, lds_bytes, kind)
                  : s->dtype == TDS_DTYPE_F64_REC32 ? tds_kernel_max_dynamic_lds<double, float>(s->lanes, s->pool_lds.NDP, lds_bytes, kind)
                                                    : tds_kernel_max_dynamic_lds<float, float>(s->lanes, s->pool_lds.NDP, lds_bytes, kind);
    if (e != 0) return fail(TDS_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed (reset pool)");
  }
  return TDS_OK;
}

// pool stream: plan the work list of a pass (after everything the step stream has enqueued so far)
int pool_plan(tds_hip_sim *s) {
  TDS_HIP_TRY(hipEventRecord(s->pool_step_ev, s->stream));
  TDS_HIP_TRY(hipStreamWaitEvent(s->pool_stream, s->pool_step_ev, 0));
  TDS_HIP_TRY(hipMemsetAsync(s->d_pool_items, 0, sizeof(int), s->pool_stream));
  const int n = s->num_envs;
  hipLaunchKernelGGL(tds_pool_plan_kernel, dim3((n + 127) / 128), dim3(128), 0, s->pool_stream, s->d_reset_count,
                     s->d_pool_filled, s->pool_depth, n, s->pool_cap, s->d_pool_items);
  if (hipGetLastError() != hipSuccess) return fail(TDS_ERR_HIP, "reset pool: plan kernel launch");
  TDS_HIP_TRY(hipMemcpyAsync(s->h_pool_nitems, s->d_pool_items, sizeof(int), hipMemcpyDeviceToHost, s->pool_stream));
  TDS_HIP_TRY(hipEventRecord(s->pool_plan_ev, s->pool_stream));
  return TDS_OK;
}

// pool stream: settle + scatter the planned work list (its size is known on the host), then signal `done`
int pool_run(tds_hip_sim *s, hipEvent_t done) {
  int n_items = *s->h_pool_nitems;
  if (n_items > s->pool_cap) n_items = s->pool_cap;
  if (n_items > 0) {
    {  // reset states of the work list into the staging records
      PoolDist dist;
      for (int i = 0; i < TDS_MAX_DOF; ++i) {
        dist.q[i] = s->model.reset_q[i];
        dist.noise[i] = s->model.reset_noise[i];
      }
      const int in_dim = s->model.input_dim, nq = s->model.dof_q, nd = s->model.dof_qd;
      const size_t total = (size_t)n_items * in_dim;
      const dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define TDS_STAGE(TT, RR)                                                                                            \
  hipLaunchKernelGGL((tds_pool_stage_kernel<TT, RR>), grid, block, 0, s->pool_stream, s->d_pool_items, n_items,         \
                     s->pool_cap, (RR *)s->d_stage_x, (const RR *)s->d_x, in_dim, nq, nd, s->model.action_dim, dist, s->seed)
      if (s->dtype == TDS_DTYPE_F64)
        TDS_STAGE(double, double);
      else if (s->dtype == TDS_DTYPE_F64_REC32)
        TDS_STAGE(double, float);
      else
        TDS_STAGE(float, float);
#undef TDS_STAGE
      if (hipGetLastError() != hipSuccess) return fail(TDS_ERR_HIP, "reset pool: stage kernel launch");
    }
    LaunchOpts o;
    o.other_stream = true;
    o.stream = s->pool_stream;
    o.lds = &s->pool_lds;
    o.ovf = s->d_pool_ovf;
    // straight-line step kernel on the staging records: zero action, state fed back in place, no y / obs record
    // (the settle steps as ONE launch of the step-loop build; option pool_settle_loop = 0: one straight-line launch per
    //  settle step — round 4's form, from when the passes ran beside the chunks)
    const bool one_launch = s->model.settle_steps > 1 && s->opt.get(TDS_OPT_POOL_SETTLE_LOOP, 1) == 1;
    for (int k = 0; k < (one_launch ? 1 : s->model.settle_steps); ++k) {
      const int rc = launch(s, s->d_stage_x, nullptr, nullptr, s->d_stage_x, nullptr, n_items,
                            one_launch ? s->model.settle_steps : 1, TDS_RESET_NONE, nullptr, nullptr, 0, &o);
      if (rc != TDS_OK) return rc;
    }
    const int w = s->model.dof_q + s->model.dof_qd;
    const size_t total = (size_t)n_items * w;
    if (s->records_f64())
      hipLaunchKernelGGL(tds_pool_scatter_kernel<double>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                         s->pool_stream, s->d_pool_items, n_items, s->pool_cap, (const double *)s->d_stage_x,
                         s->model.input_dim, w, (double *)s->d_pool, s->num_envs);
    else
      hipLaunchKernelGGL(tds_pool_scatter_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                         s->pool_stream, s->d_pool_items, n_items, s->pool_cap, (const float *)s->d_stage_x,
                         s->model.input_dim, w, (float *)s->d_pool, s->num_envs);
    if (hipGetLastError() != hipSuccess) return fail(TDS_ERR_HIP, "reset pool: scatter kernel launch");
  }
  if (done) TDS_HIP_TRY(hipEventRecord(done, s->pool_stream));
  return TDS_OK;
}

// fill every ring completely (first use, new seed, or after resets behind the pool's back); host-synchronous, rare
int pool_fill(tds_hip_sim *s) {
  int rc = pool_alloc(s);
  if (rc != TDS_OK) return rc;
  if (s->pool_planned) {  // a pass in flight: finish it first
    TDS_HIP_TRY(hipEventSynchronize(s->pool_plan_ev));
    rc = pool_run(s, nullptr);
    if (rc != TDS_OK) return rc;
    s->pool_planned = 0;
  }
  if (s->pool_discard) {
    TDS_HIP_TRY(hipMemsetAsync(s->d_pool_filled, 0, (size_t)s->num_envs * sizeof(unsigned), s->pool_stream));
    s->pool_discard = false;
  }
  for (int guard = 0; guard < 4096; ++guard) {
    rc = pool_plan(s);
    if (rc != TDS_OK) return rc;
    TDS_HIP_TRY(hipEventSynchronize(s->pool_plan_ev));
    if (*s->h_pool_nitems <= 0) break;
    rc = pool_run(s, nullptr);
    if (rc != TDS_OK) return rc;
  }
  // the step stream continues only after the last scatter
  TDS_HIP_TRY(hipEventRecord(s->pool_sync_ev, s->pool_stream));
  TDS_HIP_TRY(hipStreamWaitEvent(s->stream, s->pool_sync_ev, 0));
  s->pool_step = 0;
  s->pool_waited = 0;
  s->pool_run_pending = false;  // (the wait above covers every pass issued before it: the pool stream is in order)
  s->pool_ready = true;
  return TDS_OK;
}

// one auto-reset step through the pool
// C > 1 (inside tds_hip_step_many, between its fork and join): the step as C launches over contiguous environment ranges, chain c
// on its own stream (chain_streams) — as the graphs of the plain call run it: a chain's kernel boundary is filled by the other
// chain's workgroups (laikago_soft x 8192 with auto-reset: 25.2 -> see DESIGN 6)
int pool_step(tds_hip_sim *s, const void *actions_dev, void *obs_dev, void *y_dev = nullptr, int y_stride = 0, int C = 1) {
  int rc;
  if (s->pool_many) {  // (the step_many form keeps its own pass schedule: start again from full rings)
    s->pool_many = false;
    s->pool_ready = false;
  }
  if (!s->pool_ready) {
    rc = pool_fill(s);
    if (rc != TDS_OK) return rc;
  }
  const long long t = ++s->pool_step;
  const int R = s->pool_every, W = s->pool_lag, H = s->pool_host_lag;
  // a planned pass goes out as soon as the size of its work list has reached the host (at the latest H steps after
  // it was planned: then the host waits for the GPU to get there — the step stream still has H steps queued)
  if (s->pool_planned) {
    const bool due = t - s->pool_planned_at >= H;
    if (due) TDS_HIP_TRY(hipEventSynchronize(s->pool_plan_ev));
    if (due || hipEventQuery(s->pool_plan_ev) == hipSuccess) {
      rc = pool_run(s, s->pool_ev[s->pool_planned % tds_hip_sim::kPoolEvents]);
      if (rc != TDS_OK) return rc;
      s->pool_planned = 0;
    }
  }
  // entries this step may consume were produced by passes <= floor((t - 1 - W) / R)
  const long long jstar = (t - 1 - W) / R;
  while (t - 1 - W >= R && s->pool_waited < jstar) {
    ++s->pool_waited;
    for (int c = 0; c < C; ++c)
      TDS_HIP_TRY(hipStreamWaitEvent(c == 0 ? s->stream : s->graph_chain[c - 1], s->pool_ev[s->pool_waited % tds_hip_sim::kPoolEvents], 0));
  }
  TdsStepCtl extra;
  memset(&extra, 0, sizeof(extra));
  extra.pool = s->d_pool;
  extra.pool_depth = s->pool_depth;
  extra.pool_envs = s->num_envs;
  const int epb = 64 / s->lanes, n_blocks = (s->num_envs + epb - 1) / epb;
  for (int c = 0; c < C; ++c) {
    LaunchOpts o;
    o.extra = &extra;
    o.y_stride = y_dev ? y_stride : 0;
    int e0 = 0, e1 = s->num_envs;
    if (C > 1) {
      const int b0 = (int)((long long)n_blocks * c / C), b1 = (int)((long long)n_blocks * (c + 1) / C);
      e0 = b0 * epb;
      e1 = (b1 * epb < s->num_envs) ? b1 * epb : s->num_envs;
      if (e1 <= e0) continue;
      o.env_total = s->num_envs;
      o.env_first = e0;
      o.other_stream = true;
      o.stream = c == 0 ? s->stream : s->graph_chain[c - 1];
    }
    rc = launch(s, s->d_x, y_dev ? y_dev : s->d_y, actions_dev, s->d_x, obs_dev ? obs_dev : s->d_split, e1 - e0, 1,
                TDS_RESET_NONE, nullptr, nullptr, 0, &o);
    if (rc != TDS_OK) return rc;
  }
  if (t % R == 0) {  // pass t / R: plan now, launch when its size is known
    for (int c = 1; c < C; ++c) {  // (the plan reads every environment's reset count: behind step t of every chain)
      TDS_HIP_TRY(hipEventRecord(s->graph_join[c - 1], s->graph_chain[c - 1]));
      TDS_HIP_TRY(hipStreamWaitEvent(s->stream, s->graph_join[c - 1], 0));
    }
    rc = pool_plan(s);
    if (rc != TDS_OK) return rc;
    s->pool_planned = t / R;
    s->pool_planned_at = t;
  }
  return TDS_OK;
}

// K auto-reset steps as step-loop launches ("chunks") of up to R = pool_chunk steps each (tds_hip_step_many of a handle with auto-reset on):
// inside a launch a done environment copies its next ring entry into its LDS record and carries on (tds_kernels.hip,
// pool_r), between the launches the rings are topped up by the same passes as above.
//
// Schedule (round 5).  The two-wavefront step-loop launch holds every wave slot of the GPU (two 256-register wavefronts
// per SIMD), so a refill pass cannot run BESIDE a chunk any more: it runs between two chunks, and it takes ten settle
// launches one after the other (~12 us each however few environments they settle).  A pass per call — round 4's
// schedule — therefore cost a 20-step call 130 us on top of its 317 us (profiles/r05_ant4096_f64_default_dispatches.txt:
// auto-reset rate 0.73 of the plain rate).  Now a pass is planned only when R steps have been launched since the last
// snapshot, across calls: the same ten settle launches then settle R x (done per step) environments instead of
// 20 x (done per step), i.e. the GPU-wide latency of a pass is paid once per R steps, and the host never blocks on a plan
// behind the chunk it has just launched unless the rings would otherwise run dry:
//     total    steps launched since the rings were filled
//     visible  value of `total` at the snapshot of the newest pass the NEXT launch waits for
//     before a launch of k steps:  total + k - visible > D - 4  ->  bring a newer pass in first (plan if none is, wait
//                                  for its size, run it): an environment consumes at most one entry per step, so
//                                  D entries cannot run dry
//     after it:                    a planned pass whose size has reached the host (always, if the caller synchronised
//                                  since; otherwise only between the chunks of one call) is run behind the launch;
//                                  with none planned and total - (last snapshot) >= R, plan one
// A pass only overwrites slots whose entries were consumed before it was planned: the results are those of resetting
// inside the step, whatever the rate of resets and however the calls are cut (tests/test_auto_reset.py).
int pool_make_visible(tds_hip_sim *s) {  // the planned pass (planned here if there is none): size to the host, run it
  int rc;
  if (!s->pool_planned) {
    rc = pool_plan(s);
    if (rc != TDS_OK) return rc;
    s->pool_planned = 1;
    s->pool_snap_planned = s->pool_total;
  }
  TDS_HIP_TRY(hipEventSynchronize(s->pool_plan_ev));
  rc = pool_run(s, s->pool_sync_ev);
  if (rc != TDS_OK) return rc;
  s->pool_planned = 0;
  s->pool_run_pending = true;
  const long long snap = s->pool_snap_planned;
  while (*s->h_pool_nitems > s->pool_cap) {  // the work list was cut short: the rings must be FULL up to the snapshot
    rc = pool_plan(s);
    if (rc != TDS_OK) return rc;
    TDS_HIP_TRY(hipEventSynchronize(s->pool_plan_ev));
    rc = pool_run(s, s->pool_sync_ev);
    if (rc != TDS_OK) return rc;
  }
  s->pool_snap_visible = snap;
  return TDS_OK;
}
int pool_step_many(tds_hip_sim *s, const void *actions_dev, int act_blocks, int act_first, int n_steps, void *obs_dev,
                   const tds_hip_rings_t *rings = nullptr) {
  int rc;
  if (!s->pool_many) {
    s->pool_many = true;
    s->pool_ready = false;
  }
  if (!s->pool_ready) {
    rc = pool_fill(s);
    if (rc != TDS_OK) return rc;
    s->pool_many_chunks = 0;
    s->pool_total = s->pool_snap_visible = s->pool_snap_planned = 0;
  }
  TdsStepCtl extra;
  memset(&extra, 0, sizeof(extra));
  extra.pool = s->d_pool;
  extra.pool_depth = s->pool_depth;
  extra.pool_envs = s->num_envs;
  const size_t blk = (size_t)s->num_envs * s->model.action_dim * s->elem;
  for (int done = 0; done < n_steps;) {
    const int R = s->pool_chunk;
    const int k = n_steps - done < R ? n_steps - done : R;
    if (s->pool_total + k - s->pool_snap_visible > (long long)s->pool_depth - 4) {
      rc = pool_make_visible(s);
      if (rc != TDS_OK) return rc;
    }
    if (s->pool_run_pending) {
      TDS_HIP_TRY(hipStreamWaitEvent(s->stream, s->pool_sync_ev, 0));
      s->pool_run_pending = false;
    }
    LaunchOpts o;
    o.extra = &extra;
    o.rings = rings;
    o.ring_step0 = done;
    const void *a0 = nullptr;
    if (actions_dev) {
      const int first = (act_first + done) % act_blocks;
      a0 = (const char *)actions_dev + (size_t)first * blk;
      o.act_pool = actions_dev;
      o.act_blocks = act_blocks;
      o.act_first = first;
    }
    rc = launch(s, s->d_x, s->d_y, a0, s->d_x, obs_dev ? obs_dev : s->d_split, s->num_envs, k, TDS_RESET_NONE, nullptr,
                nullptr, 0, &o);
    if (rc != TDS_OK) return rc;
    ++s->pool_many_chunks;
    s->pool_total += k;
    done += k;
    // (between the chunks of one call the host waits for the size — the chunk just launched keeps the GPU busy meanwhile;
    //  behind the last chunk of a call it only looks: a caller that synchronises between calls finds it there next time)
    if (s->pool_planned && (done < n_steps || hipEventQuery(s->pool_plan_ev) == hipSuccess)) {
      rc = pool_make_visible(s);
      if (rc != TDS_OK) return rc;
    }
    if (!s->pool_planned && s->pool_total - s->pool_snap_visible >= R) {
      rc = pool_plan(s);  // what the launches so far have consumed
      if (rc != TDS_OK) return rc;
      s->pool_planned = 1;
      s->pool_snap_planned = s->pool_total;
    }
  }
  return TDS_OK;
}

void pool_free(tds_hip_sim *s) {
  if (s->pool_stream) (void)hipStreamSynchronize(s->pool_stream);
  if (s->d_pool) (void)hipFree(s->d_pool);
  if (s->d_pool_filled) (void)hipFree(s->d_pool_filled);
  if (s->d_pool_items) (void)hipFree(s->d_pool_items);
  if (s->h_pool_nitems) (void)hipHostFree(s->h_pool_nitems);
  if (s->d_stage_x) (void)hipFree(s->d_stage_x);
  if (s->d_pool_ovf) (void)hipFree(s->d_pool_ovf);
  for (int i = 0; i < tds_hip_sim::kPoolEvents; ++i)
    if (s->pool_ev[i]) (void)hipEventDestroy(s->pool_ev[i]);
  if (s->pool_step_ev) (void)hipEventDestroy(s->pool_step_ev);
  if (s->pool_plan_ev) (void)hipEventDestroy(s->pool_plan_ev);
  if (s->pool_sync_ev) (void)hipEventDestroy(s->pool_sync_ev);
  if (s->pool_stream) (void)hipStreamDestroy(s->pool_stream);
}
void pool_reset(tds_hip_sim *s) {
  pool_free(s);
  s->d_pool = s->d_stage_x = s->d_pool_ovf = nullptr;
  s->d_pool_filled = nullptr;
  s->d_pool_items = nullptr;
  s->h_pool_nitems = nullptr;
  for (int i = 0; i < tds_hip_sim::kPoolEvents; ++i) s->pool_ev[i] = nullptr;
  s->pool_step_ev = s->pool_plan_ev = s->pool_sync_ev = nullptr;
  s->pool_stream = nullptr;
  s->pool_depth = s->pool_every = s->pool_lag = s->pool_host_lag = s->pool_cap = s->pool_chunk = 0;
  s->pool_step = s->pool_waited = s->pool_planned = s->pool_planned_at = s->pool_many_chunks = 0;
  s->pool_many = false;
  s->pool_ready = false;
  s->pool_discard = true;
}

}  // namespace
}  // extern "C++"

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
  // Auto-reset: default = the reset pool (straight-line kernel, reset states computed ahead of time, see above).
  // TDS_HIP_AUTO_RESET_SPLIT=0: reset + settle inside the step launch (step-loop build); =1: straight-line launch
  // followed by a forced-reset launch masked with the done flags; =2 / unset: pool.  All three draw the same stream
  // of random numbers (seed, environment, reset counter) and are held to the same host emulation by the tests.
  if (s->auto_reset && substeps == 1) {
    const long long split = s->opt.get(TDS_OPT_AUTO_RESET_SPLIT, 2);
    if (split == 2) return pool_step(s, actions_dev, obs_dev);
    s->pool_ready = false;  // (entries are consumed behind the pool's back)
    if (split == 1) {
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
  if (s->auto_reset) s->pool_ready = false;
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
int graph_matches(const tds_hip_sim *s, const void *actions, int pool, int first, int n_steps, void *obs,
                  const tds_hip_rings_t *rings = nullptr) {
  tds_hip_rings_t none;
  memset(&none, 0, sizeof(none));
  return s->graph_exec[0] && s->graph_actions == actions && s->graph_pool == pool && s->graph_first == first &&
         s->graph_steps == n_steps && s->graph_obs == obs &&
         memcmp(&s->graph_rings, rings ? rings : &none, sizeof(none)) == 0;
}
// slot of a record ring that step k of a call owns
void *ring_slot(const tds_hip_sim *s, void *ring, int slots, int first, int k, size_t scalars_per_env) {
  return ring ? (char *)ring + (size_t)((first + k) % slots) * s->num_envs * scalars_per_env * s->elem : nullptr;
}
// scalars between consecutive records of a y ring (tds_hip_rings_t::y_stride; 0: packed)
size_t y_width(const tds_hip_sim *s, const tds_hip_rings_t *r) {
  return (r && r->y_stride > 0) ? (size_t)r->y_stride : (size_t)s->model.output_dim;
}
// The environments are independent and a step_many call holds K steps of each: enqueue them as C chains (contiguous
// environment ranges, one stream / graph branch each) instead of K whole-batch launches.  A chain's kernel boundary
// (launch latency ~1.1 us, workgroup dispatch ramp ~1.4 us, the wait for its slowest workgroup ~1 us: a sixth of a
// 20 us step, profiles/r02d_*) is then filled by the other chains' workgroups instead of idling the whole GPU.
int chain_count(const tds_hip_sim *s, int n_steps) {
  const int epb = 64 / s->lanes, n_blocks = (s->num_envs + epb - 1) / epb;
  // default: two chains for models with contact points (Ant x 2048 ... 16384: +4 ... +24 %, laikago_soft x 8192 +22 %;
  // a kernel as short as pendulum5's 11 us loses 15 %: profiles/r02d_graph_chains.txt); four and more chains collapse
  int c = s->chains_wanted > 0 ? s->chains_wanted : (s->model.num_geoms > 0 && s->model.has_plane ? 2 : 1);
  if (s->opt.is_set(TDS_OPT_GRAPH_CHAINS)) c = (int)s->opt.v[TDS_OPT_GRAPH_CHAINS];
  if (c > tds_hip_sim::kMaxChains) c = tds_hip_sim::kMaxChains;
  if (c > n_blocks) c = n_blocks;
  if (c < 1 || n_steps < 2) c = 1;
  return c;
}
int chain_streams(tds_hip_sim *s, int n_chains) {
  if (n_chains > 1 && !s->graph_fork) HIP_TRY(hipEventCreateWithFlags(&s->graph_fork, hipEventDisableTiming));
  for (int c = 1; c < n_chains; ++c) {
    if (!s->graph_chain[c - 1]) HIP_TRY(hipStreamCreateWithFlags(&s->graph_chain[c - 1], hipStreamNonBlocking));
    if (!s->graph_join[c - 1]) HIP_TRY(hipEventCreateWithFlags(&s->graph_join[c - 1], hipEventDisableTiming));
  }
  return TDS_OK;
}
// K steps of chain c of C on `stream`
int enqueue_chain(tds_hip_sim *s, int c, int C, hipStream_t stream, const void *actions, int pool, int first, int n_steps,
                  void *obs, const tds_hip_rings_t *rings = nullptr) {
  const size_t blk = (size_t)s->num_envs * s->model.action_dim * s->elem;
  const int epb = 64 / s->lanes, n_blocks = (s->num_envs + epb - 1) / epb;
  const int b0 = (int)((long long)n_blocks * c / C), b1 = (int)((long long)n_blocks * (c + 1) / C);
  const int e0 = b0 * epb, e1 = (b1 * epb < s->num_envs) ? b1 * epb : s->num_envs;
  if (e1 <= e0) return TDS_OK;
  LaunchOpts lo;
  lo.env_total = s->num_envs;
  lo.env_first = e0;
  lo.other_stream = true;
  lo.stream = stream;
  for (int k = 0; k < n_steps; ++k) {
    const void *a = actions ? (const char *)actions + (size_t)((first + k) % pool) * blk : nullptr;
    // (record rings: every launch is pointed at the slots of its step)
    void *const ob = (rings && rings->obs_ring) ? ring_slot(s, rings->obs_ring, rings->obs_slots, rings->obs_first, k, s->obs_width()) : obs;
    void *const yk = (rings && rings->y_ring) ? ring_slot(s, rings->y_ring, rings->y_slots, rings->y_first, k, y_width(s, rings)) : s->d_y;
    lo.y_stride = (rings && rings->y_ring && rings->y_stride > 0) ? rings->y_stride : 0;
    const int rc = launch(s, s->d_x, yk, a, s->d_x, ob, e1 - e0, 1, TDS_RESET_NONE, nullptr, nullptr, 0, &lo);
    if (rc != TDS_OK) return rc;
  }
  return TDS_OK;
}

void drop_graphs(tds_hip_sim *s) {
  for (int c = 0; c < tds_hip_sim::kMaxChains; ++c) {
    if (s->graph_exec[c]) (void)hipGraphExecDestroy(s->graph_exec[c]);
    s->graph_exec[c] = nullptr;
  }
  s->graph_chains = 0;
}

// one LINEAR graph per chain (a single graph with parallel branches costs ~2 us more per step than the same chains
// as independent launches, tools/ubench/two_streams.hip; linear graphs replay at the single-stream rate)
int build_graph(tds_hip_sim *s, const void *actions, int pool, int first, int n_steps, void *obs,
                const tds_hip_rings_t *rings = nullptr) {
  drop_graphs(s);
  if (!s->graph_stream) HIP_TRY(hipStreamCreateWithFlags(&s->graph_stream, hipStreamNonBlocking));
  const int n_chains = chain_count(s, n_steps);
  int rc = chain_streams(s, n_chains);
  if (rc != TDS_OK) return rc;
  for (int c = 0; c < n_chains; ++c) {
    // capture on a private stream (the handle's stream may be the NULL stream, which cannot be captured);
    // the launches are recorded, not executed
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamBeginCapture(s->graph_stream, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
      rc = enqueue_chain(s, c, n_chains, s->graph_stream, actions, pool, first, n_steps, obs, rings);
      e = hipStreamEndCapture(s->graph_stream, &graph);
    }
    if (e == hipSuccess && rc == TDS_OK && graph) e = hipGraphInstantiate(&s->graph_exec[c], graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    // (device-side copy of the executable graph made now, not by the first launch: a short replay — the 20-step runs of
    //  a benchmark driver — otherwise pays it inside its timed region)
    if (e == hipSuccess && rc == TDS_OK && s->graph_exec[c] && !s->opt.flag(TDS_OPT_NO_GRAPH_UPLOAD))
      (void)hipGraphUpload(s->graph_exec[c], s->graph_stream);
    if (e != hipSuccess || rc != TDS_OK || !s->graph_exec[c]) {
      if (rc == TDS_OK) snprintf(g_err, sizeof(g_err), "graph capture / instantiation failed: %s", hipGetErrorString(e));
      drop_graphs(s);
      return TDS_ERR_HIP;
    }
  }
  s->graph_actions = actions;
  s->graph_pool = pool;
  s->graph_first = first;
  s->graph_steps = n_steps;
  s->graph_obs = obs;
  s->graph_chains = n_chains;
  if (rings) s->graph_rings = *rings; else memset(&s->graph_rings, 0, sizeof(s->graph_rings));
  return TDS_OK;
}
}  // namespace
}  // extern "C++"

extern "C++" {
namespace {
// K steps as ONE launch of the step-loop build, every step taking its own action block (TdsStepCtl::act_pool): no kernel
// boundaries at all, the state stays in LDS between the steps.  Always for worlds without contact points (pendulums,
// the cartpole: ~7 us step kernels of which a boundary is a third), and for narrow kernels with contacts while the
// batch is at most three rounds of workgroups (see below).  TDS_HIP_STEP_MANY_LOOP=0 / 1 forbids / forces it.
bool step_many_as_loop(const tds_hip_sim *s, int n_steps) {
  if (n_steps < 2) return false;
  if (s->opt.is_set(TDS_OPT_STEP_MANY_LOOP)) return s->opt.v[TDS_OPT_STEP_MANY_LOOP] == 1;
  const int ncp = s->compute_f64() ? s->h64.num_cp : s->h32.num_cp;
  const bool two = s->compute_f64() ? s->h64.num_bodies >= 2 : s->h32.num_bodies >= 2;
  if (!(s->model.has_plane && ncp > 0) && !two) return true;
  // Worlds with contacts, kernels up to 16 dof (their step-loop builds fit the registers: 256 VGPR + 12 AGPR at one
  // wavefront per SIMD, 52 B of scratch at two): one launch beats the chained graphs up to three rounds of workgroups
  // (Ant x 2048 / 4096 / 8192: 14.4 / 14.9 / 20.3 us per step against 15.3 / 16.6 / 23.2; x 16384: 39.0 against 36.6).
  // Wider kernels (Laikago, 18 dof) spill in the loop build and stay with the graphs (66 against 49 us).
  const bool plain = !two && !(s->compute_f64() ? s->h64.is_floating : s->h32.is_floating) &&
                     (s->compute_f64() ? s->h64.num_spherical : s->h32.num_spherical) == 0;
  // the star-shaped legged robots (tds_quad.hip; 248 VGPR, no scratch in its step-loop form): the step-loop form while
  // EVERY workgroup of the launch is resident at once — its constant table costs LDS: six workgroups per compute unit
  // instead of the straight-line form's eight — and the chained graphs (single steps through the reset pool with
  // auto-reset on) beyond that, where the loop form would run its workgroups in two rounds of all the steps each.
  // laikago_soft (tools/quad_occupancy_sweep.sh, us per step, loop / graphs): x 4096 13.4 / 20.8, x 6144 18.3 / 23.2,
  // x 8192 32.5 / 24.7; with auto-reset: 13.2 / 21.0, 17.8 / 25.8, 30.9 / 27.4.  Option step_many_loop = 0 / 1 forces a form.
  if (s->compute_f64() && s->h64.quad) {
    return quad_loop_waves(s, s->num_envs) != 0;
  }
  // the 8-lane kernel of the stars with two-link legs (tds_oct.hip: the Ant): always one launch.  Its straight-line form costs
  // the same table copy and workgroup rounds per step plus a kernel boundary and the state's round trip through HBM, so
  // beyond one round of resident workgroups (8192 environments) R rounds of K steps still beat K launches of R rounds
  if (s->compute_f64() && (s->h64.oct || s->h64.chain)) return true;  // (and the serial-chain kernel, tds_chain.hip)
  const int n_blocks = (s->num_envs + (64 / s->lanes) - 1) / (64 / s->lanes);
  // With auto-reset on the alternative is not the chained graphs but single steps through the reset pool: the step-loop
  // launches (pool_step_many) win at every batch size (Ant x 16384 / 32768 at 5 % resets per step: 2.81e8 / 2.87e8
  // against 2.29e8 / 2.40e8; with hardly any resets 3.97e8 / 4.07e8 against 3.39e8 / 3.76e8)
  return plain && s->lds.NDP <= 16 && (n_blocks <= 3072 || s->auto_reset);
}
}  // namespace
}  // extern "C++"

extern "C++" {
namespace {
int rings_check(const tds_hip_sim *s, const tds_hip_rings_t *r, int n_steps) {
  if (!r) return TDS_OK;
  if (r->obs_ring && (r->obs_slots < 1 || r->obs_first < 0)) return fail(TDS_ERR_INVALID_ARG, "record rings: obs_slots must be >= 1, obs_first >= 0");
  if (r->y_ring && (r->y_slots < 1 || r->y_first < 0)) return fail(TDS_ERR_INVALID_ARG, "record rings: y_slots must be >= 1, y_first >= 0");
  if (r->y_stride != 0 && r->y_stride < s->model.output_dim)
    return fail(TDS_ERR_INVALID_ARG, "record rings: y_stride must be 0 (packed) or >= output_dim");
  // (a call with auto-reset on is cut into several launches, none of which signals its last step: the counter would fall
  //  one workgroup count behind per launch)
  if (r->progress && s->auto_reset)
    return fail(TDS_ERR_INVALID_ARG, "record rings: a progress counter cannot be combined with auto-reset");
  // (one counter per slot of the OBS ring: without that ring the device would index the counters modulo zero)
  if (r->progress && (!r->obs_ring || r->obs_slots < 1))
    return fail(TDS_ERR_INVALID_ARG, "record rings: a progress counter needs an obs ring (one counter per obs slot)");
  const bool loop = step_many_as_loop(s, n_steps) || (n_steps == 1 && step_many_as_loop(s, 2));
  if (r->obs_slot_envs != 0 && (r->obs_slot_envs < s->num_envs || !loop))
    return fail(TDS_ERR_INVALID_ARG, "record rings: obs_slot_envs must be 0 or >= num_envs, step-loop form only");
  if (r->progress && !loop) return fail(TDS_ERR_INVALID_ARG, "record rings: a progress counter needs the step-loop form (tds_hip_step_many_is_loop)");
  if (r->obs_f32 && s->elem == 8 && !loop)
    return fail(TDS_ERR_INVALID_ARG, "record rings: a float obs ring beside f64 records needs the step-loop form (tds_hip_step_many_is_loop)");
  return TDS_OK;
}

int step_many_prepare_impl(tds_hip_sim_t *s, const void *actions_dev, int action_blocks, int first_block, int n_steps,
                           void *obs_dev, const tds_hip_rings_t *rings) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  if (n_steps < 1 || n_steps > 4096) return fail(TDS_ERR_INVALID_ARG, "n_steps must be in 1..4096");
  if (actions_dev && action_blocks < 1) return fail(TDS_ERR_INVALID_ARG, "action_blocks must be >= 1");
  int rc = rings_check(s, rings, n_steps);
  if (rc != TDS_OK) return rc;
  if (s->auto_reset) return TDS_OK;  // (step-loop launches through the reset pool, or single steps: nothing to build)
  DeviceGuard guard(s->device);
  const int pool = actions_dev ? action_blocks : 1;
  const int first = actions_dev ? ((first_block % pool) + pool) % pool : 0;
  // (one kernel launch: nothing to build; with rings a single step is a step-loop launch too where the form exists)
  if (step_many_as_loop(s, n_steps) || (rings && n_steps == 1 && step_many_as_loop(s, 2))) return TDS_OK;
  if (graph_matches(s, actions_dev, pool, first, n_steps, obs_dev, rings)) return TDS_OK;
  return build_graph(s, actions_dev, pool, first, n_steps, obs_dev, rings);
}

int step_many_impl(tds_hip_sim_t *s, const void *actions_dev, int action_blocks, int first_block, int n_steps,
                   void *obs_dev, const tds_hip_rings_t *rings) {
  const bool eager = s && s->opt.get(TDS_OPT_STEP_MANY_EAGER, 0) == 1;  // (diagnostic: the same chains as plain stream launches)
  if (!eager) {
    int rc = step_many_prepare_impl(s, actions_dev, action_blocks, first_block, n_steps, obs_dev, rings);
    if (rc != TDS_OK) return rc;
  } else if (!s || n_steps < 1 || (actions_dev && action_blocks < 1)) {
    return fail(TDS_ERR_INVALID_ARG, "step_many: bad arguments");
  }
  DeviceGuard guard(s->device);
  TimedCall timed(s);
  const int pool = actions_dev ? action_blocks : 1;
  const int first = actions_dev ? ((first_block % pool) + pool) % pool : 0;
  const bool as_loop = !eager && (step_many_as_loop(s, n_steps) || (rings && n_steps == 1 && step_many_as_loop(s, 2)));
  // the handle's y record holds the last step's record afterwards, rings or not
  auto y_back = [&]() -> int {
    if (rings && rings->y_ring) {
      const size_t row = (size_t)s->model.output_dim * s->elem;
      HIP_TRY(hipMemcpy2DAsync(s->d_y, row, ring_slot(s, rings->y_ring, rings->y_slots, rings->y_first, n_steps - 1, y_width(s, rings)),
                               y_width(s, rings) * s->elem, row, (size_t)s->num_envs, hipMemcpyDeviceToDevice, s->stream));
    }
    return TDS_OK;
  };
  if (s->auto_reset) {
    // auto_reset_when_done after every step: step-loop launches that take the fresh states from the reset pool where
    // the plain call would be one step-loop launch (pool_step_many), else K single steps through the pool
    if (as_loop)  // (the step-loop launches write the last step's y record into d_y themselves)
      return pool_step_many(s, actions_dev, pool, first, n_steps, obs_dev, rings);
    const size_t blk = (size_t)s->num_envs * s->model.action_dim * s->elem;
    // (environment chains as in the plain call's graphs, launched eagerly: the passes of the reset pool are host-driven)
    const int C = chain_count(s, n_steps);
    if (C > 1) {
      const int rc = chain_streams(s, C);
      if (rc != TDS_OK) return rc;
      HIP_TRY(hipEventRecord(s->graph_fork, s->stream));
      for (int c = 1; c < C; ++c) HIP_TRY(hipStreamWaitEvent(s->graph_chain[c - 1], s->graph_fork, 0));
    }
    for (int k = 0; k < n_steps; ++k) {
      const void *a = actions_dev ? (const char *)actions_dev + (size_t)((first + k) % pool) * blk : nullptr;
      void *const ob = (rings && rings->obs_ring) ? ring_slot(s, rings->obs_ring, rings->obs_slots, rings->obs_first, k, s->obs_width()) : obs_dev;
      void *const yk = (rings && rings->y_ring) ? ring_slot(s, rings->y_ring, rings->y_slots, rings->y_first, k, y_width(s, rings)) : nullptr;
      const int rc = pool_step(s, a, ob, yk, (rings && rings->y_ring && rings->y_stride > 0) ? rings->y_stride : 0, C);
      if (rc != TDS_OK) return rc;
    }
    for (int c = 1; c < C; ++c) {
      HIP_TRY(hipEventRecord(s->graph_join[c - 1], s->graph_chain[c - 1]));
      HIP_TRY(hipStreamWaitEvent(s->stream, s->graph_join[c - 1], 0));
    }
    return y_back();
  }
  if (as_loop) {
    LaunchOpts lo;
    const size_t blk = (size_t)s->num_envs * s->model.action_dim * s->elem;
    const void *a0 = actions_dev ? (const char *)actions_dev + (size_t)first * blk : nullptr;
    if (actions_dev) {
      lo.act_pool = actions_dev;
      lo.act_blocks = pool;
      lo.act_first = first;
    }
    lo.rings = rings;
    // (a single step with rings is a step-loop launch too: the launcher picks that build whenever a ring is set)
    // (no copy of the last y slot behind the launch: the step-loop kernel writes the last step's record into d_y as well)
    // The 8-lane kernel beyond one round of resident two-wavefront workgroups (Ant: 8192 environments): the environments are
    // independent, so the call runs as environment ranges of that size ONE AFTER THE OTHER, each a launch of
    // all the steps in the two-wavefront build — instead of one launch of the one-wavefront build in several rounds
    // (Ant x 16384: 20.8 -> see DESIGN 2d; the exchange's launches count workgroups per slot and stay whole)
    if (s->compute_f64() && s->h64.oct != 0 && s->opt.get(TDS_OPT_OCT_W2, 1) == 1 && !(rings && rings->progress) && !s->peer_launch) {
      const int per_cu = (int)(s->lds_per_cu / (size_t)tds_oct_workgroup_bytes(s->model.input_dim));
      const int cap = 8 * s->num_cus * (per_cu < 4 ? per_cu : 4);
      if (cap > 0 && s->num_envs > cap) {
        // (full rounds first, the rest last: a step costs the same from 4097 to 8192 environments — two wavefronts on some
        //  SIMD — and less up to 4096; x 12288 as 8192 + 4096: 15.7 us per step, as 2 x 6144: 17.5)
        for (int e0 = 0; e0 < s->num_envs; e0 += cap) {
          const int e1 = e0 + cap < s->num_envs ? e0 + cap : s->num_envs;
          lo.env_first = e0;
          lo.env_total = e1 - e0;
          const int rc = launch(s, s->d_x, s->d_y, a0, s->d_x, obs_dev, e1 - e0, n_steps, TDS_RESET_NONE, nullptr, nullptr, 0, &lo);
          if (rc != TDS_OK) return rc;
        }
        return TDS_OK;
      }
    }
    return launch(s, s->d_x, s->d_y, a0, s->d_x, obs_dev, s->num_envs, n_steps, TDS_RESET_NONE, nullptr, nullptr, 0, &lo);
  }
  const int C = eager ? chain_count(s, n_steps) : s->graph_chains;
  if (eager) {
    int rc = chain_streams(s, C);
    if (rc != TDS_OK) return rc;
  }
  // chain 0 runs on the handle's stream, the others on their own streams forked from / joined to it
  if (C > 1) {
    HIP_TRY(hipEventRecord(s->graph_fork, s->stream));
    for (int c = 1; c < C; ++c) HIP_TRY(hipStreamWaitEvent(s->graph_chain[c - 1], s->graph_fork, 0));
  }
  if (eager) {
    for (int k = 0; k < n_steps; ++k)  // (step by step, so that every stream has work from the start)
      for (int c = 0; c < C; ++c) {
        tds_hip_rings_t rk;
        if (rings) {
          rk = *rings;
          rk.obs_first = rings->obs_ring ? (rings->obs_first + k) % rings->obs_slots : 0;
          rk.y_first = rings->y_ring ? (rings->y_first + k) % rings->y_slots : 0;
        }
        const int rc = enqueue_chain(s, c, C, c ? s->graph_chain[c - 1] : s->stream, actions_dev, pool, first + k, 1, obs_dev,
                                     rings ? &rk : nullptr);
        if (rc != TDS_OK) return rc;
      }
  } else {
    for (int c = C - 1; c >= 0; --c) HIP_TRY(hipGraphLaunch(s->graph_exec[c], c ? s->graph_chain[c - 1] : s->stream));
  }
  for (int c = 1; c < C; ++c) {
    HIP_TRY(hipEventRecord(s->graph_join[c - 1], s->graph_chain[c - 1]));
    HIP_TRY(hipStreamWaitEvent(s->stream, s->graph_join[c - 1], 0));
  }
  return y_back();
}
}  // namespace
}  // extern "C++"

int tds_hip_step_many_prepare(tds_hip_sim_t *s, const void *actions_dev, int action_blocks, int first_block,
                              int n_steps, void *obs_dev) {
  return step_many_prepare_impl(s, actions_dev, action_blocks, first_block, n_steps, obs_dev, nullptr);
}

int tds_hip_step_many(tds_hip_sim_t *s, const void *actions_dev, int action_blocks, int first_block, int n_steps,
                      void *obs_dev) {
  return step_many_impl(s, actions_dev, action_blocks, first_block, n_steps, obs_dev, nullptr);
}

int tds_hip_step_many_rings_prepare(tds_hip_sim_t *s, const void *actions_dev, int action_blocks, int first_block,
                                    int n_steps, const tds_hip_rings_t *rings) {
  if (!rings || (!rings->obs_ring && !rings->y_ring)) return fail(TDS_ERR_INVALID_ARG, "record rings: no ring given");
  return step_many_prepare_impl(s, actions_dev, action_blocks, first_block, n_steps, nullptr, rings);
}

int tds_hip_step_many_rings(tds_hip_sim_t *s, const void *actions_dev, int action_blocks, int first_block, int n_steps,
                            const tds_hip_rings_t *rings) {
  if (!rings || (!rings->obs_ring && !rings->y_ring)) return fail(TDS_ERR_INVALID_ARG, "record rings: no ring given");
  return step_many_impl(s, actions_dev, action_blocks, first_block, n_steps, nullptr, rings);
}

int tds_hip_step_many_rings_blocks(const tds_hip_sim_t *s) {
  if (!s) return 0;
  if (s->compute_f64() && (s->h64.oct || s->h64.chain)) return (s->num_envs + 7) / 8;  // (the 8-lane kernels: eight environments per workgroup)
  return (s->num_envs + (64 / s->lanes) - 1) / (64 / s->lanes);
}

int tds_hip_step_many_is_loop(const tds_hip_sim_t *s, int n_steps) { return s && step_many_as_loop(s, n_steps) ? 1 : 0; }

int tds_hip_set_graph_chains(tds_hip_sim_t *s, int chains) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  if (chains < 0 || chains > tds_hip_sim::kMaxChains) return fail(TDS_ERR_INVALID_ARG, "chains must be in 0..8");
  if (chains != s->chains_wanted) {
    DeviceGuard guard(s->device);
    drop_graphs(s);
  }
  s->chains_wanted = chains;
  return TDS_OK;
}

int tds_hip_step_many_tune(tds_hip_sim_t *s, const void *actions_dev, int action_blocks, int probe_steps, void *obs_dev,
                           int *chains) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  if (probe_steps < 2 || probe_steps > 4096) return fail(TDS_ERR_INVALID_ARG, "probe_steps must be in 2..4096");
  DeviceGuard guard(s->device);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  const bool timing = s->timing;
  s->timing = false;
  int best = 1, rc = TDS_OK;
  float best_ms = 0.0f;
  for (int c = 1; c <= 3 && rc == TDS_OK; ++c) {
    rc = tds_hip_set_graph_chains(s, c);
    if (rc == TDS_OK) rc = tds_hip_step_many(s, actions_dev, action_blocks, 0, probe_steps, obs_dev);  // builds, warms up
    if (rc != TDS_OK) break;
    hipError_t e = hipEventRecord(e0, s->stream);
    rc = tds_hip_step_many(s, actions_dev, action_blocks, 0, probe_steps, obs_dev);
    if (e == hipSuccess) e = hipEventRecord(e1, s->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e != hipSuccess) rc = fail(TDS_ERR_HIP, "timing the probe failed: %s", hipGetErrorString(e));
    if (rc == TDS_OK && (c == 1 || ms < best_ms)) {
      best = c;
      best_ms = ms;
    }
  }
  s->timing = timing;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc != TDS_OK) return rc;
  if (chains) *chains = best;
  return tds_hip_set_graph_chains(s, best);
}

int tds_hip_set_auto_reset(tds_hip_sim_t *s, int enable, unsigned long long seed) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  if (enable && s->model.reward_mode == TDS_REWARD_NONE)
    return fail(TDS_ERR_INVALID_ARG, "auto-reset needs a model with a termination rule (reward_mode)");
  s->auto_reset = enable != 0;
  s->seed = seed;
  s->pool_ready = false;  // (entries depend on the seed)
  s->pool_discard = true;
  return TDS_OK;
}

int tds_hip_reset(tds_hip_sim_t *s, const unsigned char *mask_dev, void *obs_dev) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  DeviceGuard guard(s->device);
  TimedCall timed(s);
  s->pool_ready = false;  // (a forced reset takes its states straight from the random stream, past the pool)
  return launch(s, s->d_x, s->d_y, nullptr, s->d_x, obs_dev, s->num_envs, 0, TDS_RESET_FORCED, mask_dev, nullptr,
                TDS_CTL_RESET_CALL);
}

int tds_hip_step(tds_hip_sim_t *s, const void *actions_dev, int substeps) {
  return tds_hip_step_obs(s, actions_dev, substeps, nullptr);
}

// ------------------------------------------------------------------------------------------------------
// Host-vector entry points for a plain C / C++ caller without HIP headers (include/tds_hip_stepper.hpp:
// tds_hip::VectorizedEnv): the STATE stays resident on the device; per call only the actions travel up and the
// [obs | reward | done] records (and, if asked for, the y records) travel down, through pinned staging memory.
// ------------------------------------------------------------------------------------------------------
extern "C++" {
namespace {
// bytes of the action staging region: the actions of a step, or the [N]-byte mask of tds_hip_reset_host, whichever is
// larger (models without actions — free bodies — still reset through a mask)
size_t stage_act_bytes(const tds_hip_sim *s) {
  const size_t n = (size_t)s->num_envs, a = n * s->model.action_dim * s->elem;
  return align256(a > n ? a : n);
}
int stage_alloc(tds_hip_sim *s) {
  if (s->stage_ready) return TDS_OK;
  const size_t n = (size_t)s->num_envs;
  const size_t b_act = stage_act_bytes(s), b_obs = align256(n * s->obs_width() * s->elem),
               b_y = align256(n * s->model.output_dim * s->elem);
  // all or nothing: a partial allocation is undone, and readiness is a flag set last (a later call then starts over
  // instead of running with NULL device staging)
  hipError_t e = hipHostMalloc(&s->h_stage, b_act + b_obs + b_y, hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc(&s->d_stage_act, b_act);
  if (e == hipSuccess) e = hipMalloc(&s->d_stage_obs, b_obs);
  if (e == hipSuccess) e = hipMemset(s->d_stage_obs, 0, b_obs);
  if (e != hipSuccess) {
    if (s->h_stage) (void)hipHostFree(s->h_stage);
    if (s->d_stage_act) (void)hipFree(s->d_stage_act);
    if (s->d_stage_obs) (void)hipFree(s->d_stage_obs);
    s->h_stage = s->d_stage_act = s->d_stage_obs = nullptr;
    snprintf(g_err, sizeof(g_err), "staging buffers of the host-vector entry points: %s", hipGetErrorString(e));
    return TDS_ERR_HIP;
  }
  s->h_stage_bytes = b_act + b_obs + b_y;
  s->stage_ready = true;
  return TDS_OK;
}
// records of the record dtype in pinned memory -> host doubles
void widen(const tds_hip_sim *s, const void *src, double *dst, size_t count) {
  if (s->records_f64()) memcpy(dst, src, count * 8);
  else for (size_t i = 0; i < count; ++i) dst[i] = (double)((const float *)src)[i];
}
}  // namespace
}  // extern "C++"

int tds_hip_step_host(tds_hip_sim_t *s, const double *actions_host, int substeps, double *obs_host, double *y_host) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  if (substeps < 1) return fail(TDS_ERR_INVALID_ARG, "substeps must be >= 1");
  DeviceGuard guard(s->device);
  int rc = stage_alloc(s);
  if (rc != TDS_OK) return rc;
  const size_t n = (size_t)s->num_envs, na = n * s->model.action_dim, no = n * s->obs_width(), ny = n * s->model.output_dim;
  const size_t b_act = stage_act_bytes(s), b_obs = align256(no * s->elem);
  char *const h_act = (char *)s->h_stage, *const h_obs = h_act + b_act, *const h_y = h_obs + b_obs;
  if (actions_host) {
    if (s->records_f64()) memcpy(h_act, actions_host, na * 8);
    else for (size_t i = 0; i < na; ++i) ((float *)h_act)[i] = (float)actions_host[i];
    HIP_TRY(hipMemcpyAsync(s->d_stage_act, h_act, na * s->elem, hipMemcpyHostToDevice, s->stream));
  }
  {
    TimedCall timed(s);
    rc = step_obs_impl(s, actions_host ? s->d_stage_act : nullptr, substeps, s->d_stage_obs);
  }
  if (rc != TDS_OK) return rc;
  if (obs_host) HIP_TRY(hipMemcpyAsync(h_obs, s->d_stage_obs, no * s->elem, hipMemcpyDeviceToHost, s->stream));
  if (y_host) HIP_TRY(hipMemcpyAsync(h_y, s->d_y, ny * s->elem, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (obs_host) widen(s, h_obs, obs_host, no);
  if (y_host) widen(s, h_y, y_host, ny);
  return TDS_OK;
}

int tds_hip_reset_host(tds_hip_sim_t *s, const unsigned char *mask_host, double *obs_host) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  DeviceGuard guard(s->device);
  int rc = stage_alloc(s);
  if (rc != TDS_OK) return rc;
  const size_t n = (size_t)s->num_envs, no = n * s->obs_width();
  const size_t b_act = stage_act_bytes(s);
  char *const h_obs = (char *)s->h_stage + b_act;
  unsigned char *mask_dev = nullptr;
  if (mask_host) {  // (the action staging buffer doubles as the mask: a reset takes no action)
    memcpy(s->h_stage, mask_host, n);
    HIP_TRY(hipMemcpyAsync(s->d_stage_act, s->h_stage, n, hipMemcpyHostToDevice, s->stream));
    mask_dev = (unsigned char *)s->d_stage_act;
  }
  rc = tds_hip_reset(s, mask_dev, s->d_stage_obs);
  if (rc != TDS_OK) return rc;
  if (obs_host) HIP_TRY(hipMemcpyAsync(h_obs, s->d_stage_obs, no * s->elem, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (obs_host) widen(s, h_obs, obs_host, no);
  return TDS_OK;
}

int tds_hip_set_states(tds_hip_sim_t *s, const double *qqd_host) {
  if (!s || !qqd_host) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard guard(s->device);
  const int w = s->model.dof_q + s->model.dof_qd, in = s->model.input_dim;
  const size_t n = (size_t)s->num_envs;
  // strided: only the [q | qd] columns of the x records change (actions and the PD variables stay as they are)
  if (s->records_f64()) {
    HIP_TRY(hipMemcpy2DAsync(s->d_x, (size_t)in * 8, qqd_host, (size_t)w * 8, (size_t)w * 8, n, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  } else {
    std::vector<float> tmp(n * w);
    for (size_t i = 0; i < n * w; ++i) tmp[i] = (float)qqd_host[i];
    HIP_TRY(hipMemcpy2DAsync(s->d_x, (size_t)in * 4, tmp.data(), (size_t)w * 4, (size_t)w * 4, n, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  s->pool_ready = false;  // (see tds_hip_set_inputs)
  return TDS_OK;
}

int tds_hip_device_alloc(tds_hip_sim_t *s, size_t bytes, void **out) {
  if (!s || !out) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard guard(s->device);
  HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
  HIP_TRY(hipMemset(*out, 0, bytes ? bytes : 1));
  return TDS_OK;
}
int tds_hip_device_free(tds_hip_sim_t *s, void *p) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  DeviceGuard guard(s->device);
  if (p) HIP_TRY(hipFree(p));
  return TDS_OK;
}
int tds_hip_device_upload(tds_hip_sim_t *s, void *dst_dev, const void *src_host, size_t bytes) {
  if (!s || !dst_dev || !src_host) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard guard(s->device);
  HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return TDS_OK;
}
int tds_hip_device_download(tds_hip_sim_t *s, void *dst_host, const void *src_dev, size_t bytes) {
  if (!s || !dst_host || !src_dev) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard guard(s->device);
  HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return TDS_OK;
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
// The by-products of Worker::rollouts (examples/ars/ars_vectorized_worker.h:88-135), optional:
//   stats  [n][od][3] = (count, mean, S) of RunningStat (running_stat.h: Knuth's recurrence) per environment and
//          observation component, pushed with the observation the policy is evaluated on, every step, done or not
//          (:88-110; the filter the reference then applies to its local copy never reaches anything);
//   traj   [n][traj_cap][out_dim] + traj_len [n]: per environment the y record (sim_states_with_graphics_) of every
//          step taken while not done; a step that ends with done repeats the previous entry, if there is one (:118-135).
template <typename T>
struct TdsRolloutExtras {
  T *stats;
  T *traj;
  int *traj_len;
  const T *y;
  int out_dim, traj_cap;
};

template <typename T>
__global__ void tds_policy_book_kernel(const T *__restrict__ x, int in_dim, int od, int adim,
                                       const T *__restrict__ policy, T *__restrict__ actions,
                                       T *__restrict__ rec, T *__restrict__ ret, int *__restrict__ cnt,
                                       unsigned char *__restrict__ frozen, T shift, int do_book, int do_policy,
                                       int raw_xy, int auto_reset, TdsRolloutExtras<T> ex, int n,
                                       int linear_policy = 1) {  // 0: the actions come from tds_policy_mlp_kernel
  // one wavefront per environment; L = 2^k lanes per action, consecutive lanes on consecutive weights
  const int env = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const int lane = threadIdx.x & 63;
  if (env >= n) return;
  if (do_book) {
    // (all lanes evaluate the flags; lane 0 writes them)
    bool fr = frozen[env] != 0;
    const bool done_now = rec[(size_t)env * (od + 2) + od + 1] != T(0);
    const T reward = rec[(size_t)env * (od + 2) + od];
    const bool count_it = !fr && !done_now;
    if (!fr && done_now && !auto_reset) fr = true;  // without auto-reset the environment stays done
    if (ex.traj != nullptr) {
      const int len = ex.traj_len[env];
      if (len < ex.traj_cap) {
        T *const dst = ex.traj + ((size_t)env * ex.traj_cap + len) * ex.out_dim;
        if (!(done_now || fr)) {
          for (int i = lane; i < ex.out_dim; i += 64) dst[i] = ex.y[(size_t)env * ex.out_dim + i];
          if (lane == 0) ex.traj_len[env] = len + 1;
        } else if (len > 0) {
          for (int i = lane; i < ex.out_dim; i += 64) dst[i] = dst[i - ex.out_dim];
          if (lane == 0) ex.traj_len[env] = len + 1;
        }
      }
    }
    if (lane == 0) {
      if (count_it) {
        ret[env] += reward - shift;
        cnt[env] += 1;
      }
      frozen[env] = fr ? 1 : 0;
      // after the last step the record's done column is the latch ("was done at some step"), as the one-launch
      // rollout leaves it
      if (!do_policy && !auto_reset) rec[(size_t)env * (od + 2) + od + 1] = fr ? T(1) : T(0);
    }
  }
  if (do_policy && ex.stats != nullptr) {
    for (int o = lane; o < od; o += 64) {
      const T xo = (o < 2 && !raw_xy) ? T(0) : x[(size_t)env * in_dim + o];
      T *const st = ex.stats + ((size_t)env * od + o) * 3;
      const T nn = st[0] + T(1);
      if (nn == T(1)) {
        st[1] = xo;
        st[2] = T(0);
      } else {
        const T m_old = st[1];
        const T m_new = m_old + (xo - m_old) / nn;
        st[2] = st[2] + (xo - m_old) * (xo - m_new);
        st[1] = m_new;
      }
      st[0] = nn;
    }
  }
  if (do_policy && linear_policy) {
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

// The environment's own policy NETWORK on the new state (NeuralNetwork::compute, src/math/neural_network.hpp:223-300):
// one wavefront per environment, the activations of two consecutive layers in LDS, the lanes split a unit's dot product
// (consecutive lanes on consecutive weights of its row) and reduce it with shuffles.
struct TdsNN {
  int n, units[TDS_NN_MAX_LAYERS], act[TDS_NN_MAX_LAYERS], bias[TDS_NN_MAX_LAYERS], nw, nb;
};

template <typename T>
__global__ void tds_policy_mlp_kernel(const T *__restrict__ x, int in_dim, const T *__restrict__ policy,
                                      T *__restrict__ actions, TdsNN nn, int raw_xy, int n) {
  __shared__ T buf[2][TDS_NN_MAX_UNITS];
  const int env = blockIdx.x, lane = threadIdx.x;  // (64 threads per block)
  if (env >= n) return;
  const T *const W = policy + (size_t)env * (nn.nw + nn.nb);
  const T *const Bs = W + nn.nw;
  int wi = 0, bi = 0, cur = 0;
  for (int o = lane; o < nn.units[0]; o += 64) {
    // obs = [q | qd] with obs[0] = obs[1] = 0 (ars_vectorized_environment.h:283-288); the input layer's bias (:251-255)
    T v = (o < 2 && !raw_xy) ? T(0) : x[(size_t)env * in_dim + o];
    if (nn.bias[0]) v += Bs[o];
    buf[0][o] = v;
  }
  if (nn.bias[0]) bi += nn.units[0];
  __syncthreads();
  for (int i = 1; i < nn.n; ++i) {
    const int P = nn.units[i - 1], Cn = nn.units[i], act = nn.act[i - 1];
    const T *const prev = buf[cur];
    T *const out = buf[cur ^ 1];
    for (int ci = 0; ci < Cn; ++ci) {
      T acc = T(0);
      for (int pi = lane; pi < P; pi += 64) acc += prev[pi] * W[wi + ci * P + pi];
      for (int d = 1; d < 64; d <<= 1) acc += __shfl_xor(acc, d, 64);
      if (lane == 0) {
        T v = acc + (nn.bias[i] ? Bs[bi + ci] : T(0));
        switch (act) {  // :267-294
          case TDS_NN_ACT_TANH: v = tanh(v); break;
          case TDS_NN_ACT_SIN: v = sin(v); break;
          case TDS_NN_ACT_RELU: v = v > T(0) ? v : T(0); break;
          case TDS_NN_ACT_SOFT_RELU: v = log(T(1) + exp(v)); break;
          case TDS_NN_ACT_ELU: v = v >= T(0) ? v : exp(v) - T(1); break;
          case TDS_NN_ACT_SIGMOID: { const T e = exp(v); v = e / (e + T(1)); break; }
          case TDS_NN_ACT_SOFTSIGN: v = v / (T(1) + (v < T(0) ? -v : v)); break;
          default: break;
        }
        out[ci] = v;
      }
    }
    wi += P * Cn;
    if (nn.bias[i]) bi += Cn;
    cur ^= 1;
    __syncthreads();
  }
  for (int a = lane; a < nn.units[nn.n - 1]; a += 64) actions[(size_t)env * nn.units[nn.n - 1] + a] = buf[cur][a];
}

template <typename T>
int rollout_per_step(tds_hip_sim *s, const void *policy_dev, int n_steps, double shift, int flags,
                     void *return_sum_dev, int *return_steps_dev, void *obs_dev, void *stats_dev, void *traj_dev,
                     int *traj_len_dev) {
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
  TdsRolloutExtras<T> ex;
  ex.stats = (T *)stats_dev;
  ex.traj = (T *)traj_dev;
  ex.traj_len = traj_len_dev;
  ex.y = (const T *)s->d_y;
  ex.out_dim = s->model.output_dim;
  ex.traj_cap = n_steps;
  if (traj_dev && (!traj_len_dev || hipMemsetAsync(traj_len_dev, 0, (size_t)n * sizeof(int), s->stream) != hipSuccess))
    return fail(TDS_ERR_INVALID_ARG, "trajectories need a traj_len buffer");
  const int threads = 256, blocks = (n + threads / 64 - 1) / (threads / 64);
  for (int t = 0; t <= n_steps; ++t) {
    hipLaunchKernelGGL(tds_policy_book_kernel<T>, dim3(blocks), dim3(threads), 0, s->stream, (const T *)s->d_x,
                       s->model.input_dim, od, adim, (const T *)policy_dev, actions, rec, ret, cnt, frozen,
                       (T)shift, t > 0 ? 1 : 0, t < n_steps ? 1 : 0, ((flags & 1) && t == 0) ? 1 : 0,
                       s->auto_reset ? 1 : 0, ex, n, s->nn_layers > 0 ? 0 : 1);
    if (hipGetLastError() != hipSuccess) return fail(TDS_ERR_HIP, "policy kernel launch");
    if (s->nn_layers > 0 && t < n_steps) {
      TdsNN nn;
      nn.n = s->nn_layers;
      for (int i = 0; i < TDS_NN_MAX_LAYERS; ++i) {
        nn.units[i] = s->nn_units[i];
        nn.act[i] = s->nn_act[i];
        nn.bias[i] = s->nn_bias[i];
      }
      nn.nw = s->nn_weights;
      nn.nb = s->nn_biases;
      hipLaunchKernelGGL(tds_policy_mlp_kernel<T>, dim3(n), dim3(64), 0, s->stream, (const T *)s->d_x, s->model.input_dim,
                         (const T *)policy_dev, actions, nn, ((flags & 1) && t == 0) ? 1 : 0, n);
      if (hipGetLastError() != hipSuccess) return fail(TDS_ERR_HIP, "policy network kernel launch");
    }
    if (t == n_steps) break;
    // (plain straight-line step, or — with auto-reset — the step through the reset pool)
    const int rc = step_obs_impl(s, actions, 1, rec);
    if (rc != TDS_OK) return rc;
  }
  return TDS_OK;
}

}  // namespace
}  // extern "C++"

int tds_hip_rollout_ex(tds_hip_sim_t *s, const void *policy_dev, int n_steps, double shift, int flags,
                       void *return_sum_dev, int *return_steps_dev, void *obs_dev, void *stats_dev, void *traj_dev,
                       int *traj_len_dev) {
  if (!s || !policy_dev) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  if (n_steps < 1) return fail(TDS_ERR_INVALID_ARG, "n_steps < 1");
  if (s->model.action_dim < 1) return fail(TDS_ERR_INVALID_ARG, "model has no actions");
  DeviceGuard guard(s->device);
  TimedCall timed(s);
  // One launch for the whole rollout: the step-loop build, compiled for one or for two wavefronts per SIMD (the
  // launcher picks by grid size) — faster than per-step launches at every batch size measured
  // (profiles/r02a_rollout_modes.txt).  flags bit 1 forces the other form: one straight-line step launch per step
  // with a small policy + bookkeeping kernel in between; the by-products of Worker::rollouts (running statistics of
  // the observations, trajectory records) are produced by that bookkeeping kernel, so asking for them selects it.
  const bool extras = stats_dev != nullptr || traj_dev != nullptr;
  const bool per_step = extras || (!s->auto_reset && (flags & 2) != 0) || s->nn_layers > 0;
  if (per_step)
    return s->records_f64()
               ? rollout_per_step<double>(s, policy_dev, n_steps, shift, flags, return_sum_dev, return_steps_dev, obs_dev,
                                          stats_dev, traj_dev, traj_len_dev)
               : rollout_per_step<float>(s, policy_dev, n_steps, shift, flags, return_sum_dev, return_steps_dev, obs_dev,
                                         stats_dev, traj_dev, traj_len_dev);
  Rollout ro;
  ro.policy = policy_dev;
  ro.ret_sum = return_sum_dev;
  ro.ret_steps = return_steps_dev;
  ro.shift = shift;
  ro.flags = flags;
  if (s->auto_reset) s->pool_ready = false;
  return launch(s, s->d_x, s->d_y, nullptr, s->d_x, obs_dev, s->num_envs, n_steps,
                s->auto_reset ? TDS_RESET_AUTO : TDS_RESET_NONE, nullptr, &ro);
}

int tds_hip_set_policy_network(tds_hip_sim_t *s, int num_layers, const int *layer_sizes, const int *activations,
                               const int *use_bias) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  if (num_layers == 0) {
    s->nn_layers = 0;
    return TDS_OK;
  }
  if (num_layers < 2 || num_layers > TDS_NN_MAX_LAYERS) return fail(TDS_ERR_INVALID_ARG, "num_layers must be 0 or 2..TDS_NN_MAX_LAYERS");
  if (!layer_sizes || !activations || !use_bias) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  if (layer_sizes[0] != s->model.dof_q + s->model.dof_qd) return fail(TDS_ERR_INVALID_ARG, "layer_sizes[0] must be the observation size dof_q + dof_qd");
  if (layer_sizes[num_layers - 1] != s->model.action_dim) return fail(TDS_ERR_INVALID_ARG, "the last layer must have action_dim units");
  int nw = 0, nb = 0;
  for (int i = 0; i < num_layers; ++i) {
    if (layer_sizes[i] < 1 || layer_sizes[i] > TDS_NN_MAX_UNITS) return fail(TDS_ERR_INVALID_ARG, "layer size out of range (1..TDS_NN_MAX_UNITS)");
    if (i > 0 && (activations[i - 1] < TDS_NN_ACT_IDENTITY || activations[i - 1] > TDS_NN_ACT_SOFTSIGN))
      return fail(TDS_ERR_INVALID_ARG, "unknown activation");
    if (i > 0) nw += layer_sizes[i - 1] * layer_sizes[i];
    nb += use_bias[i] ? layer_sizes[i] : 0;
  }
  s->nn_layers = num_layers;
  for (int i = 0; i < TDS_NN_MAX_LAYERS; ++i) {
    s->nn_units[i] = i < num_layers ? layer_sizes[i] : 0;
    s->nn_bias[i] = (i < num_layers && use_bias[i]) ? 1 : 0;
  }
  // (nn_act[i - 1] belongs to layer i, as in the reference's activations_ vector)
  for (int i = 0; i < TDS_NN_MAX_LAYERS; ++i) s->nn_act[i] = i + 1 < num_layers ? activations[i] : TDS_NN_ACT_IDENTITY;
  s->nn_weights = nw;
  s->nn_biases = nb;
  return TDS_OK;
}

int tds_hip_policy_num_parameters(const tds_hip_sim_t *s) {
  if (!s) return -1;
  if (s->nn_layers > 0) return s->nn_weights + s->nn_biases;
  return s->model.action_dim * (s->model.dof_q + s->model.dof_qd) + s->model.action_dim;
}

int tds_hip_rollout(tds_hip_sim_t *s, const void *policy_dev, int n_steps, double shift, int flags,
                    void *return_sum_dev, int *return_steps_dev, void *obs_dev) {
  return tds_hip_rollout_ex(s, policy_dev, n_steps, shift, flags, return_sum_dev, return_steps_dev, obs_dev, nullptr,
                            nullptr, nullptr);
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
  // room for 2 x 14 stamps and a grid the two-wavefront form serves: profile THAT form (14..: the helper wavefront)
  const int n_blocks = (s->num_envs + (64 / s->lanes) - 1) / (64 / s->lanes);
  const bool two_waves = n >= 2 * TDS_NUM_PHASE_STAMPS && s->w2_max_blocks > 0 && n_blocks <= s->w2_max_blocks;
  // (two-wavefront form: + the 100 MHz wall clock at the first / last stamp of EVERY workgroup, 28 + 2 b + {0, 1})
  const int ns = two_waves ? 2 * TDS_NUM_PHASE_STAMPS + 2 * n_blocks : TDS_NUM_PHASE_STAMPS;
  const TdsLds &lds = two_waves ? s->lds_w2 : s->lds;
  long long *d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(long long) * ns));
  HIP_TRY(hipMemset(d, 0, sizeof(long long) * ns));
  TdsStepCtl ctl;
  memset(&ctl, 0, sizeof(ctl));
  ctl.nsub = 1;
  ctl.y_stride = s->model.output_dim;
  if (s->opt.is_set(TDS_OPT_GRAM_STAMP_AT)) ctl.flags |= (int)s->opt.v[TDS_OPT_GRAM_STAMP_AT] << 8;  // (stamp 10 inside tds_gram_solve)
  // (profiling builds of the kernels, -DTDS_PROF_LOOP: the stamps of iteration K / 2 of a K-step launch of the two-wavefront
  //  step-loop kernel — TDS_HIP_PROF_LOOP=K; the library's own kernels have no such build and ignore the request)
  if (const char *pl = getenv("TDS_HIP_PROF_LOOP")) {
    const int k = atoi(pl);
    if (k > 1 && two_waves) {
      ctl.nsub = k;
      int it = k / 2;
      if (const char *pi = getenv("TDS_HIP_PROF_ITER")) it = atoi(pi);  // (which iteration is stamped; default the middle one)
      ctl.flags |= (it < 0 ? 0 : (it >= k ? k - 1 : it)) << 16;
    }
  }
  int rc;
  if (s->dtype == TDS_DTYPE_F64)
    rc = tds_launch_step<double, double>((const DevModel<double> *)s->d_model, s->h64, lds, s->lanes,
                                         (const double *)s->d_x, (double *)s->d_y, nullptr, nullptr, nullptr,
                                         (double *)s->d_ovf, s->num_envs, s->stream, ctl, d, two_waves ? TDS_FORM_W2 : 0);
  else if (s->dtype == TDS_DTYPE_F64_REC32)
    rc = tds_launch_step<double, float>((const DevModel<double> *)s->d_model, s->h64, lds, s->lanes,
                                        (const float *)s->d_x, (float *)s->d_y, nullptr, nullptr, nullptr,
                                        (double *)s->d_ovf, s->num_envs, s->stream, ctl, d, two_waves ? TDS_FORM_W2 : 0);
  else
    rc = tds_launch_step<float, float>((const DevModel<float> *)s->d_model, s->h32, lds, s->lanes,
                                       (const float *)s->d_x, (float *)s->d_y, nullptr, nullptr, nullptr,
                                       (float *)s->d_ovf, s->num_envs, s->stream, ctl, d, two_waves ? TDS_FORM_W2 : 0);
  if (rc != 0) {
    (void)hipFree(d);
    return fail(TDS_ERR_HIP, "profiling launch failed");
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  for (int i = 0; i < n; ++i) cycles_host[i] = 0;
  HIP_TRY(hipMemcpy(cycles_host, d, sizeof(long long) * (ns < n ? ns : n), hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return TDS_OK;
}

// The reference's profiling hook (src/base.hpp:39 SubmitProfileTiming, called at the start of a zone with its name and
// at its end with NULL: world.hpp:82-86, 293-366; mb_constraint_solver.hpp:225-247, 396-411, 439, 547-551) cannot be called
// from inside a kernel.  Its host-side counterpart: ONE step of the handle's state with the instrumented kernel build
// (tds_hip_profile_phases, one-wavefront form: workgroup 0's dependent chain), then the zones reported one after the
// other — a zone of the reference where a phase group corresponds to one (same names), the kernel's own phases otherwise.
int tds_hip_profile_zones(tds_hip_sim_t *s, tds_hip_profile_zone_fn fn, void *user) {
  if (!s || !fn) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  long long st[TDS_NUM_PHASE_STAMPS];
  const int rc = tds_hip_profile_phases(s, st, TDS_NUM_PHASE_STAMPS);
  if (rc != TDS_OK) return rc;
  int khz = 0;
  {
    DeviceGuard guard(s->device);
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, s->device) != hipSuccess || khz <= 0) khz = 2400000;
  }
  const double us_per_cycle = 1e3 / (double)khz;
  auto zone = [&](const char *name, int from, int to) { fn(name, (double)(st[to] - st[from]) * us_per_cycle, user); };
  // stamps: 0 start | 1 A | 2 B | 3 C | 4 I + M1 + D | 5 E | 6 G | 7 H | 8 F | 9 (sync) | 10 J | 11 K | 12 L | 13 M / N
  zone("forward_dynamics", 0, 8);                     // env step ahead of World::step: PD, kinematics, M = LDL^T, qdd, integrate_euler_qdd
  zone("forward_dynamics/load + PD", 0, 1);
  zone("forward_dynamics/jcalc", 1, 2);
  zone("forward_dynamics/forward_kinematics", 2, 3);
  zone("compute multi body contacts", 3, 4);          // world.hpp:324 (+ visual poses and link inertias: same stamp interval)
  zone("forward_dynamics/composite sweep", 4, 5);
  zone("inverse_mass_matrix_a", 5, 7);                // mb_constraint_solver.hpp:225 (mass-matrix rows + LDL^T)
  zone("forward_dynamics/solve", 7, 8);
  zone("solve constraints", 9, 12);                   // world.hpp:335
  zone("solve constraints/jacobian rows", 9, 10);
  zone("lcpA", 10, 11);                               // mb_constraint_solver.hpp:396 (here: the rows z~ = D^-1/2 L^-1 J^T; A is never formed)
  zone("solve_pgs", 11, 12);                          // mb_constraint_solver.hpp:439
  zone("integrate", 12, 13);                          // world.hpp:360 (+ output packing, reward / done)
  zone("step", 0, 13);
  return TDS_OK;
}

extern "C++" {
namespace {
__global__ void tds_poison_lds_kernel(unsigned pattern, int words, unsigned *sink) {
  extern __shared__ unsigned tds_poison_smem[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) tds_poison_smem[i] = pattern;
  __syncthreads();
  // stay resident for a while so that the dispatcher has to give every workgroup its own compute unit
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < 200000) {}
  if (sink != nullptr && tds_poison_smem[(threadIdx.x * 37) % words] != pattern) sink[0] = 1u;
}
}  // namespace
}  // extern "C++"

int tds_hip_debug_poison_lds(tds_hip_sim_t *s, int byte_pattern) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  DeviceGuard guard(s->device);
  const int bytes = 160 * 1024;  // one workgroup = all of a CU's LDS
  HIP_TRY(hipFuncSetAttribute((const void *)tds_poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, s->device));
  const unsigned b = (unsigned)(byte_pattern & 0xFF);
  const unsigned pattern = b | (b << 8) | (b << 16) | (b << 24);
  HIP_TRY(hipStreamSynchronize(s->stream));
  hipLaunchKernelGGL(tds_poison_lds_kernel, dim3(prop.multiProcessorCount), dim3(256), bytes, s->stream, pattern,
                     bytes / 4, (unsigned *)nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s->stream));
  return TDS_OK;
}

int tds_hip_kernel_info(const tds_hip_sim_t *s, int *lds_bytes_per_env, int *threads_per_env, int *envs_per_block) {
  if (!s) return fail(TDS_ERR_INVALID_ARG, "sim is NULL");
  if (lds_bytes_per_env) *lds_bytes_per_env = (int)(s->lds.stride * (s->compute_f64() ? 8 : 4));
  if (threads_per_env) *threads_per_env = s->lanes;
  if (envs_per_block) *envs_per_block = 64 / s->lanes;

  return TDS_OK;
}

int tds_hip_single_step_kernel(const tds_hip_sim_t *s, int *lanes_per_env, int *lds_bytes_per_env) {
  if (!s) return -1;
  const bool quad = s->compute_f64() && s->h64.quad != 0, oct = s->compute_f64() && s->h64.oct != 0;
  const bool chain = s->compute_f64() && s->h64.chain != 0;
  if (chain) {
    if (lanes_per_env) *lanes_per_env = 8;
    if (lds_bytes_per_env) *lds_bytes_per_env = tds_chain_lds_bytes(s->h64.chain);
    return 3;
  }
  if (lanes_per_env) *lanes_per_env = oct ? 8 : (quad ? 16 : s->lanes);
  if (lds_bytes_per_env)
    *lds_bytes_per_env = oct    ? tds_oct_lds_bytes(s->model.input_dim)
                         : quad ? tds_quad_lds_bytes<double>(s->model.input_dim)
                                : (int)(s->lds.stride * (s->compute_f64() ? 8 : 4));
  return oct ? 2 : (quad ? 1 : 0);
}

}  // extern "C"
