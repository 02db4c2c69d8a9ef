// tds_api_internal.h — the handle behind tds_hip_sim_t, shared by tds_api.hip (single device) and
// tds_shard.hip (multi-GPU sharding over RCCL).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <vector>

#include "tds_device_model.h"
#include "tds_hip.h"
#include "tds_kernels.h"

// Peer-store exchange of ONE step-loop launch (tds_shard.hip sets tds_hip_sim::peer_launch around its call of
// tds_hip_step_many_rings; launch() copies it into TdsStepCtl::peer_*).  Library-internal: not part of the C ABI.
struct TdsPeerLaunch {
  const void *const *rings;          // device array [n_peers]: the peers' gathered rings as mapped in this process
  unsigned long long *const *flags;  // device array [n_peers + 1]: the peers' flag arrays, this rank's own last
  unsigned int *arrive;              // arrival counters of the launch's slots (slot 0 of the launch first)
  long long ring_off;                // bytes from a ring's base to this rank's block of the launch's slot 0
  unsigned long long epoch;          // the launch's sequence number
  int n_peers, flag_off, flag_stride;
  int reward_done_only;              // option exchange_fields = 1
  int wide_ok;                       // the rings table is padded to a multiple of four entries (put_obs_wide may read past n_peers)
};

struct tds_hip_sim {
  tds_model_t model;
  int num_envs = 0, device = 0, dtype = TDS_DTYPE_F64, lanes = 64;
  TdsOptions opt;  // this handle's options (tds_options.h): snapshot at creation, tds_hip_set_option afterwards
  size_t elem = 8;  // bytes per scalar of the RECORDS in HBM (x, y, actions, obs, policy)
  hipStream_t stream = nullptr;
  void *d_model = nullptr;  // DevModel<T>, T = compute scalar
  DevModel<double> h64;
  DevModel<float> h32;
  TdsLds lds;
  // two-wavefront workgroups (plain kernels, straight-line launches whose whole grid is resident at once)
  TdsLds lds_w2;
  int w2_max_blocks = 0;  // 0: not available for this model / dtype
  int num_cus = 256;                // compute units and LDS bytes per compute unit of the handle's device (hipDeviceProp_t;
  size_t lds_per_cu = 160 * 1024;   // what the residency rules of the launch forms are computed from)
  void *d_x = nullptr, *d_y = nullptr, *d_ovf = nullptr;
  unsigned int *d_reset_count = nullptr;
  void *d_split = nullptr;  // records + done mask of the two-launch auto-reset step
  void *d_ro = nullptr;     // scratch of the per-step-launch rollout (actions | records | returns | counts | latches)
  bool auto_reset = false;
  unsigned long long seed = 0x5DEECE66Dull;
  // policy network of the rollouts (tds_hip_set_policy_network); nn_layers == 0: the default linear policy
  int nn_layers = 0, nn_units[TDS_NN_MAX_LAYERS] = {0}, nn_act[TDS_NN_MAX_LAYERS] = {0}, nn_bias[TDS_NN_MAX_LAYERS] = {0};
  int nn_weights = 0, nn_biases = 0;
  // pinned staging of the host-vector entry points (tds_hip_step_host / tds_hip_reset_host): actions up, records down
  void *h_stage = nullptr, *d_stage_act = nullptr, *d_stage_obs = nullptr;
  size_t h_stage_bytes = 0;
  bool stage_ready = false;        // the three staging buffers exist (set last by stage_alloc: all or nothing)
  const TdsPeerLaunch *peer_launch = nullptr;  // != NULL: the next ring launch is a peer-store exchange launch (see above)
  bool shard_ring_shaped = false;  // a shard layer has laid out its ring from this handle's options (tds_shard.hip: ring_alloc)
  bool timing = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool have_ms = false;
  // pre-settled reset pool (auto-reset at straight-line speed; see "reset pool" in tds_api.hip)
  static constexpr int kPoolEvents = 8;
  int pool_depth = 0, pool_every = 0, pool_lag = 0, pool_host_lag = 0, pool_cap = 0, pool_chunk = 0;  // D, R, W, H, staging capacity
  void *d_pool = nullptr;                 // [D][N][nq+nd] record dtype: ring of pre-settled reset states per env
  unsigned int *d_pool_filled = nullptr;  // [N] entries produced so far per env (valid: [count, filled))
  int *d_pool_items = nullptr;            // [1 + 2 cap]: n_items | item env | item ring slot
  int *h_pool_nitems = nullptr;           // pinned: n_items of the pass that has been planned
  void *d_pool_ovf = nullptr;             // surplus-row slab of the refill launches ([cap] environments; TDS_HIP_POOL_SLAB=0: none)
  void *d_stage_x = nullptr;              // [cap][input_dim] record dtype: the work list's records while they settle
  TdsLds pool_lds;                        // LDS layout of the refill launches: all constraint rows in LDS
  hipStream_t pool_stream = nullptr;
  hipEvent_t pool_ev[kPoolEvents] = {};   // pass j complete -> pool_ev[j % kPoolEvents]
  hipEvent_t pool_step_ev = nullptr, pool_plan_ev = nullptr, pool_sync_ev = nullptr;
  long long pool_step = 0;     // auto-reset steps since the pool was last filled completely
  long long pool_waited = 0;   // passes the step stream has been made to wait for
  long long pool_planned = 0;  // pass whose work list has been planned but not launched yet (0: none)
  long long pool_planned_at = 0;
  bool pool_many = false;      // the pool is on the pass schedule of step_many (pool_step_many), not of single steps
  bool pool_run_pending = false;  // ... a pass has been issued on the pool stream that the next chunk must wait for (pool_sync_ev)
  long long pool_many_chunks = 0;
  long long pool_total = 0, pool_snap_visible = 0, pool_snap_planned = 0;  // pool_step_many: steps launched / snapshots of the passes
  bool pool_ready = false;     // false: fill the pool completely before the next auto-reset step
  bool pool_discard = true;    // the entries in the rings are void (first use, new seed): start from empty rings
  // K-steps-per-launch graph cache (tds_hip_step_many)
  const void *graph_actions = nullptr;
  void *graph_obs = nullptr;
  int graph_pool = 0, graph_steps = 0, graph_first = 0;
  tds_hip_rings_t graph_rings = {};  // record rings the cached graphs write into (all-zero: none)
  hipStream_t graph_stream = nullptr;
  // the graph's environment chains (see build_graph): chain c > 0 is captured on graph_chain[c - 1]
  static constexpr int kMaxChains = 8;
  hipGraphExec_t graph_exec[kMaxChains] = {};  // one linear graph per chain
  hipStream_t graph_chain[kMaxChains - 1] = {};
  hipEvent_t graph_fork = nullptr, graph_join[kMaxChains - 1] = {};
  int graph_chains = 0;
  int chains_wanted = 0;  // tds_hip_set_graph_chains / _tune (0: library default)

  bool compute_f64() const { return dtype != TDS_DTYPE_F32; }
  bool records_f64() const { return dtype == TDS_DTYPE_F64; }
  int obs_width() const { return model.dof_q + model.dof_qd + 2; }
};

namespace tds_internal {

extern thread_local char g_err[512];

inline int fail(int code, const char *fmt, const char *detail = "") {
  snprintf(g_err, sizeof(g_err), fmt, detail);
  return code;
}

// Every entry point that touches the device selects the handle's device for its duration and puts the caller's
// device back: a process may hold handles on several GPUs (one per shard) and call them in any order.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) == hipSuccess && prev != device) {
      switched = hipSetDevice(device) == hipSuccess;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};

struct Rollout {
  const void *policy;
  void *ret_sum;
  int *ret_steps;
  double shift;
  int flags;
};

// what a launch may override (reset-pool machinery)
struct LaunchOpts {
  const TdsStepCtl *extra = nullptr;  // n_dev / pool fields
  bool other_stream = false;          // launch on `stream` instead of the handle's
  hipStream_t stream = nullptr;
  const TdsLds *lds = nullptr;        // LDS layout (the refill launches keep every constraint row in LDS: no slab)
  void *ovf = nullptr;                // ... or a surplus-row slab of the launch's own
  const void *act_pool = nullptr;     // step-loop launch with a different action block per step (TdsStepCtl::act_pool)
  int act_blocks = 0, act_first = 0;
  int env_first = 0;                  // this launch serves environments [env_first, env_first + n) of the records
  int env_total = 0;                  // (> 0: environments of ALL launches resident at the same time, for the form choice)
  // step-loop launch with per-step record rings (TdsStepCtl::obs_ring / y_ring / progress); ring_step0: steps of the
  // call that lie before this launch (the launch's step k owns slot (first + ring_step0 + k) % slots)
  const tds_hip_rings_t *rings = nullptr;
  int ring_step0 = 0;
  int y_stride = 0;                   // straight-line launch whose `y` is a slot of a strided y ring: scalars per record (0: packed)
};

// enqueue one launch of the step kernel on the handle's stream (device already selected by the caller)
int launch(tds_hip_sim *s, const void *x, void *y, const void *actions, void *fb, void *obs, int n, int nsub,
           int reset_mode, const unsigned char *mask, const Rollout *ro = nullptr, int ctl_flags = 0,
           const LaunchOpts *opts = nullptr);

}  // namespace tds_internal

#define TDS_HIP_TRY(expr)                                                                                         \
  do {                                                                                                            \
    hipError_t e_ = (expr);                                                                                       \
    if (e_ != hipSuccess) {                                                                                       \
      snprintf(tds_internal::g_err, sizeof(tds_internal::g_err), "%s failed: %s", #expr, hipGetErrorString(e_)); \
      return TDS_ERR_HIP;                                                                                         \
    }                                                                                                             \
  } while (0)
