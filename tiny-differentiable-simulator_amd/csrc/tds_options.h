// tds_options.h — the library's run-time options in ONE table (include/tds_hip.h: tds_hip_set_option,
// tds_hip_get_option, tds_hip_default_option).  Not part of the C ABI.
//
// Every switch that used to be a getenv() somewhere inside the library is a row here.  Precedence when a handle is
// CREATED: tds_hip_default_option(key) of this process  >  the environment variable TDS_HIP_<KEY>  >  the library default
// ("unset").  The handle keeps a snapshot; after that only tds_hip_set_option changes its run-time rows — nothing reads
// the environment while a handle is in use, and nothing is latched in a function-local static.
#pragma once
#include <limits.h>
#include <stdlib.h>
#include <string.h>

enum TdsOptKey {
  // ---- create-time rows (model build / kernel form of a handle: fixed once it exists)
  TDS_OPT_LANES_PER_ENV = 0,  // 16 / 32 / 64 lanes per environment (unset: the smallest that fits)
  TDS_OPT_NA_CAP,             // contacts whose constraint rows stay in LDS (unset: chosen for eight workgroups per CU)
  TDS_OPT_W2,                 // two-wavefront workgroups: 0 never, 1 default rule also without contact points, 2 at any grid size
  TDS_OPT_GRAM,               // 1: contact solve in Gram form on the f64 matrix cores (opt-in)
  TDS_OPT_NO_CHAIN,           // 1: every parent / child hand-over through LDS (A/B of the general path)
  TDS_OPT_NO_ROOTJOINT,       // 1: no root-chain scan
  TDS_OPT_NO_KINCHAIN,        // 1: no serial-chain prefix scans beyond the root chain
  TDS_OPT_NO_EULERROOT,       // 1: no closed-form root chain
  TDS_OPT_NO_LEGSCAN,         // 1: legs by the level loop instead of the segmented scan
  TDS_OPT_FOLD_FIXED,         // 1: fold fixed links into their parents even when the lanes would suffice
  TDS_OPT_QUAD,               // 0: a star-shaped legged robot (Laikago) stays on the general kernel instead of tds_quad.hip's 16-lane kernel
  TDS_OPT_OCT,                // 0: a star with two-link legs (the Ant) stays on the general kernel instead of tds_oct.hip's 8-lane kernel
  TDS_OPT_CHAIN,              // 0: a fixed-base serial chain without contacts (cartpole, pendulum5) stays on the general kernel instead of tds_chain.hip's
  // ---- run-time rows (may change between calls of a handle)
  TDS_OPT_LOOP_W2,            // step-loop launches: 0 one-wave build, 1 (default) two-wavefront build where it fits, 2 ... not with the reset pool
  TDS_OPT_OCT_W2,             // 8-lane kernel (tds_oct.hip): 0 one wavefront per workgroup, 1 / unset two (main + helper) while every workgroup is resident with at most two wavefronts per SIMD, 2 two at any grid size
  TDS_OPT_CHAIN_W2,           // serial-chain kernel (tds_chain.hip), step-loop launches with per-step records: 0 one wavefront per workgroup, 1 / unset a second one as the recorder while the launch is resident with at most two wavefronts per SIMD, 2 at any grid size
  TDS_OPT_QUAD_WIDE,          // 16-lane kernel (tds_quad.hip), step-loop launches: 0 one wavefront per workgroup always, 1 / unset eight wavefronts around one constant table (a workgroup per compute unit) where the one-wavefront workgroups are not all resident, 2 wherever the wide form is
  TDS_OPT_LOOP_OCC,           // step-loop build: 1 / 2 wavefronts per SIMD forced (unset: by grid size)
  TDS_OPT_EXCHANGE_W2,        // launches whose ring slots are exchanged while they run (rings->progress): 1 / unset the two-wavefront build N = 1 takes, 0 the one-wave build
  TDS_OPT_RING_NOFENCE,       // 1 (default): write-through record stores + plain wait; 0: release fence per step
  TDS_OPT_POOL_SLAB,          // 0: refill launches keep every constraint row in LDS
  TDS_OPT_POOL_SETTLE_LOOP,   // 1: the settle steps of a refill pass as one step-loop launch
  TDS_OPT_POOL_BESIDE,        // 1 / unset: the refill passes of a handle whose chunks run one wavefront per SIMD (8-lane kernel, at most two workgroups per compute unit) use the 240-register build, which fits beside them; 0: the build the grid size selects
  TDS_OPT_POOL_EVERY,         // reset pool: R
  TDS_OPT_POOL_HOST_LAG,      // H
  TDS_OPT_POOL_LAG,           // W
  TDS_OPT_POOL_CHUNK,         // steps per launch of the auto-reset step loop
  TDS_OPT_POOL_CAP,           // staging capacity (work-list entries)
  TDS_OPT_AUTO_RESET_SPLIT,   // 0 in-kernel reset, 1 split launch, 2 / unset reset pool
  TDS_OPT_GRAPH_CHAINS,       // environment chains of the step_many graphs
  TDS_OPT_NO_GRAPH_UPLOAD,    // 1: no hipGraphUpload after instantiation
  TDS_OPT_STEP_MANY_LOOP,     // 0 / 1: forbid / force K steps as one step-loop launch
  TDS_OPT_STEP_MANY_EAGER,    // 1: the chains as plain stream launches (diagnostic)
  TDS_OPT_GRAM_STAMP_AT,      // profile build: probe id of the extra stamp
  TDS_OPT_Y_STRIDE,           // scalars between consecutive y records of the rings the LIBRARY allocates (shard layer); unset: padded to 128 B
  // ---- shard layer (kept in the shard's sim handle)
  TDS_OPT_SHARD_RING,         // 0: never the ring exchange
  TDS_OPT_SHARD_WAIT_MS,      // give-up time of a ring wait (wait-kernel form)
  TDS_OPT_SHARD_GRAPH,        // 1: launch + exchanges of a chunk as one hipGraph
  TDS_OPT_SHARD_NO_GRAPH,     // 1: per-step form never from a graph
  TDS_OPT_SHARD_WAIT,         // how the communication stream follows the progress counter: 0 wait kernel, 1 hipStreamWaitValue64 (default where supported)
  TDS_OPT_SHARD_INPLACE,      // 1 (default): the launch stores its records straight into its own block of the gathered buffer (in-place all-gather)
  TDS_OPT_SHARD_REGISTER,     // 1 (default): ncclCommRegister the ring buffers where librccl offers it
  TDS_OPT_SHARD_CHUNK,        // steps per step-loop launch of the ring exchange (default 256; read when the ring is first used)
  TDS_OPT_RING_SIGNAL_LATE,   // experiment: 1 = the helper wavefront counts a step in at the top of its NEXT iteration
  TDS_OPT_ALT_BUILD,          // experiment slot k (1 .. TDS_ALT_SLOTS) of the library, where it was linked in (tds_kernels.h); f64 plain kernels
  TDS_OPT_SHARD_PEER,         // ring exchange by PEER STORES (the step kernel writes its records into the other ranks' gathered rings, IPC-mapped): unset / 1 where it can be set up on every rank (else the RCCL all-gather), 0 never, 2 required (error instead of the fallback)
  TDS_OPT_EXCHANGE_FIELDS,    // peer-store exchange: 0 / unset the whole [obs | reward | done] record travels, 1 only [reward | done]
  TDS_OPT_SHARD_PEER_RELEASE,   // peer-store exchange: 1 = system-scope release fences in front of the arrival counts and the flag stores (A/B switch for the first run on a real fabric; default: vmcnt(0) + relaxed stores)
  TDS_OPT_SHARD_PEER_COPY,      // peer exchange, STAGED form: 1 = the launch stores its records into this rank's own ring only and the communication stream pushes the launch's slots to every peer's ring with one strided device-to-device copy per peer (the runtime's copy engines: SDMA over xGMI) + one flag kernel — nothing of the exchange on a compute unit beside the launch, no store over the fabric from inside it (third form of bench.py's warm-up ladder; default 0: in-kernel peer stores)
  TDS_OPT_SHARD_PEER_LOOPBACK,  // diagnostic: k extra "peers" mapped onto scratch rings of this rank's own GPU (the kernel-side cost of k peers, measurable on one GPU)
  TDS_OPT_COUNT
};

constexpr long long TDS_OPT_UNSET = LLONG_MIN;

struct TdsOptRow {
  const char *key;   // what tds_hip_set_option takes; the environment variable is TDS_HIP_<KEY in upper case>
  bool create_time;  // fixed once the handle exists
  const char *env;   // (spelled out: two historical names do not follow the rule)
};

inline const TdsOptRow *tds_opt_rows() {
  static const TdsOptRow rows[TDS_OPT_COUNT] = {
      {"lanes_per_env", true, "TDS_HIP_LANES_PER_ENV"},
      {"na_cap", true, "TDS_HIP_NA_CAP"},
      {"w2", true, "TDS_HIP_W2"},
      {"gram", true, "TDS_HIP_GRAM"},
      {"no_chain", true, "TDS_HIP_NO_CHAIN"},
      {"no_rootjoint", true, "TDS_HIP_NO_ROOTJOINT"},
      {"no_kinchain", true, "TDS_HIP_NO_KINCHAIN"},
      {"no_eulerroot", true, "TDS_HIP_NO_EULERROOT"},
      {"no_legscan", true, "TDS_HIP_NO_LEGSCAN"},
      {"fold_fixed", true, "TDS_HIP_FOLD_FIXED"},
      {"quad", true, "TDS_HIP_QUAD"},
      {"oct", true, "TDS_HIP_OCT"},
      {"chain", true, "TDS_HIP_CHAIN"},
      {"loop_w2", false, "TDS_HIP_LOOP_W2"},
      {"oct_w2", false, "TDS_HIP_OCT_W2"},
      {"chain_w2", false, "TDS_HIP_CHAIN_W2"},
      {"quad_wide", false, "TDS_HIP_QUAD_WIDE"},
      {"loop_occ", false, "TDS_HIP_LOOP_OCC"},
      {"exchange_w2", false, "TDS_HIP_EXCHANGE_W2"},
      {"ring_nofence", false, "TDS_HIP_RING_NOFENCE"},
      {"pool_slab", false, "TDS_HIP_POOL_SLAB"},
      {"pool_settle_loop", false, "TDS_HIP_POOL_SETTLE_LOOP"},
      {"pool_beside", false, "TDS_HIP_POOL_BESIDE"},
      {"pool_every", false, "TDS_HIP_POOL_EVERY"},
      {"pool_host_lag", false, "TDS_HIP_POOL_HOST_LAG"},
      {"pool_lag", false, "TDS_HIP_POOL_LAG"},
      {"pool_chunk", false, "TDS_HIP_POOL_CHUNK"},
      {"pool_cap", false, "TDS_HIP_POOL_CAP"},
      {"auto_reset_split", false, "TDS_HIP_AUTO_RESET_SPLIT"},
      {"graph_chains", false, "TDS_HIP_GRAPH_CHAINS"},
      {"no_graph_upload", false, "TDS_HIP_NO_GRAPH_UPLOAD"},
      {"step_many_loop", false, "TDS_HIP_STEP_MANY_LOOP"},
      {"step_many_eager", false, "TDS_HIP_STEP_MANY_EAGER"},
      {"gram_stamp_at", false, "TDS_GRAM_STAMP_AT"},
      {"y_stride", false, "TDS_HIP_Y_STRIDE"},
      {"shard_ring", false, "TDS_HIP_SHARD_RING"},
      {"shard_wait_ms", false, "TDS_HIP_SHARD_WAIT_MS"},
      {"shard_graph", false, "TDS_HIP_SHARD_GRAPH"},
      {"shard_no_graph", false, "TDS_HIP_SHARD_NO_GRAPH"},
      {"shard_wait", false, "TDS_HIP_SHARD_WAIT"},
      {"shard_inplace", false, "TDS_HIP_SHARD_INPLACE"},
      {"shard_register", false, "TDS_HIP_SHARD_REGISTER"},
      {"shard_chunk", false, "TDS_HIP_SHARD_CHUNK"},
      {"ring_signal_late", false, "TDS_HIP_RING_SIGNAL_LATE"},
      {"alt_build", false, "TDS_HIP_ALT_BUILD"},
      {"shard_peer", false, "TDS_HIP_SHARD_PEER"},
      {"exchange_fields", false, "TDS_HIP_EXCHANGE_FIELDS"},
      {"shard_peer_release", false, "TDS_HIP_SHARD_PEER_RELEASE"},
      {"shard_peer_copy", false, "TDS_HIP_SHARD_PEER_COPY"},
      {"shard_peer_loopback", false, "TDS_HIP_SHARD_PEER_LOOPBACK"},
  };
  return rows;
}

inline int tds_opt_find(const char *key) {
  if (!key) return -1;
  const TdsOptRow *rows = tds_opt_rows();
  for (int i = 0; i < TDS_OPT_COUNT; ++i)
    if (strcmp(rows[i].key, key) == 0) return i;
  return -1;
}

struct TdsOptions {
  long long v[TDS_OPT_COUNT];
  bool is_set(int k) const { return v[k] != TDS_OPT_UNSET; }
  long long get(int k, long long dflt) const { return v[k] != TDS_OPT_UNSET ? v[k] : dflt; }
  bool flag(int k) const { return v[k] != TDS_OPT_UNSET && v[k] != 0; }  // "set and non-zero"
};

// process-wide overrides (tds_hip_default_option); all unset at start
inline TdsOptions &tds_opt_overrides() {
  static TdsOptions o = [] {
    TdsOptions t;
    for (int i = 0; i < TDS_OPT_COUNT; ++i) t.v[i] = TDS_OPT_UNSET;
    return t;
  }();
  return o;
}

// what a handle created NOW starts from: override > environment variable > unset
inline TdsOptions tds_opt_snapshot() {
  TdsOptions s = tds_opt_overrides();
  const TdsOptRow *rows = tds_opt_rows();
  for (int i = 0; i < TDS_OPT_COUNT; ++i) {
    if (s.v[i] != TDS_OPT_UNSET) continue;
    const char *e = getenv(rows[i].env);
    if (e && e[0]) s.v[i] = atoll(e);
  }
  return s;
}

// one row as a handle created now would see it (the model builder, which runs before a handle exists)
inline long long tds_opt_now(int k) {
  const long long o = tds_opt_overrides().v[k];
  if (o != TDS_OPT_UNSET) return o;
  const char *e = getenv(tds_opt_rows()[k].env);
  return (e && e[0]) ? atoll(e) : TDS_OPT_UNSET;
}
inline bool tds_opt_now_flag(int k) {
  const long long v = tds_opt_now(k);
  return v != TDS_OPT_UNSET && v != 0;
}
