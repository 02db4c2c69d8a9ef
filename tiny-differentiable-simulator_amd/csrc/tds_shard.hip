// tds_shard.hip — multi-GPU sharding of the environment batch behind the C ABI (include/tds_hip.h, tds_hip_shard_*).
//
// environments are independent, so rank r owns the contiguous block [r N/G, (r+1) N/G) of the global
// batch on its own GPU — model constants replicated, NO collective inside the step.  The one exchange is an all-gather
// of the [obs | reward | done] records the step kernel writes, once per policy step, straight over xGMI with RCCL
// (ncclAllGather; xGMI is a full mesh, so each shard travels once over its own link).  The reference has no analogue
// (SURVEY §2 "Parallelism strategies": none); the host side stays C/C++ as north_star asks — this file calls librccl
// directly, no torch.distributed in the data path.
//
// Streams.  The step runs on the handle's stream; the exchange on a private communication stream:
//     step k      : sim stream   — writes the records of step k into ring slot k mod S
//     gather k    : comm stream  — waits for step k's event, (converts to the wire dtype,) ncclAllGather into the
//                                  slot's gathered buffer, records the slot's done event
//     step k + S  : sim stream   — waits for gather k's done event before it overwrites the slot
// so the exchange of step k overlaps the computation of steps k+1 .. k+S-1, and nothing the step needs ever waits
// for xGMI (per-environment policies run on device; a learner consumes the gathered records one step late).
//
// librccl is loaded with dlopen at first use: a single-GPU host never needs it, and inside a PyTorch process the
// copy PyTorch already mapped is reused instead of a second one from /opt/rocm.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <new>
#include <vector>

#include "tds_api_internal.h"
#include "tds_shard_plan.h"

using namespace tds_internal;

namespace {

struct Rccl {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  // optional (RCCL >= 2.19): user-buffer registration
  ncclResult_t (*CommRegister)(const ncclComm_t, void *, size_t, void **) = nullptr;
  ncclResult_t (*CommDeregister)(const ncclComm_t, void *) = nullptr;
  bool ok = false;
};

Rccl *rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.ok ? &r : nullptr;
  tried = true;
  const char *names[] = {getenv("TDS_HIP_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  // an explicitly named library wins (RTLD_LOCAL: its nccl* symbols must not shadow anybody else's) ...
  if (names[0] && names[0][0]) r.handle = dlopen(names[0], RTLD_NOW | RTLD_LOCAL);
  // ... then a copy already mapped into the process (PyTorch's)
  for (const char *n : names)
    if (n && !r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  for (const char *n : names)
    if (n && !r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!r.handle) return nullptr;
#define TDS_SYM(field, name)                                    \
  r.field = (decltype(r.field))dlsym(r.handle, name);           \
  if (!r.field) return nullptr
  TDS_SYM(GetUniqueId, "ncclGetUniqueId");
  TDS_SYM(CommInitRank, "ncclCommInitRank");
  TDS_SYM(CommInitAll, "ncclCommInitAll");
  TDS_SYM(CommDestroy, "ncclCommDestroy");
  TDS_SYM(AllGather, "ncclAllGather");
  TDS_SYM(GroupStart, "ncclGroupStart");
  TDS_SYM(GroupEnd, "ncclGroupEnd");
  TDS_SYM(GetErrorString, "ncclGetErrorString");
  TDS_SYM(GetVersion, "ncclGetVersion");
#undef TDS_SYM
  r.CommRegister = (decltype(r.CommRegister))dlsym(r.handle, "ncclCommRegister");
  r.CommDeregister = (decltype(r.CommDeregister))dlsym(r.handle, "ncclCommDeregister");
  r.ok = true;
  return &r;
}

#define NCCL_TRY(expr)                                                                        \
  do {                                                                                        \
    ncclResult_t r_ = (expr);                                                                 \
    if (r_ != ncclSuccess) {                                                                  \
      snprintf(g_err, sizeof(g_err), "%s failed: %s", #expr, rccl()->GetErrorString(r_));     \
      return TDS_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

constexpr int kSlots = 4;  // record blocks in flight

__global__ void tds_f64_to_f32_kernel(const double *__restrict__ in, float *__restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}

// Ring exchange: ONE lane on the communication stream waits until the step-loop launch running on the step stream has
// counted `target` workgroups in (TdsStepCtl::progress: the records of a step are visible device-wide), then ends — the
// all-gather of that step's ring slot is the next thing in the stream.  Bounded: after `timeout_ticks` of the 100 MHz
// clock it raises the error latch and gives up (as does every wait behind it), so that a launch order nobody foresaw
// costs a wrong exchange that tds_hip_shard_flush reports, never a hung GPU.
__global__ void tds_ring_wait_kernel(const unsigned long long *progress, unsigned long long target, unsigned *err,
                                     long long timeout_ticks, unsigned *host_latch) {
  if (threadIdx.x != 0) return;
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  while (__hip_atomic_load(progress, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (pinned host word: the next host call of the shard sees the failure without a device round trip)
      if (host_latch) __hip_atomic_store(host_latch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    __builtin_amdgcn_s_sleep(16);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Peer-store exchange (round 5): the all-gather without a collective.  Every rank maps the other ranks' gathered rings and
// flag arrays into its address space (hipIpcGetMemHandle / hipIpcOpenMemHandle, handles exchanged once over the
// communicator); the step-loop launch stores each [obs | reward | done] record into its own block of the slot on EVERY
// rank and raises the slot's flag on every rank when its last workgroup has stored it (tds_kernels.hip: put_obs,
// peer_signal).  What is left for the streams are three one-wavefront kernels per LAUNCH (not per step), none of which
// runs beside the launch it belongs to:
//   credit   (step stream, in front of launch m) publishes "this rank has started launch m" on every rank and waits until
//            every peer has started launch m - 1: the half of the ring launch m writes holds the records of launch m - 2,
//            which are out of contract on a rank that has submitted launch m - 1 (tds_hip_shard_gathered_step: "inside the
//            most recently submitted launch") — ranks may drift apart by one launch, never by a ring;
//   arrived  (communication stream, behind launch m) waits until every slot of launch m carries every rank's flag: what
//            tds_hip_shard_gathered / _flush wait for.  Nothing of the step stream waits for it.
// All waits are bounded (option shard_wait_ms) and raise the same latch as the waits of the RCCL form.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool tds_wait_ge(const unsigned long long *p, unsigned long long target, unsigned *err,
                                            long long t0, long long timeout_ticks, unsigned *host_latch) {
  while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < target) {
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
    if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (host_latch) __hip_atomic_store(host_latch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
    __builtin_amdgcn_s_sleep(32);
  }
  return true;
}

// flags: [n_peers + 1] device table of flag arrays (the peers', then this rank's own); credit_off: index of credit[0]
__global__ void tds_peer_credit_kernel(unsigned long long *const *flags, int n_peers, long long credit_off, int rank,
                                       int world, unsigned long long seq, unsigned *err, long long timeout_ticks,
                                       unsigned *host_latch) {
  const int lane = threadIdx.x;
  if (lane <= n_peers)
    __hip_atomic_store(flags[lane] + credit_off + rank, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (seq > 1ull && lane < world && lane != rank) {
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    (void)tds_wait_ge(flags[n_peers] + credit_off + lane, seq - 1ull, err, t0, timeout_ticks, host_latch);
  }
}

__global__ void tds_peer_arrived_kernel(const unsigned long long *flags, int n, unsigned long long seq, unsigned *err,
                                        long long timeout_ticks, unsigned *host_latch) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    if (!tds_wait_ge(flags + i, seq, err, t0, timeout_ticks, host_latch)) return;
}

// staged peer exchange (option shard_peer_copy): the copies of a launch's slots into the peers' rings have completed on this
// stream; raise this rank's flag of every slot of the launch on every peer (this rank's own flags were raised by the launch)
__global__ void tds_peer_raise_kernel(unsigned long long *const *flags, int n_peers, int flag_off, int flag_stride, int n_slots,
                                      unsigned long long seq) {
  for (int i = threadIdx.x; i < n_peers * n_slots; i += blockDim.x) {
    const int pr = i / n_slots, k = i - pr * n_slots;
    __hip_atomic_store(flags[pr] + (size_t)flag_off + (size_t)k * (size_t)flag_stride, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// set-up check: a 32-bit token into the first word of THIS rank's block of slot 0 of every peer's gathered ring — through
// the very mapping the step kernel's record stores will use —, then, once those stores are acknowledged, a token into slot
// `rank` of the test row of every peer's flag array (and this rank's own): a rank that sees a peer's flag token must find
// that peer's ring token in its own ring (the protocol's visibility rule, checked once on the real memory)
__global__ void tds_peer_token_kernel(void *const *rings, unsigned long long *const *flags, int n_peers, long long ring_off,
                                      long long test_off, int rank, unsigned long long token) {
  const int lane = threadIdx.x;
  if (lane < n_peers)
    __hip_atomic_store((unsigned int *)((char *)rings[lane] + ring_off), (unsigned int)token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  if (lane <= n_peers)
    __hip_atomic_store(flags[lane] + test_off + rank, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

struct tds_hip_shard {
  tds_hip_sim *sim = nullptr;
  int rank = 0, world = 1, n_local = 0, n_global = 0;
  int wire_bytes = 4;  // bytes per scalar on the wire
  int block = 1;       // steps whose records travel in one all-gather
  ncclComm_t comm = nullptr;  // NULL: world == 1 without RCCL (device copy)
  hipStream_t comm_stream = nullptr;
  void *rec[kSlots] = {};       // [block][n_local][w]   record dtype: written by the step kernel
  void *wire[kSlots] = {};      // [block][n_local][w]   wire dtype (== rec when no conversion is needed)
  void *gathered[kSlots] = {};  // [world][block][n_local][w] wire dtype
  hipEvent_t ev_step[kSlots] = {}, ev_done[kSlots] = {};
  bool pending[kSlots] = {};
  long long steps = 0;  // steps submitted so far
  int last_slot = -1;   // slot of the most recently submitted exchange
  // K steps + their exchanges replayed from one captured hipGraph (tds_hip_shard_step_many)
  hipGraphExec_t graph_exec = nullptr;
  hipStream_t graph_stream = nullptr;
  const void *graph_actions = nullptr;
  int graph_pool = 0, graph_first = 0, graph_steps = 0, graph_block = 0, graph_slot0 = 0, graph_last_slot = -1;
  bool comm_warm = false;  // an all-gather has run eagerly on the communicator (connections are up)
  bool one_process_group = false;  // created by tds_hip_shard_create_all: collectives must be issued as a group

  // ---- ring exchange (tds_hip_shard_step_many where the sim's K steps are ONE step-loop launch): the launch writes
  //      the [obs | reward | done] record of every step into a slot of `rwire`, in the wire dtype; the communication
  //      stream sends slot k as soon as the launch has counted every workgroup in for step k
  void *rwire = nullptr;   // [2 TDS_SHARD_CHUNK][n_local][w]          wire dtype (NULL with the in-place exchange)
  void *rgath = nullptr;   // [2 TDS_SHARD_CHUNK][world][n_local][w]   wire dtype
  bool ring_uncached = false;  // rgath is uncached device memory (peers write into it)
  void *ry = nullptr;      // [TDS_SHARD_Y_SLOTS][n_local][y_stride]   record dtype: the y records (local, not exchanged)
  int ry_stride = 0;       // scalars per y record in `ry` (padded to whole 128-byte lines; option y_stride)
  bool inplace = false;    // the launch stores its records into ITS block of rgath; the all-gather is in place
  void *reg_handle = nullptr;  // ncclCommRegister handle of rgath
  // one counter per RING SLOT (2 chunk of them), + the error latch behind them.  The counters are NEVER reset: every use
  // of a slot adds n_blocks to its counter, the host keeps the number of uses (slot_uses) — no memset between launches
  unsigned long long *progress = nullptr;
  std::vector<unsigned long long> slot_uses;
  unsigned *host_latch = nullptr;  // pinned: raised by a wait that gave up (checked by every later call of the shard)
  bool wait_value = false;         // the waits are hipStreamWaitValue64 commands instead of wait kernels
  bool ring_ready = false;         // ring_alloc has completed (a failed allocation is undone as a whole)
  int chunk = TDS_SHARD_CHUNK_DEFAULT;  // steps per step-loop launch of the ring exchange (option shard_chunk, read at ring_alloc)
  int last_ring_slot0 = -1, last_ring_steps = 0, last_ring_half = 0;  // the most recently submitted launch of the ring exchange
  hipEvent_t ev_kernel[2] = {}, ev_comm[2] = {};
  hipEvent_t cap_fork = nullptr, cap_kernel = nullptr, cap_comm = nullptr;  // the same roles inside a stream capture
  bool comm_pending[2] = {};
  long long chunks = 0;    // step-loop launches submitted so far
  void *last_ptr = nullptr;       // gathered records of the most recently submitted exchange ...
  hipEvent_t last_ev = nullptr;   // ... and the event that says they have arrived
  int last_block = 1;
  static constexpr int kRingGraphs = 8;
  struct RingGraph {
    hipGraphExec_t exec = nullptr;
    const void *actions = nullptr;
    int pool = 0, first = 0, steps = 0, half = 0;
    long long used = 0;
  } rgraph[kRingGraphs];
  long long rgraph_clock = 0;

  // ---- peer-store exchange (see the kernels above)
  bool peer_mode = false;        // the ring exchange goes through peer stores (decided once, in ring_alloc, by ALL ranks together)
  int n_peers = 0;               // ranks other than this one (+ loopback "peers": scratch rings of this rank's own)
  int n_real_peers = 0;
  unsigned long long *pflags = nullptr;  // [ring slots][world] arrival flags | credit[world] | test[world]  (uncached where offered)
  unsigned int *parrive = nullptr;       // [ring slots][TDS_PEER_ARRIVE_STRIDE] arrival counters of this rank's own launches
  void *peer_ring_map[TDS_MAX_PEERS] = {};   // the peers' rgath / pflags as mapped here (hipIpcOpenMemHandle), peer order =
  void *peer_flag_map[TDS_MAX_PEERS] = {};   //   rank order without this rank
  bool peer_opened[TDS_MAX_PEERS] = {};      // (mapped through IPC: closed in ring_free; loopback rings are hipFree'd)
  void *d_peer_tab = nullptr;    // device: [n_peers] ring bases | [n_peers + 1] flag arrays (own last)
  bool reward_done_only = false;
  int exchange_form = 0;         // what the most recent tds_hip_shard_step_many ran: TDS_EXCHANGE_*

  size_t flag_count() const { return 2 * (size_t)chunk * world; }
  size_t credit_off() const { return flag_count(); }
  size_t test_off() const { return flag_count() + world; }
  size_t block_scalars() const { return (size_t)block * n_local * sim->obs_width(); }
  size_t slot_scalars() const { return (size_t)n_local * sim->obs_width(); }
  unsigned *wait_err() const { return (unsigned *)(progress + 2 * (size_t)chunk); }
};

namespace {

int alloc_ring(tds_hip_shard *sh) {
  // (a cached graph's kernel and all-gather nodes point into the rings that are about to be freed)
  if (sh->graph_exec) {
    (void)hipGraphExecDestroy(sh->graph_exec);
    sh->graph_exec = nullptr;
  }
  sh->graph_steps = 0;
  sh->graph_actions = nullptr;
  const size_t rec_b = sh->block_scalars() * sh->sim->elem;
  const size_t wire_b = sh->block_scalars() * sh->wire_bytes;
  const bool convert = (int)sh->sim->elem != sh->wire_bytes;
  for (int i = 0; i < kSlots; ++i) {
    if (sh->rec[i]) (void)hipFree(sh->rec[i]);
    if (sh->wire[i] && sh->wire[i] != sh->rec[i]) (void)hipFree(sh->wire[i]);
    if (sh->gathered[i]) (void)hipFree(sh->gathered[i]);
    sh->rec[i] = sh->wire[i] = sh->gathered[i] = nullptr;
    TDS_HIP_TRY(hipMalloc(&sh->rec[i], rec_b));
    TDS_HIP_TRY(hipMemset(sh->rec[i], 0, rec_b));
    if (convert)
      TDS_HIP_TRY(hipMalloc(&sh->wire[i], wire_b));
    else
      sh->wire[i] = sh->rec[i];
    TDS_HIP_TRY(hipMalloc(&sh->gathered[i], wire_b * sh->world));
    sh->pending[i] = false;
  }
  return TDS_OK;
}

int make_shard(const tds_model_t *model, int global_envs, int rank, int world, int device, int dtype, int wire_dtype,
               tds_hip_shard **out) {
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return fail(TDS_ERR_INVALID_ARG, "rank / world out of range");
  if (global_envs < world || global_envs % world != 0)
    return fail(TDS_ERR_INVALID_ARG, "global_envs must be a positive multiple of world (equal shards: one all-gather)");
  if (wire_dtype != TDS_DTYPE_F32 && wire_dtype != TDS_DTYPE_F64) return fail(TDS_ERR_INVALID_ARG, "wire dtype: F32 or F64");
  tds_hip_shard *sh = new (std::nothrow) tds_hip_shard();
  if (!sh) return fail(TDS_ERR_INVALID_ARG, "out of host memory");
  sh->rank = rank;
  sh->world = world;
  sh->n_global = global_envs;
  sh->n_local = global_envs / world;
  int rc = tds_hip_create(model, sh->n_local, device, dtype, &sh->sim);
  if (rc != TDS_OK) {
    delete sh;
    return rc;
  }
  sh->wire_bytes = wire_dtype == TDS_DTYPE_F64 ? 8 : 4;
  if (sh->wire_bytes > (int)sh->sim->elem) sh->wire_bytes = (int)sh->sim->elem;  // never widen on the wire
  DeviceGuard guard(device);
  hipError_t e = hipStreamCreateWithFlags(&sh->comm_stream, hipStreamNonBlocking);
  for (int i = 0; i < kSlots && e == hipSuccess; ++i) {
    e = hipEventCreateWithFlags(&sh->ev_step[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sh->ev_done[i], hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "shard stream / event creation failed: %s", hipGetErrorString(e));
    tds_hip_shard_destroy(sh);
    return TDS_ERR_HIP;
  }
  rc = alloc_ring(sh);
  if (rc != TDS_OK) {
    tds_hip_shard_destroy(sh);
    return rc;
  }
  *out = sh;
  return TDS_OK;
}

// sim stream: one step of the local shard into the current ring slot
int shard_step_local(tds_hip_shard *sh, const void *actions_dev, int substeps) {
  tds_hip_sim *s = sh->sim;
  const int slot = (int)((sh->steps / sh->block) % kSlots), j = (int)(sh->steps % sh->block);
  if (j == 0 && sh->pending[slot]) {  // the slot's previous exchange must have read its records
    TDS_HIP_TRY(hipStreamWaitEvent(s->stream, sh->ev_done[slot], 0));
    sh->pending[slot] = false;
  }
  char *rec = (char *)sh->rec[slot] + (size_t)j * sh->n_local * s->obs_width() * s->elem;
  int rc = tds_hip_step_obs(s, actions_dev, substeps, rec);
  if (rc != TDS_OK) return rc;
  sh->steps++;
  return TDS_OK;
}

// comm stream: exchange the block that has just been completed (or a partial one at a flush)
int shard_submit(tds_hip_shard *sh, int slot) {
  tds_hip_sim *s = sh->sim;
  TDS_HIP_TRY(hipEventRecord(sh->ev_step[slot], s->stream));
  TDS_HIP_TRY(hipStreamWaitEvent(sh->comm_stream, sh->ev_step[slot], 0));
  const size_t count = sh->block_scalars();
  if (sh->wire[slot] != sh->rec[slot]) {
    hipLaunchKernelGGL(tds_f64_to_f32_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, sh->comm_stream,
                       (const double *)sh->rec[slot], (float *)sh->wire[slot], count);
    if (hipGetLastError() != hipSuccess) return fail(TDS_ERR_HIP, "wire conversion launch");
  }
  if (sh->comm) {
    NCCL_TRY(rccl()->AllGather(sh->wire[slot], sh->gathered[slot], count,
                               sh->wire_bytes == 8 ? ncclFloat64 : ncclFloat32, sh->comm, sh->comm_stream));
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(sh->comm_stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) sh->comm_warm = true;
  } else {
    TDS_HIP_TRY(hipMemcpyAsync(sh->gathered[slot], sh->wire[slot], count * sh->wire_bytes, hipMemcpyDeviceToDevice,
                               sh->comm_stream));
  }
  return TDS_OK;
}
int shard_mark_done(tds_hip_shard *sh, int slot) {
  TDS_HIP_TRY(hipEventRecord(sh->ev_done[slot], sh->comm_stream));
  sh->pending[slot] = true;
  sh->last_slot = slot;
  sh->last_ptr = sh->gathered[slot];
  sh->last_ev = sh->ev_done[slot];
  sh->last_block = sh->block;
  return TDS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Ring exchange.  Where the shard's K steps are ONE launch of the step-loop kernel (tds_hip_step_many_is_loop: the Ant
// up to three rounds of workgroups, every world without contacts) there is no kernel boundary per step to hang an
// exchange on — and none is needed: the launch stores every step's [obs | reward | done] record into a ring slot in the
// wire dtype (tds_hip_step_many_rings, obs_f32) and counts its workgroups in per step; the communication stream runs
//     wait(step k counted in)  ->  ncclAllGather(slot k)            k = 0 .. c - 2
//     wait(launch complete)    ->  ncclAllGather(slot c - 1)
// beside it.  One exchange per policy step, no host call per step, nothing the step needs ever waits for
// xGMI.  A call is cut into launches of up to TDS_SHARD_CHUNK steps; the ring holds two of them, launch j + 2 waits
// for the exchanges of launch j (which used the same half).  N = 1 (tds_hip_step_many_rings alone) and N > 1 (this)
// run the SAME kernel doing the SAME work per step; the only difference is the exchange.
// ---------------------------------------------------------------------------------------------------------------
void ring_free(tds_hip_shard *sh);

// bytes all-gathered over the communicator (ncclInt8): one record per rank
struct PeerHello {
  int ok;
  int pad_;
  hipIpcMemHandle_t ring, flags;
};

int comm_all_gather_bytes(tds_hip_shard *sh, const void *mine, void *all, size_t bytes) {
  if (sh->world == 1 || !sh->comm) {
    memcpy(all, mine, bytes);
    return TDS_OK;
  }
  // (one allocation for both buffers: a single failure point, nothing to leak.  A rank that cannot allocate these few hundred
  //  bytes cannot take part in the collective at all — as with any failed rank of a communicator, the others are then in
  //  RCCL's hands)
  void *d_send = nullptr;
  const size_t send_b = (bytes + 255) & ~(size_t)255;
  TDS_HIP_TRY(hipMalloc(&d_send, send_b + bytes * sh->world));
  void *const d_recv = (char *)d_send + send_b;
  int rc = TDS_OK;
  hipError_t e = hipMemcpyAsync(d_send, mine, bytes, hipMemcpyHostToDevice, sh->comm_stream);
  if (e == hipSuccess) {
    const ncclResult_t r = rccl()->AllGather(d_send, d_recv, bytes, ncclInt8, sh->comm, sh->comm_stream);
    if (r != ncclSuccess) {
      snprintf(g_err, sizeof(g_err), "ncclAllGather (peer-store set-up) failed: %s", rccl()->GetErrorString(r));
      rc = TDS_ERR_HIP;
    } else {
      sh->comm_warm = true;
    }
  }
  if (e == hipSuccess && rc == TDS_OK) e = hipMemcpyAsync(all, d_recv, bytes * sh->world, hipMemcpyDeviceToHost, sh->comm_stream);
  if (e == hipSuccess && rc == TDS_OK) e = hipStreamSynchronize(sh->comm_stream);
  (void)hipFree(d_send);
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "peer-store set-up: %s", hipGetErrorString(e));
    return TDS_ERR_HIP;
  }
  return rc;
}

void peer_unmap(tds_hip_shard *sh) {
  for (int i = 0; i < TDS_MAX_PEERS; ++i) {
    if (sh->peer_opened[i]) {
      if (sh->peer_ring_map[i]) (void)hipIpcCloseMemHandle(sh->peer_ring_map[i]);
      if (sh->peer_flag_map[i]) (void)hipIpcCloseMemHandle(sh->peer_flag_map[i]);
    } else {  // (loopback "peers": scratch allocations of this rank's own)
      if (sh->peer_ring_map[i]) (void)hipFree(sh->peer_ring_map[i]);
      if (sh->peer_flag_map[i]) (void)hipFree(sh->peer_flag_map[i]);
    }
    sh->peer_ring_map[i] = sh->peer_flag_map[i] = nullptr;
    sh->peer_opened[i] = false;
  }
  sh->n_peers = sh->n_real_peers = 0;
}

// Decide — all ranks together — whether the ring exchange goes through peer stores, and set it up: flag / counter arrays,
// the peers' rings and flag arrays mapped through IPC, a token round trip over the mapped memory as a check that what was
// mapped is what the peers read.  Any failure on any rank (no IPC between the devices, a container without dmabuf IPC, a
// token that never arrives) leaves EVERY rank on the RCCL all-gather (option shard_peer = 2: an error instead).
int peer_setup(tds_hip_shard *sh, size_t ring_slots, size_t slot_b) {
  tds_hip_sim *s = sh->sim;
  const long long want = s->opt.get(TDS_OPT_SHARD_PEER, 1);
  sh->peer_mode = false;
  sh->reward_done_only = s->opt.get(TDS_OPT_EXCHANGE_FIELDS, 0) == 1;
  const int loopback = (int)s->opt.get(TDS_OPT_SHARD_PEER_LOOPBACK, 0);
  // (several ranks: only with the gathered ring in UNCACHED memory — in ordinary device memory a line of a reused slot that
  //  this GPU's L2 still holds would be read instead of what the peer has written since, and nothing would notice)
  const bool possible = want != 0 && sh->inplace && !sh->one_process_group && sh->world - 1 + (loopback > 0 ? loopback : 0) <= TDS_MAX_PEERS &&
                        (sh->world == 1 || (sh->comm != nullptr && sh->ring_uncached));
  char why[160] = "";
  if (want != 0 && sh->world > 1 && sh->comm != nullptr && !sh->ring_uncached)
    snprintf(why, sizeof(why), "the gathered ring is not in uncached device memory");
  else if (want != 0 && !possible)
    snprintf(why, sizeof(why), "not available for this shard: in-place ring %d, one process for all ranks %d, peers %d + %d loopback of at most %d, communicator %d",
             (int)sh->inplace, (int)sh->one_process_group, sh->world - 1, loopback > 0 ? loopback : 0, TDS_MAX_PEERS, sh->comm != nullptr);
  // own arrays first (every rank allocates them: the set-up below is a collective either way — a LOCAL failure here makes this
  // rank's verdict "no" but never keeps it away from the all-gathers the other ranks are waiting in)
  const size_t n_flags = sh->flag_count() + 2 * (size_t)sh->world;
  bool alloc_ok = true;
  {
    void *pf = nullptr;
    // fine-grained / uncached device memory where the runtime offers it: a flag written by another GPU must never be served
    // from this GPU's L2
    hipError_t e = hipExtMallocWithFlags(&pf, n_flags * sizeof(unsigned long long), hipDeviceMallocUncached);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      pf = nullptr;
      e = hipMalloc(&pf, n_flags * sizeof(unsigned long long));
    }
    sh->pflags = e == hipSuccess ? (unsigned long long *)pf : nullptr;
    if (e == hipSuccess) e = hipMemset(sh->pflags, 0, n_flags * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void **)&sh->parrive, ring_slots * TDS_PEER_ARRIVE_STRIDE * sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemset(sh->parrive, 0, ring_slots * TDS_PEER_ARRIVE_STRIDE * sizeof(unsigned int));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
      (void)hipGetLastError();
      alloc_ok = false;
      snprintf(why, sizeof(why), "flag / counter arrays: %s", hipGetErrorString(e));
    }
  }
  if (sh->world > 1 && !sh->comm) return TDS_OK;  // (no communicator, several ranks: nothing collective can be set up)
  bool ok = possible && alloc_ok;
  std::vector<PeerHello> all((size_t)sh->world);
  PeerHello mine;
  memset(&mine, 0, sizeof(mine));
  if (ok && sh->world > 1) {
    hipError_t e = hipIpcGetMemHandle(&mine.ring, sh->rgath);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&mine.flags, sh->pflags);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      ok = false;
      snprintf(why, sizeof(why), "hipIpcGetMemHandle: %s", hipGetErrorString(e));
    }
  }
  mine.ok = ok ? 1 : 0;
  if (sh->world > 1) {  // (collective: every rank of the communicator is here, whatever its own verdict)
    const int rc = comm_all_gather_bytes(sh, &mine, all.data(), sizeof(PeerHello));
    if (rc != TDS_OK) return rc;
    for (int r = 0; r < sh->world; ++r) ok = ok && all[(size_t)r].ok != 0;
  }
  // map the peers (rank order without this rank)
  int np = 0;
  if (ok) {
    for (int r = 0; r < sh->world && ok; ++r) {
      if (r == sh->rank) continue;
      void *pr = nullptr, *pf = nullptr;
      hipError_t e = hipIpcOpenMemHandle(&pr, all[(size_t)r].ring, hipIpcMemLazyEnablePeerAccess);
      const bool ring_open = e == hipSuccess;
      if (e == hipSuccess) e = hipIpcOpenMemHandle(&pf, all[(size_t)r].flags, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        if (pr) (void)hipIpcCloseMemHandle(pr);
        ok = false;
        snprintf(why, sizeof(why), "hipIpcOpenMemHandle (rank %d, %s): %s", r, ring_open ? "flags" : "ring", hipGetErrorString(e));
        break;
      }
      sh->peer_ring_map[np] = pr;
      sh->peer_flag_map[np] = pf;
      sh->peer_opened[np] = true;
      ++np;
    }
  }
  sh->n_real_peers = ok ? np : 0;
  for (int k = 0; ok && k < loopback; ++k) {  // diagnostic: scratch rings of this rank's own as further "peers"
    void *pr = nullptr, *pf = nullptr;
    hipError_t e = hipMalloc(&pr, ring_slots * slot_b * sh->world);
    if (e == hipSuccess) e = hipMalloc(&pf, n_flags * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(pf, 0, n_flags * sizeof(unsigned long long));
    if (e != hipSuccess) {
      (void)hipGetLastError();
      if (pr) (void)hipFree(pr);
      if (pf) (void)hipFree(pf);
      ok = false;
      snprintf(why, sizeof(why), "loopback ring: %s", hipGetErrorString(e));
      break;
    }
    sh->peer_ring_map[np] = pr;
    sh->peer_flag_map[np] = pf;
    sh->peer_opened[np] = false;
    ++np;
  }
  sh->n_peers = ok ? np : 0;
  // device table: [np4] ring bases (padded to a multiple of four entries: the wide store fetches them four at a time) |
  // [np + 1] flag arrays, this rank's own last
  const int np4 = (np + 3) / 4 * 4;
  if (ok) {
    std::vector<void *> tab((size_t)(np4 + np + 1));
    for (int i = 0; i < np4; ++i) tab[(size_t)i] = np > 0 ? sh->peer_ring_map[i < np ? i : np - 1] : nullptr;
    for (int i = 0; i < np; ++i) tab[(size_t)(np4 + i)] = sh->peer_flag_map[i];
    tab[(size_t)(np4 + np)] = sh->pflags;
    hipError_t e = hipMalloc(&sh->d_peer_tab, tab.size() * sizeof(void *));
    if (e == hipSuccess) e = hipMemcpy(sh->d_peer_tab, tab.data(), tab.size() * sizeof(void *), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      ok = false;
      snprintf(why, sizeof(why), "peer table: %s", hipGetErrorString(e));
    }
  }
  // token round trip: every rank writes (token base + its rank) into its slot of the test row on every rank, then waits
  // for the tokens of all ranks in its own test row
  if (ok && sh->world > 1) {
    const unsigned long long token = 0x7D5000000000ull + (unsigned long long)sh->rank + 1ull;
    unsigned long long *const *ftab = (unsigned long long *const *)((void **)sh->d_peer_tab + np4);
    // (the real peers come first in the table; loopback rings are this rank's own memory: nothing to learn from them)
    hipLaunchKernelGGL(tds_peer_token_kernel, dim3(1), dim3(64), 0, sh->comm_stream, (void *const *)sh->d_peer_tab, ftab, sh->n_real_peers,
                       (long long)((size_t)sh->rank * slot_b), (long long)sh->test_off(), sh->rank, token);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(sh->comm_stream);
    std::vector<unsigned long long> got((size_t)sh->world);
    const long long wait_ms = s->opt.get(TDS_OPT_SHARD_WAIT_MS, 2000);
    bool seen = false;
    for (long long t = 0; e == hipSuccess && t < wait_ms + 1000 && !seen; ++t) {  // (the peers may still be mapping: their set-up time counts)
      e = hipMemcpy(got.data(), sh->pflags + sh->test_off(), got.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      seen = true;
      for (int r = 0; r < sh->world; ++r) seen = seen && got[(size_t)r] == 0x7D5000000000ull + (unsigned long long)r + 1ull;
      if (!seen) {
        struct timespec ts = {0, 1000000};
        nanosleep(&ts, nullptr);
      }
    }
    if (e != hipSuccess || !seen) {
      (void)hipGetLastError();
      ok = false;
      snprintf(why, sizeof(why), "token round trip over the mapped memory failed%s%s", e != hipSuccess ? ": " : "",
               e != hipSuccess ? hipGetErrorString(e) : "");
    } else {
      // ... and every peer's ring token is in THIS rank's ring, in that peer's block of slot 0 (then zeroed again)
      for (int r = 0; r < sh->world && ok; ++r) {
        if (r == sh->rank) continue;
        unsigned int word = 0u;
        void *const at = (char *)sh->rgath + (size_t)r * slot_b;
        e = hipMemcpy(&word, at, sizeof(word), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemset(at, 0, sizeof(word));
        if (e != hipSuccess || word != (unsigned int)(0x7D5000000000ull + (unsigned long long)r + 1ull)) {
          (void)hipGetLastError();
          ok = false;
          snprintf(why, sizeof(why), "rank %d's flag token arrived before its store into the gathered ring (read %#x)", r, word);
        }
      }
    }
  }
  // the verdict, all ranks together (a rank that failed late must take the others with it)
  if (sh->world > 1) {
    PeerHello v;
    memset(&v, 0, sizeof(v));
    v.ok = ok ? 1 : 0;
    const int rc = comm_all_gather_bytes(sh, &v, all.data(), sizeof(PeerHello));
    if (rc != TDS_OK) return rc;
    for (int r = 0; r < sh->world; ++r) ok = ok && all[(size_t)r].ok != 0;
  }
  if (!ok) {
    peer_unmap(sh);
    if (sh->d_peer_tab) (void)hipFree(sh->d_peer_tab);
    sh->d_peer_tab = nullptr;
    if (want == 2) {
      snprintf(g_err, sizeof(g_err), "option shard_peer = 2: the peer-store exchange could not be set up on every rank (%s)",
               why[0] ? why : (possible ? "another rank failed" : "not available for this shard"));
      return TDS_ERR_UNSUPPORTED;
    }
    if (possible && why[0])
      fprintf(stderr, "tds_hip_shard (rank %d): peer-store exchange not available (%s) — RCCL all-gather instead\n", sh->rank, why);
    return TDS_OK;
  }
  sh->peer_mode = true;
  return TDS_OK;
}

int ring_alloc_impl(tds_hip_shard *sh) {
  tds_hip_sim *s = sh->sim;
  const size_t slot_b = sh->slot_scalars() * sh->wire_bytes;
  // In-place exchange (default): the gathered buffer is the only copy — the step-loop launch stores the records of step k
  // into THIS rank's block of slot k (tds_hip_rings_t::obs_slot_envs = world n_local), ncclAllGather runs with
  // sendbuff == recvbuff + rank * count.  On one rank nothing is left to move at all; on G ranks the local block is
  // neither copied nor sent to itself.  (option shard_inplace = 0: separate send ring, as in round 3)
  sh->inplace = s->opt.get(TDS_OPT_SHARD_INPLACE, 1) != 0;
  {
    long long c = s->opt.get(TDS_OPT_SHARD_CHUNK, TDS_SHARD_CHUNK_DEFAULT);
    sh->chunk = (int)(c < 8 ? 8 : (c > TDS_SHARD_CHUNK_MAX ? TDS_SHARD_CHUNK_MAX : c));
  }
  const size_t ring_slots = 2 * (size_t)sh->chunk;
  if (!sh->inplace) {
    TDS_HIP_TRY(hipMalloc(&sh->rwire, ring_slots * slot_b));
    TDS_HIP_TRY(hipMemset(sh->rwire, 0, ring_slots * slot_b));
  }
  // The gathered ring is written by OTHER GPUs in the peer-store exchange (system-scope stores over xGMI into this rank's
  // memory): it is allocated UNCACHED — not held in this GPU's L2 — where that can happen (several ranks, peer stores not
  // ruled out), as RCCL allocates its own intra-node buffers: a line of the ring that this GPU's L2 still holds from the
  // slot's previous use would otherwise be read instead of what the peer has written since (coarse-grained memory is coherent
  // between GPUs at kernel boundaries of its OWNER only).  One rank: ordinary device memory.
  // (TDS_HIP_SHARD_RING_UNCACHED = 0 / 1 forces either — for measuring what the uncached ring costs this rank's own stores.)
  {
    bool uncached = sh->world > 1 && s->opt.get(TDS_OPT_SHARD_PEER, 1) != 0 && s->opt.get(TDS_OPT_SHARD_RING, 1) != 0;
    if (const char *e = getenv("TDS_HIP_SHARD_RING_UNCACHED")) uncached = atoi(e) != 0;
    const size_t bytes = ring_slots * slot_b * sh->world;
    if (uncached && hipExtMallocWithFlags(&sh->rgath, bytes, hipDeviceMallocUncached) != hipSuccess) {
      (void)hipGetLastError();
      sh->rgath = nullptr;
    }
    sh->ring_uncached = sh->rgath != nullptr;
    if (!sh->rgath) TDS_HIP_TRY(hipMalloc(&sh->rgath, bytes));
    TDS_HIP_TRY(hipMemset(sh->rgath, 0, bytes));
  }
  // y records on 128-byte line boundaries (the launch then writes whole lines only)
  {
    const int per_line = 128 / (int)s->elem;
    int ys = (s->model.output_dim + per_line - 1) / per_line * per_line;
    if (s->opt.is_set(TDS_OPT_Y_STRIDE) && s->opt.v[TDS_OPT_Y_STRIDE] >= s->model.output_dim) ys = (int)s->opt.v[TDS_OPT_Y_STRIDE];
    sh->ry_stride = ys;
  }
  TDS_HIP_TRY(hipMalloc(&sh->ry, (size_t)TDS_SHARD_Y_SLOTS * sh->n_local * sh->ry_stride * s->elem));
  TDS_HIP_TRY(hipMalloc((void **)&sh->progress, (ring_slots + 2) * sizeof(unsigned long long)));
  TDS_HIP_TRY(hipMemset(sh->progress, 0, (ring_slots + 2) * sizeof(unsigned long long)));
  sh->slot_uses.assign(ring_slots, 0ull);
  TDS_HIP_TRY(hipHostMalloc((void **)&sh->host_latch, sizeof(unsigned), hipHostMallocMapped));
  *sh->host_latch = 0u;
  for (int i = 0; i < 2; ++i) {
    TDS_HIP_TRY(hipEventCreateWithFlags(&sh->ev_kernel[i], hipEventDisableTiming));
    TDS_HIP_TRY(hipEventCreateWithFlags(&sh->ev_comm[i], hipEventDisableTiming));
  }
  TDS_HIP_TRY(hipEventCreateWithFlags(&sh->cap_fork, hipEventDisableTiming));
  TDS_HIP_TRY(hipEventCreateWithFlags(&sh->cap_kernel, hipEventDisableTiming));
  TDS_HIP_TRY(hipEventCreateWithFlags(&sh->cap_comm, hipEventDisableTiming));
  // how the communication stream follows the counter: a one-lane wait kernel of our own (bounded: it gives up after
  // shard_wait_ms and raises the latch), or — option shard_wait = 1 — a hipStreamWaitValue64 command.  Measured
  // (profiles/r04_ubench_wait_value.txt, r04_ring_exchange_forms.txt): on ROCm 7.2 the latter is ALSO a one-wave kernel
  // (__amd_rocclr_streamOpsWait in the trace), not a command-processor wait — it needs a wavefront slot just the same,
  // has no timeout, and follows only counters in DEVICE memory at speed (signal / host memory: 700 us per step).
  sh->wait_value = false;
  if (s->opt.get(TDS_OPT_SHARD_WAIT, 0) == 1) {
    int can = 0;
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, s->device) == hipSuccess && can) sh->wait_value = true;
  }
  // user-buffer registration of the receive ring (RCCL >= 2.19; harmless where the transport ignores it)
  if (sh->comm && s->opt.get(TDS_OPT_SHARD_REGISTER, 1) != 0 && rccl() && rccl()->CommRegister) {
    if (rccl()->CommRegister(sh->comm, sh->rgath, ring_slots * slot_b * sh->world, &sh->reg_handle) != ncclSuccess)
      sh->reg_handle = nullptr;  // (not fatal: the collective works on unregistered buffers)
  }
  // peer-store exchange: decided here, once, by all ranks together (a collective when the shard has a communicator)
  {
    const int rc = peer_setup(sh, ring_slots, slot_b);
    if (rc != TDS_OK) return rc;
  }
  s->shard_ring_shaped = true;  // (tds_hip_set_option refuses the options read above from now on)
  // The memsets above are commands of the NULL stream; the communication stream is a non-blocking stream, which the NULL
  // stream does not order: behind a long launch of the caller's they are still pending when the first wait of the
  // exchange starts polling — a recycled allocation then shows it the counters of an earlier ring and it sends its slot
  // at once (tests/test_multi_gpu.py: ..._only_when_the_slowest_workgroup_has_stored_it, zeros in one slot).  Once per
  // shard: the ring exists — zeroed — before anything of the exchange is enqueued.
  TDS_HIP_TRY(hipDeviceSynchronize());
  return TDS_OK;
}
int ring_alloc(tds_hip_shard *sh) {
  if (sh->ring_ready) return TDS_OK;
  const int rc = ring_alloc_impl(sh);
  if (rc != TDS_OK) {
    ring_free(sh);  // a partial allocation is undone: the next call starts from nothing instead of dereferencing NULL
    return rc;
  }
  sh->ring_ready = true;
  return TDS_OK;
}

void ring_free(tds_hip_shard *sh) {
  for (auto &g : sh->rgraph)
    if (g.exec) {
      (void)hipGraphExecDestroy(g.exec);
      g.exec = nullptr;
    }
  if (sh->reg_handle && sh->comm && rccl() && rccl()->CommDeregister) (void)rccl()->CommDeregister(sh->comm, sh->reg_handle);
  sh->reg_handle = nullptr;
  peer_unmap(sh);
  if (sh->d_peer_tab) (void)hipFree(sh->d_peer_tab);
  if (sh->pflags) (void)hipFree(sh->pflags);
  if (sh->parrive) (void)hipFree(sh->parrive);
  sh->d_peer_tab = nullptr;
  sh->pflags = nullptr;
  sh->parrive = nullptr;
  sh->peer_mode = false;
  if (sh->rwire) (void)hipFree(sh->rwire);
  if (sh->rgath) (void)hipFree(sh->rgath);
  if (sh->ry) (void)hipFree(sh->ry);
  if (sh->progress) (void)hipFree(sh->progress);
  if (sh->host_latch) (void)hipHostFree(sh->host_latch);
  if (sh->cap_fork) (void)hipEventDestroy(sh->cap_fork);
  if (sh->cap_kernel) (void)hipEventDestroy(sh->cap_kernel);
  if (sh->cap_comm) (void)hipEventDestroy(sh->cap_comm);
  for (int i = 0; i < 2; ++i) {
    if (sh->ev_kernel[i]) (void)hipEventDestroy(sh->ev_kernel[i]);
    if (sh->ev_comm[i]) (void)hipEventDestroy(sh->ev_comm[i]);
    sh->ev_kernel[i] = sh->ev_comm[i] = nullptr;
  }
  sh->cap_fork = sh->cap_kernel = sh->cap_comm = nullptr;
  sh->rwire = sh->rgath = sh->ry = nullptr;
  sh->progress = nullptr;
  sh->host_latch = nullptr;
  sh->ring_ready = false;
}

bool ring_form(const tds_hip_shard *sh, int n_steps) {
  const tds_hip_sim *s = sh->sim;
  if (sh->block != 1 || s->auto_reset) return false;  // (auto-reset: the refill passes of the reset pool are host-driven)
  if (s->opt.get(TDS_OPT_SHARD_RING, 1) == 0) return false;
  // (the 16-lane kernel of the legged robots has a step-loop form, but not the exchange's part of it — progress counters,
  //  peer stores: their shards keep the per-step launches, which run on that kernel's straight-line form)
  if (s->compute_f64() && s->h64.quad) return false;
  // (the 8-lane kernel hands its exchange launches to the general kernel: tds_oct_takes; what the ring form needs is that
  //  kernel's step-loop form, judged as if the handle had no 8-lane kernel)
  return tds_hip_step_many_is_loop(s, n_steps > 1 ? n_steps : 2) != 0;
}

// one chunk on the CURRENT streams (sim->stream / comm_stream; under capture both belong to the capture)
int ring_chunk(tds_hip_shard *sh, const void *actions_dev, int pool, const TdsRingChunk &ck, bool capturing) {
  tds_hip_sim *s = sh->sim;
  const int h = ck.half;
  const size_t slot_b = sh->slot_scalars() * sh->wire_bytes;
  // (events recorded inside a capture belong to the capture: it has its own set)
  const hipEvent_t e_kernel = capturing ? sh->cap_kernel : sh->ev_kernel[h], e_comm = capturing ? sh->cap_comm : sh->ev_comm[h];
  // (peer-store exchange: nothing of the communication stream reads this rank's ring — the credit kernel in front of the
  //  launch is what keeps the ranks within a ring half of each other)
  if (!capturing && sh->comm_pending[h] && !sh->peer_mode) {  // the exchanges of the launch two back read this half of the ring
    TDS_HIP_TRY(hipStreamWaitEvent(s->stream, sh->ev_comm[h], 0));
    sh->comm_pending[h] = false;
  }
  // The counters are never reset in the eager form: the wait for step k targets (uses of its slot so far + 1) n_blocks,
  // which no earlier launch can have reached — no fill kernel, no fork event between the two streams (round 3 had both
  // per launch).  A captured chunk must replay with fixed targets: it zeroes its half's counters first, and forks the
  // communication stream off the capture's origin.
  unsigned long long *const counters = sh->progress + ck.slot0;  // (the kernel's slot k of this launch = ring slot slot0 + k)
  if (capturing) {
    TDS_HIP_TRY(hipMemsetAsync(counters, 0, (size_t)sh->chunk * sizeof(unsigned long long), s->stream));
    TDS_HIP_TRY(hipEventRecord(sh->cap_fork, s->stream));
    TDS_HIP_TRY(hipStreamWaitEvent(sh->comm_stream, sh->cap_fork, 0));
  }
  tds_hip_rings_t r;
  memset(&r, 0, sizeof(r));
  if (sh->inplace) {  // slot k of the chunk = this rank's block of gathered slot slot0 + k
    r.obs_ring = (char *)sh->rgath + ((size_t)ck.slot0 * sh->world + sh->rank) * slot_b;
    r.obs_slot_envs = sh->world * sh->n_local;
  } else {
    r.obs_ring = (char *)sh->rwire + (size_t)ck.slot0 * slot_b;
  }
  r.obs_slots = sh->chunk;
  r.obs_first = 0;
  r.obs_f32 = (sh->wire_bytes == 4 && s->elem == 8) ? 1 : 0;
  r.y_ring = sh->ry;
  r.y_slots = TDS_SHARD_Y_SLOTS;
  r.y_first = 0;
  r.y_stride = sh->ry_stride;
  // Which build the launch takes decides how its slots are exchanged.  The one-wave build (option exchange_w2 = 0) leaves
  // the exchange's kernels room beside it: the communication stream follows the slots' counters and sends slot k while
  // the launch runs step k + 1.  The two-wavefront build (the default: the kernel N = 1 runs) fills every SIMD's register
  // file — a wait or an all-gather gets onto a compute unit only when the launch retires, so per-slot waits buy nothing
  // and cost a kernel each (measured on one rank: 15.6 against 13.8 us per step, profiles/
  // r04_one_rank_exchange_with_table.txt): the launch then runs exactly as at N = 1 — no progress counter, streaming
  // stores — and the communication stream waits for its completion ONCE and sends the launch's slots as one RCCL group.
  if (sh->peer_mode && !capturing) {
    // PEER-STORE EXCHANGE: the launch is the N = 1 launch (two-wavefront build, no progress counter polled by anybody) whose
    // recorder stores every record on every rank; see the kernels at the top of the file for the three small kernels around it
    const unsigned long long seq = (unsigned long long)sh->chunks + 1ull;  // the same on every rank: all make the same calls
    const long long timeout_ticks = s->opt.get(TDS_OPT_SHARD_WAIT_MS, 2000) * 100000ll;  // 100 MHz
    const int np = sh->n_peers;
    unsigned long long *const *ftab = (unsigned long long *const *)((void **)sh->d_peer_tab + (np + 3) / 4 * 4);
    if (sh->n_real_peers > 0) {
      hipLaunchKernelGGL(tds_peer_credit_kernel, dim3(1), dim3(64), 0, s->stream, ftab, np, (long long)sh->credit_off(), sh->rank,
                         sh->world, seq, sh->wait_err(), timeout_ticks, sh->host_latch);
      if (hipGetLastError() != hipSuccess) return fail(TDS_ERR_HIP, "peer-store exchange: credit kernel launch");
    }
    // STAGED form (option shard_peer_copy = 1; full records only): the launch sees no peer — it stores into this rank's ring and
    // raises this rank's own flags — and the communication stream moves the launch's slots behind it
    const bool staged = s->opt.get(TDS_OPT_SHARD_PEER_COPY, 0) != 0 && !sh->reward_done_only && np > 0;
    TdsPeerLaunch pl;
    pl.rings = (const void *const *)sh->d_peer_tab;
    pl.flags = staged ? ftab + np : ftab;  // (the table's last entry is this rank's own flag array)
    pl.arrive = sh->parrive + (size_t)ck.slot0 * TDS_PEER_ARRIVE_STRIDE;
    pl.ring_off = (long long)(((size_t)ck.slot0 * sh->world + sh->rank) * slot_b);
    pl.epoch = seq;
    pl.n_peers = staged ? 0 : np;
    pl.flag_off = ck.slot0 * sh->world + sh->rank;
    pl.flag_stride = sh->world;
    pl.reward_done_only = sh->reward_done_only ? 1 : 0;
    pl.wide_ok = 1;
    r.progress = nullptr;
    s->peer_launch = &pl;
    const int rc = tds_hip_step_many_rings(s, actions_dev, pool, ck.act_first, ck.steps, &r);
    s->peer_launch = nullptr;
    if (rc != TDS_OK) return rc;
    TDS_HIP_TRY(hipEventRecord(e_kernel, s->stream));
    TDS_HIP_TRY(hipStreamWaitEvent(sh->comm_stream, e_kernel, 0));
    if (staged) {
      // this rank's block of slot slot0 + k lies at the same offset of every ring: ck.steps rows of slot_b bytes, a ring slot
      // (world blocks) apart — ONE strided copy per peer, issued to the runtime's copy path (SDMA engines between devices)
      const size_t off = ((size_t)ck.slot0 * sh->world + sh->rank) * slot_b, pitch = (size_t)sh->world * slot_b;
      for (int pr = 0; pr < np; ++pr)
        TDS_HIP_TRY(hipMemcpy2DAsync((char *)sh->peer_ring_map[pr] + off, pitch, (const char *)sh->rgath + off, pitch, slot_b,
                                     (size_t)ck.steps, hipMemcpyDeviceToDevice, sh->comm_stream));
      hipLaunchKernelGGL(tds_peer_raise_kernel, dim3(1), dim3(64), 0, sh->comm_stream, ftab, np, ck.slot0 * sh->world + sh->rank,
                         sh->world, ck.steps, seq);
      if (hipGetLastError() != hipSuccess) return fail(TDS_ERR_HIP, "staged peer exchange: flag kernel launch");
    }
    if (sh->n_real_peers > 0) {  // (one rank: the launch's completion IS the arrival of everything anybody stores here)
      hipLaunchKernelGGL(tds_peer_arrived_kernel, dim3(1), dim3(64), 0, sh->comm_stream,
                         (const unsigned long long *)(sh->pflags + (size_t)ck.slot0 * sh->world), ck.steps * sh->world, seq,
                         sh->wait_err(), timeout_ticks, sh->host_latch);
      if (hipGetLastError() != hipSuccess) return fail(TDS_ERR_HIP, "peer-store exchange: arrival kernel launch");
    }
    TDS_HIP_TRY(hipEventRecord(e_comm, sh->comm_stream));
    sh->exchange_form = staged ? TDS_EXCHANGE_PEER_COPY : TDS_EXCHANGE_PEER_STORES;
    return TDS_OK;
  }
  const int n_blocks = tds_hip_step_many_rings_blocks(s);
  const bool after_launch = s->opt.get(TDS_OPT_EXCHANGE_W2, 1) != 0 && s->opt.get(TDS_OPT_LOOP_W2, 1) != 0 &&
                            s->w2_max_blocks > 0 && n_blocks <= s->w2_max_blocks && s->lds_w2.NDP <= 16;
  r.progress = after_launch ? nullptr : counters;
  int rc = tds_hip_step_many_rings(s, actions_dev, pool, ck.act_first, ck.steps, &r);
  if (rc != TDS_OK) return rc;
  TDS_HIP_TRY(hipEventRecord(e_kernel, s->stream));
  if (after_launch) {
    TDS_HIP_TRY(hipStreamWaitEvent(sh->comm_stream, e_kernel, 0));
    if (sh->comm) {
      NCCL_TRY(rccl()->GroupStart());
      for (int k = 0; k < ck.steps; ++k) {
        char *const dst = (char *)sh->rgath + (size_t)(ck.slot0 + k) * slot_b * sh->world;
        const void *src = sh->inplace ? (const void *)(dst + (size_t)sh->rank * slot_b)
                                      : (const void *)((const char *)sh->rwire + (size_t)(ck.slot0 + k) * slot_b);
        const ncclResult_t gr = rccl()->AllGather(src, dst, sh->slot_scalars(), sh->wire_bytes == 8 ? ncclFloat64 : ncclFloat32,
                                                  sh->comm, sh->comm_stream);
        if (gr != ncclSuccess) {
          (void)rccl()->GroupEnd();
          snprintf(g_err, sizeof(g_err), "ncclAllGather (ring exchange, grouped) failed: %s", rccl()->GetErrorString(gr));
          return TDS_ERR_HIP;
        }
      }
      NCCL_TRY(rccl()->GroupEnd());
      if (!capturing) sh->comm_warm = true;
    } else if (!sh->inplace) {  // (one rank, no communicator: the launch's slots are contiguous in both rings)
      TDS_HIP_TRY(hipMemcpyAsync((char *)sh->rgath + (size_t)ck.slot0 * slot_b, (const char *)sh->rwire + (size_t)ck.slot0 * slot_b,
                                 (size_t)ck.steps * slot_b, hipMemcpyDeviceToDevice, sh->comm_stream));
    }
    TDS_HIP_TRY(hipEventRecord(e_comm, sh->comm_stream));
    sh->exchange_form = TDS_EXCHANGE_RCCL_AFTER_LAUNCH;
    return TDS_OK;
  }
  sh->exchange_form = TDS_EXCHANGE_RCCL_PER_SLOT;
  const long long timeout_ticks = s->opt.get(TDS_OPT_SHARD_WAIT_MS, 2000) * 100000ll;  // 100 MHz
  const bool nothing_to_move = sh->inplace && sh->world == 1 && !sh->comm;
  for (int k = 0; k < ck.steps; ++k) {
    const unsigned long long rel = tds_ring_wait_target(k, ck.steps, n_blocks);
    if (rel == 0ull) {
      TDS_HIP_TRY(hipStreamWaitEvent(sh->comm_stream, e_kernel, 0));
    } else if (nothing_to_move) {
      continue;  // (one rank without a communicator, records already where the gather would put them: the slot is
                 //  complete when the launch is — nothing follows the counter)
    } else if (sh->wait_value && !capturing) {
      const unsigned long long target = sh->slot_uses[ck.slot0 + k] * (unsigned long long)n_blocks + rel;
      TDS_HIP_TRY(hipStreamWaitValue64(sh->comm_stream, counters + k, target, hipStreamWaitValueGte, ~0ull));
    } else {
      const unsigned long long target = (capturing ? 0ull : sh->slot_uses[ck.slot0 + k] * (unsigned long long)n_blocks) + rel;
      hipLaunchKernelGGL(tds_ring_wait_kernel, dim3(1), dim3(64), 0, sh->comm_stream, counters + k, target,
                         sh->wait_err(), timeout_ticks, sh->host_latch);
      if (hipGetLastError() != hipSuccess) return fail(TDS_ERR_HIP, "ring exchange: wait kernel launch");
    }
    char *const dst = (char *)sh->rgath + (size_t)(ck.slot0 + k) * slot_b * sh->world;
    const void *src = sh->inplace ? (const void *)(dst + (size_t)sh->rank * slot_b)
                                  : (const void *)((const char *)sh->rwire + (size_t)(ck.slot0 + k) * slot_b);
    if (sh->comm) {
      NCCL_TRY(rccl()->AllGather(src, dst, sh->slot_scalars(), sh->wire_bytes == 8 ? ncclFloat64 : ncclFloat32, sh->comm,
                                 sh->comm_stream));
      if (!capturing) sh->comm_warm = true;
    } else if (!sh->inplace) {
      TDS_HIP_TRY(hipMemcpyAsync(dst, src, slot_b, hipMemcpyDeviceToDevice, sh->comm_stream));
    }
  }
  TDS_HIP_TRY(hipEventRecord(e_comm, sh->comm_stream));
  if (!capturing)
    for (int k = 0; k + 1 < ck.steps; ++k) sh->slot_uses[ck.slot0 + k]++;
  return TDS_OK;
}

// bookkeeping after a chunk has been submitted (eagerly or as a graph launch)
void ring_submitted(tds_hip_shard *sh, const TdsRingChunk &ck) {
  const size_t slot_b = sh->slot_scalars() * sh->wire_bytes;
  sh->comm_pending[ck.half] = true;
  sh->steps += ck.steps;
  sh->chunks++;
  sh->last_ptr = (char *)sh->rgath + (size_t)(ck.slot0 + ck.steps - 1) * slot_b * sh->world;
  sh->last_ev = sh->ev_comm[ck.half];
  sh->last_block = 1;
  sh->last_slot = 0;
  sh->last_ring_slot0 = ck.slot0;
  sh->last_ring_steps = ck.steps;
  sh->last_ring_half = ck.half;
}

tds_hip_shard::RingGraph *ring_graph_find(tds_hip_shard *sh, const void *actions, int pool, const TdsRingChunk &ck) {
  for (auto &g : sh->rgraph)
    if (g.exec && g.actions == actions && g.pool == pool && g.first == ck.act_first && g.steps == ck.steps && g.half == ck.half)
      return &g;
  return nullptr;
}

// capture one chunk — memset, step-loop launch, waits, all-gathers — into a graph of its own
tds_hip_shard::RingGraph *ring_graph_build(tds_hip_shard *sh, const void *actions, int pool, const TdsRingChunk &ck) {
  tds_hip_sim *s = sh->sim;
  tds_hip_shard::RingGraph *slot = &sh->rgraph[0];
  for (auto &g : sh->rgraph) {
    if (!g.exec) {
      slot = &g;
      break;
    }
    if (g.used < slot->used) slot = &g;
  }
  if (slot->exec) {
    (void)hipGraphExecDestroy(slot->exec);
    slot->exec = nullptr;
  }
  if (!sh->graph_stream && hipStreamCreateWithFlags(&sh->graph_stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
  hipStream_t user = s->stream;
  s->stream = sh->graph_stream;  // capture origin (the handle's own stream may be the NULL stream)
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamBeginCapture(sh->graph_stream, hipStreamCaptureModeThreadLocal);
  int rc = TDS_OK;
  if (e == hipSuccess) {
    rc = ring_chunk(sh, actions, pool, ck, true);
    // join: the graph is complete when the last slot has been exchanged
    if (rc == TDS_OK && hipStreamWaitEvent(sh->graph_stream, sh->cap_comm, 0) != hipSuccess) rc = TDS_ERR_HIP;
    e = hipStreamEndCapture(sh->graph_stream, &graph);
  }
  s->stream = user;
  if (e == hipSuccess && rc == TDS_OK && graph) {
    e = hipGraphInstantiate(&slot->exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) slot->exec = nullptr;
  }
  if (graph) (void)hipGraphDestroy(graph);
  if (!slot->exec) {
    (void)hipGetLastError();
    fprintf(stderr, "tds_hip_shard_step_many: capture of the ring exchange refused (%s) — submitting it eagerly\n",
            e != hipSuccess ? hipGetErrorString(e) : tds_hip_last_error());
    return nullptr;
  }
  slot->actions = actions;
  slot->pool = pool;
  slot->first = ck.act_first;
  slot->steps = ck.steps;
  slot->half = ck.half;
  return slot;
}

// a wait of the ring exchange that gave up (tds_ring_wait_kernel raised the pinned latch): the all-gathers behind it sent
// slots that may not have been written — every later call of the shard fails until tds_hip_shard_flush has reported it
int ring_latched(const tds_hip_shard *sh) {
  if (sh->host_latch && *(volatile unsigned *)sh->host_latch != 0u)
    return fail(TDS_ERR_HIP, "ring exchange: a wait for the step-loop launch timed out — gathered records are not valid "
                             "(tds_hip_shard_flush clears the condition)");
  return TDS_OK;
}

int ring_many(tds_hip_shard *sh, const void *actions_dev, int pool, int first, int n_steps, bool run) {
  tds_hip_sim *s = sh->sim;
  int rc = ring_alloc(sh);
  if (rc != TDS_OK) return rc;
  rc = ring_latched(sh);
  if (rc != TDS_OK) return rc;
  if (sh->comm && !sh->comm_warm) {
    // the first collective of a communicator sets up its transport connections between the ranks: eagerly, never inside
    // a stream capture — one warm-up all-gather of a scratch slot, on every rank
    if (sh->one_process_group)
      return fail(TDS_ERR_INVALID_ARG, "shards of tds_hip_shard_create_all: call tds_hip_shard_group_step once before "
                                       "tds_hip_shard_step_many (the first collective must be issued as a group)");
    // (slot 0 of the ring; in place: this rank's block of it)
    {
      const size_t slot_b0 = sh->slot_scalars() * sh->wire_bytes;
      const void *src0 = sh->inplace ? (const void *)((const char *)sh->rgath + (size_t)sh->rank * slot_b0) : (const void *)sh->rwire;
      NCCL_TRY(rccl()->AllGather(src0, sh->rgath, sh->slot_scalars(), sh->wire_bytes == 8 ? ncclFloat64 : ncclFloat32,
                                 sh->comm, sh->comm_stream));
    }
    TDS_HIP_TRY(hipStreamSynchronize(sh->comm_stream));
    sh->comm_warm = true;
  }
  TdsRingChunk plan[4096 / 8 + 1];
  const int nc = tds_ring_plan(sh->chunks, n_steps, first, pool, plan, (int)(sizeof(plan) / sizeof(plan[0])), sh->chunk);
  if (nc < 0) return fail(TDS_ERR_INVALID_ARG, "n_steps too large");
  // Submitted eagerly by default: 2 host calls per step (wait kernel, all-gather), issued while the launch runs.  As
  // ONE hipGraph per launch (TDS_HIP_SHARD_GRAPH=1) the same nodes replay ~10 us per step SLOWER on ROCm 7 — measured,
  // profiles/r03_ring_exchange_forms.txt: a chain of 128 dependent kernel / copy nodes pays a node-to-node latency the
  // stream does not.
  // (peer-store exchange: a launch carries its sequence number — three small kernels per launch need no graph either)
  const bool want_graph = s->opt.get(TDS_OPT_SHARD_GRAPH, 0) == 1 && !s->opt.flag(TDS_OPT_SHARD_NO_GRAPH) && !sh->peer_mode;
  for (int i = 0; i < nc; ++i) {
    const TdsRingChunk &ck = plan[i];
    tds_hip_shard::RingGraph *g = want_graph ? ring_graph_find(sh, actions_dev, pool, ck) : nullptr;
    if (want_graph && !g) g = ring_graph_build(sh, actions_dev, pool, ck);
    if (!run) continue;  // (prepare: the graphs only)
    if (g) {
      for (int h = 0; h < 2; ++h)  // a graph is ordered behind the stream it is launched into: that stream waits for
        if (sh->comm_pending[h]) {  // whatever the communication stream still has in flight
          TDS_HIP_TRY(hipStreamWaitEvent(s->stream, sh->ev_comm[h], 0));
          sh->comm_pending[h] = false;
        }
      TDS_HIP_TRY(hipGraphLaunch(g->exec, s->stream));
      g->used = ++sh->rgraph_clock;
      // (the captured chunk zeroes its half's counters and leaves n_blocks in those of its steps but the last)
      for (int k = 0; k < sh->chunk; ++k) sh->slot_uses[ck.slot0 + k] = k + 1 < ck.steps ? 1ull : 0ull;
      // consumers of tds_hip_shard_gathered wait on ev_comm[half]: record it behind the graph (complete = exchanged)
      TDS_HIP_TRY(hipEventRecord(sh->ev_comm[ck.half], s->stream));
    } else {
      rc = ring_chunk(sh, actions_dev, pool, ck, false);
      if (rc != TDS_OK) return rc;
    }
    ring_submitted(sh, ck);
  }
  return TDS_OK;
}

}  // namespace

extern "C" {

int tds_hip_shard_rccl_version(void) {
  Rccl *r = rccl();
  int v = 0;
  if (!r || r->GetVersion(&v) != ncclSuccess) return 0;
  return v;
}

int tds_hip_shard_unique_id(void *id_out) {
  if (!id_out) return fail(TDS_ERR_INVALID_ARG, "id_out is NULL");
  Rccl *r = rccl();
  if (!r) return fail(TDS_ERR_UNSUPPORTED, "librccl could not be loaded (set TDS_HIP_RCCL_LIB)");
  static_assert(sizeof(ncclUniqueId) == TDS_SHARD_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  NCCL_TRY(r->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return TDS_OK;
}

int tds_hip_shard_create(const tds_model_t *model, int global_envs, int rank, int world, int device, int dtype,
                         const void *unique_id, int wire_dtype, tds_hip_shard_t **out) {
  if (!out) return fail(TDS_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  if (world > 1 && !unique_id) return fail(TDS_ERR_INVALID_ARG, "world > 1 needs the unique id of tds_hip_shard_unique_id");
  tds_hip_shard *sh = nullptr;
  int rc = make_shard(model, global_envs, rank, world, device, dtype, wire_dtype, &sh);
  if (rc != TDS_OK) return rc;
  if (unique_id) {
    Rccl *r = rccl();
    if (!r) {
      tds_hip_shard_destroy(sh);
      return fail(TDS_ERR_UNSUPPORTED, "librccl could not be loaded (set TDS_HIP_RCCL_LIB)");
    }
    DeviceGuard guard(device);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t e = r->CommInitRank(&sh->comm, world, id, rank);
    if (e != ncclSuccess) {
      snprintf(g_err, sizeof(g_err), "ncclCommInitRank failed: %s", r->GetErrorString(e));
      sh->comm = nullptr;
      tds_hip_shard_destroy(sh);
      return TDS_ERR_HIP;
    }
  }
  *out = sh;
  return TDS_OK;
}

int tds_hip_shard_create_all(const tds_model_t *model, int global_envs, int n_devices, const int *devices, int dtype,
                             int wire_dtype, tds_hip_shard_t **out) {
  if (!out || !devices || n_devices < 1 || n_devices > 64) return fail(TDS_ERR_INVALID_ARG, "bad device list");
  for (int i = 0; i < n_devices; ++i) out[i] = nullptr;
  Rccl *r = rccl();
  if (!r) return fail(TDS_ERR_UNSUPPORTED, "librccl could not be loaded (set TDS_HIP_RCCL_LIB)");
  int rc = TDS_OK;
  for (int i = 0; i < n_devices && rc == TDS_OK; ++i)
    rc = make_shard(model, global_envs, i, n_devices, devices[i], dtype, wire_dtype, &out[i]);
  ncclComm_t comms[64];
  if (rc == TDS_OK) {
    ncclResult_t e = r->CommInitAll(comms, n_devices, devices);
    if (e != ncclSuccess) {
      snprintf(g_err, sizeof(g_err), "ncclCommInitAll failed: %s", r->GetErrorString(e));
      rc = TDS_ERR_HIP;
    }
  }
  if (rc != TDS_OK) {
    for (int i = 0; i < n_devices; ++i) {
      tds_hip_shard_destroy(out[i]);
      out[i] = nullptr;
    }
    return rc;
  }
  for (int i = 0; i < n_devices; ++i) {
    out[i]->comm = comms[i];
    out[i]->one_process_group = n_devices > 1;
  }
  return TDS_OK;
}

int tds_hip_shard_destroy(tds_hip_shard_t *sh) {
  if (!sh) return TDS_OK;
  const int device = sh->sim ? sh->sim->device : 0;
  {
    DeviceGuard guard(device);
    if (sh->comm_stream) (void)hipStreamSynchronize(sh->comm_stream);
    if (sh->graph_exec) (void)hipGraphExecDestroy(sh->graph_exec);
    ring_free(sh);
    if (sh->graph_stream) (void)hipStreamDestroy(sh->graph_stream);
    if (sh->comm && rccl()) (void)rccl()->CommDestroy(sh->comm);
    for (int i = 0; i < kSlots; ++i) {
      if (sh->wire[i] && sh->wire[i] != sh->rec[i]) (void)hipFree(sh->wire[i]);
      if (sh->rec[i]) (void)hipFree(sh->rec[i]);
      if (sh->gathered[i]) (void)hipFree(sh->gathered[i]);
      if (sh->ev_step[i]) (void)hipEventDestroy(sh->ev_step[i]);
      if (sh->ev_done[i]) (void)hipEventDestroy(sh->ev_done[i]);
    }
    if (sh->comm_stream) (void)hipStreamDestroy(sh->comm_stream);
  }
  if (sh->sim) tds_hip_destroy(sh->sim);
  delete sh;
  return TDS_OK;
}

tds_hip_sim_t *tds_hip_shard_sim(tds_hip_shard_t *sh) { return sh ? sh->sim : nullptr; }
int tds_hip_shard_rank(const tds_hip_shard_t *sh) { return sh ? sh->rank : -1; }
int tds_hip_shard_world(const tds_hip_shard_t *sh) { return sh ? sh->world : 0; }
int tds_hip_shard_local_envs(const tds_hip_shard_t *sh) { return sh ? sh->n_local : 0; }
int tds_hip_shard_first_env(const tds_hip_shard_t *sh) { return sh ? sh->rank * sh->n_local : 0; }
int tds_hip_shard_wire_bytes(const tds_hip_shard_t *sh) { return sh ? sh->wire_bytes : 0; }

int tds_hip_shard_exchange_form(const tds_hip_shard_t *sh) { return sh ? sh->exchange_form : 0; }
int tds_hip_shard_peer_count(const tds_hip_shard_t *sh) { return (sh && sh->peer_mode) ? sh->n_peers : -1; }

int tds_hip_shard_set_block(tds_hip_shard_t *sh, int steps_per_exchange) {
  if (!sh) return fail(TDS_ERR_INVALID_ARG, "shard is NULL");
  if (steps_per_exchange < 1 || steps_per_exchange > 1024) return fail(TDS_ERR_INVALID_ARG, "steps_per_exchange must be in 1..1024");
  int rc = tds_hip_shard_flush(sh);
  if (rc != TDS_OK) return rc;
  DeviceGuard guard(sh->sim->device);
  if (steps_per_exchange == sh->block) return TDS_OK;  // (nothing to reallocate)
  sh->block = steps_per_exchange;
  sh->steps = 0;
  sh->last_slot = -1;
  sh->last_ptr = nullptr;
  return alloc_ring(sh);
}

int tds_hip_shard_step(tds_hip_shard_t *sh, const void *actions_dev, int substeps) {
  if (!sh) return fail(TDS_ERR_INVALID_ARG, "shard is NULL");
  DeviceGuard guard(sh->sim->device);
  const int slot = (int)((sh->steps / sh->block) % kSlots);
  int rc = shard_step_local(sh, actions_dev, substeps);
  if (rc != TDS_OK) return rc;
  if (sh->steps % sh->block != 0) return TDS_OK;  // the block is still filling
  rc = shard_submit(sh, slot);
  if (rc != TDS_OK) return rc;
  sh->exchange_form = TDS_EXCHANGE_RCCL_PER_STEP;
  return shard_mark_done(sh, slot);
}

// One process, several devices: every shard steps, then ONE grouped RCCL call carries all their all-gathers
// (ncclGroupStart / End — issuing them one by one from a single thread would deadlock).
int tds_hip_shard_group_step(tds_hip_shard_t **shards, int n, const void *const *actions_dev, int substeps) {
  if (!shards || n < 1) return fail(TDS_ERR_INVALID_ARG, "no shards");
  for (int i = 0; i < n; ++i)
    if (!shards[i] || shards[i]->block != shards[0]->block || shards[i]->steps != shards[0]->steps)
      return fail(TDS_ERR_INVALID_ARG, "shards of one group must step together");
  const int slot = (int)((shards[0]->steps / shards[0]->block) % kSlots);
  for (int i = 0; i < n; ++i) {
    DeviceGuard guard(shards[i]->sim->device);
    int rc = shard_step_local(shards[i], actions_dev ? actions_dev[i] : nullptr, substeps);
    if (rc != TDS_OK) return rc;
  }
  if (shards[0]->steps % shards[0]->block != 0) return TDS_OK;
  Rccl *r = rccl();
  const bool grouped = r && shards[0]->comm;
  if (grouped) NCCL_TRY(r->GroupStart());
  int rc = TDS_OK;
  for (int i = 0; i < n && rc == TDS_OK; ++i) {
    DeviceGuard guard(shards[i]->sim->device);
    rc = shard_submit(shards[i], slot);
  }
  if (grouped) {
    ncclResult_t e = r->GroupEnd();
    if (e != ncclSuccess && rc == TDS_OK) {
      snprintf(g_err, sizeof(g_err), "ncclGroupEnd failed: %s", r->GetErrorString(e));
      rc = TDS_ERR_HIP;
    }
  }
  for (int i = 0; i < n && rc == TDS_OK; ++i) {
    DeviceGuard guard(shards[i]->sim->device);
    rc = shard_mark_done(shards[i], slot);
    if (grouped) shards[i]->comm_warm = true;
  }
  return rc;
}

// n_steps closed-loop steps of the shard INCLUDING their record exchanges as ONE hipGraph launch.  The two-stream
// pattern of tds_hip_shard_step (step on the sim stream, all-gather on the communication stream, ring of record
// blocks) is captured once — the cross-stream events become graph edges, the RCCL all-gathers graph nodes (RCCL
// supports stream capture) — and replayed; the host then issues one call per n_steps instead of ~8 per step, which
// is what bounds the eager form at ~20 us per step.  Every rank of the communicator must make the same call.
// n_steps must be a multiple of the exchange block.  Falls back to eager stepping when the capture is refused.
static int shard_many(tds_hip_shard_t *sh, const void *actions_dev, int action_blocks, int first_block, int n_steps,
                      bool run);
int tds_hip_shard_step_many(tds_hip_shard_t *sh, const void *actions_dev, int action_blocks, int first_block,
                            int n_steps) {
  return shard_many(sh, actions_dev, action_blocks, first_block, n_steps, true);
}
// capture + instantiate only (nothing executes): keeps the capture out of a timed region
int tds_hip_shard_step_many_prepare(tds_hip_shard_t *sh, const void *actions_dev, int action_blocks, int first_block,
                                    int n_steps) {
  return shard_many(sh, actions_dev, action_blocks, first_block, n_steps, false);
}
static int shard_many(tds_hip_shard_t *sh, const void *actions_dev, int action_blocks, int first_block, int n_steps,
                      bool run) {
  if (!sh) return fail(TDS_ERR_INVALID_ARG, "shard is NULL");
  if (n_steps < 1 || n_steps > 4096) return fail(TDS_ERR_INVALID_ARG, "n_steps must be in 1..4096");
  if (n_steps % sh->block != 0) return fail(TDS_ERR_INVALID_ARG, "n_steps must be a multiple of the exchange block");
  if (actions_dev && action_blocks < 1) return fail(TDS_ERR_INVALID_ARG, "action_blocks must be >= 1");
  tds_hip_sim *s = sh->sim;
  DeviceGuard guard(s->device);
  const int pool = actions_dev ? action_blocks : 1;
  const int first = actions_dev ? ((first_block % pool) + pool) % pool : 0;
  const size_t blk = (size_t)sh->n_local * s->model.action_dim * s->elem;
  // the K steps as step-loop launches that write a record ring, the exchange of a slot following the launch's own
  // progress counter: one launch per up to 64 steps instead of one per step (see "Ring exchange" above)
  if (ring_form(sh, n_steps)) return ring_many(sh, actions_dev, pool, first, n_steps, run);
  if (sh->comm && !sh->comm_warm && sh->one_process_group)
    return fail(TDS_ERR_INVALID_ARG, "shards of tds_hip_shard_create_all: call tds_hip_shard_group_step once before "
                                     "tds_hip_shard_step_many (the first collective must be issued as a group)");
  const bool eager_only = s->auto_reset || s->opt.flag(TDS_OPT_SHARD_NO_GRAPH);
  if (!eager_only) {
    if (sh->steps % sh->block != 0) {  // (a partially filled block of eager steps travels first)
      const int rc = tds_hip_shard_flush(sh);
      if (rc != TDS_OK) return rc;
    }
    // the graph always starts at ring slot 0 (one cached graph serves every call): exchanges of eager steps still in
    // flight are waited for by the stream the graph goes into, then the step count is realigned to a slot-0 boundary
    for (int i = 0; i < kSlots; ++i)
      if (sh->pending[i]) {
        TDS_HIP_TRY(hipStreamWaitEvent(s->stream, sh->ev_done[i], 0));
        sh->pending[i] = false;
      }
    sh->steps = 0;
  }
  const int slot0 = (int)((sh->steps / sh->block) % kSlots);
  // (auto-reset: the refill passes of the reset pool are host-driven — stepped eagerly, one exchange per block as ever)
  const bool cached = sh->graph_exec && sh->graph_actions == actions_dev && sh->graph_pool == pool &&
                      sh->graph_first == first && sh->graph_steps == n_steps && sh->graph_block == sh->block &&
                      sh->graph_slot0 == slot0;
  if (!cached && !eager_only) {
    if (sh->graph_exec) {
      (void)hipGraphExecDestroy(sh->graph_exec);
      sh->graph_exec = nullptr;
    }
    // exchanges in flight belong to the eager timeline: drain them, the capture starts from a clean ring
    TDS_HIP_TRY(hipStreamSynchronize(sh->comm_stream));
    for (int i = 0; i < kSlots; ++i) sh->pending[i] = false;
    if (!sh->graph_stream) TDS_HIP_TRY(hipStreamCreateWithFlags(&sh->graph_stream, hipStreamNonBlocking));
    if (sh->comm && !sh->comm_warm) {
      // the first collective of a communicator sets up its transport connections between the ranks: that must happen
      // eagerly, not inside a stream capture — one warm-up all-gather of a (scratch) ring slot, on every rank
      NCCL_TRY(rccl()->AllGather(sh->wire[0], sh->gathered[0], sh->block_scalars(),
                                 sh->wire_bytes == 8 ? ncclFloat64 : ncclFloat32, sh->comm, sh->comm_stream));
      TDS_HIP_TRY(hipStreamSynchronize(sh->comm_stream));
      sh->comm_warm = true;
    }
    hipStream_t user = s->stream;
    const long long steps0 = sh->steps;
    const int last0 = sh->last_slot;
    void *const last_ptr0 = sh->last_ptr;  // (shard_mark_done inside the capture sets these to events recorded in the
    const hipEvent_t last_ev0 = sh->last_ev;  //  capture only: nothing a consumer may wait on)
    const int last_block0 = sh->last_block;
    s->stream = sh->graph_stream;  // capture origin (the handle's own stream may be the NULL stream)
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamBeginCapture(sh->graph_stream, hipStreamCaptureModeThreadLocal);
    int rc = TDS_OK;
    if (e == hipSuccess) {
      for (int k = 0; k < n_steps && rc == TDS_OK; ++k) {
        const void *a = actions_dev ? (const char *)actions_dev + (size_t)((first + k) % pool) * blk : nullptr;
        const int slot = (int)((sh->steps / sh->block) % kSlots);
        rc = shard_step_local(sh, a, 1);
        if (rc == TDS_OK && sh->steps % sh->block == 0) {
          rc = shard_submit(sh, slot);
          if (rc == TDS_OK) rc = shard_mark_done(sh, slot);
        }
      }
      // join: the origin stream waits for every exchange of the graph (a capture must end with all forked work
      // joined; it also makes the graph's completion mean "all records exchanged")
      for (int i = 0; i < kSlots; ++i)
        if (sh->pending[i]) {
          if (rc == TDS_OK && hipStreamWaitEvent(sh->graph_stream, sh->ev_done[i], 0) != hipSuccess) rc = TDS_ERR_HIP;
          sh->pending[i] = false;
        }
      e = hipStreamEndCapture(sh->graph_stream, &graph);
    }
    sh->graph_last_slot = sh->last_slot;
    s->stream = user;
    sh->steps = steps0;  // (the capture executed nothing)
    sh->last_slot = last0;
    sh->last_ptr = last_ptr0;
    sh->last_ev = last_ev0;
    sh->last_block = last_block0;
    if (e == hipSuccess && rc == TDS_OK && graph) {
      e = hipGraphInstantiate(&sh->graph_exec, graph, nullptr, nullptr, 0);
      if (e != hipSuccess) sh->graph_exec = nullptr;
    }
    if (graph) (void)hipGraphDestroy(graph);
    if (sh->graph_exec) {
      sh->graph_actions = actions_dev;
      sh->graph_pool = pool;
      sh->graph_first = first;
      sh->graph_steps = n_steps;
      sh->graph_block = sh->block;
      sh->graph_slot0 = slot0;
    } else {
      (void)hipGetLastError();
      fprintf(stderr, "tds_hip_shard_step_many: graph capture refused (%s) — stepping eagerly\n",
              e != hipSuccess ? hipGetErrorString(e) : tds_hip_last_error());
    }
  }
  if (!run) return TDS_OK;
  if (sh->graph_exec && !eager_only) {
    // eager exchanges still in flight must not be overtaken by the graph's reuse of their slots
    for (int i = 0; i < kSlots; ++i)
      if (sh->pending[i]) {
        TDS_HIP_TRY(hipStreamWaitEvent(s->stream, sh->ev_done[i], 0));
        sh->pending[i] = false;
      }
    TDS_HIP_TRY(hipGraphLaunch(sh->graph_exec, s->stream));
    sh->exchange_form = TDS_EXCHANGE_RCCL_PER_STEP;
    sh->steps += n_steps;
    sh->last_slot = sh->graph_last_slot;
    // consumers of tds_hip_shard_gathered wait on ev_done[last_slot]: re-record it behind the graph
    TDS_HIP_TRY(hipEventRecord(sh->ev_done[sh->last_slot], s->stream));
    sh->last_ptr = sh->gathered[sh->last_slot];
    sh->last_ev = sh->ev_done[sh->last_slot];
    sh->last_block = sh->block;
    return TDS_OK;
  }
  for (int k = 0; k < n_steps; ++k) {
    const void *a = actions_dev ? (const char *)actions_dev + (size_t)((first + k) % pool) * blk : nullptr;
    int rc = tds_hip_shard_step(sh, a, 1);
    if (rc != TDS_OK) return rc;
  }
  return TDS_OK;
}

int tds_hip_shard_ring_plan(long long chunks_done, int n_steps, int act_first, int act_blocks, int n_blocks, int *out,
                            int cap) {
  if (!out || cap < 1 || n_steps < 1) return -1;
  TdsRingChunk plan[4096 / TDS_SHARD_CHUNK + 1];
  if (n_steps > 4096) return -1;
  const int nc = tds_ring_plan(chunks_done, n_steps, act_first, act_blocks, plan, (int)(sizeof(plan) / sizeof(plan[0])));
  if (nc < 0 || nc > cap) return -1;
  for (int i = 0; i < nc; ++i) {
    out[6 * i + 0] = plan[i].half;
    out[6 * i + 1] = plan[i].steps;
    out[6 * i + 2] = plan[i].step0;
    out[6 * i + 3] = plan[i].act_first;
    out[6 * i + 4] = plan[i].slot0;
    // what the communication stream waits for before it sends the chunk's FIRST slot (0: the launch's completion)
    out[6 * i + 5] = (int)tds_ring_wait_target(0, plan[i].steps, n_blocks);
  }
  return nc;
}
long long tds_hip_shard_gathered_offset(int global_env, int n_local, int width) {
  if (global_env < 0 || n_local < 1 || width < 1) return -1;
  return (long long)tds_gathered_offset(global_env, n_local, width);
}

int tds_hip_shard_flush(tds_hip_shard_t *sh) {
  if (!sh) return fail(TDS_ERR_INVALID_ARG, "shard is NULL");
  DeviceGuard guard(sh->sim->device);
  if (sh->steps % sh->block != 0) {  // a partially filled block travels as it is (stale tail records included)
    const int slot = (int)((sh->steps / sh->block) % kSlots);
    int rc = shard_submit(sh, slot);
    if (rc == TDS_OK) rc = shard_mark_done(sh, slot);
    if (rc != TDS_OK) return rc;
    sh->steps = (sh->steps / sh->block + 1) * sh->block;
  }
  TDS_HIP_TRY(hipStreamSynchronize(sh->sim->stream));  // (graph launches of the ring exchange live on the step stream)
  TDS_HIP_TRY(hipStreamSynchronize(sh->comm_stream));
  for (int i = 0; i < kSlots; ++i) sh->pending[i] = false;
  sh->comm_pending[0] = sh->comm_pending[1] = false;
  if (sh->progress) {  // a wait of the ring exchange that gave up (see tds_ring_wait_kernel)
    // (every wait that gives up also raises the pinned host word: no device round trip on the good path)
    unsigned err = 0;
    if (!sh->host_latch) TDS_HIP_TRY(hipMemcpy(&err, sh->wait_err(), sizeof(err), hipMemcpyDeviceToHost));
    if (err != 0u || (sh->host_latch && *(volatile unsigned *)sh->host_latch != 0u)) {
      TDS_HIP_TRY(hipMemset(sh->wait_err(), 0, sizeof(err)));
      if (sh->host_latch) *(volatile unsigned *)sh->host_latch = 0u;
      return fail(TDS_ERR_HIP, "ring exchange: a wait for the step-loop launch timed out — gathered records are not valid");
    }
  }
  return TDS_OK;
}

int tds_hip_shard_gathered_step(tds_hip_shard_t *sh, int steps_back, void *consumer_stream, void **records_dev) {
  if (!sh || !records_dev) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  if (sh->last_ring_slot0 < 0 || !sh->rgath || sh->last_ptr < sh->rgath)
    return fail(TDS_ERR_INVALID_ARG, "the most recent exchange was not a ring exchange");
  if (steps_back < 0 || steps_back >= sh->last_ring_steps)
    return fail(TDS_ERR_INVALID_ARG, "steps_back must lie inside the most recently submitted launch");
  {
    const int rc = ring_latched(sh);
    if (rc != TDS_OK) return rc;
  }
  DeviceGuard guard(sh->sim->device);
  TDS_HIP_TRY(hipStreamWaitEvent((hipStream_t)consumer_stream, sh->ev_comm[sh->last_ring_half], 0));
  const size_t slot_b = sh->slot_scalars() * sh->wire_bytes;
  *records_dev = (char *)sh->rgath + (size_t)(sh->last_ring_slot0 + sh->last_ring_steps - 1 - steps_back) * slot_b * sh->world;
  return TDS_OK;
}

int tds_hip_shard_gathered(tds_hip_shard_t *sh, void *consumer_stream, void **records_dev, int *steps_in_block) {
  if (!sh || !records_dev) return fail(TDS_ERR_INVALID_ARG, "NULL argument");
  if (!sh->last_ptr) return fail(TDS_ERR_INVALID_ARG, "no exchange submitted yet");
  {
    const int rc = ring_latched(sh);
    if (rc != TDS_OK) return rc;
  }
  DeviceGuard guard(sh->sim->device);
  // the consumer's stream waits for the exchange; the host does not
  TDS_HIP_TRY(hipStreamWaitEvent((hipStream_t)consumer_stream, sh->last_ev, 0));
  *records_dev = sh->last_ptr;
  if (steps_in_block) *steps_in_block = sh->last_block;
  return TDS_OK;
}

}  // extern "C"
