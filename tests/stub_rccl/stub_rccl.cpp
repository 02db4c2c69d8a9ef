// stub_rccl.cpp — TEST INFRASTRUCTURE: a stand-in for librccl that lets TWO PROCESSES ON ONE GPU drive the C shard layer
// (csrc/tds_shard.hip) with world == 2.  The real RCCL refuses two ranks on one device, and the GPU box of the test
// tier has exactly one — without this the multi-rank path of tds_hip_shard_* would meet a second rank for the first
// time on the scaling run.  Loaded through TDS_HIP_RCCL_LIB (tds_shard.hip resolves every nccl* symbol with dlsym).
//
// ncclAllGather here is stream-ordered like the real one: D2H copy of the send buffer into a POSIX shared-memory
// segment, a cross-process barrier run as a host function of the stream, H2D copies of every rank's block, a second
// barrier (so that nobody overwrites a block somebody else is still reading).  All of it is capturable into a hipGraph
// (memcpy nodes + host nodes): barrier generations are counted when the host function RUNS, not when it is enqueued.
// Never linked into the product.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;

static const size_t kMaxRanks = 8, kBlock = 4u << 20;  // bytes per rank in the segment

struct Segment {
  std::atomic<long long> arrived[2];  // barrier A / B: total arrivals so far
  char pad[64 - 2 * sizeof(std::atomic<long long>)];
  char data[kMaxRanks * kBlock];
};

struct StubComm {
  int rank, world;
  Segment *seg;
  long long gen[2];  // generations of barrier A / B this rank has passed (advanced when the host function runs)
  char name[128];
};
typedef StubComm *ncclComm_t;

struct BarrierCtx {
  StubComm *c;
  int which;
};

static void barrier_fn(void *p) {
  BarrierCtx *b = (BarrierCtx *)p;
  StubComm *c = b->c;
  const long long target = (c->gen[b->which] + 1) * c->world;
  c->gen[b->which]++;
  c->seg->arrived[b->which].fetch_add(1, std::memory_order_acq_rel);
  const time_t t0 = time(nullptr);
  while (c->seg->arrived[b->which].load(std::memory_order_acquire) < target) {
    usleep(20);
    if (time(nullptr) - t0 > 60) {
      fprintf(stderr, "stub_rccl: rank %d waited 60 s at barrier %d (generation %lld) — a rank is missing\n", c->rank,
              b->which, c->gen[b->which]);
      abort();
    }
  }
}

ncclResult_t ncclGetVersion(int *v) {
  *v = 99999;
  return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t) { return "stub_rccl error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/tds_stub_rccl_%d_%ld", (int)getpid(), (long)time(nullptr));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int world, ncclUniqueId id, int rank) {
  if (world < 1 || world > (int)kMaxRanks || rank < 0 || rank >= world) return ncclInvalidArgument;
  id.internal[sizeof(id.internal) - 1] = 0;
  const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  if (ftruncate(fd, sizeof(Segment)) != 0) return ncclSystemError;  // (a fresh segment reads as zeros: counters start at 0)
  void *p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  if (hipHostRegister(p, sizeof(Segment), hipHostRegisterDefault) != hipSuccess) return ncclUnhandledCudaError;
  StubComm *c = new StubComm();
  c->rank = rank;
  c->world = world;
  c->seg = (Segment *)p;
  c->gen[0] = c->gen[1] = 0;
  strncpy(c->name, id.internal, sizeof(c->name) - 1);
  *comm = c;
  return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *) { return ncclInvalidArgument; }
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  (void)hipHostUnregister(c->seg);
  munmap(c->seg, sizeof(Segment));
  if (c->rank == 0) shm_unlink(c->name);
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclGroupStart(void) { return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { return ncclSuccess; }

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t c, hipStream_t stream) {
  const size_t bytes = count * (dt == ncclFloat64 ? 8 : dt == ncclFloat32 ? 4 : 1);
  if (bytes > kBlock) return ncclInvalidArgument;
  if (hipMemcpyAsync(c->seg->data + (size_t)c->rank * kBlock, send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess)
    return ncclUnhandledCudaError;
  BarrierCtx *a = new BarrierCtx{c, 0}, *b = new BarrierCtx{c, 1};  // (live as long as a graph may replay them)
  if (hipLaunchHostFunc(stream, barrier_fn, a) != hipSuccess) return ncclUnhandledCudaError;
  for (int r = 0; r < c->world; ++r)
    if (hipMemcpyAsync((char *)recv + (size_t)r * bytes, c->seg->data + (size_t)r * kBlock, bytes, hipMemcpyHostToDevice,
                       stream) != hipSuccess)
      return ncclUnhandledCudaError;
  if (hipLaunchHostFunc(stream, barrier_fn, b) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

}  // extern "C"
