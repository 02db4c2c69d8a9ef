// tds_shard_plan.h — the host arithmetic of the ring exchange of tds_shard.hip, free of HIP: which launches a
// tds_hip_shard_step_many call is cut into, which ring slots their steps own, what the communication stream waits for
// before it sends a slot, where a global environment's record lies in a gathered slot.  Kept apart so that it runs (and
// is tested: tests/test_shard_plan.py, through tds_hip_shard_ring_plan of the C ABI) on a host without a GPU.
#pragma once
#include <stddef.h>

// steps per step-loop launch ("chunk") of the ring exchange; the obs ring holds two chunks (one being exchanged while the
// next one is written).  Default; a shard's option shard_chunk (8 .. 1024) overrides it: a launch costs ~15 us before / after
// its steps plus the gap to the next one, 0.3 us per step at 64 steps; the ring grows with it (world x 0.49 MB per slot at
// 4096 Ant environments per rank)
#define TDS_SHARD_CHUNK 64
#define TDS_SHARD_CHUNK_MAX 1024
// what a shard takes where the option is not set: measured on one rank (profiles/r04_same_box_ab_and_exchange_forms.txt,
// Ant x 4096, two-wavefront build under the exchange): 64-step launches 17.3 us per step, 256-step launches 15.1 — the
// launch's fixed ~15 us and the gap to the next one are paid once per chunk; the ring then holds 2 x 256 slots (8 ranks, f32
// wire: 2 GB of 288)
#define TDS_SHARD_CHUNK_DEFAULT 256
// slots of the shard's y ring (local records, never exchanged; step k of a chunk owns slot k % TDS_SHARD_Y_SLOTS)
#define TDS_SHARD_Y_SLOTS 16

struct TdsRingChunk {
  int half;       // which half of the obs ring the chunk writes (chunk index & 1)
  int steps;      // steps of the chunk (one step-loop launch)
  int step0;      // steps of the call that lie before the chunk
  int act_first;  // action block of the chunk's first step
  int slot0;      // ring slot of the chunk's first step (= half * TDS_SHARD_CHUNK; step k -> slot0 + k)
};

// the chunks of one call of n_steps steps, `chunks_done` chunks having been submitted before it; returns their number
// (-1: more than cap)
inline int tds_ring_plan(long long chunks_done, int n_steps, int act_first, int act_blocks, TdsRingChunk *out, int cap,
                         int chunk = TDS_SHARD_CHUNK) {
  int n = 0;
  for (int done = 0; done < n_steps; ++n) {
    if (n >= cap) return -1;
    const int c = n_steps - done < chunk ? n_steps - done : chunk;
    TdsRingChunk &k = out[n];
    k.half = (int)((chunks_done + n) & 1);
    k.steps = c;
    k.step0 = done;
    k.act_first = act_blocks > 0 ? (act_first + done) % act_blocks : 0;
    k.slot0 = k.half * chunk;
    done += c;
  }
  return n;
}

// The step-loop launch counts a workgroup in for step k — on the counter of step k's OWN ring slot — while it runs step
// k + 1 (TdsStepCtl::progress), never for its last step: slot k of a chunk of c steps may be sent when the slot's counter
// has grown by n_blocks since the slot was last used (k < c - 1) — every workgroup has then stored that step's records,
// however far apart the workgroups of the launch have drifted — or, for k == c - 1, when the launch has completed
// (returns 0: wait for the launch's event instead).  (Round 3 kept ONE running total per launch and waited for
// (k + 1) * n_blocks: met as soon as the AVERAGE workgroup had passed step k, not the slowest one.)
inline unsigned long long tds_ring_wait_target(int k, int chunk_steps, int n_blocks) {
  return k < chunk_steps - 1 ? (unsigned long long)n_blocks : 0ull;
}

// scalar offset of the record of GLOBAL environment e in a gathered slot [world][n_local][width] (ncclAllGather lays the
// ranks' blocks out in rank order; rank r owns the environments [r n_local, (r + 1) n_local))
inline size_t tds_gathered_offset(int e, int n_local, int width) {
  return ((size_t)(e / n_local) * (size_t)n_local + (size_t)(e % n_local)) * (size_t)width;
}
