"""The host-side protocol of an N-rank run (one process per GPU, torch.distributed: backend "nccl" == RCCL on the GPUs, "gloo" in
the CPU tests).  The DATA path of N > 1 is the C shard layer (csrc/tds_shard.hip: the step-loop launch stores its records into
every rank's gathered ring, or librccl's all-gather called from C); what goes through torch.distributed is only what is here:

  * the 128-byte communicator id, made on rank 0, broadcast to the others            (share_id)
  * "did every rank get through?" — one failed rank takes all ranks the same way      (any_rank)
  * the timed region: barrier on both sides, the MAXIMUM of the ranks' wall clocks     (open_region / close_region)
  * which contiguous block of the global environments a rank owns                      (block_of: the C layer's rule,
    rank r owns [r n_local, (r + 1) n_local): tds_gathered_offset, csrc/tds_shard_plan.h)

bench.py runs exactly these functions; tests/test_rank_protocol_gloo.py runs them with world_size 2 on CPU.
"""
from __future__ import annotations

import time


def _dist():
    import torch.distributed as dist

    return dist


def share_id(make_id, rank: int, world: int, device) -> bytes:
    """rank 0 calls make_id() -> 128 bytes; every rank returns the same 128 bytes"""
    import torch

    idt = torch.zeros(128, dtype=torch.uint8, device=device)
    if rank == 0:
        raw = bytes(make_id())
        assert len(raw) == 128, len(raw)
        idt.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
    if world > 1:
        _dist().broadcast(idt, src=0)
    return bytes(idt.cpu().numpy().tobytes())


def any_rank(flag: bool, world: int, device) -> bool:
    """True on EVERY rank if `flag` is true on ANY rank (a rank that failed must take the others the same way: a collective
    entered by some ranks only never returns)"""
    import torch

    if world <= 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    _dist().all_reduce(t, op=_dist().ReduceOp.MAX)
    return bool(int(t.item()))


def max_over_ranks(seconds: float, world: int, device) -> float:
    import torch

    if world <= 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    _dist().all_reduce(t, op=_dist().ReduceOp.MAX)
    return float(t.item())


def open_region(world: int, sync=None) -> float:
    """barrier + device synchronisation, then the clock: every rank starts its timed region together"""
    if world > 1:
        _dist().barrier()
    if sync is not None:
        sync()
    return time.perf_counter()


def close_region(t0: float, world: int, device, sync=None) -> float:
    """device synchronisation + barrier + synchronisation on the far side; the job's time = the slowest rank's"""
    if sync is not None:
        sync()
    if world > 1:
        _dist().barrier()
        if sync is not None:
            sync()
    return max_over_ranks(time.perf_counter() - t0, world, device)


def block_of(rank: int, n_local: int):
    """[lo, hi) of the global environment indices rank `rank` owns: equal contiguous blocks (the C layer's layout of
    a gathered slot, [world][n_local][width])"""
    return rank * n_local, (rank + 1) * n_local


def job_rate(world: int, n_local: int, steps: int, seconds: float) -> float:
    """whole-job env-steps/s: the units ALL ranks processed / the slowest rank's time"""
    return world * n_local * steps / seconds
