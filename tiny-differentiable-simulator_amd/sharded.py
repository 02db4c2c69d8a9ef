"""Multi-GPU layout of the hot path: environments are independent, so they shard across
ranks with NO collective inside the step; the only exchange is one all-gather per policy step of
the per-env record [obs | reward | done] that the step kernel writes.

One process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm, "gloo" for CPU tests).
"""
from __future__ import annotations


def shard_bounds(n_global: int, world: int, rank: int):
    """Contiguous block [lo, hi) of global environment indices owned by `rank`
    (block r = [r*N/G, (r+1)*N/G) with the remainder spread over the first ranks)."""
    base, rem = divmod(n_global, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class ObsGather:
    """All-gather of the per-env records.  Equal shards use a single
    ``all_gather_into_tensor`` (one direct exchange per peer over xGMI — every rank's shard goes
    to every peer on its own link; no ring); ragged shards pad to the largest shard."""

    def __init__(self, n_global: int, width: int, dtype, device, group=None):
        import torch
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_global = n_global
        self.width = width
        self.bounds = [shard_bounds(n_global, self.world, r) for r in range(self.world)]
        self.sizes = [hi - lo for lo, hi in self.bounds]
        self.equal = len(set(self.sizes)) == 1
        self.max_size = max(self.sizes)
        self.out = torch.zeros((n_global, width), dtype=dtype, device=device)
        if not self.equal:
            self._pad_in = torch.zeros((self.max_size, width), dtype=dtype, device=device)
            self._pad_out = torch.zeros((self.world * self.max_size, width), dtype=dtype, device=device)

    @property
    def local_size(self) -> int:
        return self.sizes[self.rank]

    def __call__(self, local):
        """local: [local_size, width] on this rank -> [n_global, width] on every rank (async on
        the current stream for nccl; the returned tensor is reused between calls)."""
        assert tuple(local.shape) == (self.local_size, self.width), (tuple(local.shape), self.local_size, self.width)
        if self.world == 1:
            self.out.copy_(local)
            return self.out
        if self.equal:
            self.dist.all_gather_into_tensor(self.out, local.contiguous(), group=self.group)
            return self.out
        self._pad_in[: self.local_size].copy_(local)
        self.dist.all_gather_into_tensor(self._pad_out, self._pad_in, group=self.group)
        for r, (lo, hi) in enumerate(self.bounds):
            self.out[lo:hi].copy_(self._pad_out[r * self.max_size: r * self.max_size + (hi - lo)])
        return self.out


class PipelinedObsGather:
    """The same exchange, overlapped with the NEXT step's kernel: the record buffer is double-buffered
    and each all-gather runs on a side stream (RCCL) / as an async work item (gloo), so that step i+1
    computes while the records of step i travel over xGMI.  Nothing in the step depends on the gathered
    records (per-environment policies run on device, `tds_hip_rollout`; a central learner consumes them
    one step late), which is what makes the overlap legal.

        g = PipelinedObsGather(n_global, width, dtype, device)
        for i in range(steps):
            slot = i % g.slots
            g.before_reuse(slot)              # the producer may overwrite local[slot] again
            produce(local[slot])              # e.g. HipSim.step(actions, 1, local[slot])
            g.submit(local[slot], slot)
        all_records = g.result(slot)          # [n_global, width] of the last submitted step
    """

    def __init__(self, n_global: int, width: int, dtype, device, group=None, slots: int = 4, wire_dtype=None):
        """wire_dtype: dtype the records travel and arrive in (default: `dtype`, the producer's).  The step computes
        and keeps its state in f64; the [obs | reward | done] records a policy / learner consumes can cross xGMI as
        f32 (SURVEY 8e sizes the exchange at 4 B per scalar): the block is converted into a per-slot staging buffer
        on the side stream, in front of the all-gather."""
        import torch

        self.torch = torch
        self.slots = slots
        self.wire_dtype = wire_dtype if wire_dtype is not None else dtype
        self.gathers = [ObsGather(n_global, width, self.wire_dtype, device, group) for _ in range(slots)]
        self.world = self.gathers[0].world
        self._stage = None
        if self.wire_dtype != dtype:
            self._stage = [torch.zeros((self.gathers[0].local_size, width), dtype=self.wire_dtype, device=device)
                           for _ in range(slots)]
        self.is_cuda = torch.device(device).type == "cuda"
        self._work = [None] * slots
        if self.is_cuda:
            self.stream = torch.cuda.Stream(device=device)
            self._done = [torch.cuda.Event() for _ in range(slots)]
            self._ready = [torch.cuda.Event() for _ in range(slots)]
            self._pending = [False] * slots

    @property
    def local_size(self) -> int:
        return self.gathers[0].local_size

    def submit(self, local, slot: int):
        torch = self.torch
        g = self.gathers[slot]
        if self.is_cuda:
            ready = self._ready[slot]
            ready.record()                    # on the producer's (current) stream
            self.stream.wait_event(ready)
            with torch.cuda.stream(self.stream):
                if self._stage is not None:
                    self._stage[slot].copy_(local)    # dtype conversion, on the side stream
                    local = self._stage[slot]
                g(local)
            self._done[slot].record(self.stream)
            self._pending[slot] = True
            return
        if self._stage is not None:
            self._stage[slot].copy_(local)
            local = self._stage[slot]
        if g.world > 1 and g.equal:
            self._work[slot] = g.dist.all_gather_into_tensor(g.out, local.contiguous(), group=g.group, async_op=True)
        else:
            g(local)

    def before_reuse(self, slot: int):
        """order the producer (current stream / host) after the gather that last read `local[slot]`"""
        if self.is_cuda:
            # a gather that the host can already see finished needs no stream-side wait (each cross-stream
            # wait costs the producer's stream a few microseconds): with a few slots in flight that is
            # the normal case
            if self._pending[slot] and not self._done[slot].query():
                self.torch.cuda.current_stream().wait_event(self._done[slot])
            self._pending[slot] = False
        elif self._work[slot] is not None:
            self._work[slot].wait()
            self._work[slot] = None

    def result(self, slot: int):
        self.before_reuse(slot)
        return self.gathers[slot].out

    def wait_all(self):
        for s in range(self.slots):
            self.before_reuse(s)
