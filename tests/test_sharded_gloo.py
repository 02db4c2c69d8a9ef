"""CPU, world_size 2 over gloo: the N > 1 path of the bench (env sharding + the single
all-gather of [obs | reward | done] records), including ragged shards."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT  # noqa: F401

import tds_amd
from tds_amd.sharded import ObsGather, PipelinedObsGather, shard_bounds


def test_shard_bounds_cover_exactly():
    for n in (1, 7, 8, 4096, 65536, 65537):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_global, width, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = ObsGather(n_global, width, torch.float64, "cpu")
        lo, hi = shard_bounds(n_global, world, rank)
        assert g.local_size == hi - lo
        ok = True
        for step in range(3):
            # record of global env e at step s: e*1000 + column + s/10  (what a rank's kernel would write)
            env = torch.arange(lo, hi, dtype=torch.float64).unsqueeze(1)
            col = torch.arange(width, dtype=torch.float64).unsqueeze(0)
            local = env * 1000 + col + step / 10.0
            out = g(local)
            env_all = torch.arange(0, n_global, dtype=torch.float64).unsqueeze(1)
            expect = env_all * 1000 + col + step / 10.0
            ok = ok and bool(torch.equal(out, expect))
        # the pipelined variant of the bench loop: double-buffered records, gather of step i in flight
        # while step i + 1 is produced
        pg = PipelinedObsGather(n_global, width, torch.float64, "cpu")
        local2 = [torch.zeros((hi - lo, width), dtype=torch.float64) for _ in range(pg.slots)]
        for step in range(9):
            slot = step % pg.slots
            pg.before_reuse(slot)
            env = torch.arange(lo, hi, dtype=torch.float64).unsqueeze(1)
            col = torch.arange(width, dtype=torch.float64).unsqueeze(0)
            local2[slot].copy_(env * 1000 + col + step / 10.0)
            pg.submit(local2[slot], slot)
            if step > 0:  # the previous step's records are complete by now or after the wait
                prev = pg.result((step - 1) % pg.slots)
                env_all = torch.arange(0, n_global, dtype=torch.float64).unsqueeze(1)
                ok = ok and bool(torch.equal(prev, env_all * 1000 + col + (step - 1) / 10.0))
        pg.wait_all()
        # records produced in f64, travelling and arriving as f32 (bench.py --gather-dtype f32): integers < 2^24 are exact
        pw = PipelinedObsGather(n_global, width, torch.float64, "cpu", wire_dtype=torch.float32)
        local3 = [torch.zeros((hi - lo, width), dtype=torch.float64) for _ in range(pw.slots)]
        env_all = torch.arange(0, n_global, dtype=torch.float64).unsqueeze(1)
        for step in range(7):
            slot = step % pw.slots
            pw.before_reuse(slot)
            local3[slot].copy_(env * 1000 + col + step)
            pw.submit(local3[slot], slot)
            if step > 0:
                prev = pw.result((step - 1) % pw.slots)
                ok = ok and prev.dtype == torch.float32
                ok = ok and bool(torch.equal(prev, (env_all * 1000 + col + (step - 1)).to(torch.float32)))
        pw.wait_all()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_global", [64, 7])
def test_obs_gather_world2_gloo(n_global):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_global, 30, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in procs)
    [p.join(timeout=60) for p in procs]
    assert res == [(0, True), (1, True)]


def test_single_process_gather_is_identity():
    g = ObsGather(5, 3, torch.float64, "cpu")
    x = torch.arange(15, dtype=torch.float64).reshape(5, 3)
    assert torch.equal(g(x), x)


@pytest.mark.gpu
def test_pipelined_gather_side_stream_single_gpu():
    """CUDA path of PipelinedObsGather (side stream + events) with world size 1: every submitted record
    set comes back intact while the producer keeps overwriting the other buffer"""
    assert torch.cuda.is_available()
    n, w = 4096, 30
    pg = PipelinedObsGather(n, w, torch.float64, "cuda:0")
    bufs = [torch.zeros((n, w), dtype=torch.float64, device="cuda") for _ in range(pg.slots)]
    base = torch.arange(n * w, dtype=torch.float64, device="cuda").reshape(n, w)
    for step in range(20):
        slot = step % pg.slots
        pg.before_reuse(slot)
        bufs[slot].copy_(base + step)
        pg.submit(bufs[slot], slot)
        if step > 0:
            assert torch.equal(pg.result((step - 1) % pg.slots), base + (step - 1))
    pg.wait_all()
    torch.cuda.synchronize()
    # f64 records on an f32 wire: converted on the side stream into a staging buffer
    pw = PipelinedObsGather(n, w, torch.float64, "cuda:0", wire_dtype=torch.float32)
    for step in range(20):
        slot = step % pw.slots
        pw.before_reuse(slot)
        bufs[slot].copy_(base + step)
        pw.submit(bufs[slot], slot)
        if step > 0:
            got = pw.result((step - 1) % pw.slots)
            assert got.dtype == torch.float32 and torch.equal(got, (base + (step - 1)).to(torch.float32))
    pw.wait_all()
    torch.cuda.synchronize()
