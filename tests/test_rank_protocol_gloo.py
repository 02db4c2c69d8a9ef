"""The N > 1 host protocol on CPU: two processes, torch.distributed backend "gloo", through the functions bench.py itself runs
(tds_amd.ranks) and the C layer's shard arithmetic (tds_hip_shard_ring_plan, the layout of a gathered slot) — no GPU.

What an N-rank run needs from the host side (the data path is csrc/tds_shard.hip, covered on the GPU): every rank ends
up with the SAME communicator id; a failure on one rank is seen by all; the timed region is bracketed by barriers and the job's
time is the slowest rank's; every rank cuts the same call into the same launches / ring slots and agrees on where each
environment's record lies in a gathered slot."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        import time

        import torch
        import torch.distributed as dist

        from tds_amd import hip_backend, ranks

        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        out = {}
        # 1. the communicator id: made on rank 0 only
        calls = []

        def make_id():
            calls.append(rank)
            return bytes((7 * i + 1) % 251 for i in range(128))

        out["uid"] = ranks.share_id(make_id, rank, world, "cpu")
        out["id_made_on"] = list(calls)
        # 2. one failing rank takes all ranks the same way
        out["any"] = [ranks.any_rank(False, world, "cpu"), ranks.any_rank(rank == 1, world, "cpu"), ranks.any_rank(rank == 0, world, "cpu")]
        # 3. the timed region: rank 1 is slower; the job's time is ITS time on every rank
        t0 = ranks.open_region(world)
        time.sleep(0.05 + 0.25 * rank)
        out["elapsed"] = ranks.close_region(t0, world, "cpu")
        out["rate"] = ranks.job_rate(world, 4096, 20, out["elapsed"])
        # 4. shard arithmetic of the C layer (no device needed): the same plan on every rank, blocks in rank order
        out["plan"] = hip_backend.shard_ring_plan(chunks_done=3, n_steps=150, act_first=5, act_blocks=16, n_blocks=512)
        out["block"] = ranks.block_of(rank, 4096)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, out))
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put((rank, "ERR " + repr(e) + traceback.format_exc()))


def test_two_rank_host_protocol_on_cpu():
    import torch.multiprocessing as mp

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=180) for _ in range(world))
    [p.join(timeout=60) for p in ps]
    for r in range(world):
        assert not isinstance(res[r], str), res[r]
    a, b = res[0], res[1]
    assert a["uid"] == b["uid"] == bytes((7 * i + 1) % 251 for i in range(128))
    assert a["id_made_on"] == [0] and b["id_made_on"] == []
    assert a["any"] == b["any"] == [False, True, True]
    # the slow rank slept 0.30 s: both ranks report the job at >= that, and the same number
    assert a["elapsed"] == b["elapsed"] and a["elapsed"] >= 0.29
    assert a["rate"] == b["rate"] == world * 4096 * 20 / a["elapsed"]
    assert a["plan"] == b["plan"] and sum(c["steps"] for c in a["plan"]) == 150
    assert [c["half"] for c in a["plan"]] == [(3 + i) & 1 for i in range(len(a["plan"]))]
    assert a["block"] == (0, 4096) and b["block"] == (4096, 8192)


def test_block_rule_matches_the_gathered_slot_layout():
    """rank r's block of a gathered slot [world][n_local][w] starts at r * n_local records: block_of is that rule"""
    sys.path.insert(0, ROOT)
    from tds_amd import ranks

    n_local, world = 4093, 8
    lo_hi = [ranks.block_of(r, n_local) for r in range(world)]
    assert lo_hi[0][0] == 0 and all(lo_hi[r][1] == lo_hi[r + 1][0] for r in range(world - 1)) and lo_hi[-1][1] == world * n_local
    e = np.arange(world * n_local)
    assert np.array_equal(e // n_local, np.repeat(np.arange(world), n_local))
