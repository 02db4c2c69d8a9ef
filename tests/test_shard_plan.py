"""Host arithmetic of the multi-GPU ring exchange (csrc/tds_shard_plan.h through the C ABI) — runs without a GPU.

tds_hip_shard_step_many cuts a call into step-loop launches of up to 64 steps; launch j writes half j & 1 of the obs
ring, its step k owns slot half * 64 + k and is sent once the launch's progress counter has reached (k + 1) * n_blocks
(the last step: once the launch has completed).  The reference has no multi-device path; what is pinned here
is that every step of a call gets exactly one slot, that two launches in flight never share a slot, and that the
gathered layout puts global environment e where rank e // n_local wrote its local environment e % n_local.
"""
import numpy as np
import pytest

from tds_amd import hip_backend

CHUNK = 64


@pytest.mark.parametrize("n_steps", [1, 5, 20, 63, 64, 65, 128, 1000, 4096])
@pytest.mark.parametrize("chunks_done", [0, 1, 7])
def test_every_step_owns_one_slot_and_halves_alternate(built, n_steps, chunks_done):
    plan = hip_backend.shard_ring_plan(chunks_done, n_steps, act_first=3, act_blocks=16, n_blocks=1024)
    assert sum(c["steps"] for c in plan) == n_steps
    assert len(plan) == -(-n_steps // CHUNK)
    done = 0
    for j, c in enumerate(plan):
        assert 1 <= c["steps"] <= CHUNK
        assert c["half"] == (chunks_done + j) & 1 and c["slot0"] == c["half"] * CHUNK
        assert c["step0"] == done
        assert c["act_first"] == (3 + done) % 16  # step k of the call takes action block (first + k) % blocks
        # consecutive launches never write the same half: the one being exchanged is left alone
        if j:
            assert plan[j - 1]["half"] != c["half"]
        # the first slot of a launch of one step waits for the launch itself, otherwise for one round of workgroups
        assert c["first_wait"] == (0 if c["steps"] == 1 else 1024)
        done += c["steps"]
    slots = [c["slot0"] + k for c in plan[-2:] for k in range(c["steps"])]
    assert len(set(slots)) == len(slots) and max(slots) < 2 * CHUNK


def test_bad_arguments(built):
    import ctypes as C

    out = (C.c_int * 6)()
    L = hip_backend.lib()
    assert L.tds_hip_shard_ring_plan(0, 0, 0, 1, 1, out, 1) == -1
    assert L.tds_hip_shard_ring_plan(0, 5000, 0, 1, 1, out, 1) == -1
    assert L.tds_hip_shard_ring_plan(0, 200, 0, 1, 1, out, 1) == -1  # four chunks do not fit one entry
    assert L.tds_hip_shard_ring_plan(0, 10, 0, 1, 1, None, 1) == -1
    assert L.tds_hip_shard_gathered_offset(-1, 4, 3) == -1


@pytest.mark.parametrize("world,n_local,width", [(1, 4096, 30), (2, 2048, 30), (8, 8192, 30), (4, 3, 38)])
def test_gathered_layout_is_global_environment_order(built, world, n_local, width):
    """ncclAllGather concatenates the ranks' [n_local][width] blocks in rank order == contiguous shards of the global batch"""
    L = hip_backend.lib()
    n = world * n_local
    e = np.unique(np.concatenate([[0, n - 1, n_local - 1, min(n_local, n - 1)], np.random.default_rng(0).integers(0, n, 64)]))
    for g in e:
        assert L.tds_hip_shard_gathered_offset(int(g), n_local, width) == int(g) * width
