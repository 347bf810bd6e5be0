"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: weight broadcast and utterance sharding."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rstnet_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(3)
        def make():
            g = torch.Generator().manual_seed(3)
            return {"a.weight": torch.randn(5, 7, generator=g), "b.bias": torch.randn(11, generator=g), "c": torch.ones(1),
                    "w.bf16": torch.randn(3, 5, generator=g).bfloat16(), "ids": torch.arange(7), "flag": torch.tensor([True, False]),
                    "empty": torch.zeros(0, 4)}
        sd = make() if rank == 0 else None
        got = parallel.broadcast_state_dict(sd, torch.device("cpu"), src=0)
        exp = make()
        ok = list(got) == list(exp) and all(got[k].dtype == exp[k].dtype and got[k].shape == exp[k].shape and torch.equal(got[k], exp[k])
                                            for k in exp)
        mine = parallel.shard_utterances(list(range(10)), rank, world)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ok = ok and sorted(sum(gathered, [])) == list(range(10)) and mine == list(range(10))[rank::world]
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}


def test_shard_utterances_partitions():
    items = list("abcdefg")
    parts = [parallel.shard_utterances(items, r, 3) for r in range(3)]
    assert sorted(sum(parts, [])) == sorted(items) and parts[0] == ["a", "d", "g"]


def test_plan_affinity_slices_are_disjoint_and_numa_local():
    allowed = list(range(128))
    flat = [parallel.plan_affinity(allowed, r, 8) for r in range(8)]
    assert all(len(s) == 16 for s in flat) and sorted(sum(flat, [])) == allowed
    # two NUMA nodes of 64 CPUs, four GPUs each: a rank's CPUs come from its GPU's node, four disjoint slices per node
    near = [parallel.plan_affinity(allowed, r % 4, 4, range(64 * (r // 4), 64 * (r // 4) + 64)) for r in range(8)]
    assert all(len(s) == 16 and s[0] // 64 == r // 4 for r, s in enumerate(near)) and sorted(sum(near, [])) == allowed
    # a cgroup that allows 8 CPUs, pool outside it: falls back to the allowed set; fewer CPUs than ranks: one each, wrapping
    assert parallel.plan_affinity(range(8), 1, 4, [100, 101]) == [2, 3]
    assert parallel.plan_affinity(range(3), 5, 8) == [2] and parallel.plan_affinity([], 0, 4) == []


def test_pin_rank_threads_when_each_rank_sees_only_its_own_gpu(monkeypatch):
    """HIP_VISIBLE_DEVICES per rank: device 0 is every rank's own GPU and the others cannot be asked.  Eight ranks, two NUMA
    nodes of 64 CPUs, GPUs 0-3 on node 0 and 4-7 on node 1: every rank stays on its GPU's node and no two slices overlap."""
    applied = {}
    monkeypatch.setattr(parallel.os, "sched_getaffinity", lambda pid: set(range(128)), raising=False)
    monkeypatch.setattr(parallel.os, "sched_setaffinity", lambda pid, cpus: applied.__setitem__("cpus", list(cpus)), raising=False)
    monkeypatch.setattr(torch, "set_num_threads", lambda n: None)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    slices = []
    for r in range(8):
        node = r // 4
        asked = []
        monkeypatch.setattr(parallel, "_gpu_numa_cpus", lambda i, node=node, asked=asked: asked.append(i) or list(range(64 * node, 64 * node + 64)))
        info = parallel.pin_rank_threads(r, 8)
        assert asked == [0] and info["numa"] is True              # only the visible device is queried
        assert all(c // 64 == node for c in applied["cpus"]) and len(applied["cpus"]) == 8
        slices.append(applied["cpus"])
    flat = sum(slices, [])
    assert len(flat) == len(set(flat))
    # all GPUs visible to every rank (the other launch style): the four ranks of a node split its 64 CPUs
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(parallel, "_gpu_numa_cpus", lambda i: list(range(64 * (i // 4), 64 * (i // 4) + 64)))
    slices = []
    for r in range(8):
        parallel.pin_rank_threads(r, 8)
        slices.append(applied["cpus"])
    assert all(len(c) == 16 and c[0] // 64 == r // 4 for r, c in enumerate(slices)) and sorted(sum(slices, [])) == list(range(128))


def _worker_big(rank, world, port, nbytes, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # three entries whose byte offsets inside the blob straddle 2^31: a small head, a body of `nbytes`, a tail after it
        def pattern(n, mul):       # byte i = i * mul % 251 (period 251: built by tiling, not from a 17 GB index vector)
            return (torch.arange(251) * mul % 251).to(torch.uint8).repeat(n // 251 + 1)[:n]
        sd = None
        if rank == 0:
            sd = {"head": torch.arange(37, dtype=torch.float32), "body": pattern(nbytes, 7), "tail.bf16": torch.arange(1000).bfloat16(),
                  "tail.i64": torch.arange(5) + (1 << 40)}
        stats = {}
        got = parallel.broadcast_state_dict(sd, torch.device("cpu"), src=0, stats=stats)
        ok = stats["bytes"] > nbytes and stats["seconds"] > 0
        ok = ok and torch.equal(got["head"], torch.arange(37, dtype=torch.float32)) and got["body"].numel() == nbytes
        # the tail entries live past the 2^31 boundary: offsets must not have wrapped
        ok = ok and torch.equal(got["tail.bf16"], torch.arange(1000).bfloat16()) and torch.equal(got["tail.i64"], torch.arange(5) + (1 << 40))
        probe = torch.tensor([0, 1, nbytes // 2, (1 << 31) - 1, 1 << 31, (1 << 31) + 12345, nbytes - 1])
        probe = probe[probe < nbytes]
        ok = ok and torch.equal(got["body"][probe], (probe * 7 % 251).to(torch.uint8))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_broadcast_world4_blob_past_2_gib():
    """world 4, a 2 GB-class blob -- entry offsets beyond 2^31 bytes (the 15 GB LM blob is far beyond)."""
    import psutil
    nbytes = (1 << 31) + (1 << 20)
    if psutil.virtual_memory().available < 6 * nbytes:
        pytest.skip("needs ~14 GB of host memory for four ranks")
    world, port = 4, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_big, args=(world, port, nbytes, ret), nprocs=world, join=True)
        assert dict(ret) == {r: True for r in range(world)}
