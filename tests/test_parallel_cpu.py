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
