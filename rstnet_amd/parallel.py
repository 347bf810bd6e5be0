"""Multi-GPU plumbing of the hot path: replica per GPU, utterances sharded, ONE weight broadcast.

Every utterance / stream is independent (own conv history, KV rings, codes), so the path shards with no data-path
collective.  The only communication is the start-up broadcast of the weight blob from rank 0
over RCCL/xGMI; after it each rank runs ``utterances[rank::world]`` on its own replica.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Dict, List, Optional, Sequence, TypeVar

import torch
import torch.distributed as dist

T = TypeVar("T")


def shard_utterances(items: Sequence[T], rank: int, world: int) -> List[T]:
    """``items[rank::world]`` -- the reference's own fan-out rule for offline tokenisation
    (MLLM_v2/egs/pretraining/local/offline_codec_tokenization.py:45-51: ``rank % device_count``)."""
    return list(items[rank::world])


def plan_affinity(allowed: Sequence[int], slot: int, slots: int, pool: Optional[Sequence[int]] = None) -> List[int]:
    """Host CPUs of one rank: the ``slot``-th of ``slots`` equal, disjoint slices of ``pool`` (the CPUs of the NUMA node the rank's GPU
    hangs off, shared by ``slots`` ranks) restricted to ``allowed`` -- or of ``allowed`` itself when no pool is known or it has fewer
    CPUs than sharers.  Pure function (unit-tested on the host)."""
    allowed = sorted(set(int(c) for c in allowed))
    if not allowed or slots <= 0:
        return []
    ok = set(allowed)
    cpus = [c for c in sorted(set(int(c) for c in pool)) if c in ok] if pool else allowed
    if len(cpus) < slots:
        cpus = allowed
    k = max(1, len(cpus) // slots)
    i = slot % max(1, len(cpus) // k)
    return cpus[i * k:(i + 1) * k]


def _gpu_numa_cpus(local_rank: int) -> Optional[List[int]]:
    """CPUs of the NUMA node of GPU ``local_rank`` (sysfs), or None when the platform does not say."""
    try:
        prop = torch.cuda.get_device_properties(local_rank)
        bus = f"{getattr(prop, 'pci_domain_id', 0):04x}:{prop.pci_bus_id:02x}:{prop.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bus}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus: List[int] = []
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus += list(range(int(lo), int(hi or lo) + 1))
        return cpus
    except (OSError, ValueError, AttributeError, RuntimeError, AssertionError):
        return None


def pin_rank_threads(local_rank: int, local_world: int) -> Dict[str, object]:
    """One process per GPU replays captured graphs and launches ~100 kernels per step from the host: eight ranks left to the
    scheduler migrate across sockets and fight over cores.  Gives this rank a disjoint slice of the host CPUs (NUMA-near its GPU
    when sysfs tells), and sizes torch's intra-op pool to it.  Returns what was applied (reported by ``bench.py --gpus N``)."""
    info: Dict[str, object] = {"cpus": None, "threads": torch.get_num_threads(), "numa": False}
    if local_world <= 1 or not hasattr(os, "sched_getaffinity"):
        return info
    try:
        allowed = sorted(os.sched_getaffinity(0))
        visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if 0 < visible < local_world:
            # every rank sees only its own GPU (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank): the other ranks' NUMA nodes
            # cannot be asked, so the rank takes slot local_rank of local_world of ITS node's CPUs -- NUMA-near, and disjoint from
            # every other rank whatever node that one sits on (same slot count everywhere)
            near = tuple(_gpu_numa_cpus(torch.cuda.current_device()) or ())
            mine = plan_affinity(allowed, local_rank, local_world, near) if near else plan_affinity(allowed, local_rank, local_world)
        else:
            nears = [tuple(_gpu_numa_cpus(r) or ()) if visible else () for r in range(local_world)]
            near = nears[local_rank]
            if near:        # the ranks whose GPUs share this NUMA node split its CPUs in rank order
                mine = plan_affinity(allowed, sum(1 for r in range(local_rank) if nears[r] == near), sum(1 for n in nears if n == near), near)
            else:
                mine = plan_affinity(allowed, local_rank, local_world)
        if mine:
            os.sched_setaffinity(0, mine)
            torch.set_num_threads(max(1, min(len(mine), 16)))
            info.update(cpus=f"{mine[0]}-{mine[-1]} ({len(mine)})", threads=torch.get_num_threads(), numa=bool(near))
    except OSError:
        pass
    return info


def broadcast_state_dict(sd: Optional[Dict[str, torch.Tensor]], device: torch.device, src: int = 0,
                         template: Optional[Callable[[], Dict[str, torch.Tensor]]] = None,
                         stats: Optional[Dict[str, float]] = None) -> Dict[str, torch.Tensor]:
    """Rank ``src`` holds ``sd`` (CPU or device tensors, any dtypes); every rank returns the same dict on ``device``.
    All tensors travel as ONE flat byte blob in a single ``dist.broadcast`` (ring collectives over xGMI are per-link bound:
    one large message beats hundreds of small ones), each at its own dtype -- a bf16 checkpoint is not widened on the wire.
    Entries are 16-byte aligned inside the blob, so the returned tensors can be views of it (no second copy of a 15 GB LM).
    Offsets are Python integers: blobs past 2^31 / 2^32 bytes (the 15 GB LM) need no special case.
    ``stats`` (optional dict) receives ``bytes`` and ``seconds`` (wall time of the broadcast itself, synchronised)."""
    rank = dist.get_rank()
    meta = [[(k, tuple(v.shape), str(v.dtype)) for k, v in sd.items()]] if rank == src else [None]
    dist.broadcast_object_list(meta, src=src, device=device if device.type == "cuda" else None)
    entries = meta[0]
    offsets, total = [], 0
    for _, shape, dtype in entries:
        dt = getattr(torch, dtype.split(".")[1])
        n = 1
        for d in shape:
            n *= d
        offsets.append(total)
        total += (n * torch.empty((), dtype=dt).element_size() + 15) // 16 * 16
    blob = torch.zeros(total, dtype=torch.uint8, device=device)
    if rank == src:
        for (k, shape, _), off in zip(entries, offsets):
            t = sd[k].detach().contiguous().reshape(-1)
            nbytes = t.numel() * t.element_size()
            blob[off:off + nbytes].copy_(t.view(torch.uint8) if t.numel() else t.new_empty(0, dtype=torch.uint8))
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    dist.broadcast(blob, src=src)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    if stats is not None:
        stats["bytes"] = stats.get("bytes", 0) + total
        stats["seconds"] = stats.get("seconds", 0.0) + (time.perf_counter() - t0)
    out = {}
    for (k, shape, dtype), off in zip(entries, offsets):
        dt = getattr(torch, dtype.split(".")[1])
        n = 1
        for d in shape:
            n *= d
        nbytes = n * torch.empty((), dtype=dt).element_size()
        out[k] = blob[off:off + nbytes].view(dt).view(shape)
    return out
