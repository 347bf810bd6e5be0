"""Multi-GPU plumbing of the hot path: replica per GPU, utterances sharded, ONE weight broadcast.

Every utterance / stream is independent (own conv history, KV rings, codes), so the path shards with no data-path
collective.  The only communication is the start-up broadcast of the weight blob from rank 0
over RCCL/xGMI; after it each rank runs ``utterances[rank::world]`` on its own replica.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, TypeVar

import torch
import torch.distributed as dist

T = TypeVar("T")


def shard_utterances(items: Sequence[T], rank: int, world: int) -> List[T]:
    """``items[rank::world]`` -- the reference's own fan-out rule for offline tokenisation
    (MLLM_v2/egs/pretraining/local/offline_codec_tokenization.py:45-51: ``rank % device_count``)."""
    return list(items[rank::world])


def broadcast_state_dict(sd: Optional[Dict[str, torch.Tensor]], device: torch.device, src: int = 0,
                         template: Optional[Callable[[], Dict[str, torch.Tensor]]] = None) -> Dict[str, torch.Tensor]:
    """Rank ``src`` holds ``sd`` (CPU or device tensors, any dtypes); every rank returns the same dict on ``device``.
    All tensors travel as ONE flat byte blob in a single ``dist.broadcast`` (ring collectives over xGMI are per-link bound:
    one large message beats hundreds of small ones), each at its own dtype -- a bf16 checkpoint is not widened on the wire.
    Entries are 16-byte aligned inside the blob, so the returned tensors can be views of it (no second copy of a 15 GB LM)."""
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
    dist.broadcast(blob, src=src)
    out = {}
    for (k, shape, dtype), off in zip(entries, offsets):
        dt = getattr(torch, dtype.split(".")[1])
        n = 1
        for d in shape:
            n *= d
        nbytes = n * torch.empty((), dtype=dt).element_size()
        out[k] = blob[off:off + nbytes].view(dt).view(shape)
    return out
