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
    """Rank ``src`` holds ``sd`` (CPU or device tensors); every rank returns the same dict on ``device``.
    All fp32 tensors travel as ONE flat blob in a single ``dist.broadcast`` (ring collectives over xGMI are
    per-link bound: one large message beats hundreds of small ones)."""
    rank = dist.get_rank()
    meta = [[(k, tuple(v.shape), str(v.dtype)) for k, v in sd.items()]] if rank == src else [None]
    dist.broadcast_object_list(meta, src=src, device=device if device.type == "cuda" else None)
    entries = meta[0]
    total = sum(int(torch.tensor(shape).prod()) if len(shape) else 1 for _, shape, _ in entries)
    blob = torch.empty(total, dtype=torch.float32, device=device)
    if rank == src:
        off = 0
        for k, shape, _ in entries:
            n = sd[k].numel()
            blob[off:off + n].copy_(sd[k].reshape(-1).to(torch.float32))
            off += n
    dist.broadcast(blob, src=src)
    out, off = {}, 0
    for k, shape, dtype in entries:
        n = 1
        for s in shape:
            n *= s
        out[k] = blob[off:off + n].view(shape).to(getattr(torch, dtype.split(".")[1])).clone()
        off += n
    return out
