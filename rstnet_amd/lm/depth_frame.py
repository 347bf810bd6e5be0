"""Pointer tables of ``rst_depth_decode_frame`` (the depth phase of a frame as one persistent launch, csrc/lm_depth.hip) for a
depth transformer with per-step weights -- ``LMModel.depformer`` (models/model.py:188-225) or ``GPT.codecformer``
(models/llama_streaming.py:560-590): host arrays of device pointers, built once per weight version."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch


class DepthFrameTables:
    def __init__(self, dep, heads: Sequence[torch.Tensor], head_bias: Sequence[Optional[torch.Tensor]], emb: Sequence[torch.Tensor]):
        """``dep``: the ``StreamingTransformer`` (weights_per_step = dep_q); ``heads[k]`` bf16 ``[card, E]``; ``head_bias[k]`` fp32 ``[card]``
        or None; ``emb[k]`` bf16 ``[rows, E]`` = the embedding table of step k's INPUT token."""
        L, Q, E = len(dep.layers), dep.weights_per_step, dep.d_model
        assert len(heads) == Q and len(emb) == Q and len(head_bias) == Q
        self.L, self.dep_q, self.E, self.H = L, Q, E, dep.num_heads
        self.Hd = dep.layers[0].gating[0].linear_out.weight.shape[1]
        self.card = heads[0].shape[0]
        keep: List[torch.Tensor] = []

        def ptrs(ts):
            arr = (C.c_void_p * len(ts))()
            for i, t in enumerate(ts):
                if t is None:
                    arr[i] = None
                    continue
                assert t.is_cuda and t.is_contiguous(), "depth-frame tables need contiguous device tensors"
                keep.append(t)
                arr[i] = t.data_ptr()
            return arr
        for layer in dep.layers:
            att = layer.self_attn
            assert att.in_proj_weight.dtype == torch.bfloat16 and tuple(att.in_proj_weight.shape) == (Q * 3 * E, E)
            assert tuple(att.out_proj.weight.shape) == (Q * E, E)
            for g in layer.gating:
                assert tuple(g.linear_in.weight.shape) == (2 * self.Hd, E) and tuple(g.linear_out.weight.shape) == (E, self.Hd)
        self.in_proj = ptrs([l.self_attn.in_proj_weight for l in dep.layers])
        self.out_proj = ptrs([l.self_attn.out_proj.weight for l in dep.layers])
        self.norm1 = ptrs([l.norm1.alpha_f32() for l in dep.layers])
        self.norm2 = ptrs([l.norm2.alpha_f32() for l in dep.layers])
        self.gate_in = ptrs([l.gating[k].linear_in.weight for l in dep.layers for k in range(Q)])
        self.gate_out = ptrs([l.gating[k].linear_out.weight for l in dep.layers for k in range(Q)])
        for h, e in zip(heads, emb):
            assert h.dtype == torch.bfloat16 and tuple(h.shape) == (self.card, E) and e.dtype == torch.bfloat16 and e.shape[1] == E
        self.heads = ptrs(list(heads))
        self.head_bias = ptrs(list(head_bias)) if any(b is not None for b in head_bias) else None
        self.emb = ptrs(list(emb))
        self.emb_rows = (C.c_int * Q)(*[int(e.shape[0]) for e in emb])
        self.eps = float(dep.layers[0].norm1.eps)
        self.context = dep.context
        dev = heads[0].device
        from .. import ops
        # 4 words (csrc/persist.h): time-out codes of the frame in flight | frames repaired by the one-workgroup launch | OR of the
        # repaired frames' codes | reserved; registered with ops.persistent_poll
        self.status = ops.new_persistent_status(dev)
        self._keep = keep

    def repairs(self) -> int:
        """Frames whose hand-offs timed out and that the repair launch recomputed (reads the device words: synchronises)."""
        return int(self.status[1].item())

    def check(self) -> None:
        """Raises if any launch so far had a timed-out hand-off, repaired or not (reads the device status words: synchronises).  The
        tokens of a repaired frame are right; the check exists for tests and ``LMGen(check=True)`` runs that want to know."""
        st = self.status.tolist()
        if st[0] or st[1]:
            raise RuntimeError(f"rst_depth_decode_frame: hand-offs timed out ({st[1]} frame(s) repaired by the one-workgroup launch, codes "
                               f"{st[2]:#x}, in flight {st[0]:#x}); the device was shared with other work or the launch was not fully resident")
