"""Streaming transformer of the Mimi codec with the reference's module surface (``modules/transformer.py`` of
the MimiCodec tokenizer copy): LayerNorm -> fused-QKV attention with interleaved RoPE and a ring KV cache ->
LayerScale residual -> LayerNorm -> GELU FFN -> LayerScale residual, all in fp32.

Kernel plan per layer (6 launches): layernorm | in_proj GEMM | rope+split(+ring append) | attention |
out_proj GEMM with fused ``x + scale * .`` epilogue | layernorm | linear1 GEMM with fused GELU |
linear2 GEMM with fused ``x + scale * .``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
from torch import nn

from .. import ops
from .conv import _to_ncl, _to_nlc
from .streaming import StreamingContainer, StreamingModule


class Linear(nn.Module):
    """``nn.Linear`` parameter layout (``weight [out, in]``) executed by rst_linear_f32."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, device=device, dtype=dtype))
        self.bias = nn.Parameter(torch.empty(out_features, device=device, dtype=dtype)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if self.bias is not None:
            nn.init.uniform_(self.bias, -1 / in_features ** 0.5, 1 / in_features ** 0.5)

    def forward(self, x, **epilogue):
        return ops.linear(x.contiguous(), self.weight, self.bias, **epilogue)


class LayerNorm(nn.Module):
    """``nn.LayerNorm(dim, eps)`` with affine parameters (``create_norm_fn('layer_norm')``, transformer.py:113-114)."""

    def __init__(self, dim: int, eps: float = 1e-5, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.normalized_shape = (dim,)
        self.weight = nn.Parameter(torch.ones(dim, device=device, dtype=dtype))
        self.bias = nn.Parameter(torch.zeros(dim, device=device, dtype=dtype))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.layernorm(x.contiguous(), self.weight, self.bias, self.eps)


class LayerScale(nn.Module):
    """Learnt diagonal rescaling of a residual branch (transformer.py:68-100); fused into GEMM epilogues by the layer."""

    def __init__(self, channels: int, init: float = 1e-4, channel_last: bool = True, device=None, dtype=None):
        super().__init__()
        self.channel_last = channel_last
        self.scale = nn.Parameter(torch.full((channels,), init, device=device, dtype=dtype))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise RuntimeError("LayerScale is applied inside the producing GEMM's epilogue; call the owning layer instead")


def create_norm_fn(norm_type: str, dim: int, **kwargs) -> nn.Module:
    if norm_type == "layer_norm":
        return LayerNorm(dim, eps=1e-5, **kwargs)
    raise NotImplementedError(f"norm {norm_type!r}: the fp32 codec transformer uses 'layer_norm'")


class RotaryEmbedding(nn.Module):
    """Holder of ``max_period``; the rotation itself is fused into the QKV split kernel (modules/rope.py)."""

    def __init__(self, max_period: float = 10000.0):
        super().__init__()
        self.max_period = max_period


@dataclass
class _MHAState:
    k_cache: torch.Tensor   # [B, H, capacity, D]
    v_cache: torch.Tensor
    offset: torch.Tensor    # int64 [1] on device (kept for API parity / graph capture)
    offset_cpu: int
    shared: Optional[torch.Tensor] = None   # the enclosing transformer's counter: one increment per step instead of one per layer

    def reset(self) -> None:
        self.offset.zero_()
        self.offset_cpu = 0


class StreamingMultiheadAttention(StreamingModule[_MHAState]):
    """``modules/transformer.py:293-423``: fused in-projection (optionally one weight set per step), RoPE, ring KV
    cache with the reference's slot->position map, masked attention, out-projection."""

    def __init__(self, embed_dim: int, num_heads: int, causal: bool = False, context: Optional[int] = None,
                 rope: Optional[RotaryEmbedding] = None, weights_per_step: int = 0, device=None, dtype=None):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.causal, self.context, self.rope = causal, context, rope
        self.weights_per_step = weights_per_step
        mult = weights_per_step if weights_per_step else 1
        if not causal:
            raise NotImplementedError("only causal attention is on the hot path")
        self.in_proj_weight = nn.Parameter(torch.empty(mult * 3 * embed_dim, embed_dim, device=device, dtype=dtype))
        nn.init.kaiming_uniform_(self.in_proj_weight, a=5 ** 0.5)
        self.in_proj_bias = None
        self.out_proj = Linear(embed_dim, mult * embed_dim, bias=False, device=device, dtype=dtype)

    def _init_streaming_state(self, batch_size: int) -> _MHAState:
        if self.context is None:
            if not self.weights_per_step:
                raise RuntimeError("Cannot create a streaming KVCache without a context to estimate capacity.")
            capacity = self.weights_per_step
        else:
            capacity = self.context
        dev = self.in_proj_weight.device
        D = self.embed_dim // self.num_heads
        shape = (batch_size, self.num_heads, capacity, D)
        return _MHAState(torch.zeros(shape, device=dev, dtype=torch.float32), torch.zeros(shape, device=dev, dtype=torch.float32),
                         torch.zeros(1, device=dev, dtype=torch.long), 0)

    def _project(self, weight: torch.Tensor, x: torch.Tensor, offset: int, ln=None, **epilogue) -> torch.Tensor:
        if not self.weights_per_step:
            return ops.linear(x, weight, ln=ln, **epilogue)
        if ln is not None:
            x = ops.layernorm(x, *ln)
        # multi_linear (transformer.py:155-179): step t uses weight chunk t + offset
        B, T, _ = x.shape
        w = weight.view(self.weights_per_step, -1, weight.shape[1])
        outs = []
        for t in range(T):
            ep = {k: (v[:, t].contiguous() if k == "res" and v is not None else v) for k, v in epilogue.items()}
            outs.append(ops.linear(x[:, t].contiguous(), w[t + offset], **ep))
        return torch.stack(outs, 1)

    def forward(self, query: torch.Tensor, key: Optional[torch.Tensor] = None, value: Optional[torch.Tensor] = None, *,
                res: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None, ln=None) -> torch.Tensor:
        """query ``[B, T, C]`` (self-attention: key / value are ignored, as in the reference).  With ``res`` the result
        is ``res + scale * out_proj(attn)`` computed in the out-projection's epilogue; ``ln = (gamma, beta, eps)``: the layer's
        ``norm1`` applied to ``query`` inside the in-projection launch."""
        state = self._streaming_state
        x = query.contiguous()
        B, T, _ = x.shape
        H = self.num_heads
        offset = state.offset_cpu if state is not None else 0
        qkv = self._project(self.in_proj_weight, x, offset, ln=ln)
        use_rope = self.rope is not None
        period = self.rope.max_period if use_rope else 10000.0
        if state is None and ops.ATTENTION_FUSED_QKV and qkv.is_cuda and (qkv.shape[2] // (3 * H)) in (32, 64, 128):
            # whole-utterance pass: q / k / v read in place from the in-projection's output, rotated on load (no split launch)
            a = ops.attention_qkv(qkv, H, rope=use_rope, max_period=period, context=self.context)
        elif state is None:
            q, k, v = ops.rope_split(qkv, H, pos0=0, rope=use_rope, max_period=period)
            a = ops.attention(q, k, v, pos0=0, ring=False, context=self.context)
        else:
            # position from the device-side counter (graph-replay safe), mirrored on the host in offset_cpu
            pos_dev = state.shared if state.shared is not None else state.offset
            if ops.attention_step_supported(qkv, H, state.k_cache.shape[2]):
                # a few new positions: split, rotation, ring append and the queries against the ring in one launch
                # (on the few-row route the result is written as the out-projection's packed operand)
                E = self.embed_dim
                packed = (not self.weights_per_step and ops.ATTENTION_STEP_PACKED and B * T > 4 and ops._few_rows(B * T, self.out_proj.weight.shape[0], E)
                          and E % 8 == 0)
                a = ops.attention_step(qkv, H, state.k_cache, state.v_cache, pos_dev, context=self.context, rope=use_rope, max_period=period,
                                       out_packed=packed)
            else:
                q, k, v = ops.rope_split(qkv, H, k=state.k_cache, v=state.v_cache, pos0=offset, pos_dev=pos_dev, ring=True,
                                         rope=use_rope, max_period=period)
                a = ops.attention(q, k, v, pos0=offset, pos_dev=pos_dev, ring=True, context=self.context)
        out = self._project(self.out_proj.weight, a, offset, res=res, scale=scale)
        if state is not None:
            if state.shared is None:
                state.offset.add_(T)
            state.offset_cpu += T
        return out


@dataclass
class _LayerState:
    offset_cpu: int

    def reset(self) -> None:
        self.offset_cpu = 0


class StreamingTransformerLayer(StreamingModule[_LayerState]):
    """``modules/transformer.py:434-592`` (gating='none' branch: plain GELU FFN without biases)."""

    def __init__(self, d_model: int, num_heads: int, dim_feedforward: int | List[int] = 2048, causal: bool = False,
                 context: Optional[int] = None, rope: Optional[RotaryEmbedding] = None, norm: str = "layer_norm",
                 layer_scale: Optional[float] = None, gating: str = "none", weights_per_step: int = 0, activation=None,
                 skip_self_attn: bool = False, device=None, dtype=None):
        super().__init__()
        fk = {"device": device, "dtype": dtype}
        if gating != "none" or weights_per_step:
            raise NotImplementedError("gated / per-step FFNs belong to the LM depth transformer (rstnet_amd.lm)")
        if not skip_self_attn:
            self.self_attn = StreamingMultiheadAttention(embed_dim=d_model, num_heads=num_heads, causal=causal, context=context,
                                                         rope=rope, weights_per_step=weights_per_step, **fk)
            self.norm1 = create_norm_fn(norm, d_model, **fk)
        self.norm2 = create_norm_fn(norm, d_model, **fk)
        self.weights_per_step = weights_per_step
        self.gating = None
        self.skip_self_attn = skip_self_attn
        assert isinstance(dim_feedforward, int)
        self.linear1 = Linear(d_model, dim_feedforward, bias=False, **fk)
        self.linear2 = Linear(dim_feedforward, d_model, bias=False, **fk)
        if layer_scale is None:
            self.layer_scale_1: nn.Module = nn.Identity()
            self.layer_scale_2: nn.Module = nn.Identity()
        else:
            self.layer_scale_1 = LayerScale(d_model, layer_scale, **fk)
            self.layer_scale_2 = LayerScale(d_model, layer_scale, **fk)

    def _init_streaming_state(self, batch_size: int) -> _LayerState:
        return _LayerState(offset_cpu=0)

    @staticmethod
    def _scale(ls: nn.Module) -> Optional[torch.Tensor]:
        return ls.scale if isinstance(ls, LayerScale) else None

    @staticmethod
    def _ln(norm: nn.Module):
        if not isinstance(norm, LayerNorm):
            raise NotImplementedError(f"norm {type(norm).__name__}")
        return (norm.weight, norm.bias, norm.eps)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.contiguous()
        # the two LayerNorms run as prologues of the in-projection / linear1 launches where that form exists
        if not self.skip_self_attn:
            x = self.self_attn(x, res=x, scale=self._scale(self.layer_scale_1), ln=self._ln(self.norm1))
        # (few-row streaming steps: linear1 hands its GELU'd result to linear2 already in linear2's operand order)
        chain = ops.linear_chains(x, self.linear1.weight, self.linear2.weight)
        h = self.linear1(x, act_out=ops.ACT_GELU, ln=self._ln(self.norm2), **({"out_packed": True} if chain else {}))
        x = self.linear2(h, res=x, scale=self._scale(self.layer_scale_2))
        state = self._streaming_state
        if state:
            state.offset_cpu += x.shape[1]
        return x


@dataclass
class _TransformerState:
    offset: torch.Tensor

    def reset(self) -> None:
        self.offset.zero_()


class StreamingTransformer(StreamingModule[_TransformerState]):
    """``modules/transformer.py:595-690``; ``[B, T, C]`` in and out."""

    def __init__(self, d_model: int, num_heads: int, num_layers: int, dim_feedforward: int | List[int] = 2048,
                 causal: bool = False, context: Optional[int] = None, positional_embedding: str = "sin",
                 max_period: float = 10_000, positional_scale: float = 1.0, betas: Optional[Tuple[float, float]] = None,
                 layer_class=StreamingTransformerLayer, device=None, dtype=None, **kwargs):
        super().__init__()
        assert d_model % num_heads == 0
        if positional_embedding not in ("rope", "none"):
            raise NotImplementedError(f"positional_embedding={positional_embedding!r}: the hot path uses 'rope' or 'none'")
        self.positional_embedding = positional_embedding
        self.max_period, self.positional_scale, self.betas = max_period, positional_scale, betas
        self.rope: Optional[RotaryEmbedding] = RotaryEmbedding(max_period=max_period) if positional_embedding == "rope" else None
        self.layers = nn.ModuleList([
            layer_class(d_model=d_model, num_heads=num_heads, dim_feedforward=dim_feedforward, causal=causal, context=context,
                        rope=self.rope, device=device, dtype=dtype, **kwargs) for _ in range(num_layers)])

    def _init_streaming_state(self, batch_size: int) -> _TransformerState:
        device = next(self.parameters()).device
        return _TransformerState(offset=torch.zeros(1, device=device, dtype=torch.long))

    def _frame_layers(self, x: torch.Tensor):
        """The per-layer tensors of ``ops.codec_transformer_frame`` if this streaming step can run as ONE persistent launch (a few
        rows, plain layers with attention + GELU FFN + LayerNorm, all of them streaming), else None."""
        state = self._streaming_state
        if state is None or not x.is_cuda or x.dtype != torch.float32 or not self.layers:
            return None
        out = []
        for layer in self.layers:
            att = getattr(layer, "self_attn", None)
            if (type(layer) is not StreamingTransformerLayer or att is None or att._streaming_state is None or att.weights_per_step
                    or not isinstance(layer.norm1, LayerNorm) or not isinstance(layer.norm2, LayerNorm) or layer.norm1.eps != layer.norm2.eps):
                return None
            ms = att._streaming_state
            out.append({"in_proj": att.in_proj_weight, "out_proj": att.out_proj.weight, "linear1": layer.linear1.weight,
                        "linear2": layer.linear2.weight, "norm1_w": layer.norm1.weight, "norm1_b": layer.norm1.bias,
                        "norm2_w": layer.norm2.weight, "norm2_b": layer.norm2.bias, "scale1": layer._scale(layer.layer_scale_1),
                        "scale2": layer._scale(layer.layer_scale_2), "k_cache": ms.k_cache, "v_cache": ms.v_cache})
        l0 = self.layers[0]
        att = l0.self_attn
        B, T, E = x.shape
        if any(t.dtype != torch.float32 for ly in out for t in ly.values() if t is not None):
            return None
        if (out[0]["scale1"] is None) != (out[0]["scale2"] is None) or any((ly["scale1"] is None) != (out[0]["scale1"] is None) for ly in out):
            return None
        if not ops.codec_transformer_frame_supported(B, T, E, att.num_heads, l0.linear1.out_features, len(out), out[0]["k_cache"].shape[2],
                                                     device=x.device):
            return None
        return out

    def forward(self, x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        T = x.shape[1]
        state = self._streaming_state
        frame = self._frame_layers(x) if not args and not kwargs else None
        if frame is not None:
            # one 80 ms frame of a few streams: all layers in ONE persistent launch (csrc/codec_tr.hip)
            att = self.layers[0].self_attn
            y = ops.codec_transformer_frame(x.contiguous(), frame, state.offset, H=att.num_heads, context=att.context,
                                            rope=att.rope is not None, max_period=att.rope.max_period if att.rope is not None else 10000.0,
                                            eps=self.layers[0].norm1.eps)
            for layer in self.layers:
                ms = layer.self_attn._streaming_state
                ms.shared = state.offset
                ms.offset_cpu += T
                if layer._streaming_state is not None:
                    layer._streaming_state.offset_cpu += T
            state.offset.add_(T)
            return y
        for layer in self.layers:
            att = getattr(layer, "self_attn", None)
            if att is not None and att._streaming_state is not None:
                # all layers read this transformer's position counter (same value during a step)
                att._streaming_state.shared = state.offset if state is not None else None
            x = layer(x, *args, **kwargs)
        if state is not None:
            state.offset.add_(T)
        return x


class ProjectedTransformer(StreamingContainer):
    """``modules/transformer.py:693-750``: optional in/out projections; ``conv_layout`` = ``[B, C, T]`` tensors."""

    def __init__(self, input_dimension: int, output_dimensions: Tuple[int, ...], d_model: int, *, conv_layout: bool = False,
                 **kwargs):
        super().__init__()
        self.transformer = StreamingTransformer(d_model=d_model, **kwargs)
        self.input_dimension, self.output_dimensions, self.conv_layout = input_dimension, output_dimensions, conv_layout
        self.input_proj = Linear(input_dimension, d_model, bias=False) if d_model != input_dimension else None
        self.output_projs = nn.ModuleList([nn.Identity() if d_model == od else Linear(d_model, od, bias=False)
                                           for od in output_dimensions])

    def forward_nlc(self, x: torch.Tensor) -> List[torch.Tensor]:
        if self.input_proj is not None:
            x = self.input_proj(x)
        z = self.transformer(x)
        return [z if isinstance(p, nn.Identity) else p(z) for p in self.output_projs]

    def forward(self, x: torch.Tensor, *args, **kwargs) -> List[torch.Tensor]:
        if self.conv_layout:
            x = _to_nlc(x)
        ys = self.forward_nlc(x)
        return [_to_ncl(y) for y in ys] if self.conv_layout else ys
