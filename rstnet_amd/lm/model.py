"""``LMModel`` / ``LMGen`` -- drop-in for the streaming generation half of ``MLLM_v2/models/model.py`` (Moshi-style
RQ-Transformer: a temporal transformer over 17 token streams + a per-codebook depth transformer).

Same constructor keywords, ``state_dict`` keys (``emb.{i}.weight``, ``transformer.layers.{l}.self_attn.in_proj_weight``,
``...gating.linear_in.weight``, ``depformer.layers.{l}.gating.{k}.linear_out.weight``, ``linears.{k}.weight`` ...),
``forward_text`` / ``forward_depformer`` signatures and ``LMGen.step`` semantics (delayed token ring cache, ``None`` for
the first ``max_delay`` steps).  Training ``forward`` is out of scope.

Execution: weights bf16 in HBM, activations fp32, one decode step (T = 1) per call through the kernels of
``csrc/lm_*.hip``; a whole ``LMGen`` frame (token-ring update, ``forward_text``, the depth steps with their samplers, ring
commit) is captured into ONE HIP graph after a warm-up -- the reference wraps ``forward_text`` and ``depformer_step`` in two
``CUDAGraphed`` wrappers (``MLLM_v2/utils/compile.py:189-277``) and does the ring arithmetic on the host in between -- and the
environment flag ``NO_CUDA_GRAPH`` disables that (``compile.py:168-174``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
from torch import nn

from .. import ops
from ..codec.conv import _PackedCache
from ..codec.streaming import StreamingContainer, StreamingModule
from ..graphs import Graphed as _Graphed


def _gating_hidden(dim: int, dim_feedforward: int) -> int:
    """modules/gating.py:40-45."""
    return (21 * dim) // 8 if dim_feedforward == 4 * dim else (2 * dim_feedforward) // 3


class _Weight(nn.Module):
    """Holder of a ``weight`` parameter (nn.Linear / nn.Embedding key layout); the arithmetic lives in the kernels."""

    def __init__(self, *shape: int, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape, device=device, dtype=dtype), requires_grad=False)
        nn.init.normal_(self.weight, std=0.02)


class ScaledEmbedding(_Weight):
    """models/model.py:67-91 -- lookup with id -1 -> zeros, executed by rst_embed_sum_bf16."""

    def __init__(self, num_embeddings: int, embedding_dim: int, zero_idx: int = -1, device=None, dtype=None, **_):
        super().__init__(num_embeddings, embedding_dim, device=device, dtype=dtype)
        self.zero_idx = zero_idx

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        """ids int64 ``[...]`` -> fp32 ``[..., D]`` (callers that look a table up directly, e.g. infer_no_streaming.py:258)."""
        flat = input.reshape(-1, 1).contiguous()
        return ops.embed_sum(flat, [self.weight], [0]).view(*input.shape, self.weight.shape[1])


class RMSNorm(nn.Module):
    """rms_norm_f32 (modules/transformer.py:49-65, eps 1e-8); key ``alpha`` [1,1,D]."""

    def __init__(self, dim: int, eps: float = 1e-8, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.alpha = nn.Parameter(torch.ones(1, 1, dim, device=device, dtype=dtype), requires_grad=False)
        self._f32 = _PackedCache()

    def alpha_f32(self) -> torch.Tensor:
        return self._f32.get((self.alpha,), lambda: self.alpha.detach().float().reshape(-1).contiguous())


class ActivationGating(nn.Module):
    """modules/gating.py:25-51 (SiLU): keys ``linear_in.weight`` [2*hidden, dim], ``linear_out.weight`` [dim, hidden]."""

    def __init__(self, dim: int, dim_feedforward: int, device=None, dtype=None):
        super().__init__()
        hidden = _gating_hidden(dim, dim_feedforward)
        self.linear_in = _Weight(2 * hidden, dim, device=device, dtype=dtype)
        self.linear_out = _Weight(dim, hidden, device=device, dtype=dtype)


class _Attention(nn.Module):
    def __init__(self, dim: int, mult: int, device=None, dtype=None):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(mult * 3 * dim, dim, device=device, dtype=dtype), requires_grad=False)
        nn.init.normal_(self.in_proj_weight, std=0.02)
        self.out_proj = _Weight(mult * dim, dim, device=device, dtype=dtype)


class _Layer(nn.Module):
    def __init__(self, dim: int, dim_feedforward: int, weights_per_step: int, device=None, dtype=None):
        super().__init__()
        fk = {"device": device, "dtype": dtype}
        self.self_attn = _Attention(dim, weights_per_step or 1, **fk)
        self.norm1 = RMSNorm(dim, **fk)
        self.norm2 = RMSNorm(dim, **fk)
        if weights_per_step:
            self.gating = nn.ModuleList([ActivationGating(dim, dim_feedforward, **fk) for _ in range(weights_per_step)])
        else:
            self.gating = ActivationGating(dim, dim_feedforward, **fk)


@dataclass
class _StepState:
    k: List[torch.Tensor]      # per layer [B, H, cap, D] fp32 ring
    v: List[torch.Tensor]
    pos: torch.Tensor          # int64 [1] on device: steps appended so far (= position of the next step)
    scratch: Optional[tuple] = None   # (split workspace, arrival counters) of the long-ring attention kernel
    offset_cpu: int = 0
    tables: object = None             # ops.TemporalFrameTables of the persistent batch-1 launch (built at the first step that takes it)
    tables_key: object = None

    def reset(self) -> None:
        self.pos.zero_()
        self.offset_cpu = 0


class StreamingTransformer(StreamingModule[_StepState]):
    """Decode-step executor with the parameter layout of modules/transformer.py:595-690 (norm rms_norm_f32, SiLU gating,
    causal, rope or no positional embedding, optional per-step weights)."""

    def __init__(self, d_model: int, num_heads: int, num_layers: int, dim_feedforward: int, context: Optional[int],
                 positional_embedding: str, max_period: float = 10000.0, weights_per_step: int = 0, device=None, dtype=None,
                 kv_dtype: torch.dtype = torch.float32):
        super().__init__()
        # precision of the KV rings: bf16 = what the reference caches (RingKVCache in the model's dtype); rings of <= 64 slots
        # (the depth transformer, tiny test models) always stay fp32 -- their bytes do not matter and their kernel is fp32
        assert kv_dtype in (torch.float32, torch.bfloat16)
        self.kv_dtype = kv_dtype
        assert d_model % num_heads == 0
        if positional_embedding not in ("rope", "none"):
            raise NotImplementedError(f"positional_embedding={positional_embedding!r}")
        self.d_model, self.num_heads, self.context = d_model, num_heads, context
        self.rope, self.max_period, self.weights_per_step = positional_embedding == "rope", max_period, weights_per_step
        self.layers = nn.ModuleList([_Layer(d_model, dim_feedforward, weights_per_step, device=device, dtype=dtype)
                                     for _ in range(num_layers)])

    def _apply_named_streaming(self, fn) -> None:
        """As the ROOT of a walk this module is always visited: the reference skips a propagate=False root but still visits
        its layers (modules/streaming.py:67-84, relied on by ``with lm.depformer.streaming(B)``), and the per-layer states of
        the reference live in this module's single state."""
        fn("", self)

    def _init_streaming_state(self, batch_size: int, capacity: Optional[int] = None) -> _StepState:
        if self.context is None and not self.weights_per_step:
            raise RuntimeError("Cannot create a streaming KVCache without a context to estimate capacity.")
        cap = capacity or (self.context if self.context is not None else self.weights_per_step)
        dev = self.layers[0].norm1.alpha.device
        shape = (batch_size, self.num_heads, cap, self.d_model // self.num_heads)
        scratch = None
        if cap > 64:
            splits = ops.lm_attn_splits(cap, batch_size * self.num_heads)
            scratch = (torch.empty(batch_size, self.num_heads, splits, shape[3] + 2, device=dev),
                       torch.zeros(batch_size, self.num_heads, device=dev, dtype=torch.int32))
        kvd = self.kv_dtype if cap > 64 else torch.float32
        return _StepState([torch.zeros(shape, device=dev, dtype=kvd) for _ in self.layers],
                          [torch.zeros(shape, device=dev, dtype=kvd) for _ in self.layers],
                          torch.zeros(1, device=dev, dtype=torch.long), scratch)

    def step(self, x: Optional[torch.Tensor], step_index: Optional[int] = None, pos: Optional[torch.Tensor] = None,
             embed: Optional[tuple] = None) -> torch.Tensor:
        """x fp32 ``[B, d_model]`` -> ``[B, d_model]``: one new time step through every layer.  ``pos`` (int64 device scalar):
        position of this step supplied by a caller that owns the loop (LMGen's depth steps are always positions 0 .. dep_q - 1);
        the module's own counter is then left alone.  ``embed = (add, table, tokens, col)`` instead of ``x``: the input is
        ``add + table[tokens[:, col]]`` (``add`` fp32 ``[B, d_model]``, any row stride) and is formed inside the first launch.

        Launches per layer -- batch <= 2: 4 for a short un-rotated ring (qkv GEMV | out-proj GEMV with the attention as its
        prologue | ffn-in GEMV with the gate | ffn-out GEMV), 5 otherwise (attention on its own); batch > 2: 5."""
        st = self._streaming_state
        if st is None:
            raise RuntimeError("the decode-step transformer only runs in streaming mode")
        E, H = self.d_model, self.num_heads
        B = x.shape[0] if x is not None else embed[0].shape[0]
        cap = st.k[0].shape[2]
        fused_attn = ops.gemv_attn_supported(B, H, E // H, cap, self.rope)
        k_idx = 0
        if self.weights_per_step:
            k_idx = st.offset_cpu if step_index is None else step_index
        pos_t = st.pos if pos is None else pos
        if B == 1 and x is not None and not self.weights_per_step and cap > 64 and ops.temporal_frame_wanted(st.offset_cpu):
            y = self._persistent_step(st, x, pos_t, cap)
            if y is not None:
                if pos is None:
                    st.pos.add_(1)
                    st.offset_cpu += 1
                return y
        # the step's rotation once for all layers (long rings: the attention launches read it instead of evaluating 24 libm calls per lane)
        rope_table = ops.lm_rope_table(pos_t, E // H, max_period=self.max_period) if self.rope and cap > 64 else None
        for l, layer in enumerate(self.layers):
            att = layer.self_attn
            if self.weights_per_step:
                w_in = att.in_proj_weight.view(self.weights_per_step, 3 * E, E)[k_idx]
                w_out = att.out_proj.weight.view(self.weights_per_step, E, E)[k_idx]
                gate = layer.gating[k_idx]
            else:
                w_in, w_out, gate = att.in_proj_weight, att.out_proj.weight, layer.gating
            if l == 0 and embed is not None:
                add, table, tokens, col = embed
                if B <= 2 and E <= 4096 and E % 8 == 0:
                    qkv, x = ops.gemv_embed(add, table, tokens, col, w_in, alpha=layer.norm1.alpha_f32(), eps=layer.norm1.eps)
                else:
                    x = ops.embed_sum(tokens, [table], [col], add=add)        # (a column block of h_all is read in place: no copy launch)
            if l > 0 or embed is None or not (B <= 2 and E <= 4096 and E % 8 == 0):
                qkv = ops.lm_linear(x, w_in, prologue=ops.PROLOGUE_RMSNORM, alpha=layer.norm1.alpha_f32(), eps=layer.norm1.eps)
            if fused_attn:
                x = ops.gemv_attn(qkv, st.k[l], st.v[l], pos_t, w_out, context=self.context, res=x)
            else:
                a = ops.lm_attn_decode(qkv, st.k[l], st.v[l], pos_t, rope=self.rope, context=self.context,
                                       max_period=self.max_period, scratch=st.scratch, packed=B > 2, rope_table=rope_table)
                x = ops.lm_linear(a, w_out, res=x)
            x = ops.lm_gated_pair(x, gate.linear_in.weight, gate.linear_out.weight, alpha=layer.norm2.alpha_f32(), eps=layer.norm2.eps,
                                  res=x)
        if pos is None:
            st.pos.add_(1)
            st.offset_cpu += 1
        return x


    def _persistent_step(self, st: _StepState, x: torch.Tensor, pos_t: torch.Tensor, cap: int) -> Optional[torch.Tensor]:
        """All layers of a batch-1 step as ONE persistent launch (csrc/lm_temporal.hip) when the library serves the shape and the
        device's persistent launches are healthy; None -> the launch-per-op chain below."""
        E, H = self.d_model, self.num_heads
        Hd = self.layers[0].gating.linear_out.weight.shape[1]
        w0 = self.layers[0].self_attn.in_proj_weight
        if w0.dtype != torch.bfloat16 or not ops.temporal_frame_supported(1, E, H, Hd, len(self.layers), cap, st.k[0].dtype == torch.bfloat16, x.device):
            return None
        key = (ops.persistent_epoch(x.device),) + tuple((ly.self_attn.in_proj_weight.data_ptr(), ly.self_attn.in_proj_weight._version,
                                                          ly.gating.linear_in.weight.data_ptr()) for ly in self.layers)
        if st.tables is None or st.tables_key != key:
            st.tables = ops.TemporalFrameTables(
                [dict(in_proj=ly.self_attn.in_proj_weight, out_proj=ly.self_attn.out_proj.weight, gate_in=ly.gating.linear_in.weight,
                      gate_out=ly.gating.linear_out.weight, norm1=ly.norm1.alpha_f32(), norm2=ly.norm2.alpha_f32(), k_cache=st.k[l],
                      v_cache=st.v[l]) for l, ly in enumerate(self.layers)],
                H=H, context=self.context, eps=self.layers[0].norm1.eps)
            st.tables_key = key
        rope_table = ops.lm_rope_table(pos_t, E // H, max_period=self.max_period) if self.rope else None
        return ops.temporal_decode_frame(st.tables, x, pos_t, rope_table)


class ModelConfig:
    def __init__(self, model_type):
        self.model_type = model_type


class LMModel(StreamingContainer):
    """models/model.py:98-225 (constructor keywords identical; only the inference-relevant ones are interpreted)."""

    def __init__(self, delays: List[int] = [0], n_q: int = 8, dep_q: int = 8, card: int = 1024, text_card: int = 32000,
                 dim: int = 128, num_heads: int = 8, hidden_scale: float = 4, norm: str = "layer_norm", norm_emb: bool = False,
                 bias_proj: bool = False, depformer_dim: int = 256, depformer_dim_feedforward=None,
                 depformer_multi_linear: bool = False, depformer_weights_per_step: bool = False, depformer_pos_emb: str = "sin",
                 existing_text_padding_id: Optional[int] = None, context: Optional[int] = None, device=None,
                 dtype=torch.bfloat16, kv_dtype: torch.dtype = torch.bfloat16, **kwargs):
        super().__init__()
        self.kv_dtype = kv_dtype      # temporal KV rings: bf16 like the reference's cache (fp32 on request, e.g. fp32 parity runs)
        if norm != "rms_norm_f32" or norm_emb or bias_proj:
            raise NotImplementedError("the decode path implements norm='rms_norm_f32', norm_emb=False, bias_proj=False")
        if kwargs.get("gating", "silu") != "silu" or kwargs.get("depformer_gating", "silu") != "silu":
            raise NotImplementedError("SiLU gating only")
        if not (depformer_multi_linear and depformer_weights_per_step and depformer_pos_emb == "none"):
            raise NotImplementedError("depth transformer: multi_linear + weights_per_step + pos_emb 'none' (the Moshi / RSTnet setup)")
        if kwargs.get("layer_scale") is not None or kwargs.get("depformer_layer_scale") is not None:
            raise NotImplementedError("layer_scale")
        self.n_q, self.dep_q, self.card, self.text_card = n_q, dep_q, card, text_card
        assert len(delays) == self.num_codebooks, "unexpected number of delays"
        self.delays, self.dim, self.context = list(delays), dim, context
        self.existing_text_padding_id = existing_text_padding_id
        fk = {"device": device, "dtype": dtype}
        self.emb = nn.ModuleList([ScaledEmbedding(card + 1, dim, **fk) for _ in range(n_q)])
        self.text_emb = ScaledEmbedding(text_card + 1, dim, **fk)
        self.text_linear = _Weight(text_card + (1 if existing_text_padding_id is None else 0), dim, **fk)
        self.transformer = StreamingTransformer(dim, num_heads, kwargs["num_layers"], int(hidden_scale * dim), context,
                                                kwargs.get("positional_embedding", "sin"), kwargs.get("max_period", 10000.0),
                                                kv_dtype=kv_dtype, **fk)
        self.out_norm = RMSNorm(dim, **fk)
        self.depformer_multi_linear = depformer_multi_linear
        self.depformer_in = nn.ModuleList([_Weight(depformer_dim, dim, **fk) for _ in range(dep_q)])
        self.depformer_emb = nn.ModuleList([ScaledEmbedding(card + 1, depformer_dim, **fk) for _ in range(dep_q - 1)])
        self.depformer_text_emb = ScaledEmbedding(text_card + 1, depformer_dim, **fk)
        if depformer_dim_feedforward is None:
            depformer_dim_feedforward = int(hidden_scale * depformer_dim)
        self.depformer = StreamingTransformer(depformer_dim, kwargs.get("depformer_num_heads", num_heads),
                                              kwargs.get("depformer_num_layers", kwargs["num_layers"]), depformer_dim_feedforward,
                                              None, "none", kwargs.get("depformer_max_period", 10000.0), weights_per_step=dep_q, **fk)
        self.depformer.set_streaming_propagate(False)
        self.linears = nn.ModuleList([_Weight(card, depformer_dim, **fk) for _ in range(dep_q)])
        self.config = ModelConfig(model_type="lora")
        self._in_cat = _PackedCache()
        self._depth_tables = _PackedCache()

    def depth_frame_tables(self):
        """Pointer tables of the persistent depth-frame launch (``lm.depth_frame.DepthFrameTables``), rebuilt when a weight changes."""
        from .depth_frame import DepthFrameTables
        dep = self.depformer
        params = [p for l in dep.layers for p in (l.self_attn.in_proj_weight, l.self_attn.out_proj.weight, l.norm1.alpha, l.norm2.alpha)]
        params += [g.linear_in.weight for l in dep.layers for g in l.gating] + [g.linear_out.weight for l in dep.layers for g in l.gating]
        params += [m.weight for m in self.linears] + [self.depformer_text_emb.weight] + [m.weight for m in self.depformer_emb]
        return self._depth_tables.get(tuple(params), lambda: DepthFrameTables(
            dep, [m.weight for m in self.linears], [None] * self.dep_q,
            [self.depformer_text_emb.weight] + [m.weight for m in self.depformer_emb]))

    def depformer_in_all(self) -> torch.Tensor:
        """``[dep_q * depformer_dim, dim]``: the dep_q ``depformer_in[k]`` matrices stacked (a second copy, built once per weight
        version): all of a frame's ``depformer_in[k](transformer_out)`` products are ONE weight-streaming launch."""
        ws = [m.weight for m in self.depformer_in]
        return self._in_cat.get(tuple(ws), lambda: torch.cat([w.detach() for w in ws], 0).contiguous())

    # ---- token-id conventions (models/model.py:226-277)
    @property
    def initial_token_id(self) -> int:
        return self.card

    @property
    def text_initial_token_id(self) -> int:
        return self.text_card

    @property
    def text_padding_token_id(self) -> int:
        return self.text_card if self.existing_text_padding_id is None else self.existing_text_padding_id

    @property
    def end_of_text_padding_id(self) -> int:
        return 0

    @property
    def zero_token_id(self) -> int:
        return -1

    @property
    def ungenerated_token_id(self) -> int:
        return -2

    @property
    def device(self):
        return next(iter(self.parameters())).device

    @property
    def num_codebooks(self) -> int:
        return self.n_q + 1

    @property
    def num_audio_codebooks(self) -> int:
        return self.n_q

    @property
    def audio_offset(self) -> int:
        return 1

    def _get_initial_token(self) -> torch.Tensor:
        tok = torch.full([1, self.num_codebooks, 1], self.initial_token_id, device=self.device, dtype=torch.long)
        tok[:, 0] = self.text_initial_token_id
        return tok

    def forward(self, sequence: torch.Tensor, masks: Optional[torch.Tensor] = None):
        raise NotImplementedError("teacher-forced training forward is out of scope; use forward_text / forward_depformer")

    # ---- decode step
    def forward_text(self, sequence: torch.Tensor, masks: Optional[torch.Tensor] = None):
        """sequence int64 ``[B, n_q+1, 1]`` -> (transformer_out fp32 ``[B,1,dim]``, text_logits fp32 ``[B,1,1,V]``)."""
        B, K, S = sequence.shape
        assert K == self.num_codebooks, f"Sequence shape {sequence.shape} must match the number of codebooks."
        assert S == 1, "the streaming decode path takes one step at a time"
        toks = sequence.reshape(B, K).contiguous()
        tables = [e.weight for e in self.emb] + [self.text_emb.weight]
        x = ops.embed_sum(toks, tables, list(range(1, K)) + [0])     # ((e_0 + e_1) + ...) + text, as the reference
        x = self.transformer.step(x)
        out = ops.rmsnorm(x, self.out_norm.alpha_f32(), self.out_norm.eps)
        logits = ops.lm_linear(out, self.text_linear.weight)
        return out.view(B, 1, self.dim), logits.view(B, 1, 1, -1)

    def forward_depformer(self, depformer_cb_index: int, sequence: torch.Tensor, transformer_out: torch.Tensor) -> torch.Tensor:
        """sequence int64 ``[B,1,1]`` (previous token), transformer_out fp32 ``[B,1,dim]`` -> logits fp32 ``[B,1,1,card]``."""
        B, K, S = sequence.shape
        assert K == 1, f"Codebooks for Depformer streaming should be passed 1 by 1, got {K}."
        assert S == 1, f"Steps for Depformer streaming should be passed 1 by 1, got {S}."
        assert transformer_out.shape[1] == 1, "Transformer out should be a for a single step."
        logits = self._depformer_logits(depformer_cb_index, sequence.reshape(B, 1).contiguous(), 0,
                                        transformer_out.reshape(B, self.dim).contiguous())
        return logits.view(B, 1, 1, -1)

    def _depformer_logits(self, k: int, tokens: torch.Tensor, col: int, h_t: Optional[torch.Tensor], pos: Optional[torch.Tensor] = None,
                          step_index: Optional[int] = None, h_all: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Depth step ``k``: previous token = ``tokens[:, col]`` (int64 ``[B, n]``), ``h_t`` fp32 ``[B, dim]`` -> logits ``[B, card]``.
        ``h_all`` (``[B, dep_q * depformer_dim]``, the stacked ``depformer_in`` products of the frame) replaces ``h_t``."""
        E = self.depformer.d_model
        h = h_all[:, k * E:(k + 1) * E] if h_all is not None else ops.lm_linear(h_t, self.depformer_in[k].weight)
        table = self.depformer_text_emb.weight if k == 0 else self.depformer_emb[k - 1].weight
        y = self.depformer.step(None, step_index=step_index, pos=pos, embed=(h, table, tokens, col))
        return ops.lm_linear(y, self.linears[k].weight)

    @classmethod
    def from_state_dict(cls, sd: Dict[str, torch.Tensor], cfg: dict, kv_dtype: torch.dtype = torch.bfloat16) -> "LMModel":
        """Model for ``cfg`` (keys of ``rstnet_amd.synth.LM_*``) with weights taken from ``sd`` WITHOUT copying them
        (a 7.7 B-parameter state dict stays a single 15 GB allocation).  ``kv_dtype``: precision of the temporal KV rings."""
        model = cls(kv_dtype=kv_dtype, causal=True, layer_scale=None, gating="silu", norm="rms_norm_f32", positional_embedding="rope",
                    depformer_causal=True, depformer_layer_scale=None, depformer_multi_linear=True, depformer_context=8,
                    depformer_gating="silu", depformer_pos_emb="none", depformer_weights_per_step=True, device="meta", **cfg)
        params = dict(model.named_parameters())
        missing = [k for k in params if k not in sd]
        unexpected = [k for k in sd if k not in params]
        if missing or unexpected:
            raise RuntimeError(f"state_dict mismatch: missing={missing[:5]} unexpected={unexpected[:5]}")
        for name, tensor in sd.items():
            mod = model
            *path, leaf = name.split(".")
            for part in path:
                mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]
            assert tuple(getattr(mod, leaf).shape) == tuple(tensor.shape), name
            setattr(mod, leaf, nn.Parameter(tensor, requires_grad=False))
        return model.eval()


@dataclass
class _LMGenState:
    cache: torch.Tensor            # int64 [B, K, max_delay + 2] token ring
    initial: torch.Tensor          # int64 [1, K, 1]
    offset_dev: torch.Tensor       # int64 [1]: the frame counter as the ring kernels see it
    graphed_frame: _Graphed
    depth: Optional[_StepState] = None     # the depth transformer's KV rings of THIS session (a captured frame points at them)
    offset: int = 0
    persist_epoch: int = 0                 # ops.persistent_epoch(device) when the frame graph was captured
    tables: object = None                  # the DepthFrameTables the captured frame points at (kept alive with the graph)
    temporal_base: Optional[int] = None    # host-side position of the temporal rings at frame 0 of this session
    temporal_choice: Optional[bool] = None  # whether the captured frame takes the persistent temporal launch

    def reset(self) -> None:
        self.offset = 0
        self.offset_dev.zero_()
        self.temporal_base = None       # (the temporal rings were reset with the session: re-read their position at the next frame)


class LMGen(StreamingModule[_LMGenState]):
    """models/model.py:443-597.  The token ring, the delay pattern and the frame counter live on the device
    (csrc/lm_ring.hip), so one frame -- ring update, temporal step, text sample, ``dep_q`` depth steps with their samples,
    ring commit + delayed gather -- is ONE captured graph fed by a single copy of the user tokens."""

    def __init__(self, lm_model: LMModel, use_sampling: bool = True, temp: float = 0.8, temp_text: float = 0.7,
                 top_k: int = 250, top_k_text: int = 25, check: bool = False):
        assert not lm_model.training, "generation shouldn't be used in training mode."
        super().__init__()
        self.lm_model = lm_model
        self.use_sampling, self.temp, self.temp_text = use_sampling, temp, temp_text
        self.top_k, self.top_k_text, self.check = top_k, top_k_text, check
        self.max_delay = max(lm_model.delays)
        self.delays_cuda = torch.tensor(lm_model.delays, device=lm_model.device, dtype=torch.long)
        self._delays_i32 = self.delays_cuda.to(torch.int32)
        self._depth_pos = torch.arange(lm_model.dep_q, device=lm_model.device, dtype=torch.long)

    def _init_streaming_state(self, batch_size: int) -> _LMGenState:
        lm = self.lm_model
        cache = torch.full((batch_size, lm.num_codebooks, self.max_delay + 2), lm.ungenerated_token_id, device=lm.device,
                           dtype=torch.long)
        disable = lm.device.type != "cuda"
        return _LMGenState(cache, lm._get_initial_token(), torch.zeros(1, device=lm.device, dtype=torch.long),
                           _Graphed(self._frame, disable=disable), depth=lm.depformer._init_streaming_state(batch_size),
                           persist_epoch=ops.persistent_epoch(lm.device))

    def _noise(self, B: int, k: int) -> Optional[torch.Tensor]:
        if not self.use_sampling:
            return None
        return torch.empty(B, k, device=self.lm_model.device, dtype=torch.float32).exponential_(1)   # utils/sampling.py:44

    def _frame(self, user_tokens: torch.Tensor):
        """user_tokens int64 ``[B, Ki]`` -> (delay-aligned output ``[B, dep_q + 1]``, model input ``[B, K]``); everything in
        between stays on the device (this is the function that is captured)."""
        state, lm = self._streaming_state, self.lm_model
        B = user_tokens.shape[0]
        input_ = ops.lm_ring_begin(state.cache, user_tokens, state.initial.reshape(-1), self._delays_i32, state.offset_dev,
                                   lm.dep_q + 1)
        # one draw of Exp(1) noise per frame for the text sampler and the dep_q audio samplers (utils/sampling.py:44-46)
        noise = self._noise(B, self.top_k_text + lm.dep_q * self.top_k)
        transformer_out, text_logits = lm.forward_text(input_.view(B, -1, 1))
        tokens = torch.empty(B, lm.dep_q + 1, device=input_.device, dtype=torch.long)
        ops.lm_sample(text_logits.view(B, -1), use_sampling=self.use_sampling, temp=self.temp_text, top_k=self.top_k_text,
                      noise=None if noise is None else noise[:, :self.top_k_text], out=tokens[:, 0])
        self._depth(tokens, transformer_out.view(B, lm.dim).contiguous(), None if noise is None else noise[:, self.top_k_text:])
        out = ops.lm_ring_commit(state.cache, tokens, self._delays_i32, state.offset_dev, self.max_delay)
        return out, input_

    def _depth(self, tokens: torch.Tensor, h_t: torch.Tensor, noise: Optional[torch.Tensor]) -> None:
        """The ``dep_q`` sequential depth-transformer steps (models/model.py:564-597): ``tokens[:, 0]`` is the text token,
        step ``cb`` embeds ``tokens[:, cb]`` and samples ``tokens[:, cb + 1]`` in place.  The depth KV rings are persistent
        buffers; the steps are positions 0 .. dep_q - 1 of a ring that restarts every frame (= the reference's fresh
        ``with depformer.streaming(B)`` context), supplied as constant device scalars instead of a counter to reset and bump."""
        B = tokens.shape[0]
        lm = self.lm_model
        dep = lm.depformer
        # the rings belong to the session (so that a frame graph captured by another live session keeps valid pointers and
        # exiting `streaming()` releases them); a bare `depformer_step` call outside any session gets throw-away rings
        # depformer_in[k](transformer_out) for all dep_q steps at once: one 8 x larger launch instead of eight
        h_all = ops.lm_linear(h_t, lm.depformer_in_all())
        E, H = dep.d_model, dep.num_heads
        Hd = dep.layers[0].gating[0].linear_out.weight.shape[1]
        if ops.depth_frame_enabled(h_t.device) and h_t.is_cuda and ops.depth_frame_supported(B, E, H, Hd, lm.card, lm.dep_q, len(dep.layers), self.top_k, device=h_t.device):
            # batch 1 / 2: the whole phase (dep_q x (L layers + head + sampler)) is ONE persistent launch whose ops hand their
            # vectors over in-kernel; the depth KV ring lives in its LDS
            tables = lm.depth_frame_tables()
            if self._streaming_state is not None:
                self._streaming_state.tables = tables       # a captured frame embeds the tables' device pointers: they live as long as it does
            ops.depth_decode_frame(tables, h_all, tokens, noise, use_sampling=self.use_sampling, temp=self.temp,
                                   top_k=self.top_k, eps=dep.layers[0].norm1.eps, context=dep.context)
            return
        state = self._streaming_state
        rings = state.depth if state is not None and state.depth is not None and state.depth.k[0].shape[0] == B \
            else dep._init_streaming_state(B)
        saved, dep._streaming_state = dep._streaming_state, rings
        try:
            for cb in range(lm.dep_q):
                logits = lm._depformer_logits(cb, tokens, cb, None, pos=self._depth_pos[cb:cb + 1], step_index=cb, h_all=h_all)
                ops.lm_sample(logits, use_sampling=self.use_sampling, temp=self.temp, top_k=self.top_k,
                              noise=None if noise is None else noise[:, cb * self.top_k:(cb + 1) * self.top_k], out=tokens[:, cb + 1])
        finally:
            dep._streaming_state = saved

    @torch.no_grad()
    def step(self, input_tokens: torch.Tensor) -> Optional[torch.Tensor]:
        state = self._streaming_state
        if state is None:
            raise RuntimeError("You should wrap those calls with a `with lm_gen.streaming(): ...`.")
        lm = self.lm_model
        assert input_tokens.dim() == 3, "Shape should be [B, K, T]."
        B, Ki, S = input_tokens.shape
        assert S == 1, "Only support being given steps one by one."
        needed = lm.num_codebooks - lm.dep_q - 1
        assert Ki == needed, f"We expect {needed} tokens from the user stream, got {Ki}."
        if state.offset % 64 == 0 and lm.device.type == "cuda":
            # health of the persistent depth launch (no synchronisation): a device that had to repair frames moves to the
            # launch-per-op chain, which needs a fresh capture
            ops.persistent_poll(lm.device)
            if state.persist_epoch != ops.persistent_epoch(lm.device):
                state.persist_epoch = ops.persistent_epoch(lm.device)
                state.graphed_frame = _Graphed(self._frame)
        tst = lm.transformer._streaming_state
        if tst is not None:
            # the temporal rings' fill as the host knows it (graph replays do not run the Python that counts steps): the persistent
            # temporal launch is chosen by it (ops.temporal_frame_wanted), and a frame captured under the other choice is re-captured
            if state.temporal_base is None:
                state.temporal_base = tst.offset_cpu - state.offset
            tst.offset_cpu = state.temporal_base + state.offset
            want = B == 1 and ops.temporal_frame_wanted(tst.offset_cpu)
            if state.temporal_choice is None:
                state.temporal_choice = want
            elif want != state.temporal_choice:
                state.temporal_choice = want
                state.graphed_frame = _Graphed(self._frame)
        out, input_ = state.graphed_frame(input_tokens.reshape(B, Ki).contiguous())
        if tst is not None:
            tst.offset_cpu = state.temporal_base + state.offset + 1
        if self.check:
            if lm._depth_tables._val is not None:
                lm._depth_tables._val.check()       # a timed-out hand-off of the persistent depth launch
            assert not (input_ == lm.ungenerated_token_id).any(), (state.offset, input_)
            assert (input_[:, lm.audio_offset:] <= lm.card).all(), input_
            assert (input_[:, :1] <= lm.text_card).all()
        state.offset += 1
        if state.offset <= self.max_delay:
            return None
        return out.view(B, lm.dep_q + 1, 1).clone()

    def depformer_step(self, text_token: torch.Tensor, transformer_out: torch.Tensor) -> torch.Tensor:
        """text_token int64 ``[B]``, transformer_out fp32 ``[B, 1, dim]`` -> the frame's ``dep_q`` audio tokens ``[B, dep_q]``
        (models/model.py:564-597)."""
        (B,) = text_token.shape
        lm = self.lm_model
        tokens = torch.empty(B, lm.dep_q + 1, device=text_token.device, dtype=torch.long)
        tokens[:, 0] = text_token
        self._depth(tokens, transformer_out.reshape(B, lm.dim).contiguous(), self._noise(B, lm.dep_q * self.top_k))
        return tokens[:, 1:]
