"""``GPT`` -- drop-in for the inference half of ``MLLM_v2/models/llama_streaming.py``: the litgpt-style speech-text backbone
(global transformer with fused grouped-query QKV, partial rotate-half RoPE, SiLU-gated MLP, optional LoRA adapters) plus the
per-codebook "codecformer" depth transformer.

Same ``Config`` field names, ``state_dict`` keys (``transformer.h.{l}.attn.attn.linear.weight``, ``...mlp.fc_1.linear.weight``,
``lm_head.linear.weight``, ``input_emb.{k}.weight``, ``codecformer.layers.{l}.gating.{k}.linear_in.weight``, ... incl. the
legacy remaps of llama_streaming.py:762-766,1000-1009,1034-1088), method signatures (``forward_global``,
``forward_codecformer``, ``forward_local``, ``forward``, ``_get_initial_token``, token-id properties) and streaming protocol
(``with gpt.streaming(B)``, ``with gpt.codecformer.streaming(B)``).

Execution (csrc/lm_step.hip, lm_attn.hip, lm_skinny.hip): bf16 weights, fp32 activations.
  * LoRA adapters are merged into the dense weights when a state dict is loaded (what the reference's ``merge_lora_weights``
    does before inference, :1120-), including the reference's zero_pad behaviour when q, k and v are all adapted.
    ``from_state_dict(..., merge_lora=False)`` keeps them apart instead, as ``LoRALinear.forward`` / ``LoRAQKVLinear.forward`` run
    them (:136-143, :373-406): every adapted linear is followed by two thin GEMMs, ``y += B_full (scaling * (A x))`` (same prologue as
    the dense product; the fused-QKV / stacked-gate B is laid out block-diagonally in the kernels' row order, rank padded to 16).
  * The fused QKV rows are re-ordered once from the GQA-interleaved ``[group: q.. k v]`` layout to ``[Q | K | V]`` and, inside
    every q / k head, from rotate-half pairing ``(i, i + n/2)`` to interleaved pairing ``(2i, 2i+1)`` -- a permutation applied to
    q and k alike leaves q.k unchanged -- so the ring-attention kernels of the Moshi path serve this model too.
  * The KV ring stores the n_query_groups key/value heads once (the reference expands them to n_head before caching, :965-971).
  * One decode step = 5 launches per layer for batch <= 2 (RMSNorm fused into the GEMVs); longer inputs (prompt prefill, batch
    > 2) use the bf16-MFMA skinny GEMM over all rows + the multi-query ring attention.  T = 1 steps are graph-captured.
"""
from __future__ import annotations

from dataclasses import dataclass, fields
from typing import Any, Dict, Optional, Tuple

import torch
from torch import nn

from .. import ops
from ..codec.conv import _PackedCache
from ..codec.streaming import StreamingModule
from ..graphs import Graphed as _Graphed
from .model import RMSNorm as _AlphaNorm  # noqa: F401  (key layout of the codecformer norms)
from .model import ScaledEmbedding, StreamingTransformer, _StepState, _Weight


@dataclass
class Config:
    """models/config.py:Config + models/llama_streaming.py:Config (:447-489).  Fields that select code paths this build does
    not implement are validated in ``__post_init__`` (NotImplementedError) instead of being silently ignored."""
    name: str = ""
    block_size: int = 4096
    n_layer: int = 16
    n_embd: int = 4096
    vocab_size: int = 50254
    padding_multiple: int = 512
    padded_vocab_size: Optional[int] = None
    norm_class_name: str = "LayerNorm"
    norm_eps: float = 1e-5
    norm_qk: bool = False
    post_attention_norm: bool = False
    post_mlp_norm: bool = False
    parallel_residual: bool = True
    shared_attention_norm: bool = False
    n_head: int = 32
    head_size: Optional[int] = None
    n_query_groups: Optional[int] = None
    attention_scores_scalar: Optional[int] = None
    sliding_window_size: Optional[int] = None
    attention_logit_softcapping: Optional[float] = None
    rope_base: int = 10000
    rotary_percentage: float = 0.25
    rope_condense_ratio: int = 1
    rope_adjustments: Optional[dict] = None
    intermediate_size: Optional[int] = None
    bias: bool = True
    mlp_class_name: str = "GptNeoxMLP"
    scale_embeddings: bool = False
    lm_head_bias: bool = False
    final_logit_softcapping: Optional[float] = None
    # lora
    lora_r: int = 0
    lora_alpha: int = 1
    lora_dropout: float = 0.0
    lora_query: bool = False
    lora_key: bool = False
    lora_value: bool = False
    lora_projection: bool = False
    lora_mlp: bool = False
    lora_head: bool = False
    # local transformer
    audio_card: int = 2048
    codecformer_dim: int = 1024
    n_q: int = 9
    dep_q: int = 8
    codecformer_heads: int = 32
    codecformer_layers: int = 6
    codecformer_hidden_scale: float = 4.5
    causal: bool = True
    codecformer_multi_linear: bool = True
    codecformer_weights_per_step: bool = True
    codecformer_dim_feedforward: int = 1024
    codecfomer_norm: str = "rms_norm_f32"
    codecformer_bias_proj: bool = False
    codecfomer_norm_emb: bool = False
    context: int = 3000

    def __post_init__(self):
        if self.head_size is None:
            assert self.n_embd % self.n_head == 0
            self.head_size = self.n_embd // self.n_head
        if self.padded_vocab_size is None:
            m = self.padding_multiple
            self.padded_vocab_size = self.vocab_size if self.vocab_size % m == 0 else self.vocab_size + m - self.vocab_size % m
        else:
            self.vocab_size = min(self.vocab_size, self.padded_vocab_size)
        if self.n_query_groups is None:
            self.n_query_groups = self.n_head
        assert self.n_head % self.n_query_groups == 0
        if self.intermediate_size is None:
            if self.mlp_class_name == "LLaMAMLP":
                raise ValueError(f"The config {self.name!r}, needs to set the `intermediate_size`")
            self.intermediate_size = 4 * self.n_embd
        self.rope_n_elem = int(self.rotary_percentage * self.head_size)
        unsupported = {
            "norm_class_name": self.norm_class_name != "RMSNorm", "mlp_class_name": self.mlp_class_name != "LLaMAMLP",
            "parallel_residual": self.parallel_residual, "shared_attention_norm": self.shared_attention_norm,
            "norm_qk": self.norm_qk, "post_attention_norm": self.post_attention_norm, "post_mlp_norm": self.post_mlp_norm,
            "attention_scores_scalar": self.attention_scores_scalar is not None,
            "sliding_window_size": self.sliding_window_size is not None,
            "attention_logit_softcapping": self.attention_logit_softcapping is not None,
            "final_logit_softcapping": self.final_logit_softcapping is not None,
            "rope_condense_ratio": self.rope_condense_ratio != 1, "rope_adjustments": self.rope_adjustments is not None,
            "scale_embeddings": self.scale_embeddings, "n_query_groups == 1 (the reference's ring cannot hold MQA either)":
                self.n_query_groups == 1 and self.n_head != 1,
            "codecformer options": not (self.codecformer_multi_linear and self.codecformer_weights_per_step and self.causal
                                        and self.codecfomer_norm == "rms_norm_f32" and not self.codecfomer_norm_emb),
            "rope_n_elem": self.rope_n_elem % 2 != 0 or self.rope_n_elem == 0,
        }
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(f"Config options outside the MI355X decode path: {bad} (supported: RMSNorm, LLaMAMLP, "
                                      "sequential residual, plain RoPE, MHA / GQA)")

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "Config":
        known = {f.name for f in fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in known})


# ---------------------------------------------------------------------------------------------------------------- loading
_LEGACY = (("lm_head.weight", "lm_head.linear.weight"), ("lm_head.bias", "lm_head.linear.bias"))
_LEGACY_SUFFIX = tuple((f".{m}.{p}", f".{m}.linear.{p}") for m in ("attn.attn", "attn.proj", "mlp.fc_1", "mlp.fc_2", "mlp.proj")
                       for p in ("weight", "bias"))


def _remap_legacy(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Base-checkpoint key names -> LoRA-wrapped names (llama_streaming.py:762-766, 1000-1009, 1076-1088)."""
    out = {}
    for k, v in sd.items():
        for old, new in _LEGACY:
            if k == old:
                k = new
        for old, new in _LEGACY_SUFFIX:
            if k.endswith(old) and not k.endswith(new):
                k = k[: -len(old)] + new
        out[k] = v
    return out


def _qkv_lora_rows(cfg: Config, enable: Tuple[bool, bool, bool]) -> Optional[torch.Tensor]:
    """Destination rows (in the GQA-interleaved fused-QKV output) of the adapter rows, which come ordered [q heads | k heads |
    v heads] over the enabled projections.  None = 'as is': the reference adds the update unscattered when all three
    projections are adapted (zero_pad, llama_streaming.py:307-308)."""
    if all(enable):
        return None
    hs, G = cfg.head_size, cfg.n_query_groups
    qpk = cfg.n_head // G
    d = torch.arange(hs)
    grp = torch.arange(G).view(G, 1, 1) * (qpk + 2)
    rows = []
    if enable[0]:
        rows.append(((grp + torch.arange(qpk).view(1, qpk, 1)) * hs + d).reshape(-1))
    if enable[1]:
        rows.append(((grp + qpk) * hs + d).reshape(-1))
    if enable[2]:
        rows.append(((grp + qpk + 1) * hs + d).reshape(-1))
    return torch.cat(rows)


def merge_lora_state_dict(sd: Dict[str, torch.Tensor], cfg: Config) -> Dict[str, torch.Tensor]:
    """``merge_lora_weights`` (llama_streaming.py:1120-) on a state dict: W += (B A) * alpha / r for every adapted linear
    (fp32 accumulate, stored back in W's dtype), adapter tensors dropped."""
    out = {k: v for k, v in sd.items() if not k.endswith((".lora_A", ".lora_B"))}
    scale = cfg.lora_alpha / cfg.lora_r if cfg.lora_r else 0.0
    for key in [k for k in sd if k.endswith(".lora_A")]:
        base = key[: -len(".lora_A")]
        A, Bm = sd[key].float(), sd[base + ".lora_B"].float()
        W = sd[base + ".linear.weight"]
        if base.endswith(".attn.attn"):
            enable = (cfg.lora_query, cfg.lora_key, cfg.lora_value)
            sizes = [n for n, e in zip((cfg.head_size * cfg.n_head, cfg.head_size * cfg.n_query_groups,
                                        cfg.head_size * cfg.n_query_groups), enable) if e]
            r = cfg.lora_r
            upd, row = [], 0
            for i, n in enumerate(sizes):      # every adapted projection owns r rows of A and its block of B rows
                upd.append(Bm[row:row + n] @ A[i * r:(i + 1) * r])
                row += n
            upd = torch.cat(upd) * scale
            rows = _qkv_lora_rows(cfg, enable)
            Wf = W.float()
            if rows is None:
                Wf = Wf + upd
            else:
                Wf = Wf.index_add(0, rows.to(Wf.device), upd)
        else:
            Wf = W.float() + (Bm @ A) * scale
        out[base + ".linear.weight"] = Wf.to(W.dtype)
    return out


def merge_lora_weights(model: "GPT") -> None:
    """API parity with llama_streaming.py:1120-: adapters are already merged when a ``GPT`` is built from a state dict."""
    return None


def _lora_bases(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Copies of the un-adapted weights of every linear that carries adapters in ``sd`` (key = ``<name>.linear.weight``)."""
    return {k[: -len(".lora_A")] + ".linear.weight": sd[k[: -len(".lora_A")] + ".linear.weight"].detach().clone()
            for k in sd if k.endswith(".lora_A")}


def _qkv_row_order(cfg: Config) -> torch.Tensor:
    """Row permutation of the fused QKV weight: GQA-interleaved [group: q_0..q_{qpk-1}, k, v] -> [Q heads | K heads | V heads],
    and within each q / k head the rotary dims from rotate-half order to interleaved pairs:
    new dim 2i <- old dim i, new dim 2i+1 <- old dim i + n/2 (i < n/2); dims >= n keep their place."""
    hs, H, G, n = cfg.head_size, cfg.n_head, cfg.n_query_groups, cfg.rope_n_elem
    qpk = H // G
    inner = torch.arange(hs)
    half = n // 2
    inner[0:n:2] = torch.arange(half)
    inner[1:n:2] = torch.arange(half) + half
    plain = torch.arange(hs)
    order = []
    for h in range(H):
        g, j = divmod(h, qpk)
        order.append((g * (qpk + 2) + j) * hs + inner)
    for g in range(G):
        order.append((g * (qpk + 2) + qpk) * hs + inner)
    for g in range(G):
        order.append((g * (qpk + 2) + qpk + 1) * hs + plain)
    return torch.cat(order)


def _pow2(v: float) -> bool:
    import math
    return v > 0 and math.frexp(v)[0] == 0.5


def _pad_rank(A: torch.Tensor, Bm: torch.Tensor, scale: float):
    """(A ``[r', in]``, B ``[out, r']``, post-scale): the rank padded with zeros to a multiple of 16 (the K granule of the thin
    second product); a power-of-two ``alpha / r`` is folded into B exactly, any other value is applied to ``A x`` in fp32."""
    r = A.shape[0]
    rp = (r + 15) // 16 * 16
    fold = _pow2(scale)
    Ap = torch.zeros(rp, A.shape[1], device=A.device, dtype=torch.bfloat16)
    Ap[:r] = A.detach().to(torch.bfloat16)
    Bp = torch.zeros(Bm.shape[0], rp, device=Bm.device, dtype=torch.bfloat16)
    Bp[:, :r] = (Bm.detach().float() * scale).to(torch.bfloat16) if fold else Bm.detach().to(torch.bfloat16)
    return Ap.contiguous(), Bp.contiguous(), (1.0 if fold else float(scale))


def _lora_add(y: torch.Tensor, x, ad, **prologue) -> torch.Tensor:
    """``y + B (scaling * (A P(x)))`` -- the unmerged LoRA branch (llama_streaming.py:136-143) as two thin products."""
    if ad is None:
        return y
    A, Bm, scale = ad
    a = ops.lm_linear(x, A, **prologue)
    if scale != 1.0:
        a = a * scale
    return ops.lm_linear(a, Bm, res=y)


# ----------------------------------------------------------------------------------------------------------------- modules
class _Linear(nn.Module):
    """``<name>.linear.{weight,bias}`` holder (a merged LoRALinear)."""

    def __init__(self, in_f: int, out_f: int, bias: bool, device=None, dtype=None):
        super().__init__()
        self.linear = nn.Module()
        self.linear.weight = nn.Parameter(torch.empty(out_f, in_f, device=device, dtype=dtype), requires_grad=False)
        self.linear.bias = nn.Parameter(torch.zeros(out_f, device=device, dtype=dtype), requires_grad=False) if bias else None
        self._b32 = _PackedCache()
        self.register_parameter("lora_A", None)      # unmerged adapters (GPT.from_state_dict(..., merge_lora=False))
        self.register_parameter("lora_B", None)
        self.lora_scale = 0.0
        self._ad = _PackedCache()

    def set_adapter(self, A: Optional[torch.Tensor], Bm: Optional[torch.Tensor], scale: float) -> None:
        self.lora_A = None if A is None else nn.Parameter(A, requires_grad=False)
        self.lora_B = None if Bm is None else nn.Parameter(Bm, requires_grad=False)
        self.lora_scale = float(scale)
        self._ad = _PackedCache()

    def adapter(self):
        """``(A, B, post_scale)`` in kernel form, or None for a linear without (unmerged) adapters."""
        if self.lora_A is None:
            return None
        return self._ad.get((self.lora_A, self.lora_B), lambda: _pad_rank(self.lora_A, self.lora_B, self.lora_scale))

    @property
    def weight(self) -> torch.Tensor:
        return self.linear.weight

    def bias_f32(self) -> Optional[torch.Tensor]:
        if self.linear.bias is None:
            return None
        return self._b32.get((self.linear.bias,), lambda: self.linear.bias.detach().float().contiguous())


class _PlainLinear(_Weight):
    """nn.Linear key layout (``weight`` / ``bias``) for codecformer_in / audio_linears."""

    def __init__(self, out_f: int, in_f: int, bias: bool = False, device=None, dtype=None):
        super().__init__(out_f, in_f, device=device, dtype=dtype)
        self.bias = nn.Parameter(torch.zeros(out_f, device=device, dtype=dtype), requires_grad=False) if bias else None
        self._b32 = _PackedCache()

    def bias_f32(self) -> Optional[torch.Tensor]:
        if self.bias is None:
            return None
        return self._b32.get((self.bias,), lambda: self.bias.detach().float().contiguous())


class _LitNorm(nn.Module):
    """lit_model.RMSNorm (:693-717): key ``weight`` [D]."""

    def __init__(self, dim: int, eps: float, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim, device=device, dtype=dtype), requires_grad=False)
        self._f32 = _PackedCache()

    def gain_f32(self) -> torch.Tensor:
        return self._f32.get((self.weight,), lambda: self.weight.detach().float().contiguous())


class CausalSelfAttention(nn.Module):
    """Weights of llama_streaming.py:867-998 (``attn`` fused QKV, ``proj``) + their kernel-side packing."""

    def __init__(self, config: Config, device=None, dtype=None):
        super().__init__()
        self.config = config
        fk = {"device": device, "dtype": dtype}
        self.attn = _Linear(config.n_embd, (config.n_head + 2 * config.n_query_groups) * config.head_size, config.bias, **fk)
        self.proj = _Linear(config.head_size * config.n_head, config.n_embd, config.bias, **fk)
        self._packed = _PackedCache()

    def packed_qkv(self) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """(weight, bias) with rows in [Q | K | V] order and interleaved rotary pairs (see the module docstring)."""
        w, b = self.attn.linear.weight, self.attn.linear.bias

        def build():
            order = _qkv_row_order(self.config).to(w.device)
            return (w.detach().index_select(0, order).contiguous(),
                    None if b is None else b.detach().float().index_select(0, order).contiguous())
        return self._packed.get((w,) if b is None else (w, b), build)

    def packed_qkv_adapter(self):
        """The fused-QKV adapter in kernel form: ``B_full [rows of packed_qkv, r * enabled]`` holds each adapted projection's B block
        in its own r columns (LoRAQKVLinear.conv1d, :310-354), on the rows ``zero_pad`` sends them to (:259-308, incl. its 'as is'
        return when q, k and v are all adapted), then in the [Q | K | V] / interleaved-rotary row order of ``packed_qkv``."""
        lin, c = self.attn, self.config
        if lin.lora_A is None:
            return None
        if not hasattr(self, "_packed_ad"):
            self._packed_ad = _PackedCache()

        def build():
            enable = (c.lora_query, c.lora_key, c.lora_value)
            sizes = [n for n, e in zip((c.head_size * c.n_head, c.head_size * c.n_query_groups, c.head_size * c.n_query_groups), enable) if e]
            r = c.lora_r
            Bm = lin.lora_B.detach()
            rows = _qkv_lora_rows(c, enable)
            rows = torch.arange(Bm.shape[0]) if rows is None else rows
            full = torch.zeros(lin.linear.weight.shape[0], r * len(sizes), device=Bm.device, dtype=Bm.dtype)
            row = 0
            for i, n in enumerate(sizes):
                full[rows[row:row + n].to(Bm.device), i * r:(i + 1) * r] = Bm[row:row + n]
                row += n
            full = full.index_select(0, _qkv_row_order(c).to(Bm.device))
            return _pad_rank(lin.lora_A, full, lin.lora_scale)
        return self._packed_ad.get((lin.lora_A, lin.lora_B), build)


class LLaMAMLP(nn.Module):
    """Weights of llama_streaming.py:1046-1088 / lit_model.py:391-403; fc_1 | fc_2 are stacked for the gated GEMV."""

    def __init__(self, config: Config, device=None, dtype=None):
        super().__init__()
        fk = {"device": device, "dtype": dtype}
        self.fc_1 = _Linear(config.n_embd, config.intermediate_size, config.bias, **fk)
        self.fc_2 = _Linear(config.n_embd, config.intermediate_size, config.bias, **fk)
        self.proj = _Linear(config.intermediate_size, config.n_embd, config.bias, **fk)
        self._packed = _PackedCache()

    def packed_fc(self) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        w1, w2, b1, b2 = self.fc_1.linear.weight, self.fc_2.linear.weight, self.fc_1.linear.bias, self.fc_2.linear.bias

        def build():
            return (torch.cat([w1.detach(), w2.detach()]).contiguous(),
                    None if b1 is None else torch.cat([b1.detach().float(), b2.detach().float()]).contiguous())
        return self._packed.get((w1, w2) if b1 is None else (w1, w2, b1, b2), build)

    def packed_fc_adapter(self):
        """Adapters of fc_1 | fc_2 for the stacked product: A's stacked, B's block-diagonal (either may be absent)."""
        l1, l2 = self.fc_1, self.fc_2
        if l1.lora_A is None and l2.lora_A is None:
            return None
        if not hasattr(self, "_packed_ad"):
            self._packed_ad = _PackedCache()
        present = [l for l in (l1, l2) if l.lora_A is not None]

        def build():
            I = l1.linear.weight.shape[0]
            scale = present[0].lora_scale
            As = [l.lora_A.detach() for l in present]
            full = torch.zeros(2 * I, sum(a.shape[0] for a in As), device=As[0].device, dtype=As[0].dtype)
            col = 0
            for l in present:
                r = l.lora_A.shape[0]
                o = 0 if l is l1 else I
                full[o:o + I, col:col + r] = l.lora_B.detach()
                col += r
            return _pad_rank(torch.cat(As), full, scale)
        return self._packed_ad.get(tuple(p for l in present for p in (l.lora_A, l.lora_B)), build)


class Block(nn.Module):
    """llama_streaming.py:810-853 (sequential residual)."""

    def __init__(self, config: Config, device=None, dtype=None):
        super().__init__()
        fk = {"device": device, "dtype": dtype}
        self.norm_1 = _LitNorm(config.n_embd, config.norm_eps, **fk)
        self.attn = CausalSelfAttention(config, **fk)
        self.norm_2 = _LitNorm(config.n_embd, config.norm_eps, **fk)
        self.mlp = LLaMAMLP(config, **fk)


class LLAMAStreamingTransformer(StreamingModule[_StepState]):
    """llama_streaming.py:775-800: ``wte`` + blocks + ``ln_f``; the streaming state holds the grouped KV rings."""

    def __init__(self, config: Config, device=None, dtype=None):
        super().__init__()
        self.config = config
        fk = {"device": device, "dtype": dtype}
        self.wte = _Weight(config.padded_vocab_size, config.n_embd, **fk)
        self.h = nn.ModuleList([Block(config, **fk) for _ in range(config.n_layer)])
        self.ln_f = _LitNorm(config.n_embd, config.norm_eps, **fk)
        self.fp8 = False      # opt-in: run the block linears on the fp8 (e4m3, per-row scales) matrix-core path

    def _make_state(self, batch_size: int, capacity: int) -> _StepState:
        c = self.config
        dev = self.ln_f.weight.device
        shape = (batch_size, c.n_query_groups, capacity, c.head_size)
        scratch = None
        if capacity > 64:
            splits = ops.lm_attn_splits(capacity, batch_size * c.n_head)
            scratch = (torch.empty(batch_size, c.n_head, splits, c.head_size + 2, device=dev),
                       torch.zeros(batch_size, c.n_head, device=dev, dtype=torch.int32))
        return _StepState([torch.zeros(shape, device=dev) for _ in self.h], [torch.zeros(shape, device=dev) for _ in self.h],
                          torch.zeros(1, device=dev, dtype=torch.long), scratch)

    def _init_streaming_state(self, batch_size: int) -> _StepState:
        if self.config.context is None:
            raise RuntimeError("Cannot create a streaming KVCache without a context to estimate capacity.")
        return self._make_state(batch_size, self.config.context)

    def run(self, x: torch.Tensor, B: int, T: int, st: _StepState) -> torch.Tensor:
        """x fp32 ``[B*T, n_embd]`` (T new positions per stream) -> final-normed hidden ``[B*T, n_embd]``."""
        c = self.config
        H, G, hs, n = c.n_head, c.n_query_groups, c.head_size, c.rope_n_elem
        f8 = self.fp8
        rope_table = None
        if T == 1 and st.k[0].shape[2] > 64:      # the step's rotation once for all blocks (long rings)
            rope_table = ops.lm_rope_table(st.pos, hs, max_period=float(c.rope_base), rope_dims=n)
        for l, blk in enumerate(self.h):
            wqkv, bqkv = blk.attn.packed_qkv()
            n1 = dict(prologue=ops.PROLOGUE_RMSNORM, alpha=blk.norm_1.gain_f32(), eps=blk.norm_1.eps)
            qkv = _lora_add(ops.lm_linear(x, wqkv, bias=bqkv, fp8=f8, **n1), x, blk.attn.packed_qkv_adapter(), **n1)
            if T == 1:
                a = ops.lm_attn_decode(qkv, st.k[l], st.v[l], st.pos, rope=True, context=c.context, max_period=float(c.rope_base),
                                       scratch=st.scratch, heads=H, rope_dims=n, packed=B > 2 and not f8, rope_table=rope_table)
            else:
                q = ops.lm_rope_append(qkv.view(B, T, -1), st.k[l], st.v[l], st.pos, heads=H, rope=True,
                                       max_period=float(c.rope_base), rope_dims=n)
                a = ops.attention(q, st.k[l], st.v[l], pos_dev=st.pos, ring=True, context=c.context).view(B * T, H * hs)
            x = _lora_add(ops.lm_linear(a, blk.attn.proj.weight, res=x, bias=blk.attn.proj.bias_f32(), fp8=f8), a, blk.attn.proj.adapter())
            wfc, bfc = blk.mlp.packed_fc()
            ad_fc, ad_proj = blk.mlp.packed_fc_adapter(), blk.mlp.proj.adapter()
            if ad_fc is None and ad_proj is None:
                x = ops.lm_gated_pair(x, wfc, blk.mlp.proj.weight, alpha=blk.norm_2.gain_f32(), eps=blk.norm_2.eps, res=x, bias_in=bfc,
                                      bias_out=blk.mlp.proj.bias_f32(), fp8=f8)
            else:       # unmerged adapters: the gate follows fc_1 / fc_2 WITH their updates, so the pair runs as two plain products
                n2 = dict(prologue=ops.PROLOGUE_RMSNORM, alpha=blk.norm_2.gain_f32(), eps=blk.norm_2.eps)
                u = _lora_add(ops.lm_linear(x, wfc, bias=bfc, fp8=f8, **n2), x, ad_fc, **n2)
                x = _lora_add(ops.lm_linear(u, blk.mlp.proj.weight, prologue=ops.PROLOGUE_SILU_GATE, res=x, bias=blk.mlp.proj.bias_f32(),
                                            fp8=f8), u, ad_proj, prologue=ops.PROLOGUE_SILU_GATE)
        st.pos.add_(T)
        st.offset_cpu += T
        return ops.rmsnorm(x, self.ln_f.gain_f32(), self.ln_f.eps)


@dataclass
class _GPTState:
    graphed_global: _Graphed
    graphed_depth: Optional[_Graphed] = None

    def reset(self) -> None:
        pass


class GPT(StreamingModule[_GPTState]):
    """The audio-text LLM of llama_streaming.py:520-767 (inference)."""

    def __init__(self, config: Config, device=None, dtype=torch.bfloat16):
        super().__init__()
        assert config.padded_vocab_size is not None
        self.config = config
        fk = {"device": device, "dtype": dtype}
        self.lm_head = _Linear(config.n_embd, config.padded_vocab_size, config.lm_head_bias, **fk)
        self._lora_base: Optional[Dict[str, torch.Tensor]] = None      # from_state_dict(..., keep_lora_base=True)
        self._unmerged = False                                         # from_state_dict(..., merge_lora=False)
        self.dep_q = config.dep_q
        self.transformer = LLAMAStreamingTransformer(config, **fk)
        self.max_seq_length = config.block_size
        self.input_emb = nn.ModuleList([ScaledEmbedding(config.audio_card + 1, config.n_embd, **fk) for _ in range(config.n_q)])
        self.codecformer_in = nn.ModuleList([_PlainLinear(config.codecformer_dim, config.n_embd, **fk) for _ in range(config.dep_q)])
        self.codecformer_emb = nn.ModuleList([ScaledEmbedding(config.audio_card + 1, config.codecformer_dim, **fk)
                                              for _ in range(config.dep_q - 1)])
        self.codecformer_text_emb = ScaledEmbedding(config.padded_vocab_size, config.codecformer_dim, **fk)
        self.codecformer = StreamingTransformer(config.codecformer_dim, config.codecformer_heads, config.codecformer_layers,
                                                config.codecformer_dim_feedforward, None, "none", 10000.0,
                                                weights_per_step=config.dep_q, **fk)
        self.codecformer.set_streaming_propagate(False)
        self.audio_linears = nn.ModuleList([_PlainLinear(config.audio_card, config.codecformer_dim, config.codecformer_bias_proj, **fk)
                                            for _ in range(config.dep_q)])

    def use_fp8(self, enabled: bool = True) -> "GPT":
        """Opt into the fp8 matrix-core path for the linears of the global transformer blocks (BASELINE.json configs[4]);
        weights are quantised lazily (e4m3, one scale per row), activations per step (one scale per batch row)."""
        self.transformer.fp8 = bool(enabled)
        return self

    # ---- token-id conventions (llama_streaming.py:591-649)
    @property
    def zero_token_id(self) -> int:
        return -1

    @property
    def text_initial_token_id(self) -> int:
        return 151655

    @property
    def initial_token_id(self) -> int:
        return self.config.audio_card

    @property
    def num_codebooks(self) -> int:
        return self.config.n_q + 1

    @property
    def num_audio_codebooks(self) -> int:
        return self.config.n_q

    @property
    def audio_offset(self) -> int:
        return 1

    @property
    def ungenerated_token_id(self) -> int:
        return -2

    @property
    def device(self):
        return next(iter(self.parameters())).device

    def _get_initial_token(self) -> torch.Tensor:
        tok = torch.full([1, self.num_codebooks, 1], self.initial_token_id, device=self.device, dtype=torch.long)
        tok[:, 0] = self.text_initial_token_id
        return tok

    # ---- streaming state
    def _init_streaming_state(self, batch_size: int) -> _GPTState:
        disable = self.device.type != "cuda"
        return _GPTState(_Graphed(self._global_step, disable=disable))

    # ---- global transformer
    def _embed(self, toks: torch.Tensor) -> torch.Tensor:
        K = toks.shape[1]
        tables = [e.weight for e in self.input_emb] + [self.transformer.wte.weight]
        return ops.embed_sum(toks, tables, list(range(1, K)) + [0])       # audio streams first, then text, as :680-686

    def _head(self, h: torch.Tensor) -> torch.Tensor:
        return _lora_add(ops.lm_linear(h, self.lm_head.weight, bias=self.lm_head.bias_f32()), h, self.lm_head.adapter())

    def _global_step(self, toks: torch.Tensor):
        """One streamed position: toks int64 ``[B, n_q+1]`` -> (hidden ``[B, n_embd]``, text logits ``[B, V]``)."""
        st = self.transformer._streaming_state
        h = self.transformer.run(self._embed(toks), toks.shape[0], 1, st)
        return h, self._head(h)

    @torch.no_grad()
    def forward_global(self, sequence: torch.Tensor):
        """sequence int64 ``[B, n_q+1, T]`` -> (transformer_out fp32 ``[B,T,n_embd]``, text_logits fp32 ``[B,T,V]``).
        Outside ``streaming()`` the T positions are 0..T-1 (plain causal + context mask, no ring effects); inside, they follow
        the positions already streamed (T = 1 steps replay a captured graph)."""
        B, K, T = sequence.shape
        assert K == self.num_codebooks, f"Sequence shape {sequence.shape} must match the number of codebooks."
        if self.max_seq_length < T:
            raise ValueError(f"Cannot forward sequence of length {T}, max seq length is only {self.max_seq_length}.")
        c = self.config
        st = self.transformer._streaming_state
        if st is not None and T == 1 and self._streaming_state is not None:
            h, logits = self._streaming_state.graphed_global(sequence.reshape(B, K).contiguous())
            return h.view(B, 1, c.n_embd), logits.view(B, 1, -1)
        if st is None:
            st = self.transformer._make_state(B, T + 1)      # a ring that never fills: positions 0..T-1 all addressable
        toks = sequence.permute(0, 2, 1).reshape(B * T, K).contiguous()
        h = self.transformer.run(self._embed(toks), B, T, st)
        logits = self._head(h)
        return h.view(B, T, c.n_embd), logits.view(B, T, -1)

    # ---- local (depth) transformer
    def codecformer_in_all(self) -> torch.Tensor:
        """``[dep_q * codecformer_dim, n_embd]``: the dep_q ``codecformer_in[k]`` matrices stacked (built once per weight version),
        so that a frame's dep_q products with ``transformer_out`` are one weight-streaming launch."""
        if not hasattr(self, "_in_cat"):
            from ..codec.conv import _PackedCache
            self._in_cat = _PackedCache()
        ws = [m.weight for m in self.codecformer_in]
        return self._in_cat.get(tuple(ws), lambda: torch.cat([w.detach() for w in ws], 0).contiguous())

    def depth_frame_tables(self):
        """Pointer tables of the persistent depth-frame launch (``lm.depth_frame.DepthFrameTables``), rebuilt when a weight changes."""
        from ..codec.conv import _PackedCache
        from .depth_frame import DepthFrameTables
        if not hasattr(self, "_depth_tables"):
            self._depth_tables = _PackedCache()
        dep = self.codecformer
        params = [p for l in dep.layers for p in (l.self_attn.in_proj_weight, l.self_attn.out_proj.weight, l.norm1.alpha, l.norm2.alpha)]
        params += [g.linear_in.weight for l in dep.layers for g in l.gating] + [g.linear_out.weight for l in dep.layers for g in l.gating]
        params += [m.weight for m in self.audio_linears] + [m.bias for m in self.audio_linears]
        params += [self.codecformer_text_emb.weight] + [m.weight for m in self.codecformer_emb]
        return self._depth_tables.get(tuple(params), lambda: DepthFrameTables(
            dep, [m.weight for m in self.audio_linears], [m.bias_f32() for m in self.audio_linears],
            [self.codecformer_text_emb.weight] + [m.weight for m in self.codecformer_emb][:self.config.dep_q - 1]))

    def _codec_step(self, k: int, prev: torch.Tensor, h: Optional[torch.Tensor], h_all: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One depth step: (codecformer_in[k](h) + embedding of the previous token ``prev`` int64 [N]) through the codecformer ->
        ``[N, codecformer_dim]``; the sum is formed inside the first launch of the step.  ``h_all``: ``h @ codecformer_in_all().T``."""
        E = self.config.codecformer_dim
        add = h_all[:, k * E:(k + 1) * E] if h_all is not None else ops.lm_linear(h, self.codecformer_in[k].weight)
        table = self.codecformer_text_emb.weight if k == 0 else self.codecformer_emb[k - 1].weight
        return self.codecformer.step(None, embed=(add, table, prev.reshape(-1, 1).contiguous(), 0))

    def _codec_in(self, k: int, prev: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
        """codecformer_in[k](h) + embedding of the previous token (text embedding for k = 0): prev int64 [N], h fp32 [N, n_embd]."""
        x = ops.lm_linear(h, self.codecformer_in[k].weight)
        table = self.codecformer_text_emb.weight if k == 0 else self.codecformer_emb[k - 1].weight
        return ops.embed_sum(prev.reshape(-1, 1).contiguous(), [table], [0], add=x)

    @torch.no_grad()
    def forward_codecformer(self, codecformer_cb_index: int, sequence: torch.Tensor, transformer_out: torch.Tensor) -> torch.Tensor:
        """sequence int64 ``[B,1,1]`` (previous token), transformer_out fp32 ``[B,1,n_embd]`` -> logits fp32 ``[B,1,1,card]``;
        the caller holds ``with gpt.codecformer.streaming(B)`` for the dep_q steps of a frame, as with the reference."""
        B, K, S = sequence.shape
        assert K == 1, f"Codebooks for Depformer streaming should be passed 1 by 1, got {K}."
        assert S == 1, f"Steps for Depformer streaming should be passed 1 by 1, got {S}."
        assert transformer_out.shape[1] == 1, "Transformer out should be a for a single step."
        k = codecformer_cb_index
        y = self._codec_step(k, sequence.reshape(B), transformer_out.reshape(B, -1).float().contiguous())
        head = self.audio_linears[k]
        return ops.lm_linear(y, head.weight, bias=head.bias_f32()).view(B, 1, 1, -1)

    @torch.no_grad()
    def forward_local(self, local_start_token: torch.Tensor, sequence: torch.Tensor, transformer_out: torch.Tensor) -> torch.Tensor:
        """Teacher-forced depth logits (llama_streaming.py:694-725).  ``local_start_token``: the ``codecformer_text_emb``
        embedding of the text ids, float ``[B,T,D]`` as in the reference, or the int64 ids ``[B,T]`` themselves (the lookup then
        stays inside the kernel); ``sequence`` int64 ``[B,dep_q,T]``, ``transformer_out`` fp32 ``[B,T,n_embd]`` ->
        ``[B,T,dep_q,card]``.  Runs dep_q steps over B*T rows on a ring that never fills (the non-streaming reference path has
        no ring, hence no Q1 slot quirk at the last codebook)."""
        B, K, T = sequence.shape
        assert K == self.config.dep_q, f"Sequence shape {sequence.shape} must match the moshi stream output."
        dep, N = self.codecformer, B * T
        saved = dep._streaming_state
        dep._streaming_state = dep._init_streaming_state(N, capacity=self.config.dep_q + 1)
        try:
            h = transformer_out.reshape(N, -1).float().contiguous()
            outs = []
            for k in range(K):
                if k == 0 and local_start_token.dtype != torch.long:
                    x = ops.lm_linear(h, self.codecformer_in[0].weight, res=local_start_token.reshape(N, -1).float().contiguous())
                else:
                    x = self._codec_in(k, local_start_token.reshape(N) if k == 0 else sequence[:, k - 1].reshape(N), h)
                y = dep.step(x)
                head = self.audio_linears[k]
                outs.append(ops.lm_linear(y, head.weight, bias=head.bias_f32()).view(B, T, 1, -1))
        finally:
            dep._streaming_state = saved
        return torch.cat(outs, dim=2)

    @torch.no_grad()
    def forward(self, sequence: torch.Tensor, input_pos: Optional[torch.Tensor] = None, lm_head_chunk_size: int = 0):
        """llama_streaming.py:651-663 (inference only): sequence ``[B, n_q+1, S]`` -> (audio_logits ``[B,S,dep_q,card]``,
        text_logits ``[B,S,V]``), the global input being the sequence shifted right behind the initial frame."""
        B, K, S = sequence.shape
        start = self._get_initial_token().repeat(B, 1, 1)
        transformer_out, text_logits = self.forward_global(torch.cat([start, sequence[:, :, :-1]], dim=2))
        audio_logits = self.forward_local(sequence[:, 0, :], sequence[:, 1:self.config.dep_q + 1, :], transformer_out)
        return audio_logits, text_logits

    # ---- loading
    @classmethod
    def from_state_dict(cls, sd: Dict[str, torch.Tensor], config: Config, keep_lora_base: bool = False, merge_lora: bool = True) -> "GPT":
        """Model for ``config`` with weights taken from ``sd`` (reference key names, legacy base-checkpoint names accepted,
        LoRA adapters merged) without copying the dense tensors.  ``keep_lora_base``: also keep a copy of the un-adapted weight
        of every adapted linear, so that ``load_adapters`` can swap adapter sets later (the reference keeps adapters unmerged
        for that, llama_streaming.py:113-143; here a swap re-merges in place).  ``merge_lora=False``: the dense weights stay as
        they are and the adapters run as their own thin products behind every adapted linear (the reference's forward before
        ``merge_lora_weights``; ``state_dict()`` then carries ``<name>.lora_A`` / ``<name>.lora_B`` like the reference's, and
        ``load_adapters`` swaps the tensors without touching a dense weight)."""
        sd = _remap_legacy(dict(sd))
        sd = {k: v for k, v in sd.items() if not k.endswith(("cos", "sin", "_lora_ind"))}
        bases = None
        if any(k.endswith(".lora_A") for k in sd):
            if config.lora_r <= 0:
                raise RuntimeError("state dict carries LoRA adapters but config.lora_r == 0")
            if not merge_lora:
                is_ad = lambda k: k.endswith((".lora_A", ".lora_B"))
                model = cls._from_merged({k: v for k, v in sd.items() if not is_ad(k)}, config)
                model._unmerged = True
                model._set_adapters({k: v for k, v in sd.items() if is_ad(k)})
                return model
            if keep_lora_base:
                bases = _lora_bases(sd)
            sd = merge_lora_state_dict(sd, config)
        model = cls._from_merged(sd, config)
        model._lora_base = bases
        return model

    def load_adapters(self, adapters: Optional[Dict[str, torch.Tensor]]) -> None:
        """Swap the LoRA adapter set of a model built with ``keep_lora_base=True``: every adapted linear becomes
        ``W_base + (B A) * alpha / r`` of the new ``<name>.lora_A`` / ``<name>.lora_B`` tensors, written IN PLACE into the live
        weights (same fp32 accumulate + rounding as at load, incl. the fused-QKV row scatter); linears of the base set that the new
        set does not adapt -- or all of them, ``adapters=None`` -- return to their base weights.  The kernel-side packed copies
        follow the weights' ``_version``, so the next call re-packs what changed; captured decode graphs hold the OLD packed
        copies, hence swaps are only allowed between sessions (outside ``streaming()`` / ``GPTGen.begin``)."""
        if self._lora_base is None and not self._unmerged:
            raise RuntimeError("load_adapters needs a model built with from_state_dict(..., keep_lora_base=True) or (..., merge_lora=False) "
                               "from a state dict with adapters")
        if self._streaming_state is not None or self.transformer._streaming_state is not None:
            raise RuntimeError("load_adapters: swap adapters between sessions, not inside streaming()")
        if self._unmerged:      # (captured decode graphs hold the launches of the OLD adapter set, hence the same between-sessions rule)
            self._set_adapters({} if adapters is None else {k: v for k, v in adapters.items() if k.endswith((".lora_A", ".lora_B"))})
            return
        adapters = {} if adapters is None else {k: v for k, v in adapters.items() if k.endswith((".lora_A", ".lora_B"))}
        unknown = [k for k in adapters if k.endswith(".lora_A") and k[: -len(".lora_A")] + ".linear.weight" not in self._lora_base]
        if unknown:
            raise RuntimeError(f"load_adapters: no base weight kept for {unknown[:3]} (adapt the same linears as the set the model was built with)")
        params = dict(self.named_parameters())
        tmp = dict(adapters)
        tmp.update({k: v for k, v in self._lora_base.items()})
        merged = merge_lora_state_dict({k: v.to(self._lora_base[next(iter(self._lora_base))].device) for k, v in tmp.items()}, self.config)
        with torch.no_grad():
            for name in self._lora_base:
                params[name].copy_(merged[name])

    def _adapted_linears(self) -> Dict[str, "_Linear"]:
        """The linears the config adapts (llama_streaming.py:870-882 attn / proj, :1049-1072 mlp, :530-537 lm_head), by key prefix."""
        c = self.config
        out: Dict[str, _Linear] = {}
        if c.lora_r <= 0:
            return out
        if c.lora_head:
            out["lm_head"] = self.lm_head
        for l, blk in enumerate(self.transformer.h):
            p = f"transformer.h.{l}"
            if c.lora_query or c.lora_key or c.lora_value:
                out[f"{p}.attn.attn"] = blk.attn.attn
            if c.lora_projection:
                out[f"{p}.attn.proj"] = blk.attn.proj
            if c.lora_mlp:
                out.update({f"{p}.mlp.fc_1": blk.mlp.fc_1, f"{p}.mlp.fc_2": blk.mlp.fc_2, f"{p}.mlp.proj": blk.mlp.proj})
        return out

    def _set_adapters(self, adapters: Dict[str, torch.Tensor]) -> None:
        """Install ``<name>.lora_A`` / ``<name>.lora_B`` on the linears the config adapts (shapes as the reference allocates them,
        :107-110, :205-221); linears without an entry run without a LoRA branch."""
        c = self.config
        lin = self._adapted_linears()
        names = {k[: -len(".lora_A")] for k in adapters if k.endswith(".lora_A")} | {k[: -len(".lora_B")] for k in adapters if k.endswith(".lora_B")}
        unknown = sorted(n for n in names if n not in lin)
        if unknown:
            raise RuntimeError(f"adapters for linears the config does not adapt: {unknown[:3]}")
        dev = self.lm_head.weight.device
        for name, m in lin.items():
            if name not in names:
                m.set_adapter(None, None, 0.0)
                continue
            if f"{name}.lora_A" not in adapters or f"{name}.lora_B" not in adapters:
                raise RuntimeError(f"{name}: lora_A and lora_B come as a pair")
            A, Bm = adapters[f"{name}.lora_A"], adapters[f"{name}.lora_B"]
            n_en = sum((c.lora_query, c.lora_key, c.lora_value)) if name.endswith(".attn.attn") else 1
            out_f = m.linear.weight.shape[0]
            if name.endswith(".attn.attn"):
                out_f = sum(n for n, e in zip((c.head_size * c.n_head, c.head_size * c.n_query_groups, c.head_size * c.n_query_groups),
                                              (c.lora_query, c.lora_key, c.lora_value)) if e)
            if tuple(A.shape) != (c.lora_r * n_en, m.linear.weight.shape[1]) or tuple(Bm.shape) != (out_f, c.lora_r):
                raise RuntimeError(f"{name}: adapter shapes {tuple(A.shape)} / {tuple(Bm.shape)} do not fit r={c.lora_r}")
            m.set_adapter(A.to(dev), Bm.to(dev), c.lora_alpha / c.lora_r)
        for blk in self.transformer.h:      # kernel-side forms are rebuilt from the new tensors
            blk.attn._packed_ad = _PackedCache()
            blk.mlp._packed_ad = _PackedCache()

    @classmethod
    def _from_merged(cls, sd: Dict[str, torch.Tensor], config: Config) -> "GPT":
        model = cls(config, device="meta")
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
