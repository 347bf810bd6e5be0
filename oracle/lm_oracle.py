"""CPU oracle for the RQ-Transformer decode step (temporal transformer + per-codebook depth transformer + LMGen
token plumbing) -- TEST INFRASTRUCTURE ONLY.

Functional torch-CPU restatement of the reference's streaming generation path
(paths relative to ``/root/reference/MLLM_v2``):

    models/model.py:364-389    LMModel.forward_text        (sum of 17 embeddings -> N-layer transformer step -> norm -> text head)
    models/model.py:392-428    LMModel.forward_depformer   (per-codebook in-proj + previous-token embedding -> depth transformer step -> head)
    models/model.py:490-597    LMGen.step / depformer_step (token ring cache, delays, greedy / top-k sampling)
    modules/transformer.py     StreamingTransformerLayer / StreamingMultiheadAttention / RingKVCache (incl. the `delta <= 0` slot map)
    modules/gating.py:12-51    ActivationGating (SiLU), hidden = 2*ff/3 or 21*dim/8
    utils/sampling.py:15-105   sample_token

Arithmetic is fp32 on whatever weights are given (the product stores weights in bf16; the tests feed the oracle the same
bf16-rounded values up-cast to fp32).  Pinned against the imported reference by tests/golden/make_golden.py
(fixture ``lm_tiny.npz``): logits <= 1e-5 relative, greedy tokens exact.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


@dataclass
class LMConfig:
    """Constructor keywords of LMModel (models/model.py:119-143) that matter for inference."""
    dim: int = 4096
    text_card: int = 32000
    existing_text_padding_id: Optional[int] = 3
    n_q: int = 16
    dep_q: int = 8
    card: int = 2048
    num_heads: int = 32
    num_layers: int = 32
    hidden_scale: float = 4.125
    context: int = 3000
    max_period: float = 10000.0
    depformer_dim: int = 1024
    depformer_dim_feedforward: int = 4224
    depformer_num_heads: int = 16
    depformer_num_layers: int = 6
    delays: List[int] = field(default_factory=lambda: [0, 0, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1])

    @property
    def num_codebooks(self) -> int:
        return self.n_q + 1

    @property
    def initial_token_id(self) -> int:
        return self.card

    @property
    def text_initial_token_id(self) -> int:
        return self.text_card


def rms_norm(x: torch.Tensor, alpha: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """_rms_norm with dtype=float32 (modules/transformer.py:34-46): x * alpha * rsqrt(eps + mean(x^2))."""
    x = x.float()
    var = eps + torch.mean(x ** 2, dim=-1, keepdim=True)
    return x * (alpha.float().view(-1) * torch.rsqrt(var))


def scaled_embedding(w: torch.Tensor, tok: torch.Tensor) -> torch.Tensor:
    """ScaledEmbedding.forward (models/model.py:83-91): id -1 -> exact zeros, ids clamped at 0."""
    y = F.embedding(tok.clamp(min=0), w.float())
    return torch.where((tok == -1)[..., None], torch.zeros(1), y)


def rope_interleaved_1(x: torch.Tensor, pos: int, max_period: float) -> torch.Tensor:
    """apply_rope for one step, x [B,H,1,D] (modules/rope.py:11-68)."""
    D = x.shape[-1]
    ds = torch.arange(D // 2, dtype=torch.float32)
    freqs = torch.exp(ds * (-math.log(max_period) * 2 / D))
    ang = freqs * (torch.tensor([pos]).float() + torch.arange(1, dtype=torch.float32)).view(-1, 1)
    xr, xi = x.reshape(*x.shape[:-1], D // 2, 2).unbind(-1)
    return torch.stack([xr * torch.cos(ang) - xi * torch.sin(ang), xr * torch.sin(ang) + xi * torch.cos(ang)], -1).reshape(x.shape)


class RingKV:
    """RingKVCache (modules/transformer.py:198-278) for T = 1 appends."""

    def __init__(self, B: int, H: int, D: int, capacity: int, dtype: torch.dtype = torch.float32):
        """``dtype``: the precision the cache stores at (the reference allocates it in the model's dtype, bf16 for the released
        checkpoints: modules/transformer.py:228); values are kept here as fp32 numbers that are exactly representable in it."""
        self.capacity = capacity
        self.dtype = dtype
        self.k = torch.zeros(B, H, capacity, D)
        self.v = torch.zeros(B, H, capacity, D)
        self.end_offset = 0

    def complete(self, k: torch.Tensor, v: torch.Tensor):
        idx = self.end_offset % self.capacity
        self.k[:, :, idx] = k[:, :, 0].to(self.dtype).float()
        self.v[:, :, idx] = v[:, :, 0].to(self.dtype).float()
        self.end_offset += 1
        slots = torch.arange(self.capacity)
        end_index = self.end_offset % self.capacity
        delta = slots - end_index
        pos = torch.where(delta <= 0, self.end_offset + delta, self.end_offset + delta - self.capacity)
        pos = torch.where(slots >= self.end_offset, torch.full_like(pos, -1), pos)
        return self.k, self.v, pos


@dataclass
class TransformerState:
    kv: List[RingKV]
    offset: int = 0


def new_transformer_state(B: int, n_layers: int, H: int, D: int, capacity: int) -> TransformerState:
    return TransformerState([RingKV(B, H, D, capacity) for _ in range(n_layers)])


def gating_hidden(dim: int, dim_feedforward: int) -> int:
    """modules/gating.py:40-45."""
    return (21 * dim) // 8 if dim_feedforward == 4 * dim else (2 * dim_feedforward) // 3


def transformer_step(sd: SD, prefix: str, x: torch.Tensor, st: TransformerState, *, num_heads: int, context: Optional[int],
                     rope: bool, max_period: float, weights_per_step: int = 0) -> torch.Tensor:
    """One streaming step (T = 1) of StreamingTransformer (modules/transformer.py:551-690): x [B,1,C] -> [B,1,C].
    With ``weights_per_step`` the in/out projections and the gating of step ``st.offset`` are used."""
    B, T, C = x.shape
    assert T == 1
    H, off = num_heads, st.offset
    n_layers = len(st.kv)
    for l in range(n_layers):
        p = f"{prefix}.layers.{l}"
        w_in, w_out = sd[f"{p}.self_attn.in_proj_weight"].float(), sd[f"{p}.self_attn.out_proj.weight"].float()
        if weights_per_step:
            w_in = w_in.view(weights_per_step, -1, C)[off]
            w_out = w_out.view(weights_per_step, -1, C)[off]
            g_in = sd[f"{p}.gating.{off}.linear_in.weight"].float()
            g_out = sd[f"{p}.gating.{off}.linear_out.weight"].float()
        else:
            g_in, g_out = sd[f"{p}.gating.linear_in.weight"].float(), sd[f"{p}.gating.linear_out.weight"].float()
        h = rms_norm(x, sd[f"{p}.norm1.alpha"])
        q, k, v = F.linear(h, w_in).view(B, 1, 3, H, C // H).permute(2, 0, 3, 1, 4)
        if rope:
            q, k = rope_interleaved_1(q, off, max_period), rope_interleaved_1(k, off, max_period)
        keys, vals, pos_k = st.kv[l].complete(k, v)
        delta = off - pos_k
        mask = (pos_k >= 0) & (delta >= 0)
        if context is not None:
            mask = mask & (delta < context)
        a = F.scaled_dot_product_attention(q, keys, vals, mask.view(1, -1), dropout_p=0.0)
        x = x + F.linear(a.permute(0, 2, 1, 3).reshape(B, 1, C), w_out)
        h = F.linear(rms_norm(x, sd[f"{p}.norm2.alpha"]), g_in).view(B, 1, 2, -1)
        x = x + F.linear(F.silu(h[..., 0, :]) * h[..., 1, :], g_out)
    st.offset += 1
    return x


def forward_text(sd: SD, cfg: LMConfig, tokens: torch.Tensor, st: TransformerState):
    """LMModel.forward_text for one step: tokens [B, n_q+1, 1] -> (transformer_out [B,1,dim], text_logits [B,1,1,V])."""
    x = None
    for cb in range(cfg.n_q):
        e = scaled_embedding(sd[f"emb.{cb}.weight"], tokens[:, cb + 1])
        x = e if x is None else x + e
    x = x + scaled_embedding(sd["text_emb.weight"], tokens[:, 0])
    out = transformer_step(sd, "transformer", x, st, num_heads=cfg.num_heads, context=cfg.context, rope=True,
                           max_period=cfg.max_period)
    out = rms_norm(out, sd["out_norm.alpha"])
    return out, F.linear(out, sd["text_linear.weight"].float())[:, None]


def forward_depformer(sd: SD, cfg: LMConfig, cb_index: int, prev_token: torch.Tensor, transformer_out: torch.Tensor,
                      st: TransformerState) -> torch.Tensor:
    """LMModel.forward_depformer: prev_token [B,1,1], transformer_out [B,1,dim] -> logits [B,1,1,card]."""
    x = F.linear(transformer_out, sd[f"depformer_in.{cb_index}.weight"].float())
    table = sd["depformer_text_emb.weight"] if cb_index == 0 else sd[f"depformer_emb.{cb_index - 1}.weight"]
    x = x + scaled_embedding(table, prev_token[:, 0])
    y = transformer_step(sd, "depformer", x, st, num_heads=cfg.depformer_num_heads, context=None, rope=False,
                         max_period=cfg.max_period, weights_per_step=cfg.dep_q)
    return F.linear(y, sd[f"linears.{cb_index}.weight"].float())[:, None]


def sample_token(logits: torch.Tensor, use_sampling: bool, temp: float, top_k: int, noise: Optional[torch.Tensor] = None,
                 top_p: float = 0.0):
    """sample_token (utils/sampling.py:85-105) with the exponential noise of `multinomial` (:44-46) passed in.  ``top_p > 0``:
    sample_top_p (:66-82) -- ``noise`` then has one entry per SORTED vocabulary position (full width); ties sort lowest id first."""
    if use_sampling and temp > 0.0:
        probs = torch.softmax(logits / temp, dim=-1)
        if top_p > 0.0:
            ps, idx = torch.sort(probs, dim=-1, descending=True, stable=True)
            mask = torch.cumsum(ps, dim=-1) - ps > top_p
            ps = ps * (~mask).float()
            ps = ps / ps.sum(dim=-1, keepdim=True)
            return idx.gather(-1, (ps / noise).argmax(dim=-1, keepdim=True))[..., 0]
        p, idx = torch.topk(probs, top_k, dim=-1)
        choice = (p / noise).argmax(dim=-1, keepdim=True)
        return idx.gather(-1, choice)[..., 0]
    return torch.argmax(logits, dim=-1)


class LMGenOracle:
    """LMGen (models/model.py:443-597): step-by-step generation with the delayed token ring cache."""

    def __init__(self, sd: SD, cfg: LMConfig, batch_size: int, use_sampling: bool = False, temp: float = 0.8,
                 temp_text: float = 0.7, top_k: int = 250, top_k_text: int = 25):
        self.sd, self.cfg, self.B = sd, cfg, batch_size
        self.use_sampling, self.temp, self.temp_text, self.top_k, self.top_k_text = use_sampling, temp, temp_text, top_k, top_k_text
        self.max_delay = max(cfg.delays)
        self.cache = torch.full((batch_size, cfg.num_codebooks, self.max_delay + 2), -2, dtype=torch.long)
        self.initial = torch.tensor([cfg.text_initial_token_id] + [cfg.initial_token_id] * cfg.n_q).view(1, -1, 1)
        self.offset = 0
        self.main = new_transformer_state(batch_size, cfg.num_layers, cfg.num_heads, cfg.dim // cfg.num_heads, cfg.context)

    def step(self, input_tokens: torch.Tensor, noise_text=None, noise_audio=None):
        cfg, CT = self.cfg, self.cache.shape[2]
        B, Ki, S = input_tokens.shape
        assert S == 1 and Ki == cfg.num_codebooks - cfg.dep_q - 1
        for q_other in range(Ki):
            k = cfg.dep_q + 1 + q_other
            self.cache[:, k, (self.offset + cfg.delays[k]) % CT] = input_tokens[:, q_other, 0]
        position = self.offset % CT
        for k, delay in enumerate(cfg.delays):
            if self.offset <= delay:
                self.cache[:, k, position] = self.initial[:, k, 0]
        input_ = self.cache[:, :, position:position + 1]
        out, text_logits = forward_text(self.sd, cfg, input_, self.main)
        text_token = sample_token(text_logits, self.use_sampling, self.temp_text, self.top_k_text, noise_text)[:, 0, 0]
        audio = self.depformer_step(text_token, out, noise_audio)
        self.offset += 1
        position = self.offset % CT
        self.cache[:, 0, position] = text_token
        self.cache[:, 1:cfg.dep_q + 1, position] = audio
        if self.offset <= self.max_delay:
            return None
        d = torch.tensor(cfg.delays[:cfg.dep_q + 1])
        index = ((self.offset - self.max_delay + d) % CT).view(1, -1, 1).expand(B, -1, 1)
        return self.cache.gather(2, index)

    def depformer_step(self, text_token: torch.Tensor, transformer_out: torch.Tensor, noise_audio=None) -> torch.Tensor:
        cfg = self.cfg
        # a fresh streaming context per frame: `with lm_model.depformer.streaming(B)` (models/model.py:577)
        st = new_transformer_state(self.B, cfg.depformer_num_layers, cfg.depformer_num_heads,
                                   cfg.depformer_dim // cfg.depformer_num_heads, cfg.dep_q)
        prev, toks = text_token, []
        for cb in range(cfg.dep_q):
            logits = forward_depformer(self.sd, cfg, cb, prev[:, None, None], transformer_out, st)
            nz = None if noise_audio is None else noise_audio[cb]
            prev = sample_token(logits, self.use_sampling, self.temp, self.top_k, nz)[:, 0, 0]
            toks.append(prev)
        return torch.stack(toks, 1)
