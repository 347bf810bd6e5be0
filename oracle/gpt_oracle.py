"""CPU oracle for the litgpt-style speech-text backbone ``GPT`` of MLLM_v2 (global transformer with LoRA / grouped-query
attention / rotate-half RoPE + the per-codebook "codecformer" depth transformer) -- TEST INFRASTRUCTURE ONLY.

Functional torch-CPU restatement of (paths relative to ``/root/reference/MLLM_v2``):

    models/llama_streaming.py:60-143     LoRALinear (forward, get_lora_AB, merge)
    models/llama_streaming.py:146-406    LoRAQKVLinear (lora_ind, zero_pad, grouped conv1d, get_lora_AB, forward)
    models/llama_streaming.py:520-749    GPT: forward_global / forward_local / forward_codecformer / forward
    models/llama_streaming.py:775-853    LLAMAStreamingTransformer, Block
    models/llama_streaming.py:867-998    CausalSelfAttention (GQA-interleaved fused QKV, partial rotate-half RoPE, ring KV, mask)
    models/lit_model.py:399-403          LLaMAMLP
    models/lit_model.py:441-488,560-573  build_rope_cache / apply_rope
    models/lit_model.py:600-660          RingKVCache.complete (same `delta <= 0` slot map as the Moshi ring)
    models/lit_model.py:693-717          RMSNorm

Reference quirks restated on purpose:
  * zero_pad returns its input unchanged when q, k and v are ALL LoRA-enabled (llama_streaming.py:307-308), so the update
    computed in [q-all | k-all | v-all] order is added to the GQA-interleaved rows as is.
  * In streaming mode the RoPE table is indexed with the single state offset (:975-977), so only T = 1 steps carry the
    right positions; this oracle (and the product) feed streaming steps one position at a time.

Arithmetic is fp32 on whatever weights are given.  Pinned against the imported reference by tests/golden/make_golden.py
(fixture ``gpt_tiny.npz``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .lm_oracle import TransformerState, new_transformer_state, scaled_embedding, transformer_step

SD = Dict[str, torch.Tensor]


@dataclass
class GPTConfig:
    """The fields of models/config.py:Config + models/llama_streaming.py:Config (:447-489) that shape inference."""
    n_layer: int = 24
    n_embd: int = 1024
    n_head: int = 16
    n_query_groups: Optional[int] = None
    head_size: Optional[int] = None
    padded_vocab_size: int = 151936
    block_size: int = 4096
    rotary_percentage: float = 1.0
    rope_base: int = 10000
    rope_condense_ratio: int = 1
    norm_eps: float = 1e-5
    bias: bool = False
    lm_head_bias: bool = False
    intermediate_size: int = 2816
    # lora
    lora_r: int = 0
    lora_alpha: int = 1
    lora_query: bool = False
    lora_key: bool = False
    lora_value: bool = False
    lora_projection: bool = False
    lora_mlp: bool = False
    lora_head: bool = False
    # local transformer
    audio_card: int = 2048
    codecformer_dim: int = 1024
    n_q: int = 8
    dep_q: int = 8
    codecformer_heads: int = 16
    codecformer_layers: int = 6
    codecformer_dim_feedforward: int = 4224
    codecformer_bias_proj: bool = False
    context: int = 3000

    def __post_init__(self):
        if self.head_size is None:
            self.head_size = self.n_embd // self.n_head
        if self.n_query_groups is None:
            self.n_query_groups = self.n_head
        self.rope_n_elem = int(self.rotary_percentage * self.head_size)

    @property
    def num_codebooks(self) -> int:
        return self.n_q + 1


# ------------------------------------------------------------------------------------------------------------------ LoRA
def lora_qkv_ind(cfg: GPTConfig, enable: Tuple[bool, bool, bool]) -> torch.Tensor:
    """LoRAQKVLinear.lora_ind (llama_streaming.py:236-257): rows of the interleaved QKV output touched by the enabled parts."""
    hs, group = cfg.head_size, cfg.n_head // cfg.n_query_groups + 2
    out_features = (cfg.n_head + 2 * cfg.n_query_groups) * hs
    rows = range(out_features)
    ind: List[int] = []
    if enable[0]:
        ind += [x for x in rows if (x // hs) % group < group - 2]
    if enable[1]:
        ind += [x for x in rows if (x // hs) % group == group - 2]
    if enable[2]:
        ind += [x for x in rows if (x // hs) % group == group - 1]
    return torch.tensor(ind, dtype=torch.long)


def lora_qkv_delta(cfg: GPTConfig, A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """LoRAQKVLinear.get_lora_AB (:356-366): the [out_features, in_features] update of the fused QKV weight."""
    enable = (cfg.lora_query, cfg.lora_key, cfg.lora_value)
    n_en = sum(enable)
    r = cfg.lora_r
    shapes = [s for s in (cfg.head_size * cfg.n_head * enable[0], cfg.head_size * cfg.n_query_groups * enable[1],
                          cfg.head_size * cfg.n_query_groups * enable[2]) if s]
    # conv1d (:320-354): each enabled part's B block multiplies its own r rows of A
    parts, row = [], 0
    for i, n in enumerate(shapes):
        parts.append(B[row:row + n].float() @ A[i * r:(i + 1) * r].float())
        row += n
    lora = torch.cat(parts, 0) * (cfg.lora_alpha / r)          # [sum(shapes), in]
    if n_en == 3:                                              # zero_pad's early return (:307-308)
        return lora
    out_features = (cfg.n_head + 2 * cfg.n_query_groups) * cfg.head_size
    full = torch.zeros(out_features, A.shape[1])
    full[lora_qkv_ind(cfg, enable)] = lora
    return full


def lora_linear_weight(sd: SD, prefix: str, r: int, alpha: int) -> torch.Tensor:
    """LoRALinear.merge (:113-134): W + (B @ A) * alpha / r (W alone when the layer has no adapter)."""
    w = sd[f"{prefix}.linear.weight"].float()
    if r > 0 and f"{prefix}.lora_A" in sd:
        w = w + (sd[f"{prefix}.lora_B"].float() @ sd[f"{prefix}.lora_A"].float()) * (alpha / r)
    return w


def merged_state(sd: SD, cfg: GPTConfig) -> SD:
    """merge_lora_weights (:1120-): every LoRA(QKV)Linear collapsed to a plain weight; other entries are passed through."""
    out = {k: v for k, v in sd.items() if "lora_" not in k}
    r, a = cfg.lora_r, cfg.lora_alpha
    for l in range(cfg.n_layer):
        p = f"transformer.h.{l}"
        w = sd[f"{p}.attn.attn.linear.weight"].float()
        if r > 0 and any((cfg.lora_query, cfg.lora_key, cfg.lora_value)):
            w = w + lora_qkv_delta(cfg, sd[f"{p}.attn.attn.lora_A"], sd[f"{p}.attn.attn.lora_B"])
        out[f"{p}.attn.attn.linear.weight"] = w
        out[f"{p}.attn.proj.linear.weight"] = lora_linear_weight(sd, f"{p}.attn.proj", r if cfg.lora_projection else 0, a)
        for name in ("fc_1", "fc_2", "proj"):
            out[f"{p}.mlp.{name}.linear.weight"] = lora_linear_weight(sd, f"{p}.mlp.{name}", r if cfg.lora_mlp else 0, a)
    out["lm_head.linear.weight"] = lora_linear_weight(sd, "lm_head", r if cfg.lora_head else 0, a)
    return out


def lora_linear_forward(x: torch.Tensor, sd: SD, prefix: str, r: int, alpha: int) -> torch.Tensor:
    """LoRALinear.forward, unmerged (:136-143)."""
    y = F.linear(x, sd[f"{prefix}.linear.weight"].float(), sd.get(f"{prefix}.linear.bias"))
    if r > 0 and f"{prefix}.lora_A" in sd:
        y = y + (x @ sd[f"{prefix}.lora_A"].float().T @ sd[f"{prefix}.lora_B"].float().T) * (alpha / r)
    return y


def lora_qkv_forward(x: torch.Tensor, sd: SD, prefix: str, cfg: GPTConfig) -> torch.Tensor:
    """LoRAQKVLinear.forward, unmerged (:373-406)."""
    y = F.linear(x, sd[f"{prefix}.linear.weight"].float(), sd.get(f"{prefix}.linear.bias"))
    enable = (cfg.lora_query, cfg.lora_key, cfg.lora_value)
    if cfg.lora_r == 0 or not any(enable):
        return y
    return y + F.linear(x, lora_qkv_delta(cfg, sd[f"{prefix}.lora_A"], sd[f"{prefix}.lora_B"]))


# ----------------------------------------------------------------------------------------------------------- primitives
def fp8_e4m3_rows(t: torch.Tensor) -> torch.Tensor:
    """Per-row symmetric e4m3 quantisation as the product's opt-in fp8 path applies it (one scale = amax / 448 per weight row, one
    dynamic scale per activation row; csrc/lm_skinny.hip rst_skinny_pack_{weight,act}_fp8), returned de-quantised in fp32."""
    sc = t.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30) / 448.0
    return (t / sc).to(torch.float8_e4m3fn).float() * sc


def fp8_linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 product of the e4m3-quantised operands (+ the unquantised bias): what BASELINE.json configs[4]'s "fp8 MFMA GEMMs" compute."""
    return F.linear(fp8_e4m3_rows(x.float()), fp8_e4m3_rows(w.float()), b)


_block_linear = F.linear      # the linears inside the global blocks (merged weights); swapped by `fp8_blocks()`


class fp8_blocks:
    """``with fp8_blocks():`` -- the block linears of ``forward_global(..., merged=True)`` run as ``fp8_linear`` (the oracle side of
    ``GPT.use_fp8()``: fused QKV, attention projection, fc_1 / fc_2, MLP projection; embeddings, norms, attention and the LM head
    stay fp32 exactly as in the product)."""

    def __enter__(self):
        global _block_linear
        self._saved, _block_linear = _block_linear, fp8_linear
        return self

    def __exit__(self, *exc):
        global _block_linear
        _block_linear = self._saved
        return False


def lit_rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """lit_model.RMSNorm.forward (:707-714) without the unit offset."""
    x = x.float()
    return x * torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + eps) * weight.float()


def build_rope_cache(seq_len: int, n_elem: int, base: int = 10000, condense_ratio: int = 1):
    """lit_model.build_rope_cache (:441-488) without the Llama-3 frequency adjustment."""
    theta = 1.0 / (base ** (torch.arange(0, n_elem, 2).float() / n_elem))
    idx_theta = torch.outer(torch.arange(seq_len) / condense_ratio, theta).repeat(1, 2)
    return torch.cos(idx_theta), torch.sin(idx_theta)


def apply_rope_half(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """lit_model.apply_rope (:560-573): rotate-half pairs (i, i + n/2); x [B,nh,T,n], cos/sin [T,n]."""
    n = x.size(-1)
    rotated = torch.cat((-x[..., n // 2:], x[..., : n // 2]), dim=-1)
    return x * cos + rotated * sin


class RingKV:
    """lit_model.RingKVCache (:600-660), any T per append."""

    def __init__(self, B: int, H: int, D: int, capacity: int):
        self.capacity = capacity
        self.k = torch.zeros(B, H, capacity, D)
        self.v = torch.zeros(B, H, capacity, D)
        self.end_offset = 0

    def complete(self, k: torch.Tensor, v: torch.Tensor):
        T = k.shape[2]
        idx = (torch.arange(T) + self.end_offset) % self.capacity
        self.k.index_copy_(2, idx, k)
        self.v.index_copy_(2, idx, v)
        self.end_offset += T
        slots = torch.arange(self.capacity)
        delta = slots - self.end_offset % self.capacity
        pos = torch.where(delta <= 0, self.end_offset + delta, self.end_offset + delta - self.capacity)
        pos = torch.where(slots >= self.end_offset, torch.full_like(pos, -1), pos)
        return self.k, self.v, pos


@dataclass
class GlobalState:
    kv: List[RingKV]
    offset: int = 0


def new_global_state(cfg: GPTConfig, B: int) -> GlobalState:
    """CausalSelfAttention._init_streaming_state (:905-926): one ring of n_head (expanded) heads per layer."""
    return GlobalState([RingKV(B, cfg.n_head, cfg.head_size, cfg.context) for _ in range(cfg.n_layer)])


def attention(x: torch.Tensor, sd: SD, p: str, cfg: GPTConfig, cos: torch.Tensor, sin: torch.Tensor, ring: Optional[RingKV],
              offset: int, merged: bool) -> torch.Tensor:
    """CausalSelfAttention.forward (:935-998)."""
    B, T, _ = x.shape
    G, hs = cfg.n_query_groups, cfg.head_size
    q_per_kv = cfg.n_head // G
    if merged:
        qkv = _block_linear(x, sd[f"{p}.attn.linear.weight"].float(), sd.get(f"{p}.attn.linear.bias"))
    else:
        qkv = lora_qkv_forward(x, sd, f"{p}.attn", cfg)
    qkv = qkv.view(B, T, G, q_per_kv + 2, hs).permute(0, 2, 3, 1, 4)
    q, k, v = qkv.split((q_per_kv, 1, 1), dim=2)
    if G != cfg.n_head and G != 1:
        k = k.expand(B, G, q_per_kv, T, hs)
        v = v.expand(B, G, q_per_kv, T, hs)
    q, k, v = q.reshape(B, -1, T, hs), k.reshape(B, -1, T, hs), v.reshape(B, -1, T, hs)
    if ring is not None:
        assert T == 1, "streaming steps carry one position (see the module docstring)"
        c, s = cos[offset:offset + 1], sin[offset:offset + 1]
    else:
        c, s = cos[:T], sin[:T]
    n = cfg.rope_n_elem
    q = torch.cat((apply_rope_half(q[..., :n], c, s), q[..., n:]), dim=-1)
    k = torch.cat((apply_rope_half(k[..., :n], c, s), k[..., n:]), dim=-1)
    if ring is None:
        pos_k = torch.arange(T)
    else:
        k, v, pos_k = ring.complete(k, v)
    delta = (offset + torch.arange(T)).view(-1, 1) - pos_k.view(1, -1)
    mask = (pos_k.view(1, -1) >= 0) & (delta >= 0)
    if cfg.context is not None:
        mask = mask & (delta < cfg.context)
    y = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, scale=1.0 / hs ** 0.5)
    y = y.transpose(1, 2).reshape(B, T, hs * cfg.n_head)
    if merged:
        return _block_linear(y, sd[f"{p}.proj.linear.weight"].float(), sd.get(f"{p}.proj.linear.bias"))
    return lora_linear_forward(y, sd, f"{p}.proj", cfg.lora_r if cfg.lora_projection else 0, cfg.lora_alpha)


def mlp(x: torch.Tensor, sd: SD, p: str, cfg: GPTConfig, merged: bool) -> torch.Tensor:
    """LLaMAMLP.forward (lit_model.py:399-403) over (LoRA) linears."""
    if merged:
        lin = lambda t, name: _block_linear(t, sd[f"{p}.{name}.linear.weight"].float(), sd.get(f"{p}.{name}.linear.bias"))     # noqa: E731
        return lin(F.silu(lin(x, "fc_1")) * lin(x, "fc_2"), "proj")
    r = cfg.lora_r if cfg.lora_mlp else 0
    a = lora_linear_forward(x, sd, f"{p}.fc_1", r, cfg.lora_alpha)
    b = lora_linear_forward(x, sd, f"{p}.fc_2", r, cfg.lora_alpha)
    return lora_linear_forward(F.silu(a) * b, sd, f"{p}.proj", r, cfg.lora_alpha)


def forward_global(sd: SD, cfg: GPTConfig, sequence: torch.Tensor, st: Optional[GlobalState] = None, merged: bool = False):
    """GPT.forward_global (:665-692): sequence [B, n_q+1, T] -> (transformer_out [B,T,n_embd], text_logits [B,T,V]).
    ``merged`` says whether ``sd`` went through merged_state (plain weights) or still carries lora_A / lora_B."""
    B, K, T = sequence.shape
    assert K == cfg.num_codebooks
    if cfg.block_size < T:
        raise ValueError(f"Cannot forward sequence of length {T}, max seq length is only {cfg.block_size}.")
    cos, sin = build_rope_cache(cfg.block_size, cfg.rope_n_elem, cfg.rope_base, cfg.rope_condense_ratio)
    x = None
    for cb in range(cfg.n_q):
        e = scaled_embedding(sd[f"input_emb.{cb}.weight"], sequence[:, cb + 1])
        x = e if x is None else x + e
    x = x + F.embedding(sequence[:, 0], sd["transformer.wte.weight"].float())
    offset = 0 if st is None else st.offset
    for l in range(cfg.n_layer):
        p = f"transformer.h.{l}"
        h = lit_rms_norm(x, sd[f"{p}.norm_1.weight"], cfg.norm_eps)
        x = attention(h, sd, f"{p}.attn", cfg, cos, sin, None if st is None else st.kv[l], offset, merged) + x
        x = mlp(lit_rms_norm(x, sd[f"{p}.norm_2.weight"], cfg.norm_eps), sd, f"{p}.mlp", cfg, merged) + x
    x = lit_rms_norm(x, sd["transformer.ln_f.weight"], cfg.norm_eps)
    if st is not None:
        st.offset += T
    r = 0 if merged or not cfg.lora_head else cfg.lora_r
    return x, lora_linear_forward(x, sd, "lm_head", r, cfg.lora_alpha)


def new_codecformer_state(cfg: GPTConfig, B: int) -> TransformerState:
    return new_transformer_state(B, cfg.codecformer_layers, cfg.codecformer_heads, cfg.codecformer_dim // cfg.codecformer_heads,
                                 cfg.dep_q)


def forward_codecformer(sd: SD, cfg: GPTConfig, cb_index: int, sequence: torch.Tensor, transformer_out: torch.Tensor,
                        st: TransformerState) -> torch.Tensor:
    """GPT.forward_codecformer (:727-749): previous token [B,1,1] + transformer_out [B,1,n_embd] -> logits [B,1,1,audio_card]."""
    x = F.linear(transformer_out.float(), sd[f"codecformer_in.{cb_index}.weight"].float())
    table = sd["codecformer_text_emb.weight"] if cb_index == 0 else sd[f"codecformer_emb.{cb_index - 1}.weight"]
    x = x + scaled_embedding(table, sequence[:, 0])
    y = transformer_step(sd, "codecformer", x, st, num_heads=cfg.codecformer_heads, context=None, rope=False, max_period=10000.0,
                         weights_per_step=cfg.dep_q)
    return F.linear(y, sd[f"audio_linears.{cb_index}.weight"].float(), sd.get(f"audio_linears.{cb_index}.bias"))[:, None]


def codecformer_full(sd: SD, cfg: GPTConfig, x: torch.Tensor) -> torch.Tensor:
    """The depth transformer NOT in streaming mode (modules/transformer.py:551-690 with state None): x [N, dep_q, C], plain
    causal attention over the dep_q axis, step k using the k-th slice of the per-step weights (multi_linear, :155-179)."""
    from .lm_oracle import rms_norm
    N, K, C = x.shape
    H = cfg.codecformer_heads
    causal = torch.tril(torch.ones(K, K, dtype=torch.bool))
    for l in range(cfg.codecformer_layers):
        p = f"codecformer.layers.{l}"
        w_in = sd[f"{p}.self_attn.in_proj_weight"].float().view(cfg.dep_q, 3 * C, C)
        w_out = sd[f"{p}.self_attn.out_proj.weight"].float().view(cfg.dep_q, C, C)
        h = rms_norm(x, sd[f"{p}.norm1.alpha"])
        qkv = torch.stack([F.linear(h[:, k], w_in[k]) for k in range(K)], 1)          # [N, K, 3C]
        q, k_, v = qkv.view(N, K, 3, H, C // H).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k_, v, causal, dropout_p=0.0).permute(0, 2, 1, 3).reshape(N, K, C)
        x = x + torch.stack([F.linear(a[:, k], w_out[k]) for k in range(K)], 1)
        h = rms_norm(x, sd[f"{p}.norm2.alpha"])
        ys = []
        for k in range(K):
            u = F.linear(h[:, k], sd[f"{p}.gating.{k}.linear_in.weight"].float()).view(N, 2, -1)
            ys.append(F.linear(F.silu(u[:, 0]) * u[:, 1], sd[f"{p}.gating.{k}.linear_out.weight"].float()))
        x = x + torch.stack(ys, 1)
    return x


def forward_local(sd: SD, cfg: GPTConfig, text_tokens: torch.Tensor, sequence: torch.Tensor, transformer_out: torch.Tensor):
    """GPT.forward_local (:694-725) on teacher-forced tokens: text_tokens [B,T], sequence [B,dep_q,T], transformer_out
    [B,T,n_embd] -> logits [B,T,dep_q,audio_card].  The depth transformer runs NON-streaming over the dep_q axis per (b, t):
    unlike dep_q streamed steps it has no ring, hence no `delta <= 0` slot-map quirk at the last codebook."""
    B, K, T = sequence.shape
    assert K == cfg.dep_q
    h = transformer_out.reshape(B * T, -1).float()
    xs = []
    for cb in range(cfg.dep_q):
        table = sd["codecformer_text_emb.weight"] if cb == 0 else sd[f"codecformer_emb.{cb - 1}.weight"]
        prev = text_tokens.reshape(B * T) if cb == 0 else sequence[:, cb - 1].reshape(B * T)
        xs.append(F.linear(h, sd[f"codecformer_in.{cb}.weight"].float()) + scaled_embedding(table, prev))
    y = codecformer_full(sd, cfg, torch.stack(xs, 1))
    out = [F.linear(y[:, cb], sd[f"audio_linears.{cb}.weight"].float(), sd.get(f"audio_linears.{cb}.bias")).view(B, T, 1, -1)
           for cb in range(cfg.dep_q)]
    return torch.cat(out, dim=2)
