"""CPU oracle for the MimiCodec hot path -- TEST INFRASTRUCTURE ONLY.

This file is a functional, CPU-only (PyTorch fp32 / int64) restatement of the
reference's MimiCodec encode / decode algorithm.  It exists so that the HIP
kernels in ``rstnet_amd/csrc`` can be checked for parity on a machine where
``/root/reference`` does not exist.  Only ``tests/``, ``__graft_entry__.smoke``
and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``rstnet_amd`` never does.

Pinning: ``tests/golden/make_golden.py`` imports the real reference
(``/root/reference/MLLM_v2/tools/tokenizer/MimiCodec``) in the build container,
loads the same seeded weights and stores its outputs as fixtures under
``tests/golden``; ``tests/test_oracle_golden.py`` checks this file against them
(codes: exact, floats: <=1e-5 relative).

Every function cites the reference lines it restates (paths relative to
``/root/reference/MLLM_v2/tools/tokenizer/MimiCodec/model``).  All tensors use
the reference layout ``[B, C, T]`` at function boundaries.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


@dataclass
class MimiConfig:
    """Hyper-parameters of models/MimiCodec.py:26-72 (+ mimi_config.yaml)."""
    sample_rate: int = 24000
    n_filters: int = 64
    ratios: List[int] = field(default_factory=lambda: [8, 6, 5, 4])  # decoder order
    kernel_size: int = 7
    last_kernel_size: int = 3
    residual_kernel_size: int = 3
    compress: int = 2
    latent_dim: int = 512
    codebook_size: int = 2048
    codebook_dim: int = 256
    rvq_layers: int = 8
    n_q_semantic: int = 1
    num_heads: int = 8
    num_layers: int = 8
    context: int = 250
    dim_feedforward: int = 2048
    max_period: float = 10000.0
    resample_stride: int = 2  # encoder_frame_rate / target_frame_rate

    @property
    def hop_length(self) -> int:
        return int(math.prod(self.ratios)) * self.resample_stride


# --------------------------------------------------------------------------
# convolutions (modules/conv.py, modules/streaming.py)
# --------------------------------------------------------------------------

def extra_padding_for_conv1d(length: int, kernel_eff: int, stride: int, padding_total: int) -> int:
    """modules/conv.py:50-57 -- right padding so that the last window is full."""
    n_frames = (length - kernel_eff + padding_total) / stride + 1
    ideal = (math.ceil(n_frames) - 1) * stride + (kernel_eff - padding_total)
    return ideal - length


def causal_conv1d(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], stride: int = 1,
                  dilation: int = 1, pad_mode: str = "constant", groups: int = 1) -> torch.Tensor:
    """Non-streaming StreamingConv1d.forward, causal branch (modules/conv.py:232-254)."""
    k_eff = (w.shape[-1] - 1) * dilation + 1
    pad_total = k_eff - stride
    extra = extra_padding_for_conv1d(x.shape[-1], k_eff, stride, pad_total)
    x = F.pad(x, (pad_total, extra), mode=pad_mode)
    return F.conv1d(x, w, b, stride=stride, dilation=dilation, groups=groups)


def causal_convtr1d(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], stride: int,
                    groups: int = 1) -> torch.Tensor:
    """Non-streaming StreamingConvTranspose1d.forward, causal, trim_right_ratio=1 (modules/conv.py:305-329)."""
    k = w.shape[-1]
    y = F.conv_transpose1d(x, w, b, stride=stride, groups=groups)
    trim = k - stride
    return y[..., : y.shape[-1] - trim] if trim > 0 else y


def resnet_block(sd: SD, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """SEANetResnetBlock.forward with true_skip (modules/seanet.py:21-94): x + k1(ELU(k3(ELU(x))))."""
    h = causal_conv1d(F.elu(x), sd[f"{prefix}.block.1.conv.conv.weight"], sd[f"{prefix}.block.1.conv.conv.bias"])
    h = causal_conv1d(F.elu(h), sd[f"{prefix}.block.3.conv.conv.weight"], sd[f"{prefix}.block.3.conv.conv.bias"])
    return x + h


def seanet_encoder(sd: SD, cfg: MimiConfig, x: torch.Tensor, prefix: str = "encoder") -> torch.Tensor:
    """SEANetEncoder.forward (modules/seanet.py:97-241), n_residual_layers=1, norm none."""
    i = 0
    x = causal_conv1d(x, sd[f"{prefix}.model.{i}.conv.conv.weight"], sd[f"{prefix}.model.{i}.conv.conv.bias"])
    i += 1
    for ratio in reversed(cfg.ratios):
        x = resnet_block(sd, f"{prefix}.model.{i}", x)
        i += 2  # resblock, ELU
        x = causal_conv1d(F.elu(x), sd[f"{prefix}.model.{i}.conv.conv.weight"],
                          sd[f"{prefix}.model.{i}.conv.conv.bias"], stride=ratio)
        i += 1
    i += 1  # ELU
    return causal_conv1d(F.elu(x), sd[f"{prefix}.model.{i}.conv.conv.weight"], sd[f"{prefix}.model.{i}.conv.conv.bias"])


def seanet_decoder(sd: SD, cfg: MimiConfig, z: torch.Tensor, prefix: str = "decoder") -> torch.Tensor:
    """SEANetDecoder.forward (modules/seanet.py:244-395)."""
    i = 0
    x = causal_conv1d(z, sd[f"{prefix}.model.{i}.conv.conv.weight"], sd[f"{prefix}.model.{i}.conv.conv.bias"])
    i += 1
    for ratio in cfg.ratios:
        i += 1  # ELU
        x = causal_convtr1d(F.elu(x), sd[f"{prefix}.model.{i}.convtr.convtr.weight"],
                            sd[f"{prefix}.model.{i}.convtr.convtr.bias"], stride=ratio)
        i += 1
        x = resnet_block(sd, f"{prefix}.model.{i}", x)
        i += 1
    i += 1  # ELU
    return causal_conv1d(F.elu(x), sd[f"{prefix}.model.{i}.conv.conv.weight"], sd[f"{prefix}.model.{i}.conv.conv.bias"])


def downsample(sd: SD, cfg: MimiConfig, x: torch.Tensor) -> torch.Tensor:
    """ConvDownsample1d learnt, dense, replicate pad, no bias (modules/resample.py:14-65)."""
    return causal_conv1d(x, sd["downsample.conv.conv.conv.weight"], None, stride=cfg.resample_stride,
                         pad_mode="replicate")


def upsample(sd: SD, cfg: MimiConfig, x: torch.Tensor) -> torch.Tensor:
    """ConvTrUpsample1d learnt, channel-wise (modules/resample.py:68-119)."""
    return causal_convtr1d(x, sd["upsample.convtr.convtr.convtr.weight"], None, stride=cfg.resample_stride,
                           groups=x.shape[1])


# --------------------------------------------------------------------------
# transformer (modules/transformer.py, modules/rope.py)
# --------------------------------------------------------------------------

def rope_interleaved(q: torch.Tensor, k: torch.Tensor, offset: int, max_period: float):
    """apply_rope, [B,H,T,D] layout, interleaved (real, imag) pairs, fp32 (modules/rope.py:11-68)."""
    B, H, T, D = q.shape
    ds = torch.arange(D // 2, dtype=torch.float32)
    freqs = torch.exp(ds * (-math.log(max_period) * 2 / D))
    ts = (torch.tensor([offset]).float() + torch.arange(T, dtype=torch.float32)).view(1, -1, 1)
    rotr, roti = torch.cos(freqs * ts), torch.sin(freqs * ts)
    out = []
    for t in (q, k):
        t = t.view(B, H, T, D // 2, 2)
        tr, ti = t[..., 0], t[..., 1]
        out.append(torch.stack([tr * rotr - ti * roti, tr * roti + ti * rotr], dim=-1).view(B, H, T, D))
    return out[0], out[1]


def attention_mask(T: int, context: Optional[int]) -> torch.Tensor:
    """Non-streaming mask of StreamingMultiheadAttention.forward (modules/transformer.py:404-414)."""
    pos = torch.arange(T)
    delta = pos.view(-1, 1) - pos.view(1, -1)
    mask = delta >= 0
    if context is not None:
        mask = mask & (delta < context)
    return mask


def transformer_layer(sd: SD, p: str, cfg: MimiConfig, x: torch.Tensor) -> torch.Tensor:
    """StreamingTransformerLayer.forward, gating none, layer_norm, LayerScale (modules/transformer.py:551-592)."""
    B, T, C = x.shape
    H = cfg.num_heads
    h = F.layer_norm(x, (C,), sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], eps=1e-5)
    proj = F.linear(h, sd[f"{p}.self_attn.in_proj_weight"])
    q, k, v = proj.view(B, T, 3, H, C // H).permute(2, 0, 3, 1, 4)  # "b t (p h d) -> p b h t d"
    q, k = rope_interleaved(q, k, 0, cfg.max_period)
    a = F.scaled_dot_product_attention(q, k, v, attention_mask(T, cfg.context), dropout_p=0.0)
    a = a.permute(0, 2, 1, 3).reshape(B, T, C)
    x = x + sd[f"{p}.layer_scale_1.scale"] * F.linear(a, sd[f"{p}.self_attn.out_proj.weight"])
    h = F.layer_norm(x, (C,), sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], eps=1e-5)
    u = F.linear(F.gelu(F.linear(h, sd[f"{p}.linear1.weight"])), sd[f"{p}.linear2.weight"])
    return x + sd[f"{p}.layer_scale_2.scale"] * u


def projected_transformer(sd: SD, prefix: str, cfg: MimiConfig, x: torch.Tensor) -> torch.Tensor:
    """ProjectedTransformer.forward, conv_layout, no projections (modules/transformer.py:738-750)."""
    x = x.transpose(1, 2)
    for layer in range(cfg.num_layers):
        x = transformer_layer(sd, f"{prefix}.transformer.layers.{layer}", cfg, x)
    return x.transpose(1, 2)


# --------------------------------------------------------------------------
# streaming transformer: ring KV cache (modules/transformer.py:211-278, 376-423)
# --------------------------------------------------------------------------

class RingKVCache:
    """RingKVCache for appends of T >= 1 steps (modules/transformer.py:211-278), positions incl. the `delta <= 0` branch that
    gives the slot at end_index the position end_offset once the ring has wrapped."""

    def __init__(self, B: int, H: int, D: int, capacity: int):
        self.capacity = capacity
        self.k = torch.zeros(B, H, capacity, D)
        self.v = torch.zeros(B, H, capacity, D)
        self.end_offset = 0

    def complete(self, k: torch.Tensor, v: torch.Tensor):
        """k, v [B,H,T,D] -> (keys [B,H,cap,D], values, positions [cap] int64)."""
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


def ring_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, ring: RingKVCache, offset: int,
                   context: Optional[int], max_period: Optional[float]) -> torch.Tensor:
    """The streaming branch of StreamingMultiheadAttention.forward (modules/transformer.py:392-416): RoPE at `offset`, ring
    append, position mask, SDPA.  q, k, v [B,H,T,D] -> [B,T,H*D]."""
    B, H, T, D = q.shape
    if max_period is not None:
        q, k = rope_interleaved(q.contiguous(), k.contiguous(), offset, max_period)
    keys, vals, pos_k = ring.complete(k, v)
    delta = (offset + torch.arange(T)).view(-1, 1) - pos_k.view(1, -1)
    mask = (pos_k.view(1, -1) >= 0) & (delta >= 0)
    if context is not None:
        mask = mask & (delta < context)
    a = F.scaled_dot_product_attention(q, keys, vals, mask, dropout_p=0.0)
    return a.permute(0, 2, 1, 3).reshape(B, T, H * D)


class TransformerStream:
    """ProjectedTransformer in streaming mode (modules/transformer.py:595-750): one ring per layer, shared offset."""

    def __init__(self, sd: SD, prefix: str, cfg: MimiConfig, B: int):
        self.sd, self.prefix, self.cfg, self.offset = sd, prefix, cfg, 0
        self.rings = [RingKVCache(B, cfg.num_heads, cfg.latent_dim // cfg.num_heads, cfg.context) for _ in range(cfg.num_layers)]

    def step(self, x: torch.Tensor) -> torch.Tensor:
        """x [B,C,T] (conv layout) -> [B,C,T]."""
        sd, cfg = self.sd, self.cfg
        x = x.transpose(1, 2)
        B, T, C = x.shape
        H = cfg.num_heads
        for l, ring in enumerate(self.rings):
            p = f"{self.prefix}.transformer.layers.{l}"
            h = F.layer_norm(x, (C,), sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], eps=1e-5)
            q, k, v = F.linear(h, sd[f"{p}.self_attn.in_proj_weight"]).view(B, T, 3, H, C // H).permute(2, 0, 3, 1, 4)
            a = ring_attention(q, k, v, ring, self.offset, cfg.context, cfg.max_period)
            x = x + sd[f"{p}.layer_scale_1.scale"] * F.linear(a, sd[f"{p}.self_attn.out_proj.weight"])
            h = F.layer_norm(x, (C,), sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], eps=1e-5)
            u = F.linear(F.gelu(F.linear(h, sd[f"{p}.linear1.weight"])), sd[f"{p}.linear2.weight"])
            x = x + sd[f"{p}.layer_scale_2.scale"] * u
        self.offset += T
        return x.transpose(1, 2)


def encode_latent_streamed(sd: SD, cfg: MimiConfig, audio: torch.Tensor, frames_per_chunk: int = 1) -> torch.Tensor:
    """The latent of `MimiModel.encode` called chunk by chunk inside `streaming()` (moshi/models/compression.py:368-380).  The
    causal convolutions give the same values streamed or not (modules/streaming.py:205-303), so only the transformer -- whose
    ring cache hides one key once it has wrapped -- is actually stepped; audio length must be a multiple of the hop."""
    B, _, T = audio.shape
    assert T % cfg.hop_length == 0
    z = seanet_encoder(sd, cfg, audio)
    tr = TransformerStream(sd, "encoder_transformer", cfg, B)
    n = cfg.resample_stride * frames_per_chunk
    z = torch.cat([tr.step(z[:, :, i:i + n]) for i in range(0, z.shape[-1], n)], -1)
    return downsample(sd, cfg, z)


def encode_streamed(sd: SD, cfg: MimiConfig, audio: torch.Tensor, frames_per_chunk: int = 1) -> torch.Tensor:
    return rvq_encode(sd, cfg, encode_latent_streamed(sd, cfg, audio, frames_per_chunk))


def decode_streamed(sd: SD, cfg: MimiConfig, codes: torch.Tensor, frames_per_chunk: int = 1) -> torch.Tensor:
    """`MimiModel.decode` chunk by chunk inside `streaming()` (compression.py:398-419); see `encode_streamed`."""
    B = codes.shape[0]
    z = upsample(sd, cfg, rvq_decode(sd, cfg, codes))
    tr = TransformerStream(sd, "decoder_transformer", cfg, B)
    n = cfg.resample_stride * frames_per_chunk
    z = torch.cat([tr.step(z[:, :, i:i + n]) for i in range(0, z.shape[-1], n)], -1)
    return seanet_decoder(sd, cfg, z)


# --------------------------------------------------------------------------
# residual vector quantiser (quantization/vq.py, quantization/core_vq.py)
# --------------------------------------------------------------------------

def codebook(sd: SD, p: str, epsilon: float = 1e-5) -> torch.Tensor:
    """EuclideanCodebook.embedding (quantization/core_vq.py:142-150)."""
    return sd[f"{p}._codebook.embedding_sum"] / sd[f"{p}._codebook.cluster_usage"].clamp(min=epsilon)[:, None]


def nearest_code(x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """EuclideanCodebook._quantize (quantization/core_vq.py:179-185): cdist + argmin, [N,D]->[N] int64."""
    return torch.cdist(x[None], emb[None], p=2)[0].argmin(dim=-1)


def rvq_levels_encode(sd: SD, p: str, n_q: int, x: torch.Tensor) -> torch.Tensor:
    """ResidualVectorQuantization.encode (quantization/core_vq.py:365-376). x [B,D,T] -> [n_q,B,T]."""
    residual = x.transpose(1, 2)  # "b d n -> b n d"
    out = []
    for j in range(n_q):
        emb = codebook(sd, f"{p}.vq.layers.{j}")
        idx = nearest_code(residual.reshape(-1, residual.shape[-1]), emb).view(residual.shape[:-1])
        residual = residual - F.embedding(idx, emb)
        out.append(idx)
    return torch.stack(out)


def rvq_levels_decode(sd: SD, p: str, codes: torch.Tensor) -> torch.Tensor:
    """ResidualVectorQuantization.decode (quantization/core_vq.py:378-384). codes [n_q,B,T] -> [B,D,T]."""
    q = torch.zeros([1])[0]
    for j, c in enumerate(codes):
        q = q + F.embedding(c, codebook(sd, f"{p}.vq.layers.{j}"))
    return q.transpose(1, 2)


def rvq_encode(sd: SD, cfg: MimiConfig, z: torch.Tensor, prefix: str = "quantizer") -> torch.Tensor:
    """SplitResidualVectorQuantizer.encode (quantization/vq.py:305-315, 134-147). z [B,512,T] -> [B,K,T] int64."""
    if z.shape[-1] == 0:
        return torch.empty((z.shape[0], cfg.rvq_layers, 0), dtype=torch.int64)
    first = rvq_levels_encode(sd, f"{prefix}.rvq_first", cfg.n_q_semantic,
                              F.conv1d(z, sd[f"{prefix}.rvq_first.input_proj.weight"]))
    rest = rvq_levels_encode(sd, f"{prefix}.rvq_rest", cfg.rvq_layers - cfg.n_q_semantic,
                             F.conv1d(z, sd[f"{prefix}.rvq_rest.input_proj.weight"]))
    return torch.cat([first.transpose(0, 1), rest.transpose(0, 1)], dim=1)


def rvq_decode(sd: SD, cfg: MimiConfig, codes: torch.Tensor, prefix: str = "quantizer") -> torch.Tensor:
    """SplitResidualVectorQuantizer.decode (quantization/vq.py:317-323, 149-155)."""
    ns = cfg.n_q_semantic
    q = F.conv1d(rvq_levels_decode(sd, f"{prefix}.rvq_first", codes[:, :ns].transpose(0, 1)),
                 sd[f"{prefix}.rvq_first.output_proj.weight"])
    if codes.shape[1] > ns:
        q = q + F.conv1d(rvq_levels_decode(sd, f"{prefix}.rvq_rest", codes[:, ns:].transpose(0, 1)),
                         sd[f"{prefix}.rvq_rest.output_proj.weight"])
    return q


# --------------------------------------------------------------------------
# model (models/MimiCodec.py:93-110)
# --------------------------------------------------------------------------

def encode_latent(sd: SD, cfg: MimiConfig, audio: torch.Tensor) -> torch.Tensor:
    z = seanet_encoder(sd, cfg, audio)
    z = projected_transformer(sd, "encoder_transformer", cfg, z)
    return downsample(sd, cfg, z)


def encode(sd: SD, cfg: MimiConfig, audio: torch.Tensor) -> torch.Tensor:
    """MimiCodec.encode: audio [B,1,T] fp32 -> codes [B,K,ceil(T/hop)] int64."""
    return rvq_encode(sd, cfg, encode_latent(sd, cfg, audio))


def decode(sd: SD, cfg: MimiConfig, codes: torch.Tensor) -> torch.Tensor:
    """MimiCodec.decode: codes [B,K,F] int64 -> wav [B,1,F*hop] fp32 (untrimmed)."""
    z = rvq_decode(sd, cfg, codes)
    z = upsample(sd, cfg, z)
    z = projected_transformer(sd, "decoder_transformer", cfg, z)
    return seanet_decoder(sd, cfg, z)
