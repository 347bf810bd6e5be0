"""Seeded synthetic MimiCodec weights (random-init, reference ``state_dict`` keys).

There is no network and no checkpoint on the build/GPU boxes, so every test, fixture and
benchmark runs on weights generated here.  The generator is pure CPU torch with an explicit
``torch.Generator`` so the very same numbers are produced in the build container (where the
reference is imported to make golden fixtures) and on the GPU box.

Recipe: Xavier-uniform conv / linear weights as in the reference's own
tests (``MLLM_v2/moshi/modules/conv_test.py:53-60``), small random biases, LayerScale 0.01,
LayerNorm affine near identity, and *independent* Gaussian codebooks per RVQ level whose scale
follows the residual at that level (data-sampled codebooks create exact fp32 ties and are useless
for parity).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

import os

import numpy as np

# Per-level centre (mean residual vector, [K, D]) and scale (std of the centred residual, [K]) of
# the default-config codebooks: measured offline on a calibration clip by
# tests/golden/calibrate_codebooks.py (CPU oracle) and stored as data next to this file.  A
# codebook is  centre_l + scale_l * N_l  with N_l ~ randn drawn from the seeded generator, so the
# codes are spread over many entries and top-2 gaps are well conditioned.
_CALIB_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_calib.npz")
_FALLBACK_STD = 0.08


def load_codebook_calibration():
    if not os.path.exists(_CALIB_FILE):
        return None
    d = np.load(_CALIB_FILE)
    return {"seed": int(d["seed"]), "center": torch.from_numpy(d["center"]), "scale": torch.from_numpy(d["scale"])}


def _xavier(gen: torch.Generator, *shape: int) -> torch.Tensor:
    """nn.init.xavier_uniform_ semantics for conv / linear weight shapes."""
    receptive = 1
    for s in shape[2:]:
        receptive *= s
    fan_in, fan_out = shape[1] * receptive, shape[0] * receptive
    bound = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(*shape, generator=gen) * 2 - 1) * bound


def mimi_state_dict(seed: int = 0, *, n_filters: int = 64, ratios: Optional[List[int]] = None,
                    latent_dim: int = 512, kernel_size: int = 7, last_kernel_size: int = 3,
                    residual_kernel_size: int = 3, compress: int = 2, codebook_size: int = 2048,
                    codebook_dim: int = 256, rvq_layers: int = 8, num_layers: int = 8,
                    dim_feedforward: int = 2048, layer_scale: float = 0.01,
                    resample_stride: int = 2, calib: Optional[dict] = "default") -> Dict[str, torch.Tensor]:
    """Returns a CPU fp32 state_dict with exactly the keys of the reference ``MimiCodec``
    (``MLLM_v2/tools/tokenizer/MimiCodec/model/models/MimiCodec.py:26-72``), minus the
    training-only ``semantic_mapping_layer``."""
    ratios = list(ratios) if ratios is not None else [8, 6, 5, 4]
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(key: str, cout: int, cin: int, k: int, bias: bool = True):
        sd[f"{key}.weight"] = _xavier(g, cout, cin, k)
        if bias:
            sd[f"{key}.bias"] = 0.02 * torch.randn(cout, generator=g)

    def convtr(key: str, cin: int, cout: int, k: int):
        sd[f"{key}.weight"] = _xavier(g, cin, cout, k)
        sd[f"{key}.bias"] = 0.02 * torch.randn(cout, generator=g)

    def resblock(prefix: str, dim: int):
        conv(f"{prefix}.block.1.conv.conv", dim // compress, dim, residual_kernel_size)
        conv(f"{prefix}.block.3.conv.conv", dim, dim // compress, 1)

    # SEANet encoder (modules/seanet.py:184-235)
    i, mult = 0, 1
    conv(f"encoder.model.{i}.conv.conv", n_filters, 1, kernel_size)
    i += 1
    for r in reversed(ratios):
        resblock(f"encoder.model.{i}", mult * n_filters)
        i += 2
        conv(f"encoder.model.{i}.conv.conv", mult * n_filters * 2, mult * n_filters, 2 * r)
        i += 1
        mult *= 2
    i += 1
    conv(f"encoder.model.{i}.conv.conv", latent_dim, mult * n_filters, last_kernel_size)

    # SEANet decoder (modules/seanet.py:317-390)
    i, mult = 0, 2 ** len(ratios)
    conv(f"decoder.model.{i}.conv.conv", mult * n_filters, latent_dim, kernel_size)
    i += 1
    for r in ratios:
        i += 1
        convtr(f"decoder.model.{i}.convtr.convtr", mult * n_filters, mult * n_filters // 2, 2 * r)
        i += 1
        resblock(f"decoder.model.{i}", mult * n_filters // 2)
        i += 1
        mult //= 2
    i += 1
    conv(f"decoder.model.{i}.conv.conv", 1, n_filters, last_kernel_size)

    # resampling (modules/resample.py)
    conv("downsample.conv.conv.conv", latent_dim, latent_dim, 2 * resample_stride, bias=False)
    sd["upsample.convtr.convtr.convtr.weight"] = _xavier(g, latent_dim, 1, 2 * resample_stride)

    # transformers (modules/transformer.py:434-750)
    for name in ("encoder_transformer", "decoder_transformer"):
        for l in range(num_layers):
            p = f"{name}.transformer.layers.{l}"
            sd[f"{p}.self_attn.in_proj_weight"] = _xavier(g, 3 * latent_dim, latent_dim)
            sd[f"{p}.self_attn.out_proj.weight"] = _xavier(g, latent_dim, latent_dim)
            for n in ("norm1", "norm2"):
                sd[f"{p}.{n}.weight"] = 1.0 + 0.05 * torch.randn(latent_dim, generator=g)
                sd[f"{p}.{n}.bias"] = 0.05 * torch.randn(latent_dim, generator=g)
            sd[f"{p}.linear1.weight"] = _xavier(g, dim_feedforward, latent_dim)
            sd[f"{p}.linear2.weight"] = _xavier(g, latent_dim, dim_feedforward)
            sd[f"{p}.layer_scale_1.scale"] = torch.full((latent_dim,), layer_scale)
            sd[f"{p}.layer_scale_2.scale"] = torch.full((latent_dim,), layer_scale)

    # split RVQ (quantization/vq.py:200-246, core_vq.py:104-127)
    if calib == "default":
        calib = load_codebook_calibration()
        if calib is not None and (calib["seed"] != seed or tuple(calib["center"].shape) != (rvq_layers, codebook_dim)):
            calib = None  # calibration was measured for another seed / shape

    def rvq(prefix: str, n_q: int, first_level: int):
        conv(f"{prefix}.input_proj", codebook_dim, latent_dim, 1, bias=False)
        conv(f"{prefix}.output_proj", latent_dim, codebook_dim, 1, bias=False)
        for j in range(n_q):
            p = f"{prefix}.vq.layers.{j}._codebook"
            usage = 0.5 + 1.5 * torch.rand(codebook_size, generator=g)
            noise = torch.randn(codebook_size, codebook_dim, generator=g)
            if calib is not None:
                emb = calib["center"][first_level + j][None, :] + calib["scale"][first_level + j] * noise
            else:
                emb = _FALLBACK_STD * noise
            sd[f"{p}._initialized"] = torch.ones(1)
            sd[f"{p}.cluster_usage"] = usage
            sd[f"{p}.embedding_sum"] = emb * usage[:, None]

    rvq("quantizer.rvq_first", 1, 0)
    rvq("quantizer.rvq_rest", rvq_layers - 1, 1)
    return sd


def synth_audio(batch: int, samples: int, seed: int = 0) -> torch.Tensor:
    """0.1 * randn(B, 1, T) fp32 -- the synthetic 24 kHz mono input of BASELINE.md section 3."""
    g = torch.Generator().manual_seed(1000 + seed)
    return 0.1 * torch.randn(batch, 1, samples, generator=g)


# ----------------------------------------------------------------------------------------------------------------------
# speech-text LM (LMModel of MLLM_v2/models/model.py:98-225, Moshi-style temporal + depth transformer)
# ----------------------------------------------------------------------------------------------------------------------

LM_MOSHI_7B = dict(dim=4096, text_card=32000, existing_text_padding_id=3, n_q=16, dep_q=8, card=2048, num_heads=32,
                   num_layers=32, hidden_scale=4.125, context=3000, max_period=10000.0, depformer_dim=1024,
                   depformer_dim_feedforward=4224, depformer_num_heads=16, depformer_num_layers=6,
                   delays=[0, 0, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1])   # MLLM_v2/moshi/models/loaders.py:68-98

# small config for parity fixtures (the imported reference runs it on CPU in seconds); head dims 64 / 64 like the
# depth transformer of the real model, ring capacity 10 so that the temporal KV ring wraps within a short test
LM_TINY = dict(dim=256, text_card=50, existing_text_padding_id=3, n_q=4, dep_q=2, card=32, num_heads=4, num_layers=2,
               hidden_scale=4.125, context=10, max_period=10000.0, depformer_dim=128, depformer_dim_feedforward=528,
               depformer_num_heads=2, depformer_num_layers=2, delays=[0, 0, 1, 0, 1])

# the tiny LM with Moshi's stream layout (16 audio streams of 2048 codes, 8 of them generated, the delay pattern of
# loaders.py:97): small enough for tests, shaped so that it plugs into the real Mimi (8 codebooks x 2048) end to end
LM_TINY_16Q = dict(LM_TINY, card=2048, n_q=16, dep_q=8, delays=[0, 0, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1])


def _gating_hidden(dim: int, dim_feedforward: int) -> int:
    return (21 * dim) // 8 if dim_feedforward == 4 * dim else (2 * dim_feedforward) // 3


def lm_state_dict(cfg: dict, seed: int = 0, device: str = "cpu", dtype: torch.dtype = torch.bfloat16) -> Dict[str, torch.Tensor]:
    """Random-init LMModel weights with the reference's ``state_dict`` keys.  Linear weights ~ U(+-sqrt(3/fan_in)),
    embeddings ~ 0.5*N(0,1), norm gains 1 + 0.1*N(0,1); stored in ``dtype`` (bf16 like the released checkpoints).
    On ``cpu`` the values are bit-reproducible across machines (parity fixtures); on ``cuda`` they are generated on the
    device (the 7.7 B-parameter benchmark model never touches host memory)."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def lin(key: str, out_f: int, in_f: int):
        b = math.sqrt(3.0 / in_f)
        sd[key] = ((torch.rand(out_f, in_f, generator=g, device=device) * 2 - 1) * b).to(dtype)

    def emb(key: str, n: int, d: int):
        sd[key] = (0.5 * torch.randn(n, d, generator=g, device=device)).to(dtype)

    def alpha(key: str, d: int):
        sd[key] = (1.0 + 0.1 * torch.randn(1, 1, d, generator=g, device=device)).to(dtype)

    dim, ddim, dep_q = cfg["dim"], cfg["depformer_dim"], cfg["dep_q"]
    for i in range(cfg["n_q"]):
        emb(f"emb.{i}.weight", cfg["card"] + 1, dim)
    emb("text_emb.weight", cfg["text_card"] + 1, dim)
    lin("text_linear.weight", cfg["text_card"] + (1 if cfg.get("existing_text_padding_id") is None else 0), dim)
    hid = _gating_hidden(dim, int(cfg["hidden_scale"] * dim))
    for l in range(cfg["num_layers"]):
        p = f"transformer.layers.{l}"
        lin(f"{p}.self_attn.in_proj_weight", 3 * dim, dim)
        lin(f"{p}.self_attn.out_proj.weight", dim, dim)
        alpha(f"{p}.norm1.alpha", dim)
        alpha(f"{p}.norm2.alpha", dim)
        lin(f"{p}.gating.linear_in.weight", 2 * hid, dim)
        lin(f"{p}.gating.linear_out.weight", dim, hid)
    alpha("out_norm.alpha", dim)
    for k in range(dep_q):
        lin(f"depformer_in.{k}.weight", ddim, dim)
    for k in range(dep_q - 1):
        emb(f"depformer_emb.{k}.weight", cfg["card"] + 1, ddim)
    emb("depformer_text_emb.weight", cfg["text_card"] + 1, ddim)
    dhid = _gating_hidden(ddim, cfg["depformer_dim_feedforward"])
    for l in range(cfg["depformer_num_layers"]):
        p = f"depformer.layers.{l}"
        lin(f"{p}.self_attn.in_proj_weight", dep_q * 3 * ddim, ddim)
        lin(f"{p}.self_attn.out_proj.weight", dep_q * ddim, ddim)
        alpha(f"{p}.norm1.alpha", ddim)
        alpha(f"{p}.norm2.alpha", ddim)
        for k in range(dep_q):
            lin(f"{p}.gating.{k}.linear_in.weight", 2 * dhid, ddim)
            lin(f"{p}.gating.{k}.linear_out.weight", ddim, dhid)
    for k in range(dep_q):
        lin(f"linears.{k}.weight", cfg["card"], ddim)
    return sd


# ---- litgpt-style backbone (MLLM_v2/models/llama_streaming.py) -------------------------------------------------------
# Keys = fields of models/config.py:Config + models/llama_streaming.py:Config (:447-489).
# BASELINE.json configs[4]: Qwen-1.5-0.5B-shaped backbone + LoRA r=32 alpha=16 on q,k,v,proj,mlp,head (egs recipe run.sh),
# codecformer 1024 / 16 heads / 6 layers / ff 4224.
GPT_QWEN_0_5B = dict(block_size=4096, n_layer=24, n_embd=1024, n_head=16, n_query_groups=16, padded_vocab_size=151936,
                     rotary_percentage=1.0, rope_base=1000000, norm_class_name="RMSNorm", norm_eps=1e-6, bias=False,
                     lm_head_bias=False, parallel_residual=False, mlp_class_name="LLaMAMLP", intermediate_size=2816,
                     lora_r=32, lora_alpha=16, lora_query=True, lora_key=True, lora_value=True, lora_projection=True,
                     lora_mlp=True, lora_head=True, audio_card=2050, codecformer_dim=1024, n_q=8, dep_q=8,
                     codecformer_heads=16, codecformer_layers=6, codecformer_dim_feedforward=4224, context=3000)
# grouped-query attention, partial RoPE, LoRA on q and v only (exercises the lora_ind scatter), no biases
GPT_TINY_GQA = dict(block_size=64, n_layer=2, n_embd=256, n_head=4, n_query_groups=2, padded_vocab_size=320,
                    rotary_percentage=0.5, rope_base=10000, norm_class_name="RMSNorm", norm_eps=1e-5, bias=False,
                    lm_head_bias=False, parallel_residual=False, mlp_class_name="LLaMAMLP", intermediate_size=384,
                    lora_r=4, lora_alpha=8, lora_query=True, lora_key=False, lora_value=True, lora_projection=True,
                    lora_mlp=True, lora_head=True, audio_card=32, codecformer_dim=128, n_q=4, dep_q=3,
                    codecformer_heads=2, codecformer_layers=2, codecformer_dim_feedforward=528, context=10)
# multi-head attention, full RoPE, LoRA on q, k and v (the zero_pad early-return quirk), biases everywhere
GPT_TINY_MHA = dict(block_size=64, n_layer=2, n_embd=256, n_head=4, n_query_groups=4, padded_vocab_size=320,
                    rotary_percentage=1.0, rope_base=10000, norm_class_name="RMSNorm", norm_eps=1e-5, bias=True,
                    lm_head_bias=True, parallel_residual=False, mlp_class_name="LLaMAMLP", intermediate_size=384,
                    lora_r=4, lora_alpha=8, lora_query=True, lora_key=True, lora_value=True, lora_projection=False,
                    lora_mlp=False, lora_head=False, audio_card=32, codecformer_dim=128, n_q=4, dep_q=3,
                    codecformer_heads=2, codecformer_layers=2, codecformer_dim_feedforward=528, context=10,
                    codecformer_bias_proj=True)

# the smallest config the reference's OFFLINE generation loop runs on unchanged: infer_no_streaming.py hard-codes 8 depth steps,
# text ids 128002 / 128003 / 151655 and the 2048 / 2049 audio-id rules, so the vocabularies keep their real sizes
GPT_GEN_TINY = dict(block_size=64, n_layer=2, n_embd=64, n_head=2, n_query_groups=2, padded_vocab_size=151936,
                    rotary_percentage=1.0, rope_base=10000, norm_class_name="RMSNorm", norm_eps=1e-5, bias=False,
                    lm_head_bias=False, parallel_residual=False, mlp_class_name="LLaMAMLP", intermediate_size=128,
                    lora_r=0, audio_card=2050, codecformer_dim=64, n_q=8, dep_q=8, codecformer_heads=2, codecformer_layers=2,
                    codecformer_dim_feedforward=132, context=32)


def gpt_state_dict(cfg: dict, seed: int = 0, device: str = "cpu", dtype: torch.dtype = torch.bfloat16,
                   lora: bool = True) -> Dict[str, torch.Tensor]:
    """Random-init weights with the ``state_dict`` keys of ``models.llama_streaming.GPT`` (same distributions as
    ``lm_state_dict``; LoRA A ~ U(+-sqrt(3/in)), B ~ U(+-sqrt(3/r)) -- non-zero so that the adapters matter).  ``lora=False``
    leaves the adapters out (a checkpoint that was merged before saving)."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def lin(key: str, out_f: int, in_f: int, bias: bool = False, scale: float = 1.0):
        b = math.sqrt(3.0 / in_f) * scale
        sd[f"{key}.weight" if not key.endswith(("lora_A", "lora_B")) else key] = \
            ((torch.rand(out_f, in_f, generator=g, device=device) * 2 - 1) * b).to(dtype)
        if bias:
            sd[f"{key}.bias"] = (0.1 * torch.randn(out_f, generator=g, device=device)).to(dtype)

    def emb(key: str, n: int, d: int):
        sd[key] = (0.5 * torch.randn(n, d, generator=g, device=device)).to(dtype)

    def gain(key: str, shape):
        sd[key] = (1.0 + 0.1 * torch.randn(*shape, generator=g, device=device)).to(dtype)

    E, H, G = cfg["n_embd"], cfg["n_head"], cfg["n_query_groups"]
    hs = cfg.get("head_size") or E // H
    r = cfg.get("lora_r", 0) if lora else 0
    V, I = cfg["padded_vocab_size"], cfg["intermediate_size"]

    def lora_linear(key: str, out_f: int, in_f: int, bias: bool, enabled: bool):
        lin(f"{key}.linear", out_f, in_f, bias)
        if r > 0 and enabled:
            lin(f"{key}.lora_A", r, in_f)
            lin(f"{key}.lora_B", out_f, r, scale=0.5)

    lora_linear("lm_head", V, E, cfg.get("lm_head_bias", False), cfg.get("lora_head", False))
    emb("transformer.wte.weight", V, E)
    for l in range(cfg["n_layer"]):
        p = f"transformer.h.{l}"
        gain(f"{p}.norm_1.weight", (E,))
        lin(f"{p}.attn.attn.linear", (H + 2 * G) * hs, E, cfg.get("bias", False))
        en = (cfg.get("lora_query", False), cfg.get("lora_key", False), cfg.get("lora_value", False))
        if r > 0 and any(en):
            lin(f"{p}.attn.attn.lora_A", r * sum(en), E)
            lin(f"{p}.attn.attn.lora_B", hs * (H * en[0] + G * en[1] + G * en[2]), r, scale=0.5)
        lora_linear(f"{p}.attn.proj", E, hs * H, cfg.get("bias", False), cfg.get("lora_projection", False))
        gain(f"{p}.norm_2.weight", (E,))
        for name, (o, i) in (("fc_1", (I, E)), ("fc_2", (I, E)), ("proj", (E, I))):
            lora_linear(f"{p}.mlp.{name}", o, i, cfg.get("bias", False), cfg.get("lora_mlp", False))
    gain("transformer.ln_f.weight", (E,))
    card, ddim, dep_q = cfg["audio_card"], cfg["codecformer_dim"], cfg["dep_q"]
    for k in range(cfg["n_q"]):
        emb(f"input_emb.{k}.weight", card + 1, E)
    for k in range(dep_q):
        lin(f"codecformer_in.{k}", ddim, E)
    for k in range(dep_q - 1):
        emb(f"codecformer_emb.{k}.weight", card + 1, ddim)
    emb("codecformer_text_emb.weight", V, ddim)
    dhid = _gating_hidden(ddim, cfg["codecformer_dim_feedforward"])
    for l in range(cfg["codecformer_layers"]):
        p = f"codecformer.layers.{l}"
        sd[f"{p}.self_attn.in_proj_weight"] = ((torch.rand(dep_q * 3 * ddim, ddim, generator=g, device=device) * 2 - 1)
                                               * math.sqrt(3.0 / ddim)).to(dtype)
        lin(f"{p}.self_attn.out_proj", dep_q * ddim, ddim)
        gain(f"{p}.norm1.alpha", (1, 1, ddim))
        gain(f"{p}.norm2.alpha", (1, 1, ddim))
        for k in range(dep_q):
            lin(f"{p}.gating.{k}.linear_in", 2 * dhid, ddim)
            lin(f"{p}.gating.{k}.linear_out", ddim, dhid)
    for k in range(dep_q):
        lin(f"audio_linears.{k}", card, ddim, cfg.get("codecformer_bias_proj", False))
    return sd
