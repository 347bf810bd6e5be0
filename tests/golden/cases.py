"""Seeded inputs shared by tests/golden/make_golden.py (reference side) and the tests (oracle / HIP side).

Everything here is deterministic CPU torch, so both sides see identical inputs and weights without
any of them being stored.
"""
import zlib

import torch

from rstnet_amd import synth

MIMI_SEED = 0
TRANSFORMER_LAYER_SCALE = 0.25  # large enough that attention / FFN errors are visible in the output

# name -> (batch, samples, audio seed).  cfg1 = BASELINE.json configs[0]; ragged = length that is not a
# multiple of the 1920-sample hop (ceil(T/hop) frames, decode returns F*1920 samples).
MIMI_E2E = {
    "cfg1": (1, 24000, 0),
    "ragged": (2, 30001, 1),
}

# (batch, frames, audio seed) of the long streamed clip: 150 frames = 300 transformer positions, i.e. the 250-slot KV rings of
# the codec's transformers wrap at frame 125 (SURVEY Q1 / fixture F4); the last MIMI_STREAM_LONG_TAIL frames of waveform are stored
MIMI_STREAM_LONG = (2, 150, 7)
MIMI_STREAM_LONG_TAIL = 20

# name -> (B, Cin, Cout, T, K, stride); first four = MLLM_v2/moshi/modules/conv_test.py:11-28, the strided
# ones are the SEANet encoder shapes at reduced width.
CONV_CASES = {
    "small1": (3, 4, 5, 10, 6, 1),
    "small2": (4, 5, 6, 10, 7, 1),
    "small3": (5, 6, 7, 10, 2, 1),
    "large1": (1, 512, 512, 256, 7, 1),
    "stride4": (2, 16, 32, 103, 8, 4),
    "stride5": (2, 32, 64, 57, 10, 5),
    "stride6": (1, 64, 128, 40, 12, 6),
    "stride8": (2, 128, 256, 35, 16, 8),
}
# name -> (B, Cin, Cout, T, K, stride); first three = conv_test.py:30-48 (its fourth set, 512 -> 512 channels x 256 steps at K = 7, S = 2, is
# small2's geometry at 1 MB of output: dropped from the fixtures in round 6 -- the 512-wide layers are covered by CONV_CASES["large1"],
# RESBLOCK_CASES["dim512"] and the end-to-end fixtures)
CONVTR_CASES = {
    "small1": (3, 4, 5, 10, 6, 1),
    "small2": (4, 5, 6, 10, 7, 2),
    "small3": (5, 6, 7, 10, 4, 3),
    "stride8": (2, 256, 128, 9, 16, 8),
    "stride6": (1, 128, 64, 21, 12, 6),
    "stride5": (2, 64, 32, 33, 10, 5),
    "stride4": (2, 32, 16, 50, 8, 4),
}
# name -> (B, dim, T)   (seanet_test.py:111-160 shapes + SEANet widths)
RESBLOCK_CASES = {
    "dim8": (2, 8, 40),
    "dim64": (2, 64, 300),
    "dim512": (1, 512, 50),
}


def _seed(name: str) -> int:
    return 41 + (zlib.crc32(name.encode()) & 0xFFFF)


def layer_tensors(name, wshape, nbias, xshape):
    """Xavier-uniform weight (as conv_test.py:53-60, generator seeded from the case name), small random
    bias and torch.rand input."""
    g = torch.Generator().manual_seed(_seed(name))
    w = synth._xavier(g, *wshape)
    b = 0.1 * torch.randn(nbias, generator=g)
    x = torch.rand(*xshape, generator=g)
    return w, b, x


def rvq_latent(sd, n_batch: int = 4, n_frames: int = 250):
    """A seeded latent [B,512,F] with the statistics of the real encoder output: the calibrated centre of
    level 0 pulled back through the (pseudo-inverse of the) input projection plus white noise."""
    g = torch.Generator().manual_seed(77)
    calib = synth.load_codebook_calibration()
    w = sd["quantizer.rvq_first.input_proj.weight"][:, :, 0]  # [256, 512]
    centre = torch.linalg.lstsq(w, calib["center"][0][:, None]).solution[:, 0]  # [512]
    z = centre[None, :, None] + 0.07 * torch.randn(n_batch, 512, n_frames, generator=g)
    return z


def transformer_input(batch: int = 1, frames: int = 300):
    g = torch.Generator().manual_seed(78)
    return torch.randn(batch, 512, frames, generator=g)


# ---- the codec's streaming transformer at other head dims (tests/golden/transformer_dims.npz) -----------------------------------
# name: (d_model E, heads H, feed-forward F, layers L, ring = context, streams B, positions per step)
TRANSFORMER_DIMS = {
    "d16": (64, 4, 128, 2, 12, 2, (2, 1, 2, 2, 1) * 5),            # head dim 16 (steps of <= 4 rows: the launch-per-op attention has no 16-dim form)
    "d32": (128, 4, 256, 2, 20, 1, (3, 4, 1, 2) * 5),              # 32
    "d128": (256, 2, 512, 2, 24, 2, (2,) * 20),                    # 128
    "d256": (512, 2, 512, 1, 12, 1, (4, 1) * 8),                   # 256 (<= 4 rows per step as well)
}
TRANSFORMER_DIMS_LAYER_SCALE = 0.3


def transformer_dims_state(name: str, prefix: str = "tr"):
    """Seeded weights of a ProjectedTransformer whose input / output width equals d_model (no projections), keyed as the module's
    state_dict is (`transformer.layers.N...`) under `prefix`."""
    E, H, F, L, _, _, _ = TRANSFORMER_DIMS[name]
    g = torch.Generator().manual_seed(1000 + E + H)
    sd = {}
    for l in range(L):
        p = f"{prefix}.transformer.layers.{l}"
        sd[f"{p}.norm1.weight"] = 1 + 0.1 * torch.randn(E, generator=g)
        sd[f"{p}.norm1.bias"] = 0.1 * torch.randn(E, generator=g)
        sd[f"{p}.norm2.weight"] = 1 + 0.1 * torch.randn(E, generator=g)
        sd[f"{p}.norm2.bias"] = 0.1 * torch.randn(E, generator=g)
        sd[f"{p}.self_attn.in_proj_weight"] = torch.randn(3 * E, E, generator=g) / E ** 0.5
        sd[f"{p}.self_attn.out_proj.weight"] = torch.randn(E, E, generator=g) / E ** 0.5
        sd[f"{p}.linear1.weight"] = torch.randn(F, E, generator=g) / E ** 0.5
        sd[f"{p}.linear2.weight"] = torch.randn(E, F, generator=g) / F ** 0.5
        sd[f"{p}.layer_scale_1.scale"] = torch.full((E,), TRANSFORMER_DIMS_LAYER_SCALE)
        sd[f"{p}.layer_scale_2.scale"] = torch.full((E,), TRANSFORMER_DIMS_LAYER_SCALE)
    return sd


def transformer_dims_input(name: str):
    """[B, E, sum(positions)] in conv layout; the steps cut it along the last axis."""
    E, _, _, _, _, B, chunks = TRANSFORMER_DIMS[name]
    g = torch.Generator().manual_seed(2000 + E)
    return torch.randn(B, E, sum(chunks), generator=g)


# ---- LM (tiny config, CPU-feasible for the imported reference) ------------------------------------------------------
LM_SEED = 3
LM_STEPS = 14       # > context (10) so the temporal ring wraps, and > max_delay
LM_BATCH = 2


LM_SAMPLING = dict(seed=21, temp=0.8, temp_text=0.7, top_k=8, top_k_text=6)   # top-k below the tiny cardinalities (32 / 50)


def lm_user_tokens(cfg: dict, steps: int = LM_STEPS, batch: int = LM_BATCH):
    """Tokens of the "other" stream fed to LMGen.step: [steps][B, n_q - dep_q, 1]."""
    g = torch.Generator().manual_seed(79)
    return torch.randint(0, cfg["card"], (steps, batch, cfg["n_q"] - cfg["dep_q"], 1), generator=g)


# ---- litgpt-style backbone (tiny configs) ---------------------------------------------------------------------------
GPT_SEED = 5
GPT_STEPS = 14      # > context (10): the ring wraps
GPT_BATCH = 2
GPT_T_FULL = 9      # non-streaming forward_global length (< context, so the context mask is also exercised by GPT_STEPS only)


def gpt_tokens(cfg: dict, steps: int = GPT_STEPS, batch: int = GPT_BATCH):
    """Input frames [B, n_q + 1, steps]: row 0 text ids, rows 1.. audio ids (one column is the -1 'no input' id)."""
    g = torch.Generator().manual_seed(80)
    text = torch.randint(0, cfg["padded_vocab_size"], (batch, 1, steps), generator=g)
    audio = torch.randint(0, cfg["audio_card"] + 1, (batch, cfg["n_q"], steps), generator=g)
    audio[0, 1, 2] = -1
    return torch.cat([text, audio], 1)


# ---- samplers (utils/sampling.py): name -> (function, B, V, top_k, temp, seed)  ----------------------------------------
# "hot" ids: logits forced up so that the id blanking of sample_token_audio (>= 2049) / _2048 (>= 2048) decides the draw.
# No case puts the top-k boundary inside a run of EQUAL logits: which of the tied ids torch.topk keeps is unspecified (and
# differs between the CPU and GPU implementations of the reference's own dependency), so there is no reference answer to pin;
# the build's rule (lowest index first) is tested on its own in tests/test_lm_gpu.py.
SAMPLING_CASES = {
    "text": ("sample_token", 3, 32000, 25, 0.7, 1),
    "audio": ("sample_token", 4, 2048, 250, 0.8, 2),
    "audio_blank2049": ("sample_token_audio", 4, 2050, 250, 0.8, 3),
    "audio_blank2048": ("sample_token_audio_2048", 4, 2050, 200, 1.0, 4),
    "qwen_vocab": ("sample_token", 2, 151936, 25, 0.7, 6),
}


def sampling_logits(name: str) -> torch.Tensor:
    """fp32 logits ``[B, 1, 1, V]`` of a sampling case."""
    fn, B, V, k, temp, seed = SAMPLING_CASES[name]
    g = torch.Generator().manual_seed(900 + seed)
    lg = 3.0 * torch.randn(B, 1, 1, V, generator=g)
    if name.startswith("audio_blank"):
        lg[..., 2048:] += 12.0       # would win every draw if it were not blanked
    return lg


# nucleus sampling (sample_token(..., top_p=p) -> sample_top_p, utils/sampling.py:66-82,96-97): name -> (B, V, top_p, temp, seed, logit
# scale).  The noise is one Exp(1) draw per SORTED vocabulary position ([B, V]); it is not stored but re-drawn from the seed on the
# CPU generator (`sampling_top_p_noise`), exactly as make_golden.py re-draws what the reference's multinomial drew.
SAMPLING_TOP_P_CASES = {
    "audio": (3, 2048, 0.9, 0.8, 21, 1.0),           # nuclei of ~1000 entries
    "text": (2, 32000, 0.8, 0.7, 22, 1.5),           # several thousand
    "qwen_vocab": (2, 151936, 0.6, 0.7, 23, 2.5),    # a few hundred of 151 936
    "tight": (4, 2048, 0.3, 1.0, 24, 3.0),           # a handful
}


def sampling_top_p_logits(name: str) -> torch.Tensor:
    B, V, top_p, temp, seed, scale = SAMPLING_TOP_P_CASES[name]
    g = torch.Generator().manual_seed(930 + seed)
    return scale * torch.randn(B, 1, 1, V, generator=g)


def sampling_top_p_noise(name: str) -> torch.Tensor:
    """The Exp(1) tensor `multinomial` draws for this case: global CPU generator seeded as make_golden.py seeds it (restores the state)."""
    B, V, top_p, temp, seed, scale = SAMPLING_TOP_P_CASES[name]
    state = torch.get_rng_state()
    torch.manual_seed(seed)
    noise = torch.empty(B, V).exponential_(1)
    torch.set_rng_state(state)
    return noise


# ---- reverse_delay (infer_no_streaming.py:311-323): name -> shape; [8, L] as generated, [L, 8] exercises the transpose branch
REVERSE_DELAY_CASES = {"k_major": (8, 13), "t_major": (21, 8), "two_frames": (8, 2)}


def reverse_delay_input(name: str) -> torch.Tensor:
    g = torch.Generator().manual_seed(950 + len(name))
    return torch.randint(0, 2048, REVERSE_DELAY_CASES[name], generator=g)


# ---- the offline generation loop (infer_no_streaming.py:168-308, task TTS -- the only task whose result the reference returns)
GEN_SEED = 7
# name -> (L, text positions, RNG seed of the samplers, temp_text, top_k_text, temp, top_k)
GEN_CASES = {"tts_a": (12, 5, 11, 0.7, 25, 0.8, 250), "tts_b": (15, 4, 12, 1.0, 10, 1.0, 100)}


def gen_sequence(name: str) -> torch.Tensor:
    """[9, L] utterance in the reference's TTS layout: n_text text ids, then text_empty (128002) on row 0; audio rows < 2048."""
    L, n_text = GEN_CASES[name][:2]
    g = torch.Generator().manual_seed(970 + L)
    seq = torch.randint(0, 2048, (9, L), generator=g)
    seq[0, :n_text] = torch.randint(0, 128000, (n_text,), generator=g)
    seq[0, n_text:] = 128002
    return seq
