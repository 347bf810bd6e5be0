"""Pin the CPU oracle (oracle/mimi_oracle.py) against fixtures produced by the real reference
(tests/golden/make_golden.py).  Integer codes must match exactly, floats to 1e-5 relative."""
import os

import numpy as np
import pytest
import torch

from oracle import mimi_oracle as O
from rstnet_amd import synth
from tests.golden import cases
from tests.parity import codes_match_up_to_near_ties

G = os.path.join(os.path.dirname(__file__), "golden")


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def mimi_sd():
    return synth.mimi_state_dict(cases.MIMI_SEED)


@pytest.mark.parametrize("name", list(cases.CONV_CASES))
def test_conv_matches_reference(name):
    B, cin, cout, T, K, S = cases.CONV_CASES[name]
    w, b, x = cases.layer_tensors(name, (cout, cin, K), cout, (B, cin, T))
    ref = torch.from_numpy(np.load(os.path.join(G, "layers.npz"))[f"conv.{name}"])
    y = O.causal_conv1d(x, w, b, stride=S)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < 1e-6


@pytest.mark.parametrize("name", list(cases.CONVTR_CASES))
def test_convtr_matches_reference(name):
    B, cin, cout, T, K, S = cases.CONVTR_CASES[name]
    w, b, x = cases.layer_tensors(name, (cin, cout, K), cout, (B, cin, T))
    ref = torch.from_numpy(np.load(os.path.join(G, "layers.npz"))[f"convtr.{name}"])
    y = O.causal_convtr1d(x, w, b, stride=S)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < 1e-6


@pytest.mark.parametrize("name", list(cases.RESBLOCK_CASES))
def test_resblock_matches_reference(name):
    B, dim, T = cases.RESBLOCK_CASES[name]
    w1, b1, x = cases.layer_tensors(name + ".1", (dim // 2, dim, 3), dim // 2, (B, dim, T))
    w2, b2, _ = cases.layer_tensors(name + ".3", (dim, dim // 2, 1), dim, (1, 1, 1))
    sd = {"p.block.1.conv.conv.weight": w1, "p.block.1.conv.conv.bias": b1,
          "p.block.3.conv.conv.weight": w2, "p.block.3.conv.conv.bias": b2}
    ref = torch.from_numpy(np.load(os.path.join(G, "layers.npz"))[f"resblock.{name}"])
    assert rel_err(O.resnet_block(sd, "p", x), ref) < 1e-6


def test_rvq_matches_reference(mimi_sd):
    cfg = O.MimiConfig()
    g = np.load(os.path.join(G, "rvq.npz"))
    z = cases.rvq_latent(mimi_sd)
    codes = O.rvq_encode(mimi_sd, cfg, z)
    assert codes.dtype == torch.int64
    assert torch.equal(codes, torch.from_numpy(g["codes"]).long())
    zq = O.rvq_decode(mimi_sd, cfg, codes[:1])
    assert rel_err(zq, torch.from_numpy(g["zq0"])) < 1e-6


def test_rvq_empty(mimi_sd):
    codes = O.rvq_encode(mimi_sd, O.MimiConfig(), torch.zeros(2, 512, 0))
    assert codes.shape == (2, 8, 0) and codes.dtype == torch.int64


def test_transformer_matches_reference():
    sd = synth.mimi_state_dict(cases.MIMI_SEED, layer_scale=cases.TRANSFORMER_LAYER_SCALE)
    ref = torch.from_numpy(np.load(os.path.join(G, "transformer.npz"))["y"])
    y = O.projected_transformer(sd, "encoder_transformer", O.MimiConfig(), cases.transformer_input())
    assert rel_err(y, ref) < 1e-5


@pytest.mark.parametrize("name", list(cases.TRANSFORMER_DIMS))
def test_transformer_stream_matches_reference_at_other_head_dims(name):
    """tests/golden/transformer_dims.npz: the REFERENCE's ProjectedTransformer streamed in steps of 1-4 positions past the ring wrap at
    head dims 16 / 32 / 128 / 256 (modules/transformer.py:376-423,595-750) vs the oracle's TransformerStream."""
    E, H, F, L, ctx, B, chunks = cases.TRANSFORMER_DIMS[name]
    ref = torch.from_numpy(np.load(os.path.join(G, "transformer_dims.npz"))[name])
    cfg = O.MimiConfig(latent_dim=E, num_heads=H, num_layers=L, context=ctx, dim_feedforward=F)
    ts = O.TransformerStream(cases.transformer_dims_state(name), "tr", cfg, B)
    x = cases.transformer_dims_input(name)
    ys, i = [], 0
    with torch.no_grad():
        for T in chunks:
            ys.append(ts.step(x[:, :, i:i + T]))
            i += T
    y = torch.cat(ys, -1)
    assert tuple(y.shape) == tuple(ref.shape)
    assert float((y - ref).abs().max() / ref.abs().max()) < 2e-5



@pytest.mark.parametrize("name", list(cases.MIMI_E2E))
def test_mimi_encode_decode_matches_reference(mimi_sd, name):
    cfg = O.MimiConfig()
    g = np.load(os.path.join(G, "mimi_e2e.npz"))
    B, T, aseed = cases.MIMI_E2E[name]
    audio = synth.synth_audio(B, T, aseed)
    with torch.no_grad():
        z = O.encode_latent(mimi_sd, cfg, audio)
        codes = O.encode(mimi_sd, cfg, audio)
        wav = O.decode(mimi_sd, cfg, codes)
    ref_codes = torch.from_numpy(g[f"{name}.codes"]).long()
    frames = -(-T // cfg.hop_length)  # ceil
    assert codes.shape == (B, 8, frames) and wav.shape == (B, 1, frames * cfg.hop_length)
    assert rel_err(z, torch.from_numpy(g[f"{name}.latent"])) < 1e-5
    assert torch.equal(codes, ref_codes)
    assert rel_err(wav, torch.from_numpy(g[f"{name}.wav"])) < 1e-5


def test_mimi_long_stream_ring_wrap_matches_reference():
    """The oracle's ring cache (RingKVCache.complete incl. the `delta <= 0` slot) against the real moshi MimiModel
    streamed 150 frames -- 300 transformer positions through 250-slot rings: codes exact, waveform past the wrap 1e-5."""
    cfg = O.MimiConfig()
    mimi_sd = synth.mimi_state_dict(cases.MIMI_SEED, layer_scale=cases.TRANSFORMER_LAYER_SCALE)   # LayerScale 0.25: attention matters
    g = np.load(os.path.join(G, "mimi_stream_long.npz"))
    B, frames, seed = cases.MIMI_STREAM_LONG
    audio = synth.synth_audio(B, 1920 * frames, seed=seed)
    ref_codes = torch.from_numpy(g["codes"]).long()
    with torch.no_grad():
        z = O.encode_latent_streamed(mimi_sd, cfg, audio)
        codes = O.rvq_encode(mimi_sd, cfg, z)
        wav = O.decode_streamed(mimi_sd, cfg, ref_codes)
        batch_codes = O.encode(mimi_sd, cfg, audio)
    assert codes_match_up_to_near_ties(codes, ref_codes, torch.from_numpy(g["rel_gap"])) <= 2
    tail = cases.MIMI_STREAM_LONG_TAIL
    assert rel_err(z[:, :, -tail:], torch.from_numpy(g["latent_tail"])) < 1e-5
    assert rel_err(wav[:, :, :1920 * 4], torch.from_numpy(g["wav_head"])) < 1e-5
    assert rel_err(wav[:, :, -1920 * tail:], torch.from_numpy(g["wav_tail"])) < 1e-5
    # the wrap matters: the non-streaming pass (250-key window, no hidden slot) gives other codes past frame 125
    assert torch.equal(batch_codes[:, :, :120], ref_codes[:, :, :120]) and not torch.equal(batch_codes[:, :, 126:], ref_codes[:, :, 126:])


def test_lm_oracle_matches_reference():
    """oracle/lm_oracle.py vs the imported LMModel / LMGen (greedy) on the tiny config: logits 1e-5, tokens exact."""
    from oracle import lm_oracle as L
    cfg_d = dict(synth.LM_TINY)
    sd = {k: v.float() for k, v in synth.lm_state_dict(cfg_d, cases.LM_SEED).items()}
    cfg = L.LMConfig(**cfg_d)
    g = np.load(os.path.join(G, "lm_tiny.npz"))
    user = cases.lm_user_tokens(cfg_d)
    gen = L.LMGenOracle(sd, cfg, cases.LM_BATCH)
    text_logits, dep_logits, outs = [], [], []
    ft, fd = L.forward_text, L.forward_depformer

    def ft_hook(*a, **k):
        r = ft(*a, **k)
        text_logits.append(r[1][:, 0])
        return r

    def fd_hook(*a, **k):
        r = fd(*a, **k)
        dep_logits.append(r[:, 0])
        return r
    L.forward_text, L.forward_depformer = ft_hook, fd_hook
    try:
        with torch.no_grad():
            for s in range(cases.LM_STEPS):
                o = gen.step(user[s])
                outs.append(torch.full((cases.LM_BATCH, cfg.dep_q + 1, 1), -9, dtype=torch.long) if o is None else o)
    finally:
        L.forward_text, L.forward_depformer = ft, fd
    assert torch.equal(torch.cat(outs, -1), torch.from_numpy(g["tokens"]).long())
    assert rel_err(torch.stack(text_logits), torch.from_numpy(g["text_logits"])) < 1e-5
    assert rel_err(torch.stack(dep_logits), torch.from_numpy(g["dep_logits"])) < 1e-5


def _gpt_cfg(cfg_d):
    from oracle import gpt_oracle as Gp
    keep = {f for f in Gp.GPTConfig.__dataclass_fields__}
    return Gp.GPTConfig(**{k: v for k, v in cfg_d.items() if k in keep})


@pytest.mark.parametrize("name", ["gqa", "mha"])
def test_gpt_oracle_matches_reference(name):
    """oracle/gpt_oracle.py vs the imported models.llama_streaming.GPT on the tiny configs (fixture gpt_tiny.npz):
    non-streaming forward_global with unmerged LoRA, the merged weights, streamed T = 1 steps across the ring wrap,
    forward_codecformer steps and the teacher-forced forward_local -- all within 2e-5 relative."""
    from oracle import gpt_oracle as Gp
    cfg_d = synth.GPT_TINY_GQA if name == "gqa" else synth.GPT_TINY_MHA
    cfg = _gpt_cfg(cfg_d)
    sd = {k: v.float() for k, v in synth.gpt_state_dict(cfg_d, cases.GPT_SEED).items()}
    g = np.load(os.path.join(G, "gpt_tiny.npz"))
    toks = cases.gpt_tokens(cfg_d)
    T, B = cases.GPT_T_FULL, cases.GPT_BATCH
    with torch.no_grad():
        h_full, lg_full = Gp.forward_global(sd, cfg, toks[:, :, :T])
        assert rel_err(h_full, torch.from_numpy(g[f"{name}.full.h"])) < 2e-5
        assert rel_err(lg_full, torch.from_numpy(g[f"{name}.full.logits"])) < 2e-5
        msd = Gp.merged_state(sd, cfg)
        w0 = msd["transformer.h.0.attn.attn.linear.weight"].double()
        assert rel_err(w0.sum(1).float(), torch.from_numpy(g[f"{name}.merged.qkv0_rowsum"])) < 1e-5
        assert rel_err(w0.sum(0).float(), torch.from_numpy(g[f"{name}.merged.qkv0_colsum"])) < 1e-5
        _, lg_m = Gp.forward_global(msd, cfg, toks[:, :, :T], merged=True)
        assert rel_err(lg_m, torch.from_numpy(g[f"{name}.merged.logits"])) < 2e-5
        st = Gp.new_global_state(cfg, B)
        hs, ls, dep = [], [], []
        for t in range(cases.GPT_STEPS):
            h, lg = Gp.forward_global(msd, cfg, toks[:, :, t:t + 1], st, merged=True)
            hs.append(h)
            ls.append(lg)
            cst = Gp.new_codecformer_state(cfg, B)
            for k in range(cfg.dep_q):
                prev = toks[:, 0:1, t:t + 1] if k == 0 else toks[:, k:k + 1, t:t + 1]
                dep.append(Gp.forward_codecformer(msd, cfg, k, prev, h, cst))
        assert rel_err(torch.cat(hs, 1), torch.from_numpy(g[f"{name}.stream.h"])) < 2e-5
        assert rel_err(torch.cat(ls, 1), torch.from_numpy(g[f"{name}.stream.logits"])) < 2e-5
        assert rel_err(torch.stack(dep).view(cases.GPT_STEPS, cfg.dep_q, B, -1), torch.from_numpy(g[f"{name}.stream.dep_logits"])) < 2e-5
        local = Gp.forward_local(sd, cfg, toks[:, 0, :T], toks[:, 1:cfg.dep_q + 1, :T], h_full)
        assert rel_err(local, torch.from_numpy(g[f"{name}.local.logits"])) < 2e-5


_BLANK = {"sample_token": 0, "sample_token_audio": 2049, "sample_token_audio_2048": 2048}


@pytest.mark.parametrize("name", list(cases.SAMPLING_CASES))
def test_sampling_oracle_matches_reference(name):
    """F9: the oracle samplers (oracle/gpt_generate_oracle.py:sample_token = sample_token / sample_token_audio(_2048) with the
    Exp(1) draws passed in, oracle/lm_oracle.py:sample_token) against the tokens the reference's utils/sampling.py returned
    for the stored noise (fixture tests/golden/sampling.npz), incl. id blanking and the 151 936-entry vocabulary; greedy too."""
    from oracle import gpt_generate_oracle as GO
    from oracle import lm_oracle as L
    g = np.load(os.path.join(G, "sampling.npz"))
    fn, B, V, k, temp, seed = cases.SAMPLING_CASES[name]
    lg = cases.sampling_logits(name)
    noise = torch.from_numpy(g[f"{name}.noise"]).view(B, 1, 1, k)
    want = torch.from_numpy(g[f"{name}.tokens"]).long()
    got = GO.sample_token(lg.clone(), True, temp, k, noise, _BLANK[fn])
    assert torch.equal(got, want)
    assert torch.equal(GO.sample_token(lg.clone(), False, temp, k, None, _BLANK[fn]), torch.from_numpy(g[f"{name}.greedy"]).long())
    if not _BLANK[fn]:
        assert torch.equal(L.sample_token(lg.clone(), True, temp, k, noise), want)


@pytest.mark.parametrize("name", list(cases.SAMPLING_TOP_P_CASES))
def test_top_p_oracle_matches_reference(name):
    """Nucleus sampling: oracle/lm_oracle.py:sample_token(top_p=...) against the tokens the reference's sample_token(top_p=...) ->
    sample_top_p (utils/sampling.py:66-82) returned, with the full-width Exp(1) noise its multinomial drew re-drawn from the seed."""
    from oracle import lm_oracle as L
    g = np.load(os.path.join(G, "sampling.npz"))
    B, V, top_p, temp, seed, scale = cases.SAMPLING_TOP_P_CASES[name]
    lg = cases.sampling_top_p_logits(name)
    noise = cases.sampling_top_p_noise(name).view(B, 1, 1, V)
    want = torch.from_numpy(g[f"top_p.{name}.tokens"]).long()
    assert torch.equal(L.sample_token(lg, True, temp, 0, noise, top_p=top_p), want)


@pytest.mark.parametrize("name", list(cases.REVERSE_DELAY_CASES))
def test_reverse_delay_matches_reference(name):
    """Oracle and product `reverse_delay` against the reference function's own outputs (tests/golden/reverse_delay.npz)."""
    from oracle import gpt_generate_oracle as GO
    from rstnet_amd.lm.generate import reverse_delay
    want = torch.from_numpy(np.load(os.path.join(G, "reverse_delay.npz"))[name])
    x = cases.reverse_delay_input(name)
    assert torch.equal(GO.reverse_delay(x.clone()), want)
    assert torch.equal(reverse_delay(x.clone()), want)


@pytest.mark.parametrize("name", list(cases.GEN_CASES))
def test_generation_loop_oracle_matches_reference(name):
    """oracle/gpt_generate_oracle.py:generate against the codes the reference's own `InferenceImp.__call__` returned on the real
    reference GPT (tests/golden/gpt_generate.npz, generated by executing the class from infer_no_streaming.py unchanged), fed
    the Exp(1) noise that run drew.  Pins the loop: prompt split, per-frame global + 8 local passes, the 2049 / 2048 sampler
    choice per step, reverse_delay."""
    from oracle import gpt_generate_oracle as GG
    from oracle import gpt_oracle as Gp
    g = np.load(os.path.join(G, "gpt_generate.npz"))
    L, n_text, seed, temp_text, k_text, temp, k = cases.GEN_CASES[name]
    cfg = dict(synth.GPT_GEN_TINY)
    sd = {kk: v.float() for kk, v in synth.gpt_state_dict(cfg, cases.GEN_SEED, lora=False).items()}
    ocfg = Gp.GPTConfig(**{kk: v for kk, v in cfg.items() if kk in Gp.GPTConfig.__dataclass_fields__})
    nt, na = torch.from_numpy(g[f"{name}.noise_text"]), torch.from_numpy(g[f"{name}.noise_audio"])
    out = GG.generate(sd, ocfg, cases.gen_sequence(name), "TTS", temp=temp, top_k=k, temp_text=temp_text, top_k_text=k_text,
                      noise=lambda kind, gi, li: nt[gi].view(1, 1, -1) if kind == "text" else na[gi, li].view(1, 1, 1, -1))
    assert torch.equal(out["codes"], torch.from_numpy(g[f"{name}.codes"]).long())


def test_lm_oracle_sampling_matches_reference():
    """LMGenOracle with use_sampling=True against the reference LMGen's token streams under a seeded RNG, fed the noise that run
    drew (tests/golden/lm_tiny_sampling.npz)."""
    from oracle import lm_oracle as L
    cfg_d = dict(synth.LM_TINY)
    sd = {k: v.float() for k, v in synth.lm_state_dict(cfg_d, cases.LM_SEED).items()}
    g = np.load(os.path.join(G, "lm_tiny_sampling.npz"))
    sp = cases.LM_SAMPLING
    nt, na = torch.from_numpy(g["noise_text"]), torch.from_numpy(g["noise_audio"])
    B, dep_q = cases.LM_BATCH, cfg_d["dep_q"]
    gen = L.LMGenOracle(sd, L.LMConfig(**cfg_d), B, use_sampling=True, temp=sp["temp"], temp_text=sp["temp_text"],
                        top_k=sp["top_k"], top_k_text=sp["top_k_text"])
    user = cases.lm_user_tokens(cfg_d)
    outs = []
    with torch.no_grad():
        for s in range(cases.LM_STEPS):
            o = gen.step(user[s], noise_text=nt[s].view(B, 1, 1, -1), noise_audio=[na[s, c].view(B, 1, 1, -1) for c in range(dep_q)])
            outs.append(torch.full((B, dep_q + 1, 1), -9, dtype=torch.long) if o is None else o)
    assert torch.equal(torch.cat(outs, -1), torch.from_numpy(g["tokens"]).long())
