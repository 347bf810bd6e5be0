"""GPU parity of the litgpt-style backbone (rstnet_amd/lm/gpt.py over csrc/lm_*.hip) against the CPU oracle
(oracle/gpt_oracle.py, pinned to models.llama_streaming.GPT by tests/golden/gpt_tiny.npz).

Both sides compute in fp32 on the SAME bf16 weights: the oracle is fed the product's merged state dict up-cast, so logits
agree to summation-order noise (tolerance 1e-3 relative as the north star asks, observed ~1e-6).  Against the raw fixture
(fp32-merged weights on the reference side, bf16-rounded merged weights here) the tolerance is the bf16 weight rounding."""
import os

import numpy as np
import pytest
import torch

from oracle import gpt_oracle as Gp
from rstnet_amd import ops, synth
from rstnet_amd.lm.gpt import GPT, Config
from tests.golden import cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
TOL = 1e-3
CFGS = {"gqa": synth.GPT_TINY_GQA, "mha": synth.GPT_TINY_MHA}


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def build(name, batch_cfg=None):
    cfg_d = dict(CFGS[name])
    if batch_cfg:
        cfg_d.update(batch_cfg)
    sd = {k: v.to(DEV) for k, v in synth.gpt_state_dict(cfg_d, cases.GPT_SEED).items()}
    model = GPT.from_state_dict(sd, Config.from_dict(cfg_d))
    keep = set(Gp.GPTConfig.__dataclass_fields__)
    ocfg = Gp.GPTConfig(**{k: v for k, v in cfg_d.items() if k in keep})
    osd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    return model, ocfg, osd, cfg_d


@pytest.mark.parametrize("name", ["gqa", "mha"])
def test_forward_global_full_sequence(name):
    model, ocfg, osd, cfg_d = build(name)
    T = cases.GPT_T_FULL
    toks = cases.gpt_tokens(cfg_d)[:, :, :T]
    h, lg = model.forward_global(toks.to(DEV))
    with torch.no_grad():
        h_o, lg_o = Gp.forward_global(osd, ocfg, toks, merged=True)
    assert rel_err(h, h_o) < TOL and rel_err(lg, lg_o) < TOL
    g = np.load(os.path.join(GOLD, "gpt_tiny.npz"))
    assert rel_err(lg, torch.from_numpy(g[f"{name}.merged.logits"])) < 3e-2      # bf16 rounding of the merged weights


@pytest.mark.parametrize("name", ["gqa", "mha"])
@pytest.mark.parametrize("graphs", [False, True])
def test_streaming_steps_and_codecformer(name, graphs, monkeypatch):
    """T = 1 steps across the ring wrap (context 10, 14 steps), with the dep_q codecformer steps of every frame."""
    if not graphs:
        monkeypatch.setenv("NO_CUDA_GRAPH", "1")
    model, ocfg, osd, cfg_d = build(name)
    toks = cases.gpt_tokens(cfg_d)
    B = cases.GPT_BATCH
    st = Gp.new_global_state(ocfg, B)
    with model.streaming(B), torch.no_grad():
        for t in range(cases.GPT_STEPS):
            frame = toks[:, :, t:t + 1]
            h, lg = model.forward_global(frame.to(DEV))
            h, lg = h.clone(), lg.clone()
            h_o, lg_o = Gp.forward_global(osd, ocfg, frame, st, merged=True)
            assert rel_err(h, h_o) < TOL and rel_err(lg, lg_o) < TOL, t
            cst = Gp.new_codecformer_state(ocfg, B)
            with model.codecformer.streaming(B):
                for k in range(ocfg.dep_q):
                    prev = toks[:, 0:1, t:t + 1] if k == 0 else toks[:, k:k + 1, t:t + 1]
                    d = model.forward_codecformer(k, prev.to(DEV), h)
                    d_o = Gp.forward_codecformer(osd, ocfg, k, prev, h_o, cst)
                    assert rel_err(d, d_o) < TOL, (t, k)


@pytest.mark.parametrize("name", ["gqa", "mha"])
def test_prefill_equals_single_steps(name):
    """A T = 6 chunk followed by single steps == the same positions streamed one by one (the product rotates every
    position of a chunk by its own angle; see the oracle docstring for the reference's T > 1 streaming quirk)."""
    model, ocfg, osd, cfg_d = build(name)
    toks = cases.gpt_tokens(cfg_d).to(DEV)
    B = cases.GPT_BATCH
    with model.streaming(B):
        ref = [model.forward_global(toks[:, :, t:t + 1])[1].clone() for t in range(12)]
    with model.streaming(B):
        _, chunk = model.forward_global(toks[:, :, :6])
        rest = [model.forward_global(toks[:, :, t:t + 1])[1].clone() for t in range(6, 12)]
    assert rel_err(chunk, torch.cat(ref[:6], 1)) < 1e-4
    assert rel_err(torch.cat(rest, 1), torch.cat(ref[6:], 1)) < 1e-4


@pytest.mark.parametrize("name", ["gqa", "mha"])
def test_batch_above_four_uses_the_skinny_path(name):
    model, ocfg, osd, cfg_d = build(name)
    B = 6
    toks = cases.gpt_tokens(cfg_d, steps=4, batch=B)
    st = Gp.new_global_state(ocfg, B)
    with model.streaming(B), torch.no_grad():
        for t in range(4):
            h, lg = model.forward_global(toks[:, :, t:t + 1].to(DEV))
            h_o, lg_o = Gp.forward_global(osd, ocfg, toks[:, :, t:t + 1], st, merged=True)
            assert rel_err(h, h_o) < TOL and rel_err(lg, lg_o) < TOL, t


@pytest.mark.parametrize("name", ["gqa", "mha"])
def test_forward_local_and_forward(name):
    model, ocfg, osd, cfg_d = build(name)
    T = cases.GPT_T_FULL
    toks = cases.gpt_tokens(cfg_d)[:, :, :T]
    with torch.no_grad():
        h_o, _ = Gp.forward_global(osd, ocfg, toks, merged=True)
        loc_o = Gp.forward_local(osd, ocfg, toks[:, 0], toks[:, 1:ocfg.dep_q + 1], h_o)
    h, _ = model.forward_global(toks.to(DEV))
    loc = model.forward_local(toks[:, 0].to(DEV), toks[:, 1:ocfg.dep_q + 1].to(DEV), h)
    assert rel_err(loc, loc_o) < TOL
    emb = model.codecformer_text_emb(toks[:, 0].to(DEV))          # the reference's calling convention: embeddings
    assert rel_err(model.forward_local(emb, toks[:, 1:ocfg.dep_q + 1].to(DEV), h), loc_o) < TOL
    g = np.load(os.path.join(GOLD, "gpt_tiny.npz"))
    assert rel_err(loc, torch.from_numpy(g[f"{name}.local.logits"])) < 3e-2


def _attn_ref(q, kc, vc, pos, context, H, G, n, base):
    """Single-query GQA attention over an un-wrapped cache [B,G,pos+1,D] with interleaved partial RoPE already applied."""
    B, _, L, D = kc.shape
    qpk = H // G
    k = kc.repeat_interleave(qpk, 1)
    v = vc.repeat_interleave(qpk, 1)
    s = torch.einsum("bhd,bhld->bhl", q, k) / D ** 0.5
    idx = torch.arange(L)
    mask = (pos - idx >= 0) & (pos - idx < context)
    s = s.masked_fill(~mask, float("-inf"))
    return torch.einsum("bhl,bhld->bhd", torch.softmax(s, -1), v)


@pytest.mark.parametrize("D,H,G,n,cap,context", [(64, 4, 2, 32, 300, 300), (128, 8, 2, 128, 200, 150), (64, 4, 4, 64, 64, 64),
                                                  (128, 6, 3, 64, 700, 700)])
def test_gqa_partial_rope_decode_attention(D, H, G, n, cap, context):
    """rst_lm_attn_decode_f32 with grouped kv heads and partial interleaved RoPE, long-ring (split) and short-ring kernels,
    against a dense fp64 reference over 40 appended steps."""
    torch.manual_seed(D + H + cap)
    B, steps, base = 2, 40, 10000.0
    kc = torch.zeros(B, G, cap, D, device=DEV)
    vc = torch.zeros(B, G, cap, D, device=DEV)
    pos = torch.zeros(1, dtype=torch.long, device=DEV)
    kh = torch.zeros(B, G, steps, D, dtype=torch.float64)
    vh = torch.zeros(B, G, steps, D, dtype=torch.float64)
    theta = torch.exp(torch.arange(n // 2, dtype=torch.float64) * (-np.log(base) * 2 / n))

    def rot(x, p):
        x = x.clone()
        xr, xi = x[..., 0:n:2].clone(), x[..., 1:n:2].clone()
        x[..., 0:n:2] = xr * torch.cos(theta * p) - xi * torch.sin(theta * p)
        x[..., 1:n:2] = xr * torch.sin(theta * p) + xi * torch.cos(theta * p)
        return x
    for t in range(steps):
        qkv = torch.randn(B, (H + 2 * G) * D)
        out = ops.lm_attn_decode(qkv.to(DEV), kc, vc, pos, rope=True, context=context, max_period=base, heads=H, rope_dims=n)
        q = rot(qkv[:, :H * D].double().view(B, H, D), t)
        kh[:, :, t] = rot(qkv[:, H * D:(H + G) * D].double().view(B, G, D), t)
        vh[:, :, t] = qkv[:, (H + G) * D:].double().view(B, G, D)
        ref = _attn_ref(q, kh[:, :, :t + 1], vh[:, :, :t + 1], t, context, H, G, n, base).reshape(B, H * D)
        assert rel_err(out, ref) < 1e-4, t
        pos += 1
    assert rel_err(kc[:, :, :steps], kh) < 1e-4 and rel_err(vc[:, :, :steps], vh) == 0.0


# ---- streaming generation vs the O(T^2) restatement of infer_no_streaming.py ------------------------------------------------
def _tts_prompt(cfg_d, L=16, n_text=5, n_pad=2, seed=0):
    """A [K, L] utterance in the reference's TTS layout: text ids then `text_empty` on row 0, audio rows, trailing padding."""
    g = torch.Generator().manual_seed(seed)
    K = cfg_d["n_q"] + 1
    seq = torch.randint(0, 30, (K, L), generator=g)
    seq[0, :n_text] = torch.randint(0, 300, (n_text,), generator=g)
    seq[0, n_text:] = 318
    seq[1:, L - n_pad:] = 31
    return seq


def _noise_fn(k_text, k_audio):
    def fn(kind, g_idx, l_idx):
        g = torch.Generator().manual_seed(1000 * g_idx + 10 * l_idx + (kind == "text"))
        k = k_text if kind == "text" else k_audio
        return -torch.log(torch.rand(1, k, generator=g).clamp_min(1e-9))
    return fn


@pytest.mark.parametrize("name", ["gqa", "mha"])
@pytest.mark.parametrize("task,use_sampling", [("TTS", False), ("TTS", True), ("audio_only", True), ("ASR", False)])
def test_streaming_generate_matches_offline_loop(name, task, use_sampling):
    from oracle import gpt_generate_oracle as GG
    from rstnet_amd.lm.generate import GenIds, InferenceImp
    model, ocfg, osd, cfg_d = build(name)
    seq = _tts_prompt(cfg_d)
    if task == "ASR":
        seq[0, :4] = 318          # ASR layout: leading `text_empty` frames
        seq[0, 4:] = torch.arange(seq.shape[1] - 4) % 300
        seq[0, -2:] = 319         # text padding
    ids = dict(text_pad_token=319, text_empty_token=318, semantic_pad_token=31, n_audio_codes=30)
    noise = _noise_fn(5, 8)
    kw = dict(use_sampling=use_sampling, temp=0.8, top_k=8, temp_text=0.7, top_k_text=5)
    ref = GG.generate(osd, ocfg, seq, task, ids=GG.GenIds(text_initial_token_id=model.text_initial_token_id % 320, **ids),
                      noise=lambda kind, g, l: noise(kind, g, l).view(1, 1, -1) if kind == "text" else noise(kind, g, l).view(1, 1, 1, -1),
                      **kw)
    imp = InferenceImp(None, model, "sample", kw["temp_text"], kw["top_k_text"], kw["temp"], kw["top_k"], task,
                       use_sampling=use_sampling, ids=GenIds(text_initial_token_id=model.text_initial_token_id % 320, **ids),
                       noise=noise)
    out = imp.generate(seq)
    assert out["frames"].shape == ref["frames"].shape and out["frames"].shape[0] > 0
    assert torch.equal(out["frames"].cpu(), ref["frames"])
    assert torch.equal(out["text"].cpu(), ref["text"])


@pytest.mark.parametrize("name", list(cases.GEN_CASES))
def test_streaming_generate_matches_reference_loop_fixture(name):
    """The streaming O(T) `InferenceImp` of the build against the codes the REFERENCE's offline O(T^2) loop returned on the real
    reference GPT (tests/golden/gpt_generate.npz: `InferenceImp.__call__` of infer_no_streaming.py executed unchanged), fed the
    Exp(1) noise that run drew: real vocabularies (151 936 text ids, 2050 audio ids), the literal special ids, both id-blanking
    samplers, reverse_delay."""
    from rstnet_amd.lm.generate import InferenceImp
    g = np.load(os.path.join(GOLD, "gpt_generate.npz"))
    L, n_text, seed, temp_text, k_text, temp, k = cases.GEN_CASES[name]
    cfg_d = dict(synth.GPT_GEN_TINY)
    sd = {kk: v.to(DEV) for kk, v in synth.gpt_state_dict(cfg_d, cases.GEN_SEED, lora=False).items()}
    model = GPT.from_state_dict(sd, Config.from_dict(cfg_d))
    nt, na = torch.from_numpy(g[f"{name}.noise_text"]), torch.from_numpy(g[f"{name}.noise_audio"])
    imp = InferenceImp(None, model, "sample", temp_text, k_text, temp, k, "TTS",
                       noise=lambda kind, gi, li: nt[gi:gi + 1] if kind == "text" else na[gi, li:li + 1])
    out = imp.generate(cases.gen_sequence(name))
    assert torch.equal(out["codes"].cpu(), torch.from_numpy(g[f"{name}.codes"]).long())


def test_streaming_generate_graphed_greedy_equals_eager():
    """The production configuration (captured graphs, device-side Exp(1) draws) in greedy mode reproduces the eager loop."""
    from rstnet_amd.lm.generate import GenIds, InferenceImp
    model, ocfg, osd, cfg_d = build("gqa")
    seq = _tts_prompt(cfg_d, L=20)
    ids = GenIds(text_pad_token=319, text_empty_token=318, semantic_pad_token=31, n_audio_codes=30, text_initial_token_id=7)
    a = InferenceImp(None, model, "sample", 0.7, 5, 0.8, 8, "TTS", use_sampling=False, ids=ids).generate(seq)
    b = InferenceImp(None, model, "sample", 0.7, 5, 0.8, 8, "TTS", use_sampling=False, ids=ids, noise=lambda *a: torch.ones(1, 8)).generate(seq)
    assert torch.equal(a["frames"], b["frames"]) and torch.equal(a["text"], b["text"])
    c = InferenceImp(None, model, "sample", 0.7, 5, 0.8, 8, "TTS", use_sampling=True, ids=ids).generate(seq)
    assert c["frames"].shape == a["frames"].shape and int(c["frames"][1:, 0].max()) < 30      # blanked ids never sampled at l = 0


@pytest.mark.parametrize("B", [1, 2, 5])
@pytest.mark.parametrize("name", ["gqa", "mha"])
def test_gptgen_fused_step_equals_frame_then_advance(name, B, monkeypatch):
    """`GPTGen.step` (round 5: advance of the completed frame + text sample + dep_q depth steps as ONE captured graph on a
    device-resident token column) against the `frame` / `advance` pair `InferenceImp` uses -- sampled with the same injected noise
    (eager), and the captured graph in greedy mode; batch 1 / 2 take the persistent depth launch, batch 5 the launch-per-op chain."""
    from rstnet_amd.lm.generate import GPTGen
    model, ocfg, osd, cfg_d = build(name)
    n_codes = cfg_d["audio_card"] - 2
    g = torch.Generator().manual_seed(B)
    prompt = torch.randint(0, n_codes, (B, cfg_d["n_q"] + 1, 6), generator=g).to(DEV)
    k_text, k_audio = 5, 8

    def noise(kind, g_idx, l_idx):
        gg = torch.Generator().manual_seed(1000 * g_idx + 10 * l_idx + (kind == "text"))
        return -torch.log(torch.rand(B, k_text if kind == "text" else k_audio, generator=gg).clamp_min(1e-9))

    def run(fused, use_sampling, hook):
        gen = GPTGen(model, use_sampling=use_sampling, temp=0.8, temp_text=0.7, top_k=k_audio, top_k_text=k_text, n_audio_codes=n_codes,
                     noise=hook)
        gen.begin(B)
        gen.set_blanking([False] + [True] * (cfg_d["dep_q"] - 1))
        frames = []
        try:
            h, logits = gen.prefill(prompt)
            if fused:
                text, audio = gen.start(h, logits, 0)
                frames.append(torch.cat([text[:, None], audio], 1).clone())
                for _ in range(7):
                    text, audio = gen.step()
                    frames.append(torch.cat([text[:, None], audio], 1).clone())
            else:
                for g_idx in range(8):
                    text, audio = gen.frame(h.contiguous(), logits.contiguous(), g_idx)
                    frames.append(torch.cat([text[:, None], audio], 1).clone())
                    h, logits = gen.advance(text, audio)
        finally:
            gen.end()
        return torch.stack(frames).cpu()

    ref = run(False, True, noise)
    assert torch.equal(run(True, True, noise), ref)                 # same draws, same tokens (eager on both sides)
    greedy = run(False, False, None)
    assert torch.equal(run(True, False, None), greedy)              # the captured fused graph, greedy
    monkeypatch.setenv("NO_CUDA_GRAPH", "1")
    assert torch.equal(run(True, False, None), greedy)              # and the same function un-captured
    assert int(ref[:, :, 1].max()) < n_codes                        # sampling: the blanked ids of codebook 0 never come out


def test_sampler_id_blanking():
    from oracle.gpt_generate_oracle import sample_token
    torch.manual_seed(3)
    B, V, k = 3, 2050, 250
    logits = torch.randn(B, V) * 3
    logits[:, 2048:] += 6.0          # the blanked ids would otherwise dominate
    noise = -torch.log(torch.rand(B, k).clamp_min(1e-9))
    for limit in (2048, 2049, 0):
        ref = sample_token(logits.view(B, 1, 1, V), True, 0.8, k, noise.view(B, 1, 1, k), limit)[:, 0, 0]
        got = ops.lm_sample(logits.to(DEV), use_sampling=True, temp=0.8, top_k=k, noise=noise.to(DEV), limit=limit)
        assert torch.equal(got.cpu(), ref), limit
        lim_dev = torch.tensor([limit], dtype=torch.int32, device=DEV)
        got = ops.lm_sample(logits.to(DEV), use_sampling=True, temp=0.8, top_k=k, noise=noise.to(DEV), limit_dev=lim_dev)
        assert torch.equal(got.cpu(), ref), limit


@pytest.mark.parametrize("name", ["gqa", "mha"])
def test_fp8_blocks_stay_close_to_the_oracle(name):
    """Opt-in fp8 (e4m3) linears in the global blocks: streamed logits stay within fp8 accuracy of the fp32 oracle and the
    greedy text token agrees on most steps (tiny random model: logits are far apart)."""
    model, ocfg, osd, cfg_d = build(name)
    model.use_fp8(True)
    B = 6
    toks = cases.gpt_tokens(cfg_d, steps=5, batch=B)
    st = Gp.new_global_state(ocfg, B)
    agree = 0
    with model.streaming(B), torch.no_grad():
        for t in range(5):
            h, lg = model.forward_global(toks[:, :, t:t + 1].to(DEV))
            h_o, lg_o = Gp.forward_global(osd, ocfg, toks[:, :, t:t + 1], st, merged=True)
            assert rel_err(lg, lg_o) < 0.15, t
            agree += int((lg.argmax(-1).cpu() == lg_o.argmax(-1)).sum())
    assert agree >= 0.8 * 5 * B


def test_load_adapters_repacks_kernel_side_copies():
    """Adapter swap on a live model (GPT.load_adapters): the weights change in place, the kernels' packed copies (QKV row order,
    stacked gate, skinny operand order) must follow -- logits of the swapped model == logits of a model built from [base ; new set],
    for the full-sequence forward and for graph-captured T = 1 steps of a new session."""
    cfg_d = dict(CFGS["gqa"])
    cfg = Config.from_dict(cfg_d)
    is_lora = lambda k: k.endswith((".lora_A", ".lora_B"))
    sd1 = {k: v.to(DEV) for k, v in synth.gpt_state_dict(cfg_d, 21).items()}
    ad2 = {k: v.to(DEV) for k, v in synth.gpt_state_dict(cfg_d, 22).items() if is_lora(k)}
    base = {k: v for k, v in sd1.items() if not is_lora(k)}
    toks = cases.gpt_tokens(cfg_d)[:, :, :cases.GPT_T_FULL].to(DEV)
    model = GPT.from_state_dict({k: v.clone() for k, v in sd1.items()}, cfg, keep_lora_base=True)

    def logits_of(m):
        h, lg = m.forward_global(toks)
        with m.streaming(toks.shape[0]):
            steps = [m.forward_global(toks[:, :, t:t + 1])[1] for t in range(4)]
        return lg, torch.cat(steps, 1)

    first = logits_of(model)                                  # packs (and captures) everything once with adapter set 1
    ref1 = logits_of(GPT.from_state_dict({k: v.clone() for k, v in sd1.items()}, cfg))
    assert torch.equal(first[0], ref1[0]) and torch.equal(first[1], ref1[1])
    model.load_adapters(ad2)
    ref2 = logits_of(GPT.from_state_dict({**{k: v.clone() for k, v in base.items()}, **ad2}, cfg))
    got2 = logits_of(model)
    assert torch.equal(got2[0], ref2[0]) and torch.equal(got2[1], ref2[1])
    assert not torch.equal(got2[0], first[0])
    model.load_adapters(None)
    ref0 = logits_of(GPT.from_state_dict({k: v.clone() for k, v in base.items()}, cfg))
    got0 = logits_of(model)
    assert torch.equal(got0[0], ref0[0]) and torch.equal(got0[1], ref0[1])
    with model.streaming(2):
        with pytest.raises(RuntimeError):
            model.load_adapters(ad2)


@pytest.mark.parametrize("name,extra", [("gqa", {}), ("mha", {}), ("gqa", {"lora_alpha": 6, "lora_key": True})],
                         ids=["gqa", "mha", "gqa_qkv_alpha6"])
@pytest.mark.parametrize("B", [2, 5])
def test_unmerged_lora_forward_matches_the_oracle(name, extra, B):
    """``GPT.from_state_dict(..., merge_lora=False)``: LoRALinear.forward / LoRAQKVLinear.forward with the adapters apart
    (llama_streaming.py:136-143, 373-406) -- the full-sequence forward and graph-captured T = 1 steps across the ring wrap against the
    oracle's UNMERGED path (the one gpt_tiny.npz pins to the reference's unmerged forward), at batch 2 (weight-streaming products)
    and batch 5 (matrix-core products, packed attention output); the second case has q, k and v adapted under grouped queries (the
    zero_pad 'as is' return) and an alpha / r that is not a power of two (applied to A x in fp32)."""
    cfg_d = {**CFGS[name], **extra}
    sd = {k: v.to(DEV) for k, v in synth.gpt_state_dict(cfg_d, cases.GPT_SEED).items()}
    model = GPT.from_state_dict(sd, Config.from_dict(cfg_d), merge_lora=False)
    assert any(k.endswith(".lora_A") for k in model.state_dict())
    keep = set(Gp.GPTConfig.__dataclass_fields__)
    ocfg = Gp.GPTConfig(**{k: v for k, v in cfg_d.items() if k in keep})
    osd = {k: v.detach().float().cpu() for k, v in sd.items()}
    T = cases.GPT_T_FULL
    toks = cases.gpt_tokens(cfg_d, steps=cases.GPT_STEPS, batch=B)
    h, lg = model.forward_global(toks[:, :, :T].to(DEV))
    with torch.no_grad():
        h_o, lg_o = Gp.forward_global(osd, ocfg, toks[:, :, :T])
    assert rel_err(h, h_o) < TOL and rel_err(lg, lg_o) < TOL
    if not extra and B == cases.GPT_BATCH:
        g = np.load(os.path.join(GOLD, "gpt_tiny.npz"))
        assert rel_err(lg, torch.from_numpy(g[f"{name}.full.logits"])) < TOL      # same bf16 weights on both sides: no merge rounding
    merged = GPT.from_state_dict({k: v.clone() for k, v in sd.items()}, Config.from_dict(cfg_d))
    assert rel_err(merged.forward_global(toks[:, :, :T].to(DEV))[1], lg) < 3e-2   # vs the merged build: bf16 rounding of W + BA
    st = Gp.new_global_state(ocfg, B)
    with model.streaming(B), torch.no_grad():
        for t in range(cases.GPT_STEPS):
            h, lg = model.forward_global(toks[:, :, t:t + 1].to(DEV))
            h_o, lg_o = Gp.forward_global(osd, ocfg, toks[:, :, t:t + 1], st)
            assert rel_err(h, h_o) < TOL and rel_err(lg, lg_o) < TOL, t


def test_unmerged_lora_adapter_swap():
    """``load_adapters`` on an unmerged model replaces the adapter tensors only: logits == a model built from [base ; new set]."""
    cfg_d = dict(CFGS["gqa"])
    cfg = Config.from_dict(cfg_d)
    is_lora = lambda k: k.endswith((".lora_A", ".lora_B"))
    sd1 = {k: v.to(DEV) for k, v in synth.gpt_state_dict(cfg_d, 21).items()}
    ad2 = {k: v.to(DEV) for k, v in synth.gpt_state_dict(cfg_d, 22).items() if is_lora(k)}
    toks = cases.gpt_tokens(cfg_d, steps=4, batch=3).to(DEV)

    def logits_of(m):
        with m.streaming(3):
            return torch.cat([m.forward_global(toks[:, :, t:t + 1])[1] for t in range(4)], 1)

    model = GPT.from_state_dict({k: v.clone() for k, v in sd1.items()}, cfg, merge_lora=False)
    w_ptr = model.transformer.h[0].attn.attn.linear.weight.data_ptr()
    first = logits_of(model)
    model.load_adapters(ad2)
    assert model.transformer.h[0].attn.attn.linear.weight.data_ptr() == w_ptr
    want = logits_of(GPT.from_state_dict({**{k: v.clone() for k, v in sd1.items() if not is_lora(k)}, **ad2}, cfg, merge_lora=False))
    got = logits_of(model)
    assert torch.equal(got, want) and not torch.equal(got, first)
    model.load_adapters(None)
    base = logits_of(GPT.from_state_dict({k: v.clone() for k, v in sd1.items() if not is_lora(k)}, Config.from_dict({**cfg_d, "lora_r": 0})))
    assert torch.equal(logits_of(model), base)
