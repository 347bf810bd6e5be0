"""Generate the golden fixtures under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference, read-only, imported in place; nothing
from it is copied).  The GPU box and the CPU test-suite only ever read the .npz files written here.

    NO_TORCH_COMPILE=1 python -B tests/golden/make_golden.py

Weights are never stored: they are regenerated from ``rstnet_amd.synth.mimi_state_dict(seed)``
(or a seeded Xavier init for the layer-level cases), which is bit-reproducible on CPU.
"""
import os
import sys

os.environ["NO_TORCH_COMPILE"] = "1"  # eager fp32, no Inductor
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference/MLLM_v2")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from rstnet_amd import synth  # noqa: E402
from tests.golden import cases  # noqa: E402

from tools.tokenizer.MimiCodec.model.models.MimiCodec import MimiCodec  # noqa: E402
from tools.tokenizer.MimiCodec.model.modules.conv import StreamingConv1d, StreamingConvTranspose1d  # noqa: E402
from tools.tokenizer.MimiCodec.model.modules.seanet import SEANetResnetBlock  # noqa: E402


def ref_mimi(sd):
    m = MimiCodec(encoder_rates=[8, 6, 5, 4], codebook_size=2048, codebook_dim=256, rvq_layers=8).eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("semantic_mapping_layer") for k in missing), (missing, unexpected)
    return m


@torch.no_grad()
def gen_mimi_e2e():
    """F5: MimiCodec.encode / decode (models/MimiCodec.py:93-110) on config-1 style clips."""
    sd = synth.mimi_state_dict(cases.MIMI_SEED)
    m = ref_mimi(sd)
    out = {}
    for name, (B, T, aseed) in cases.MIMI_E2E.items():
        audio = synth.synth_audio(B, T, aseed)
        z = m.downsample(m.encoder_transformer(m.encoder(audio))[0])
        codes = m.encode(audio)
        wav = m.decode(codes)
        # top-2 gap of every RVQ decision, so that a mismatch can be attributed to a near tie
        gaps = []
        for rvq in (m.quantizer.rvq_first, m.quantizer.rvq_rest):
            r = rvq.input_proj(z).transpose(1, 2).reshape(-1, 256)
            for layer in rvq.vq.layers:
                d = torch.cdist(r[None], layer.embedding[None])[0]
                t2 = d.topk(2, largest=False)
                gaps.append(((t2.values[:, 1] - t2.values[:, 0]) / t2.values[:, 0]).view(B, -1))
                r = r - layer.embedding[t2.indices[:, 0]]
        out[f"{name}.latent"] = z.numpy()
        out[f"{name}.codes"] = codes.numpy().astype(np.int16)
        out[f"{name}.wav"] = wav.numpy()
        out[f"{name}.rel_gap"] = torch.stack(gaps, 1).numpy()
        print(name, tuple(codes.shape), tuple(wav.shape), "min rel gap %.2e" % torch.stack(gaps).min())
    np.savez(os.path.join(HERE, "mimi_e2e.npz"), **out)


@torch.no_grad()
def gen_rvq():
    """F3: SplitResidualVectorQuantizer.encode/decode (quantization/vq.py:305-323) on a seeded latent."""
    sd = synth.mimi_state_dict(cases.MIMI_SEED)
    m = ref_mimi(sd)
    z = cases.rvq_latent(sd)
    codes = m.quantizer.encode(z)
    zq = m.quantizer.decode(codes)
    np.savez(os.path.join(HERE, "rvq.npz"), codes=codes.numpy().astype(np.int16), zq0=zq[:1].numpy())  # first item only (size)
    print("rvq", tuple(codes.shape), tuple(zq.shape))


@torch.no_grad()
def gen_transformer():
    """F4: ProjectedTransformer (modules/transformer.py:738-750), T > context so the window matters."""
    sd = synth.mimi_state_dict(cases.MIMI_SEED, layer_scale=cases.TRANSFORMER_LAYER_SCALE)
    m = ref_mimi(sd)
    x = cases.transformer_input()
    y = m.encoder_transformer(x)[0]
    np.savez(os.path.join(HERE, "transformer.npz"), y=y.numpy())
    print("transformer", tuple(y.shape), float(y.abs().max()))


@torch.no_grad()
def gen_transformer_dims():
    """The reference's ProjectedTransformer (modules/transformer.py:595-750) in STREAMING mode at head dims 16 / 32 / 128 / 256, several
    positions per step, past the ring wrap: what the oracle's TransformerStream and the persistent HIP launch are held to at shapes
    other than Mimi's 64-dim heads."""
    from moshi.modules import transformer
    out = {}
    for name, (E, H, F, L, ctx, B, chunks) in cases.TRANSFORMER_DIMS.items():
        m = transformer.ProjectedTransformer(input_dimension=E, output_dimensions=(E,), d_model=E, num_heads=H, num_layers=L,
                                             dim_feedforward=F, causal=True, context=ctx, conv_layout=True, max_period=10000,
                                             gating="none", norm="layer_norm", positional_embedding="rope",
                                             layer_scale=cases.TRANSFORMER_DIMS_LAYER_SCALE, device="cpu").eval()
        sd = {k[len("tr."):]: v for k, v in cases.transformer_dims_state(name).items()}
        missing, unexpected = m.load_state_dict(sd, strict=True)
        assert not missing and not unexpected
        x = cases.transformer_dims_input(name)
        ys, i = [], 0
        with m.streaming(B):
            for T in chunks:
                ys.append(m(x[:, :, i:i + T])[0])
                i += T
        assert i == x.shape[-1] > ctx
        out[name] = torch.cat(ys, -1).numpy()
        print("transformer_dims", name, out[name].shape, float(np.abs(out[name]).max()))
    np.savez_compressed(os.path.join(HERE, "transformer_dims.npz"), **out)


@torch.no_grad()
def gen_layers():
    """F2: the reference's own conv / conv-transpose / resnet-block test shapes
    (MLLM_v2/moshi/modules/conv_test.py:11-48, seanet_test.py) with seed-41 Xavier weights."""
    out = {}
    for name, (B, cin, cout, T, K, S) in cases.CONV_CASES.items():
        layer = StreamingConv1d(cin, cout, K, stride=S, causal=True, norm="none", pad_mode="constant")
        w, b, x = cases.layer_tensors(name, (cout, cin, K), cout, (B, cin, T))
        layer.conv.conv.weight.copy_(w)
        layer.conv.conv.bias.copy_(b)
        out[f"conv.{name}"] = layer(x).numpy()
    for name, (B, cin, cout, T, K, S) in cases.CONVTR_CASES.items():
        layer = StreamingConvTranspose1d(cin, cout, K, S, causal=True, norm="none")
        w, b, x = cases.layer_tensors(name, (cin, cout, K), cout, (B, cin, T))
        layer.convtr.convtr.weight.copy_(w)
        layer.convtr.convtr.bias.copy_(b)
        out[f"convtr.{name}"] = layer(x).numpy()
    for name, (B, dim, T) in cases.RESBLOCK_CASES.items():
        blk = SEANetResnetBlock(dim, kernel_sizes=[3, 1], dilations=[1, 1], causal=True, pad_mode="constant", compress=2)
        w1, b1, x = cases.layer_tensors(name + ".1", (dim // 2, dim, 3), dim // 2, (B, dim, T))
        w2, b2, _ = cases.layer_tensors(name + ".3", (dim, dim // 2, 1), dim, (1, 1, 1))
        blk.block[1].conv.conv.weight.copy_(w1)
        blk.block[1].conv.conv.bias.copy_(b1)
        blk.block[3].conv.conv.weight.copy_(w2)
        blk.block[3].conv.conv.bias.copy_(b2)
        out[f"resblock.{name}"] = blk(x).numpy()
    np.savez(os.path.join(HERE, "layers.npz"), **out)
    print("layers", len(out))


@torch.no_grad()
def gen_lm_tiny():
    """F6 + F8: LMModel.forward_text / forward_depformer logits and the greedy LMGen.step token streams
    (MLLM_v2/models/model.py:364-597) on the tiny config, fp32 arithmetic on bf16-rounded weights."""
    from models.model import LMModel, LMGen
    cfg = dict(synth.LM_TINY)
    sd = {k: v.float() for k, v in synth.lm_state_dict(cfg, cases.LM_SEED).items()}
    m = LMModel(causal=True, layer_scale=None, gating="silu", norm="rms_norm_f32", positional_embedding="rope",
                depformer_causal=True, depformer_layer_scale=None, depformer_multi_linear=True, depformer_context=8,
                depformer_max_period=10000, depformer_gating="silu", depformer_pos_emb="none",
                depformer_weights_per_step=True, **cfg).eval()
    missing, unexpected = m.load_state_dict(sd, strict=True)
    gen = LMGen(m, use_sampling=False)
    user = cases.lm_user_tokens(cfg)
    outs, text_logits, dep_logits = [], [], []
    # capture logits through forward hooks on the two heads
    h1 = m.text_linear.register_forward_hook(lambda mod, i, o: text_logits.append(o.clone()))
    hs = [l.register_forward_hook(lambda mod, i, o: dep_logits.append(o.clone())) for l in m.linears]
    with gen.streaming(cases.LM_BATCH):
        for s in range(cases.LM_STEPS):
            o = gen.step(user[s])
            outs.append(torch.full((cases.LM_BATCH, cfg["dep_q"] + 1, 1), -9, dtype=torch.long) if o is None else o)
    h1.remove()
    [h.remove() for h in hs]
    np.savez(os.path.join(HERE, "lm_tiny.npz"), tokens=torch.cat(outs, -1).numpy().astype(np.int32),
             text_logits=torch.stack(text_logits).numpy(), dep_logits=torch.stack(dep_logits).numpy())
    print("lm_tiny", torch.cat(outs, -1).shape, torch.stack(text_logits).shape, torch.stack(dep_logits).shape)


@torch.no_grad()
def gen_lm_tiny_sampling():
    """F8 with sampling: LMGen.step (models/model.py:490-597) with use_sampling=True under a seeded RNG.  The Exp(1) noise of
    `multinomial` is re-drawn from the same seed in call order ([B, top_k_text] for the text token, then dep_q x [B, top_k]) and
    stored with the token streams; the script asserts that the oracle fed that noise reproduces them."""
    from models.model import LMModel, LMGen
    from oracle import lm_oracle as L
    cfg = dict(synth.LM_TINY)
    sd = {k: v.float() for k, v in synth.lm_state_dict(cfg, cases.LM_SEED).items()}
    m = LMModel(causal=True, layer_scale=None, gating="silu", norm="rms_norm_f32", positional_embedding="rope",
                depformer_causal=True, depformer_layer_scale=None, depformer_multi_linear=True, depformer_context=8,
                depformer_max_period=10000, depformer_gating="silu", depformer_pos_emb="none",
                depformer_weights_per_step=True, **cfg).eval()
    m.load_state_dict(sd, strict=True)
    sp = cases.LM_SAMPLING
    gen = LMGen(m, use_sampling=True, temp=sp["temp"], temp_text=sp["temp_text"], top_k=sp["top_k"], top_k_text=sp["top_k_text"])
    user = cases.lm_user_tokens(cfg)
    B, dep_q = cases.LM_BATCH, cfg["dep_q"]
    outs = []
    torch.manual_seed(sp["seed"])
    with gen.streaming(B):
        for s in range(cases.LM_STEPS):
            o = gen.step(user[s])
            outs.append(torch.full((B, dep_q + 1, 1), -9, dtype=torch.long) if o is None else o)
    tokens = torch.cat(outs, -1)
    torch.manual_seed(sp["seed"])
    nt, na = [], []
    for s in range(cases.LM_STEPS):
        nt.append(torch.empty(B, sp["top_k_text"]).exponential_(1))
        na.append(torch.stack([torch.empty(B, sp["top_k"]).exponential_(1) for _ in range(dep_q)]))
    nt, na = torch.stack(nt), torch.stack(na)          # [steps, B, k_text], [steps, dep_q, B, k]
    og = L.LMGenOracle(sd, L.LMConfig(**cfg), B, use_sampling=True, temp=sp["temp"], temp_text=sp["temp_text"], top_k=sp["top_k"],
                       top_k_text=sp["top_k_text"])
    o_outs = []
    for s in range(cases.LM_STEPS):
        o = og.step(user[s], noise_text=nt[s].view(B, 1, 1, -1), noise_audio=[na[s, c].view(B, 1, 1, -1) for c in range(dep_q)])
        o_outs.append(torch.full((B, dep_q + 1, 1), -9, dtype=torch.long) if o is None else o)
    assert torch.equal(torch.cat(o_outs, -1), tokens)
    np.savez(os.path.join(HERE, "lm_tiny_sampling.npz"), tokens=tokens.numpy().astype(np.int32), noise_text=nt.numpy(),
             noise_audio=na.numpy())
    print("lm_tiny_sampling", tuple(tokens.shape), tokens[0, :, -4:].tolist())


@torch.no_grad()
def gen_gpt_tiny():
    """F6 + F7: models.llama_streaming.GPT on the two tiny configs -- non-streaming forward_global (LoRA unmerged and after
    merge_lora_weights), streamed T = 1 forward_global steps across the ring wrap, forward_codecformer steps and the
    teacher-forced forward_local.  fp32 arithmetic on bf16-rounded weights."""
    from models.llama_streaming import GPT, Config, merge_lora_weights
    out = {}
    for name, cfg in (("gqa", synth.GPT_TINY_GQA), ("mha", synth.GPT_TINY_MHA)):
        sd = {k: v.float() for k, v in synth.gpt_state_dict(cfg, cases.GPT_SEED).items()}
        m = GPT(Config(name=name, **cfg)).eval()
        missing, unexpected = m.load_state_dict(sd, strict=True)
        toks = cases.gpt_tokens(cfg)
        B, dep_q = cases.GPT_BATCH, cfg["dep_q"]
        h_full, logits_full = m.forward_global(toks[:, :, :cases.GPT_T_FULL])
        out[f"{name}.full.h"], out[f"{name}.full.logits"] = h_full.numpy(), logits_full.numpy()
        hs, ls, dep = [], [], []
        with m.streaming(B):
            for t in range(cases.GPT_STEPS):
                h, lg = m.forward_global(toks[:, :, t:t + 1])
                hs.append(h)
                ls.append(lg)
                with m.codecformer.streaming(B):
                    for k in range(dep_q):
                        prev = toks[:, 0:1, t:t + 1] if k == 0 else toks[:, k:k + 1, t:t + 1]
                        dep.append(m.forward_codecformer(k, prev, h))
        out[f"{name}.stream.h"], out[f"{name}.stream.logits"] = torch.cat(hs, 1).numpy(), torch.cat(ls, 1).numpy()
        out[f"{name}.stream.dep_logits"] = torch.stack(dep).view(cases.GPT_STEPS, dep_q, B, -1).numpy()
        T = cases.GPT_T_FULL
        local = m.forward_local(m.codecformer_text_emb(toks[:, 0, :T]), toks[:, 1:dep_q + 1, :T], h_full)
        out[f"{name}.local.logits"] = local.numpy()
        merge_lora_weights(m)
        h_m, logits_m = m.forward_global(toks[:, :, :T])
        out[f"{name}.merged.logits"] = logits_m.numpy()
        # the merged fused-QKV weight of block 0 through its row and column sums (the matrix itself was 1.3 MB of the fixture set; the
        # merged logits above pin the merge as well)
        w0 = m.transformer.h[0].attn.attn.linear.weight.detach().double()
        out[f"{name}.merged.qkv0_rowsum"], out[f"{name}.merged.qkv0_colsum"] = w0.sum(1).float().numpy(), w0.sum(0).float().numpy()
        print("gpt_tiny", name, logits_full.shape, out[f"{name}.stream.dep_logits"].shape, local.shape,
              float((logits_m - logits_full).abs().max()))
    np.savez(os.path.join(HERE, "gpt_tiny.npz"), **out)


@torch.no_grad()
def gen_sampling():
    """F9: the reference samplers (MLLM_v2/utils/sampling.py:85-158) under a known RNG state.  `multinomial` (:44-46) draws
    its Exp(1) noise with `torch.empty_like(top-k probs).exponential_(1)` right after the seed is set, so the same seed
    reproduces that noise tensor; it is stored next to the tokens the reference returned (and greedy tokens)."""
    import utils.sampling as S
    out = {}
    for name, (fn, B, V, k, temp, seed) in cases.SAMPLING_CASES.items():
        lg = cases.sampling_logits(name)
        torch.manual_seed(seed)
        tok = getattr(S, fn)(lg.clone(), use_sampling=True, temp=temp, top_k=k)
        torch.manual_seed(seed)
        noise = torch.empty(B, k).exponential_(1)
        greedy = getattr(S, fn)(lg.clone(), use_sampling=False)
        out[f"{name}.tokens"], out[f"{name}.noise"] = tok.numpy().astype(np.int32), noise.numpy()
        out[f"{name}.greedy"] = greedy.numpy().astype(np.int32)
        print("sampling", name, tuple(tok.shape), tok.flatten().tolist())
    # nucleus sampling: sample_token(top_p=p) -> sample_top_p (:66-82); its multinomial draws [B, 1, 1, V] noise right after the seed
    for name, (B, V, top_p, temp, seed, scale) in cases.SAMPLING_TOP_P_CASES.items():
        lg = cases.sampling_top_p_logits(name)
        torch.manual_seed(seed)
        tok = S.sample_token(lg.clone(), use_sampling=True, temp=temp, top_k=0, top_p=top_p)
        probs = torch.softmax(lg / temp, -1).sort(-1, descending=True).values
        nucleus = ((probs.cumsum(-1) - probs) <= top_p).sum(-1).flatten()
        out[f"top_p.{name}.tokens"] = tok.numpy().astype(np.int32)
        out[f"top_p.{name}.nucleus"] = nucleus.numpy().astype(np.int32)
        print("sampling top_p", name, tuple(tok.shape), tok.flatten().tolist(), "nucleus sizes", nucleus.tolist())
    np.savez(os.path.join(HERE, "sampling.npz"), **out)


def gen_reverse_delay():
    """`reverse_delay` of MLLM_v2/infer_no_streaming.py:311-323.  The module itself cannot be imported here (torchaudio,
    huggingface_hub ... at its top level), so the function definition alone is taken from the parsed source of the file where it
    lies and executed as is -- no stand-in modules, nothing copied: only its outputs on seeded inputs are stored."""
    import ast
    path = "/root/reference/MLLM_v2/infer_no_streaming.py"
    tree = ast.parse(open(path).read(), path)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "reverse_delay")
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    out = {}
    for name, shape in cases.REVERSE_DELAY_CASES.items():
        x = cases.reverse_delay_input(name)
        out[name] = ns["reverse_delay"](x.clone()).numpy()
        print("reverse_delay", name, tuple(x.shape), "->", out[name].shape)
    np.savez(os.path.join(HERE, "reverse_delay.npz"), **out)


@torch.no_grad()
def gen_mimi_model():
    """The composition form of the codec, `moshi.models.compression.MimiModel` built as `moshi.models.loaders.get_mimi` builds
    it (loaders.py:105-139; 8 trained codebooks here instead of 32, seeded weights): state_dict keys, properties, batch encode
    with 8 and with 4 active codebooks, frame-by-frame streaming encode and decode."""
    from moshi.models import loaders
    from moshi.models.compression import MimiModel
    from moshi.modules import SEANetDecoder, SEANetEncoder, transformer
    from moshi.quantization import SplitResidualVectorQuantizer
    sd = synth.mimi_state_dict(cases.MIMI_SEED)
    enc, dec = SEANetEncoder(**loaders._seanet_kwargs), SEANetDecoder(**loaders._seanet_kwargs)
    m = MimiModel(enc, dec, SplitResidualVectorQuantizer(**{**loaders._quantizer_kwargs, "n_q": 8}), channels=1,
                  sample_rate=loaders.SAMPLE_RATE, frame_rate=loaders.FRAME_RATE, encoder_frame_rate=loaders.SAMPLE_RATE / enc.hop_length,
                  causal=True, resample_method="conv",
                  encoder_transformer=transformer.ProjectedTransformer(device="cpu", **loaders._transformer_kwargs),
                  decoder_transformer=transformer.ProjectedTransformer(device="cpu", **loaders._transformer_kwargs)).eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    out = {"keys": np.array(sorted(m.state_dict().keys())),
           "props": np.array([m.frame_rate, m.sample_rate, m.channels, m.num_codebooks, m.total_codebooks, m.cardinality])}
    B, T, seed = cases.MIMI_E2E["ragged"]
    audio = synth.synth_audio(B, T, seed=seed)
    out["codes8"] = m.encode(audio).numpy().astype(np.int16)
    m.set_num_codebooks(4)
    out["codes4"] = m.encode(audio).numpy().astype(np.int16)
    out["wav4"] = m.decode(torch.from_numpy(out["codes4"]).long()).numpy()      # decode of 4 of the 8 trained codebooks
    m.set_num_codebooks(8)
    a1 = synth.synth_audio(1, 1920 * 12, seed=cases.MIMI_E2E["cfg1"][2])
    cs, ws = [], []
    with m.streaming(1):
        for f in range(12):
            c = m.encode(a1[:, :, f * 1920:(f + 1) * 1920])
            cs.append(c)
            ws.append(m.decode(c))
    out["stream_codes"], out["stream_wav"] = torch.cat(cs, -1).numpy().astype(np.int16), torch.cat(ws, -1).numpy()
    assert torch.equal(torch.cat(cs, -1), m.encode(a1))
    print("mimi_model", out["codes8"].shape, out["codes4"].shape, out["wav4"].shape, out["stream_codes"].shape, out["stream_wav"].shape,
          out["props"].tolist())
    np.savez_compressed(os.path.join(HERE, "mimi_model.npz"), **out)


@torch.no_grad()
def gen_mimi_stream_long():
    """F4 at model level: the moshi `MimiModel` streamed frame by frame for MIMI_STREAM_LONG frames, batch 2 -- past the point
    (frame 125 = position 250) where the 250-slot KV rings of the encoder / decoder transformers wrap
    (modules/transformer.py:211-278).  Stored: all codes, the waveform of the last frames (past the wrap) and of a
    few early ones, and the top-2 relative gap of every RVQ decision of the streamed latent."""
    from moshi.models import loaders
    from moshi.models.compression import MimiModel
    from moshi.modules import SEANetDecoder, SEANetEncoder, transformer
    from moshi.quantization import SplitResidualVectorQuantizer
    sd = synth.mimi_state_dict(cases.MIMI_SEED, layer_scale=cases.TRANSFORMER_LAYER_SCALE)   # attention must matter
    enc, dec = SEANetEncoder(**loaders._seanet_kwargs), SEANetDecoder(**loaders._seanet_kwargs)
    m = MimiModel(enc, dec, SplitResidualVectorQuantizer(**{**loaders._quantizer_kwargs, "n_q": 8}), channels=1,
                  sample_rate=loaders.SAMPLE_RATE, frame_rate=loaders.FRAME_RATE, encoder_frame_rate=loaders.SAMPLE_RATE / enc.hop_length,
                  causal=True, resample_method="conv",
                  encoder_transformer=transformer.ProjectedTransformer(device="cpu", **loaders._transformer_kwargs),
                  decoder_transformer=transformer.ProjectedTransformer(device="cpu", **loaders._transformer_kwargs)).eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    B, frames, seed = cases.MIMI_STREAM_LONG
    audio = synth.synth_audio(B, 1920 * frames, seed=seed)
    cs, ws, gaps, zs = [], [], [], []
    with m.streaming(B):
        for f in range(frames):
            x = audio[:, :, f * 1920:(f + 1) * 1920]
            z = m._to_framerate(m.encoder_transformer(m.encoder(x))[0])
            c = m.quantizer.encode(z)
            zs.append(z)
            g = []
            for rvq in (m.quantizer.rvq_first, m.quantizer.rvq_rest):
                r = rvq.input_proj(z).transpose(1, 2).reshape(-1, 256)
                for layer in list(rvq.vq.layers)[:(1 if rvq is m.quantizer.rvq_first else 7)]:
                    d = torch.cdist(r[None], layer.embedding[None])[0]
                    t2 = d.topk(2, largest=False)
                    g.append(((t2.values[:, 1] - t2.values[:, 0]) / t2.values[:, 0]).view(B, -1))
                    r = r - layer.embedding[t2.indices[:, 0]]
            gaps.append(torch.stack(g, 1))
            cs.append(c)
            ws.append(m.decode(c))
    codes, wav, gaps = torch.cat(cs, -1), torch.cat(ws, -1), torch.cat(gaps, -1)
    tail = cases.MIMI_STREAM_LONG_TAIL
    out = {"codes": codes.numpy().astype(np.int16), "wav_head": wav[:, :, :1920 * 4].numpy(),
           "wav_tail": wav[:, :, -1920 * tail:].numpy(), "rel_gap": gaps.numpy(),
           "latent_tail": torch.cat(zs, -1)[:, :, -tail:].numpy()}
    print("mimi_stream_long", tuple(codes.shape), tuple(wav.shape), "min rel gap %.2e" % gaps.min(),
          "decisions below 1e-5:", int((gaps < 1e-5).sum()))
    np.savez_compressed(os.path.join(HERE, "mimi_stream_long.npz"), **out)


@torch.no_grad()
def gen_tokenizer():
    """`MimiTokenizer.tokenize / detokenize / tokenize2 / find_length` (tools/tokenizer/MimiCodec/mimi_tokenizer.py:47-82).  The
    module imports omegaconf / torchaudio / huggingface_hub at its top level (absent here) and its constructor downloads a
    checkpoint; the four methods need neither for a 24 kHz tensor input.  The class definition is taken from the parsed source
    of the file where it lies, instantiated without its constructor and given the real reference MimiCodec with the seeded
    weights of the other fixtures."""
    import ast
    from tools.tokenizer.abs_tokenizer import AbsTokenizer
    path = "/root/reference/MLLM_v2/tools/tokenizer/MimiCodec/mimi_tokenizer.py"
    tree = ast.parse(open(path).read(), path)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "MimiTokenizer")
    ns = {"torch": torch, "AbsTokenizer": AbsTokenizer}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), path, "exec"), ns)
    tok = ns["MimiTokenizer"].__new__(ns["MimiTokenizer"])
    torch.nn.Module.__init__(tok)
    tok.device, tok.sr = torch.device("cpu"), 24000
    tok.model = ref_mimi(synth.mimi_state_dict(cases.MIMI_SEED))
    B, T, seed = cases.MIMI_E2E["ragged"]
    wav = synth.synth_audio(B, T, seed=seed)[1]            # [1, T]: one utterance, T not a multiple of the hop
    codes = tok.tokenize(wav, 24000)
    # (the reference's detokenize wants the int64 ids of tokenize2: F.embedding rejects the int16 storage form)
    out = {"codes": codes.numpy(), "wav": tok.detokenize(tok.tokenize2(codes)).numpy(), "tokenize2": tok.tokenize2(codes).numpy(),
           "find_length": np.array(tok.find_length(codes)), "passthrough": tok.tokenize(codes[0].clone(), 24000).numpy()}
    assert codes.dtype == torch.int16
    print("tokenizer", tuple(codes.shape), codes.dtype, out["wav"].shape, out["tokenize2"].dtype, int(out["find_length"]))
    np.savez_compressed(os.path.join(HERE, "tokenizer.npz"), **out)


@torch.no_grad()
def gen_gpt_generate():
    """The reference's OFFLINE generation loop itself: class `InferenceImp` and `reverse_delay` of
    MLLM_v2/infer_no_streaming.py:149-323, taken from the parsed source of the file where it lies (its top-level imports --
    torchaudio, huggingface_hub, the dataloader -- are absent here, its class body needs none of them) and executed unchanged on
    the REAL models.llama_streaming.GPT and the REAL utils.sampling functions.  Stored: the codes it returns and the Exp(1)
    noise its samplers drew (re-drawn from the same seed in call order; the script asserts that the build's restatement of
    the loop, fed that noise, reproduces the reference's result)."""
    import ast
    import utils.sampling as S
    from models.llama_streaming import GPT, Config
    from oracle import gpt_generate_oracle as GG
    from oracle import gpt_oracle as Gp
    path = "/root/reference/MLLM_v2/infer_no_streaming.py"
    tree = ast.parse(open(path).read(), path)
    keep = [n for n in tree.body if (isinstance(n, ast.ClassDef) and n.name == "InferenceImp")
            or (isinstance(n, ast.FunctionDef) and n.name == "reverse_delay")]
    assert len(keep) == 2
    ns = {"torch": torch, "sample_token": S.sample_token, "sample_token_audio": S.sample_token_audio,
          "sample_token_audio_2048": S.sample_token_audio_2048}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    cfg = dict(synth.GPT_GEN_TINY)
    sd = {k: v.float() for k, v in synth.gpt_state_dict(cfg, cases.GEN_SEED, lora=False).items()}
    m = GPT(Config(name="gen", **cfg)).eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("lora" in k for k in missing), (missing, unexpected)
    ocfg = Gp.GPTConfig(**{k: v for k, v in cfg.items() if k in Gp.GPTConfig.__dataclass_fields__})
    out = {}
    for name, (L, n_text, seed, temp_text, k_text, temp, k) in cases.GEN_CASES.items():
        seq = cases.gen_sequence(name)
        imp = ns["InferenceImp"](None, m, "generate", temp_text, k_text, temp, k, "TTS")
        torch.manual_seed(seed)
        codes = imp(seq.clone(), torch.ones_like(seq))
        n = L - n_text
        torch.manual_seed(seed)     # the same draws again, in the order the loop made them: text, then 8 audio, per frame
        nt, na = [], []
        for _ in range(n):
            nt.append(torch.empty(1, k_text).exponential_(1))
            na.append(torch.stack([torch.empty(1, k).exponential_(1) for _ in range(8)]))
        nt, na = torch.stack(nt), torch.stack(na)      # [n, 1, k_text], [n, 8, 1, k]
        ref = GG.generate(sd, ocfg, seq, "TTS", temp=temp, top_k=k, temp_text=temp_text, top_k_text=k_text,
                          noise=lambda kind, g, l: nt[g].view(1, 1, -1) if kind == "text" else na[g, l].view(1, 1, 1, -1))
        assert torch.equal(ref["codes"], codes), (name, ref["codes"], codes)
        out[f"{name}.codes"] = codes.numpy().astype(np.int32)
        out[f"{name}.noise_text"], out[f"{name}.noise_audio"] = nt[:, 0].numpy(), na[:, :, 0].numpy()
        print("gpt_generate", name, tuple(codes.shape), codes[:, :3].tolist())
    np.savez_compressed(os.path.join(HERE, "gpt_generate.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["layers", "rvq", "transformer", "mimi_e2e", "lm_tiny", "gpt_tiny", "sampling", "reverse_delay",
                             "gpt_generate", "tokenizer", "lm_tiny_sampling", "mimi_model", "mimi_stream_long", "transformer_dims"]
    for w in which:
        globals()[f"gen_{w}"]()
