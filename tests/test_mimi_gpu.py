"""GPU parity of the MimiCodec drop-in (module surface) against the reference-generated fixtures and the CPU oracle,
plus mirrors of the reference's own module tests (MLLM_v2/moshi/modules/conv_test.py, seanet_test.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import mimi_oracle as O
from rstnet_amd import ops, synth
from rstnet_amd.codec.conv import StreamingConv1d, StreamingConvTranspose1d
from rstnet_amd.codec.mimi import MimiCodec
from rstnet_amd.codec.seanet import SEANetDecoder, SEANetEncoder, SEANetResnetBlock
from tests.golden import cases

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def mimi():
    sd = synth.mimi_state_dict(cases.MIMI_SEED)
    return sd, MimiCodec.from_state_dict(sd).to(DEV)


@pytest.mark.parametrize("name", list(cases.MIMI_E2E))
def test_encode_decode_matches_reference_fixture(mimi, name):
    sd, model = mimi
    g = np.load(os.path.join(G, "mimi_e2e.npz"))
    B, T, aseed = cases.MIMI_E2E[name]
    audio = synth.synth_audio(B, T, aseed).to(DEV)
    z = model.encode_latent(audio)
    codes = model.encode(audio)
    ref_codes = torch.from_numpy(g[f"{name}.codes"]).long()
    assert codes.dtype == torch.int64 and tuple(codes.shape) == tuple(ref_codes.shape)
    assert rel_err(z.transpose(1, 2), torch.from_numpy(g[f"{name}.latent"])) < 1e-3
    # bit-exact codes: every decision of these fixtures has a top-2 gap >= 1e-5 relative (stored in rel_gap)
    mism = (codes.cpu() != ref_codes)
    assert not mism.any(), f"{int(mism.sum())} code mismatches; min fixture gap {g[f'{name}.rel_gap'].min():.2e}"
    wav = model.decode(ref_codes.to(DEV))
    ref_wav = torch.from_numpy(g[f"{name}.wav"])
    assert tuple(wav.shape) == tuple(ref_wav.shape)
    assert rel_err(wav, ref_wav) < 1e-3


def test_state_dict_keys_match_reference(mimi):
    sd, model = mimi
    assert set(model.state_dict().keys()) == set(sd.keys())


def test_decode_fewer_codebooks(mimi):
    sd, model = mimi
    cfg = O.MimiConfig()
    codes = torch.randint(0, 2048, (2, 8, 5), generator=torch.Generator().manual_seed(3))
    for n in (1, 3, 8):
        zq = model.quantizer.decode_nlc(codes[:, :n].contiguous().to(DEV))
        ref = O.rvq_decode(sd, cfg, codes[:, :n])
        assert rel_err(zq.transpose(1, 2), ref) < 1e-5


def test_streaming_encode_decode_equals_batch(mimi):
    """Frame-by-frame streaming (1920-sample chunks) == one batch call: codes identical, waveform to fp32 round-off."""
    sd, model = mimi
    B, frames = 2, 6
    audio = synth.synth_audio(B, frames * 1920, seed=5).to(DEV)
    codes_full = model.encode(audio)
    wav_full = model.decode(codes_full)
    codes_s, wav_s = [], []
    with model.streaming(B):
        for f in range(frames):
            c = model.encode(audio[:, :, f * 1920:(f + 1) * 1920].contiguous())
            assert c.shape == (B, 8, 1)
            codes_s.append(c)
            wav_s.append(model.decode(c))
    codes_s, wav_s = torch.cat(codes_s, -1), torch.cat(wav_s, -1)
    assert torch.equal(codes_s, codes_full)
    assert rel_err(wav_s, wav_full) < 1e-4
    assert not model.is_streaming


# ---- mirrors of the reference's own unit tests -------------------------------------------------------------------

CONV1D_DATA = [(3, 4, 5, 10, 6), (4, 5, 6, 10, 7), (5, 6, 7, 10, 2), (1, 512, 512, 256, 7)]
CONVTR_DATA = [(3, 4, 5, 10, 6, 1), (4, 5, 6, 10, 7, 2), (5, 6, 7, 10, 4, 3), (1, 512, 512, 256, 7, 2)]


def _init_weights(module, generator):
    for name, p in module.named_parameters():
        if "bias" in name:
            torch.nn.init.constant_(p, 0.0)
        else:
            torch.nn.init.xavier_uniform_(p, generator=generator)


def _close(a, b):
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("batch_size, in_channels, out_channels, seq_len, kernel_size", CONV1D_DATA)
def test_conv1d_causal_and_streaming(batch_size, in_channels, out_channels, seq_len, kernel_size):
    """conv_test.py:63-109: outputs do not change when more input arrives; chunked streaming == full."""
    layer = StreamingConv1d(in_channels, out_channels, kernel_size, causal=True, norm="none", pad_mode="constant")
    _init_weights(layer, torch.Generator().manual_seed(41))
    layer = layer.to(DEV)
    x = torch.rand(batch_size, in_channels, seq_len).to(DEV)
    expected = layer(x)
    for end in range(kernel_size, seq_len + 1, 3):
        actual = layer(x[..., :end].contiguous())
        _close(actual, expected[..., :actual.shape[-1]])
    outs, start = [], 0
    with layer.streaming(batch_size=batch_size):
        for end in range(kernel_size, seq_len + 1):
            outs.append(layer(x[..., start:end].contiguous()))
            start = end
    _close(torch.cat(outs, -1), expected)


@pytest.mark.parametrize("batch_size, in_channels, out_channels, seq_len, kernel_size, stride", CONVTR_DATA)
def test_conv1d_transpose_causal_and_streaming(batch_size, in_channels, out_channels, seq_len, kernel_size, stride):
    """conv_test.py:112-157."""
    layer = StreamingConvTranspose1d(in_channels, out_channels, kernel_size, stride, causal=True, norm="none")
    _init_weights(layer, torch.Generator().manual_seed(41))
    layer = layer.to(DEV)
    x = torch.rand(batch_size, in_channels, seq_len).to(DEV)
    expected = layer(x)
    for end in range(kernel_size, seq_len + 1, 3):
        actual = layer(x[..., :end].contiguous())
        _close(actual, expected[..., :actual.shape[-1]])
    outs, start = [], 0
    with layer.streaming(batch_size=batch_size):
        for end in range(kernel_size, seq_len + 1):
            outs.append(layer(x[..., start:end].contiguous()))
            start = end
    _close(torch.cat(outs, -1), expected)


@pytest.mark.parametrize("dim,T", [(8, 20), (64, 33)])
def test_resnet_block_streaming(dim, T):
    """seanet_test.py:111-160."""
    blk = SEANetResnetBlock(dim, kernel_sizes=[3, 1], dilations=[1, 1], causal=True, pad_mode="constant", compress=2)
    _init_weights(blk, torch.Generator().manual_seed(41))
    blk = blk.to(DEV)
    x = torch.rand(2, dim, T).to(DEV)
    expected = blk(x)
    outs = []
    with blk.streaming(2):
        for t in range(0, T, 4):
            outs.append(blk(x[..., t:t + 4].contiguous()))
    _close(torch.cat(outs, -1), expected)


@pytest.mark.parametrize("dimension,n_filters,ratios,T", [(8, 4, [5], 10), (8, 4, [5], 1), (512, 64, [8, 6, 5, 4], 2), (512, 64, [8, 6, 5, 4], 10)])
def test_nonstreaming_causal_decode(dimension, n_filters, ratios, T):
    """seanet_test.py:163-187: decoding a prefix gives a prefix of the decoded sequence."""
    dec = SEANetDecoder(channels=1, dimension=dimension, n_filters=n_filters, n_residual_layers=1, ratios=ratios,
                        kernel_size=7, residual_kernel_size=3, last_kernel_size=3, causal=True, pad_mode="constant",
                        true_skip=True, compress=2)
    _init_weights(dec, torch.Generator().manual_seed(41))
    dec = dec.to(DEV)
    z = torch.rand(1, dimension, T).to(DEV)
    full = dec(z)
    hop = int(np.prod(ratios))
    assert full.shape == (1, 1, T * hop)
    for end in range(1, T + 1, max(1, T // 3)):
        part = dec(z[..., :end].contiguous())
        _close(part, full[..., :end * hop])


def test_seanet_encoder_streaming_small():
    enc = SEANetEncoder(channels=1, dimension=16, n_filters=4, n_residual_layers=1, ratios=[4, 2], kernel_size=7,
                        residual_kernel_size=3, last_kernel_size=3, causal=True, pad_mode="constant", true_skip=True, compress=2)
    _init_weights(enc, torch.Generator().manual_seed(41))
    enc = enc.to(DEV)
    x = torch.rand(2, 1, 64).to(DEV)
    expected = enc(x)
    outs = []
    with enc.streaming(2):
        for t in range(0, 64, 8):
            outs.append(enc(x[..., t:t + 8].contiguous()))
    _close(torch.cat(outs, -1), expected)


def test_tokenizer_batch_equals_single():
    """MimiTokenizer: ragged zero-padded batches give bit-identical int16 codes to one-utterance calls; detokenize inverts
    the layout (mimi_tokenizer.py:56-82)."""
    from rstnet_amd.codec.tokenizer import MimiTokenizer
    model = MimiCodec.from_state_dict(synth.mimi_state_dict(0)).to(DEV)
    tok = MimiTokenizer(model)
    lens = [24000, 30001, 1920, 5000, 47999]
    wavs = [synth.synth_audio(1, n, seed=30 + i)[0, 0] for i, n in enumerate(lens)]
    single = [tok.tokenize(w[None], 24000) for w in wavs]
    batched = tok.tokenize_batch(wavs, 24000, max_batch_seconds=3.0)
    for n, a, b in zip(lens, single, batched):
        assert a.dtype == torch.int16 and tuple(a.shape) == (8, -(-n // 1920))
        assert torch.equal(a, b), n
    d = tok.tokenize_scp({f"utt{i}": w for i, w in enumerate(wavs)})
    assert list(d) == [f"utt{i}" for i in range(len(wavs))] and torch.equal(d["utt1"], single[1])
    wav = tok.detokenize(single[0])
    assert tuple(wav.shape) == (1, single[0].shape[1] * 1920)
    assert tok.tokenize(single[0][0].long(), 24000).dim() == 1       # 1-D input = offline codes, passed through


def test_tokenizer_matches_reference_fixture():
    """MimiTokenizer.tokenize / tokenize2 / find_length / detokenize against the outputs of the REFERENCE class
    (tests/golden/tokenizer.npz: the class of mimi_tokenizer.py executed unchanged around the real reference MimiCodec, one
    ragged utterance): int16 codes exact, dtypes and shapes, waveform <= 1e-3 relative."""
    from rstnet_amd.codec.tokenizer import MimiTokenizer
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tokenizer.npz"))
    model = MimiCodec.from_state_dict(synth.mimi_state_dict(cases.MIMI_SEED)).to(DEV)
    tok = MimiTokenizer(model)
    B, T, seed = cases.MIMI_E2E["ragged"]
    wav = synth.synth_audio(B, T, seed=seed)[1]
    codes = tok.tokenize(wav, 24000)
    assert codes.dtype == torch.int16 and not codes.is_cuda and torch.equal(codes, torch.from_numpy(g["codes"]))
    ids = tok.tokenize2(codes)
    assert ids.dtype == torch.int64 and torch.equal(ids, torch.from_numpy(g["tokenize2"]))
    assert tok.find_length(codes) == int(g["find_length"])
    out = tok.detokenize(ids)
    want = torch.from_numpy(g["wav"])
    assert out.shape == want.shape and not out.is_cuda
    assert float((out - want).abs().max() / want.abs().max()) < 1e-3
    assert torch.equal(tok.tokenize(codes[0].clone(), 24000), torch.from_numpy(g["passthrough"]))


def test_mimi_model_matches_moshi_fixture():
    """`MimiModel` (the composition form of moshi/models/compression.py, built by `codec.loaders.get_mimi`) against the REAL
    moshi MimiModel (tests/golden/mimi_model.npz): batch encode with 8 and 4 active codebooks, decode of 4, frame-by-frame
    streaming encode / decode."""
    from rstnet_amd.codec.loaders import get_mimi
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mimi_model.npz"))
    m = get_mimi(synth.mimi_state_dict(cases.MIMI_SEED), device=DEV)
    B, T, seed = cases.MIMI_E2E["ragged"]
    audio = synth.synth_audio(B, T, seed=seed).to(DEV)
    assert torch.equal(m.encode(audio).cpu(), torch.from_numpy(g["codes8"]).long())
    m.set_num_codebooks(4)
    c4 = m.encode(audio)
    assert m.num_codebooks == 4 and torch.equal(c4.cpu(), torch.from_numpy(g["codes4"]).long())
    w4, want = m.decode(c4).cpu(), torch.from_numpy(g["wav4"])
    assert w4.shape == want.shape and float((w4 - want).abs().max() / want.abs().max()) < 1e-3
    m.set_num_codebooks(8)
    a1 = synth.synth_audio(1, 1920 * 12, seed=cases.MIMI_E2E["cfg1"][2]).to(DEV)
    cs, ws = [], []
    with m.streaming(1):
        for f in range(12):
            c = m.encode(a1[:, :, f * 1920:(f + 1) * 1920].contiguous())
            cs.append(c)
            ws.append(m.decode(c))
    assert torch.equal(torch.cat(cs, -1).cpu(), torch.from_numpy(g["stream_codes"]).long())
    sw, want = torch.cat(ws, -1).cpu(), torch.from_numpy(g["stream_wav"])
    assert sw.shape == want.shape and float((sw - want).abs().max() / want.abs().max()) < 1e-3


def test_mimi_model_32_trained_8_active():
    """The canonical checkpoint layout (32 trained codebooks, 8 active -- loaders.py:44-50,138): extra trained levels must not
    change the 8-level result, and `set_num_codebooks` widens it."""
    from rstnet_amd.codec.loaders import get_mimi
    sd8 = synth.mimi_state_dict(cases.MIMI_SEED)
    g = torch.Generator().manual_seed(5)
    sd32 = dict(sd8)
    for lvl in range(7, 31):
        for k in ("embedding_sum", "cluster_usage", "_initialized"):
            src = sd8[f"quantizer.rvq_rest.vq.layers.6._codebook.{k}"]
            sd32[f"quantizer.rvq_rest.vq.layers.{lvl}._codebook.{k}"] = (torch.randn(src.shape, generator=g) * 0.05).to(src.dtype) \
                if k == "embedding_sum" else src.clone()
    m8, m32 = get_mimi(sd8, device=DEV), get_mimi(sd32, device=DEV)
    assert (m32.total_codebooks, m32.num_codebooks) == (32, 8)
    audio = synth.synth_audio(2, 1920 * 5 + 700, seed=3).to(DEV)
    c8, c32 = m8.encode(audio), m32.encode(audio)
    assert c32.shape == c8.shape and torch.equal(c32, c8)
    assert torch.equal(m32.decode(c32), m8.decode(c8))
    with m32.streaming(2):
        cs = torch.cat([m32.encode(audio[:, :, f * 1920:(f + 1) * 1920].contiguous()) for f in range(5)], -1)
    assert torch.equal(cs, c8[:, :, :5])
    m32.set_num_codebooks(12)
    c12 = m32.encode(audio)
    assert c12.shape[1] == 12 and torch.equal(c12[:, :8], c8) and int(c12.max()) < 2048
    assert m32.decode(c12).shape == (2, 1, 6 * 1920)


def test_offline_tokenization_cli(tmp_path):
    """tools/offline_codec_tokenization.py end to end: PCM WAV list in, {utt: int16 codes} .pt out, equal to one-by-one
    tokenisation of the same (16-bit quantised) waveforms; a 16 kHz file is resampled to 24 kHz on the host first."""
    import importlib.util
    import wave
    from safetensors.torch import save_file
    from rstnet_amd.codec import offline
    from rstnet_amd.codec.loaders import get_mimi
    from rstnet_amd.codec.tokenizer import MimiTokenizer
    sd = synth.mimi_state_dict(cases.MIMI_SEED)
    wpath = os.path.join(tmp_path, "mimi.safetensors")
    save_file({k: v.contiguous() for k, v in sd.items()}, wpath)
    lens, lines = [24000, 5003, 41000], []
    for i, n in enumerate(lens + [8000]):
        x = synth.synth_audio(1, n, seed=60 + i)[0, 0].numpy()
        path = os.path.join(tmp_path, f"u{i}.wav")
        with wave.open(path, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(24000 if i < 3 else 16000)
            w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())
        lines.append(f"utt{i} {path}")
    scp = os.path.join(tmp_path, "wav.1.scp")
    with open(scp, "w") as f:
        f.write("\n".join(lines) + "\n")
    spec = importlib.util.spec_from_file_location("offline_cli", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools",
                                                                                "offline_codec_tokenization.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    out = os.path.join(tmp_path, "codec.1.pt")
    cli.main(["--input-file", scp, "--output-file", out, "--tokenizer", "mimi", "--rank", "1", "--weights", wpath, "--batch-seconds", "2.5"])
    data = torch.load(out)
    assert list(data) == ["utt0", "utt1", "utt2", "utt3"]
    assert data["utt3"].dtype == torch.int16 and tuple(data["utt3"].shape) == (8, -(-8000 * 3 // 2 // 1920))   # 16 kHz -> 24 kHz
    tok = MimiTokenizer(get_mimi(sd, DEV))
    for i, n in enumerate(lens):
        wav, sr = offline.read_audio(os.path.join(tmp_path, f"u{i}.wav"))
        want = tok.tokenize(wav[None], sr)
        assert data[f"utt{i}"].dtype == torch.int16 and torch.equal(data[f"utt{i}"], want), i


@pytest.mark.parametrize("b3", [True, False], ids=["three_plane_bf16", "f32_instruction"])
def test_encode_decode_8x10s_full_size_kernels_vs_oracle(mimi, b3, monkeypatch):
    """Eight 10 s clips through encode + decode against the CPU oracle: at this size every GEMM of the SEANet stacks runs the
    kernel instances of the headline benchmark (the three-plane bf16 GEMM for every unfused conv / linear, fused res-blocks at
    240 000 steps), which the one-second fixtures never reach -- and once more with the large GEMMs kept on the f32 matrix
    instruction (`ops.GEMM_B3 = False`).  Codes must equal the oracle's wherever the oracle's own top-2 gap is not a near tie."""
    from tests.parity import codes_match_up_to_near_ties
    monkeypatch.setattr(ops, "GEMM_B3", b3)
    sd, model = mimi
    cfg = O.MimiConfig()
    B, T = 8, 240000
    audio = synth.synth_audio(B, T, seed=77)
    codes = model.encode(audio.to(DEV))
    with torch.no_grad():
        z = O.encode_latent(sd, cfg, audio)
        ref_codes = O.rvq_encode(sd, cfg, z)
        # top-2 relative gap of every oracle decision (as make_golden.py records it for the fixtures)
        gaps = []
        for p, n_q in (("quantizer.rvq_first", 1), ("quantizer.rvq_rest", 7)):
            r = torch.nn.functional.conv1d(z, sd[f"{p}.input_proj.weight"]).transpose(1, 2).reshape(-1, 256)
            for j in range(n_q):
                emb = O.codebook(sd, f"{p}.vq.layers.{j}")
                t2 = torch.cdist(r[None], emb[None])[0].topk(2, largest=False)
                gaps.append(((t2.values[:, 1] - t2.values[:, 0]) / t2.values[:, 0]).view(B, -1))
                r = r - emb[t2.indices[:, 0]]
        gaps = torch.stack(gaps, 1)
    assert codes.shape == ref_codes.shape == (B, 8, 125)
    excused = codes_match_up_to_near_ties(codes.cpu(), ref_codes, gaps)
    match = float((codes.cpu() == ref_codes).float().mean())
    print(f"8 x 10 s (GEMM_B3={b3}): code exact-match {match:.6f}; {excused} frames differ, all at decisions with a top-2 gap < 2e-5 "
          f"(min gap {float(gaps.min()):.2e})")
    assert match > 0.999
    wav = model.decode(ref_codes.to(DEV))
    with torch.no_grad():
        ref_wav = O.decode(sd, cfg, ref_codes)
    assert wav.shape == ref_wav.shape == (B, 1, 240000)
    assert rel_err(wav, ref_wav) < 1e-3


def _oracle_gaps(sd, z, B):
    """Top-2 relative gap of every RVQ decision of the oracle for latent `z` (as make_golden.py records it for the fixtures)."""
    gaps = []
    for p, n_q in (("quantizer.rvq_first", 1), ("quantizer.rvq_rest", 7)):
        r = torch.nn.functional.conv1d(z, sd[f"{p}.input_proj.weight"]).transpose(1, 2).reshape(-1, 256)
        for j in range(n_q):
            emb = O.codebook(sd, f"{p}.vq.layers.{j}")
            t2 = torch.cdist(r[None], emb[None])[0].topk(2, largest=False)
            gaps.append(((t2.values[:, 1] - t2.values[:, 0]) / t2.values[:, 0]).view(B, -1))
            r = r - emb[t2.indices[:, 0]]
    return torch.stack(gaps, 1)


def test_encode_decode_64x10s_headline_batch_vs_oracle(mimi):
    """BASELINE configs[1] at its real size: the 64 x 10 s batch of the headline benchmark through encode + decode on the GPU, clips
    0 / 31 / 63 against the CPU oracle.  At this batch the 25 / 12.5 Hz layers (M = 16 000 / 8 000 rows: both transformers' linears, the
    512 -> 1024 k16 convolution, the 1024 -> 512 transposed convolution, the k3 / k7 convolutions around them) run the three-plane bf16
    kernel -- at 8 clips (the test above) they have fewer than 4096 rows and stay on the f32 instruction.  The profile rows are labelled
    from the library's own route predicate (`ops._b3_route`; rst_gemm_win_b3_f32 refuses anything it would not run), so the assertion
    names the kernel that ran."""
    from tests.parity import codes_match_up_to_near_ties
    sd, model = mimi
    cfg = O.MimiConfig()
    B, T = 64, 240000
    audio = synth.synth_audio(B, T, seed=100)
    audio_dev = audio.to(DEV)
    ops.PROFILE = []
    codes = model.encode(audio_dev)
    enc_rows, ops.PROFILE = ops.PROFILE, []
    wav = model.decode(codes)
    torch.cuda.synchronize()
    dec_rows, ops.PROFILE = ops.PROFILE, None
    assert codes.shape == (B, 8, 125) and wav.shape == (B, 1, T)
    for what, rows in (("encode", enc_rows), ("decode", dec_rows)):
        low = [(r[0], r[5]) for r in rows if r[0].startswith("gemm_win") and r[5][0] in (16000, 8000)]
        assert len(low) >= 33, (what, len(low))          # 8 layers x 4 linears + the convolutions next to the transformer
        f32 = [(n, s) for n, s in low if n != "gemm_win_b3"]
        # the only 25 / 12.5 Hz GEMM left on the f32 instruction: the 512-channel downsample convolution (replicate padding; encode)
        assert all(s == (8000, 512, 2048) for _, s in f32) and len(f32) <= 1, (what, f32)
        assert not [r for r in rows if r[0] == "gemm_win_b3" and r[5][0] <= 4096]
    sel = [0, 31, 63]
    with torch.no_grad():
        z = O.encode_latent(sd, cfg, audio[sel])
        ref_codes = O.rvq_encode(sd, cfg, z)
        gaps = _oracle_gaps(sd, z, len(sel))
        ref_wav = O.decode(sd, cfg, ref_codes)
    got = codes[sel].cpu()
    excused = codes_match_up_to_near_ties(got, ref_codes, gaps)
    match = float((got == ref_codes).float().mean())
    print(f"64 x 10 s, clips {sel}: code exact-match {match:.6f}; {excused} frames differ at oracle top-2 gaps < 2e-5 (min gap {float(gaps.min()):.2e})")
    assert match > 0.999
    # decode at B = 64 of the GPU's own codes: comparable with the oracle's waveform for every clip whose codes are the oracle's
    same = [i for i, b in enumerate(sel) if torch.equal(got[i], ref_codes[i])]
    assert same, "no selected clip reproduced the oracle's codes exactly"
    err = rel_err(wav[sel].cpu()[same], ref_wav[same])
    print(f"64 x 10 s decode (B = 64 plan), clips {[sel[i] for i in same]}: wav rel err {err:.2e}")
    assert err < 1e-3
    # and the decode of the oracle's codes inside a 64-clip batch (rows 0 / 31 / 63 replaced): every selected clip
    codes2 = codes.clone()
    codes2[sel] = ref_codes.to(DEV)
    wav2 = model.decode(codes2)
    assert rel_err(wav2[sel].cpu(), ref_wav) < 1e-3


def test_audiocodec_twin_code_layout(mimi):
    """The AudioCodec/MimiCodec twin (AudioCodec/MimiCodec/models/MimiCodec.py:94-111) moves codes as [B, T, K] (vq_dc.py:148-162
    concatenates the per-level indices on the last axis): `code_layout="btk"` gives exactly the transposed tensors, both ways,
    also frame by frame."""
    sd, model = mimi
    twin = MimiCodec.from_state_dict(sd, code_layout="btk").to(DEV)
    audio = synth.synth_audio(2, 1920 * 4 + 500, seed=17).to(DEV)
    codes = model.encode(audio)
    codes_t = twin.encode(audio)
    assert codes_t.shape == (2, 5, 8) and codes_t.is_contiguous() and torch.equal(codes_t, codes.transpose(1, 2))
    assert torch.equal(twin.decode(codes_t), model.decode(codes))
    with twin.streaming(2):
        c0 = twin.encode(audio[:, :, :1920].contiguous())
        assert c0.shape == (2, 1, 8) and torch.equal(c0[:, 0], codes[:, :, 0])
        assert twin.decode(c0).shape == (2, 1, 1920)
    with pytest.raises(ValueError):
        MimiCodec(code_layout="tkb")


@pytest.mark.parametrize("dim,B,chunk", [(64, 2, 1920), (128, 1, 480), (128, 3, 7), (256, 2, 96)])
def test_resnet_block_streaming_equals_batch(dim, B, chunk):
    """SEANetResnetBlock in streaming mode (one fused launch per chunk for C = 64 / 128: the k3 convolution's two-step history goes
    straight into rst_seanet_resblock_f32) == the non-streaming block."""
    blk = SEANetResnetBlock(dim, kernel_sizes=[3, 1], dilations=[1, 1], causal=True, pad_mode="constant", compress=2)
    _init_weights(blk, torch.Generator().manual_seed(41))
    for name, p in blk.named_parameters():
        if "bias" in name:
            torch.nn.init.normal_(p, std=0.05, generator=torch.Generator().manual_seed(7))
    blk = blk.to(DEV)
    x = (torch.rand(B, dim, 4 * chunk + 3) * 2 - 1).to(DEV)
    expected = blk(x)
    outs = []
    with blk.streaming(B):
        for s in range(0, x.shape[-1], chunk):
            outs.append(blk(x[..., s:s + chunk].contiguous()))
    _close(torch.cat(outs, -1), expected)
