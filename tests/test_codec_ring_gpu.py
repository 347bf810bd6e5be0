"""GPU parity of the CODEC's streaming attention past the ring wrap (VERDICT r1 "what's weak" #1; SURVEY fixture F4).

The encoder / decoder transformers of Mimi keep 250-slot KV rings (context 250 at 25 Hz): after 125 frames = 10 s of audio
-- every real session -- the ring wraps and `RingKVCache.complete` (modules/transformer.py:211-278) hides the slot at
`end_index` (`delta <= 0`).  Three levels:
  * the kernels (`rst_rope_split_f32` ring append, `rst_attn_decode_multi_f32`, `rst_attention_f32` ring walk) step by step
    against the oracle's ring for capacities 8 / 10 / 250 and more than two wraps;
  * `ProjectedTransformer` on the reference fixture `transformer.npz` (T = 300 > context) and streamed in chunks against the
    oracle's `TransformerStream`;
  * the whole codec streamed 150 frames (batch 2) against the REAL moshi MimiModel's outputs (`mimi_stream_long.npz`).
"""
import os

import numpy as np
import pytest
import torch

from oracle import mimi_oracle as O
from rstnet_amd import ops, synth
from rstnet_amd.codec.mimi import MimiCodec
from tests.golden import cases
from tests.parity import codes_match_up_to_near_ties

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("H,D,cap,context,chunks", [
    (8, 64, 250, 250, [2] * 270),                     # the codec's own shape: 540 positions = two wraps and a bit
    (8, 64, 250, 250, [1, 3, 2, 5, 8, 7, 4, 6] * 16),  # ragged chunk sizes (all on the few-query route)
    (8, 64, 250, 250, [12, 2, 30, 9, 64, 2] * 5),      # chunks > 8 queries: the tile-walking ring kernel
    (2, 64, 8, 8, [1] * 20 + [2] * 6 + [3] * 4),
    (4, 64, 10, 10, [2] * 14 + [1] * 5),
    (2, 128, 10, 7, [1, 2, 3] * 6),                    # context shorter than the ring
    (2, 64, 16, None, [1] * 40),                       # no context limit: only the ring bounds the window
])
def test_codec_ring_attention_across_wraps(H, D, cap, context, chunks):
    """ops.rope_split(ring=True) + ops.attention(ring=True) step by step vs RingKVCache.complete + the position mask."""
    g = torch.Generator().manual_seed(H * D + cap)
    B = 2
    ring = O.RingKVCache(B, H, D, cap)
    kc = torch.zeros(B, H, cap, D, device=DEV)
    vc = torch.zeros(B, H, cap, D, device=DEV)
    pos = torch.zeros(1, dtype=torch.long, device=DEV)
    offset = 0
    assert sum(chunks) > 2 * cap
    for T in chunks:
        qkv = torch.randn(B, T, 3 * H * D, generator=g)
        q, k, v = qkv.view(B, T, 3, H, D).permute(2, 0, 3, 1, 4)
        ref = O.ring_attention(q, k, v, ring, offset, context, 10000.0)
        qg, _, _ = ops.rope_split(qkv.to(DEV), H, k=kc, v=vc, pos0=offset, pos_dev=pos, ring=True)
        out = ops.attention(qg, kc, vc, pos0=offset, pos_dev=pos, ring=True, context=context)
        pos.add_(T)
        offset += T
        assert rel_err(out, ref) < 1e-4, f"offset {offset} (T={T})"
    assert rel_err(kc, ring.k) < 1e-4 and torch.equal(vc.cpu(), ring.v)


def _transformer(sd, prefix="encoder_transformer"):
    m = MimiCodec.from_state_dict(sd).to(DEV)
    return m, getattr(m, prefix)


def test_projected_transformer_matches_reference_fixture():
    """tests/golden/transformer.npz: the REFERENCE's ProjectedTransformer on T = 300 positions (context 250, LayerScale 0.25)."""
    sd = synth.mimi_state_dict(cases.MIMI_SEED, layer_scale=cases.TRANSFORMER_LAYER_SCALE)
    _, tr = _transformer(sd)
    x = cases.transformer_input()
    ref = torch.from_numpy(np.load(os.path.join(G, "transformer.npz"))["y"])
    y = tr(x.to(DEV))[0]
    assert tuple(y.shape) == tuple(ref.shape) and rel_err(y, ref) < 1e-3
    assert rel_err(y, O.projected_transformer(sd, "encoder_transformer", O.MimiConfig(), x)) < 1e-3


@pytest.mark.parametrize("chunk,persistent", [(2, True), (1, True), (2, False), (1, False), (5, True), (16, True)])
def test_projected_transformer_streamed_across_the_wrap(chunk, persistent, monkeypatch):
    """Streamed in chunks for 300 positions (the ring wraps at 250) vs the oracle's TransformerStream.  Chunks of 1 / 2 positions
    (x 2 streams = 2 / 4 rows) run all 8 layers as ONE persistent launch (rst_codec_transformer_frame) unless RST_DEPTH_FRAME=0
    selects the launch-per-op layer loop; longer chunks always take the loop."""
    monkeypatch.setenv("RST_DEPTH_FRAME", "1" if persistent else "0")
    sd = synth.mimi_state_dict(cases.MIMI_SEED, layer_scale=cases.TRANSFORMER_LAYER_SCALE)
    model, tr = _transformer(sd)
    x = cases.transformer_input(batch=2)
    ts = O.TransformerStream(sd, "encoder_transformer", O.MimiConfig(), 2)
    worst = 0.0
    with tr.streaming(2):
        for i in range(0, x.shape[-1], chunk):
            xc = x[:, :, i:i + chunk].contiguous()
            with torch.no_grad():
                ref = ts.step(xc)
            y = tr(xc.to(DEV))[0]
            worst = max(worst, rel_err(y, ref))
    assert worst < 1e-3, worst
    assert ops.codec_transformer_status(torch.device(DEV)).tolist()[:3] == [0, 0, 0], "a hand-off of the persistent transformer launch timed out"


@pytest.mark.parametrize("streams,chunk,one_launch", [(5, 2, True), (5, 2, False), (3, 3, True), (9, 1, True)])
def test_projected_transformer_many_streams_across_the_wrap(streams, chunk, one_launch, monkeypatch):
    """More than two streams per step: the layer loop on the few-row route (round 6: every attention step as ONE launch,
    rst_attention_step_f32, and the out-projection reading its rows in place, rst_linear_few_rows_f32 -- or the launches they replace
    with the switches off) vs the oracle's TransformerStream for 300 positions (the ring wraps at 250)."""
    monkeypatch.setattr(ops, "ATTENTION_STEP", one_launch)
    monkeypatch.setattr(ops, "SKINNY_F32_ROWS", one_launch)
    sd = synth.mimi_state_dict(cases.MIMI_SEED, layer_scale=cases.TRANSFORMER_LAYER_SCALE)
    model, tr = _transformer(sd)
    x = cases.transformer_input(batch=streams)
    ts = O.TransformerStream(sd, "encoder_transformer", O.MimiConfig(), streams)
    worst = 0.0
    with tr.streaming(streams):
        for i in range(0, x.shape[-1], chunk):
            xc = x[:, :, i:i + chunk].contiguous()
            with torch.no_grad():
                ref = ts.step(xc)
            y = tr(xc.to(DEV))[0]
            worst = max(worst, rel_err(y, ref))
    assert worst < 1e-3, worst


def test_codec_transformer_frame_mixed_with_layer_loop():
    """One session that alternates between the persistent launch (chunks of 2 positions) and the layer loop (a chunk of 7): both
    append to the same rings and advance the same position counter."""
    sd = synth.mimi_state_dict(cases.MIMI_SEED, layer_scale=cases.TRANSFORMER_LAYER_SCALE)
    model, tr = _transformer(sd, "decoder_transformer")
    x = cases.transformer_input(batch=1, frames=40)
    ts = O.TransformerStream(sd, "decoder_transformer", O.MimiConfig(), 1)
    worst, i = 0.0, 0
    with tr.streaming(1):
        for n in (2, 2, 7, 2, 1, 7, 2, 2, 3, 2, 4, 6):
            xc = x[:, :, i:i + n].contiguous()
            i += n
            with torch.no_grad():
                ref = ts.step(xc)
            worst = max(worst, rel_err(tr(xc.to(DEV))[0], ref))
    assert i == 40 and worst < 1e-3, worst
    assert ops.codec_transformer_status(torch.device(DEV)).tolist()[:3] == [0, 0, 0]


def test_mimi_long_stream_matches_moshi_fixture():
    """150 streamed frames, batch 2, against the real moshi MimiModel (tests/golden/mimi_stream_long.npz): eager and
    graph-replayed encode give the same codes, codes equal the reference's (near ties excused by the recorded top-2 gaps),
    latent and waveform past the wrap (frame 125) within 1e-3."""
    sd = synth.mimi_state_dict(cases.MIMI_SEED, layer_scale=cases.TRANSFORMER_LAYER_SCALE)
    model = MimiCodec.from_state_dict(sd).to(DEV)
    g = np.load(os.path.join(G, "mimi_stream_long.npz"))
    B, frames, seed = cases.MIMI_STREAM_LONG
    tail = cases.MIMI_STREAM_LONG_TAIL
    audio = synth.synth_audio(B, 1920 * frames, seed=seed).to(DEV)
    ref_codes = torch.from_numpy(g["codes"]).long()
    zs, cs_eager = [], []
    with model.streaming(B):
        for f in range(frames):
            z = model.encode_latent(audio[:, :, f * 1920:(f + 1) * 1920].contiguous())
            zs.append(z)
            cs_eager.append(model.quantizer.encode_nlc(z))
    z = torch.cat(zs, 1).transpose(1, 2)            # [B, 512, frames]
    assert rel_err(z[:, :, -tail:], torch.from_numpy(g["latent_tail"])) < 1e-3
    cs, ws = [], []
    with model.streaming(B):
        for f in range(frames):
            cs.append(model.encode(audio[:, :, f * 1920:(f + 1) * 1920].contiguous()))
            ws.append(model.decode(ref_codes[:, :, f:f + 1].contiguous().to(DEV)))
    codes, wav = torch.cat(cs, -1).cpu(), torch.cat(ws, -1).cpu()
    assert torch.equal(codes, torch.cat(cs_eager, -1).cpu()), "graph-replayed frames differ from eager frames"
    excused = codes_match_up_to_near_ties(codes, ref_codes, torch.from_numpy(g["rel_gap"]))
    n_near = int((g["rel_gap"] < 2e-5).sum())
    print(f"long stream: {int((codes != ref_codes).sum())} of {codes.numel()} code entries differ in {excused} frames, all at recorded "
          f"near ties ({n_near} decisions of the fixture have a top-2 gap < 2e-5)")
    assert excused <= n_near
    assert rel_err(wav[:, :, :1920 * 4], torch.from_numpy(g["wav_head"])) < 1e-3
    assert rel_err(wav[:, :, -1920 * tail:], torch.from_numpy(g["wav_tail"])) < 1e-3
    assert ops.codec_transformer_status(torch.device(DEV)).tolist()[:3] == [0, 0, 0]
