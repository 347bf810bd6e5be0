"""End-to-end streaming loop (rstnet_amd.pipeline) against the composed CPU oracles: Mimi encode -> LMGen (greedy) -> Mimi
decode, frame by frame on the GPU vs batch oracle encode + oracle LMGen + batch oracle decode."""
import pytest
import torch

from oracle import lm_oracle as L
from oracle import mimi_oracle as O
from rstnet_amd import synth
from rstnet_amd.codec.mimi import MimiCodec
from rstnet_amd.lm.model import LMGen, LMModel
from rstnet_amd.pipeline import StreamingPipeline

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_streaming_pipeline_matches_composed_oracles():
    B, frames = 2, 6
    cfg = dict(synth.LM_TINY, card=2048, n_q=16, dep_q=8, delays=[0, 0, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1])
    mimi_sd = synth.mimi_state_dict(0)
    lm_sd = synth.lm_state_dict(cfg, seed=9)
    mimi = MimiCodec.from_state_dict(mimi_sd).to(DEV)
    model = LMModel.from_state_dict({k: v.to(DEV) for k, v in lm_sd.items()}, cfg)
    gen = LMGen(model, use_sampling=False)
    pcm = synth.synth_audio(B, frames * 1920, seed=21)
    outs = []
    with StreamingPipeline(mimi, gen, B) as pipe:
        for f in range(frames):
            o = pipe.step(pcm[:, :, f * 1920:(f + 1) * 1920].contiguous().to(DEV))
            outs.append(o)
    assert outs[0] is None and all(o is not None and o.shape == (B, 1, 1920) for o in outs[1:])
    got = torch.cat([o.cpu() for o in outs[1:]], -1)

    mcfg = O.MimiConfig()
    with torch.no_grad():
        codes = O.encode(mimi_sd, mcfg, pcm)                     # [B, 8, frames] (streaming == batch, tested elsewhere)
        og = L.LMGenOracle({k: v.float() for k, v in lm_sd.items()}, L.LMConfig(**cfg), B)
        toks = [og.step(codes[:, :, f:f + 1]) for f in range(frames)]
        gen_codes = torch.cat([t[:, 1:] for t in toks[1:]], -1)  # [B, 8, frames-1]
        ref = O.decode(mimi_sd, mcfg, gen_codes)
    assert got.shape == ref.shape
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 1e-3, err
