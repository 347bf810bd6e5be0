"""The websocket loop on the GPU: 20 frames of PCM through rstnet_amd.server.ServerState (real Mimi codec + the tiny 16-stream LM,
greedy) must give (1) the audio the composed CPU oracles give for the same input -- oracle Mimi encode -> oracle LMGen -> oracle Mimi
decode, the parity statement (moshi/server.py:122-136 is exactly that composition) -- and (2) sample for sample what the
StreamingPipeline gives: the server adds transport and framing, no arithmetic."""
import asyncio

import numpy as np
import pytest
import torch

from oracle import lm_oracle as L
from oracle import mimi_oracle as O
from rstnet_amd import server as S
from rstnet_amd import synth
from rstnet_amd.codec.loaders import get_mimi
from rstnet_amd.lm.model import LMGen, LMModel
from rstnet_amd.pipeline import StreamingPipeline

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FRAME = 1920


def test_server_session_equals_pipeline():
    from aiohttp.test_utils import TestClient, TestServer
    frames = 20
    cfg = dict(synth.LM_TINY_16Q)
    mimi_sd = synth.mimi_state_dict(0)
    lm_sd = synth.lm_state_dict(cfg, seed=9)
    pcm = synth.synth_audio(1, frames * FRAME, seed=33)

    def models():
        return get_mimi(mimi_sd, device=DEV), LMModel.from_state_dict({k: v.to(DEV) for k, v in lm_sd.items()}, cfg)

    mimi, lm = models()
    want = []
    with StreamingPipeline(mimi, LMGen(lm, use_sampling=False), 1) as pipe:
        for f in range(frames):
            o = pipe.step(pcm[:, :, f * FRAME:(f + 1) * FRAME].contiguous().to(DEV))
            if o is not None:
                want.append(o[0, 0].cpu().numpy())

    async def scenario():
        mimi2, lm2 = models()
        st = S.ServerState(mimi2, lm2, DEV, use_sampling=False)
        st.warmup()                                        # 4 silent frames; the session below starts from reset states
        client = TestClient(TestServer(S.make_app(st)))
        await client.start_server()
        try:
            ws = await client.ws_connect("/api/chat")
            assert (await ws.receive_bytes()) == b"\x00"
            raw = pcm[0, 0].numpy().astype("<f4").tobytes()
            step = 3000 * 4                                # pieces that straddle frame boundaries
            for a in range(0, len(raw), step):
                await ws.send_bytes(b"\x01" + raw[a:a + step])
            got = []
            while len(got) < len(want):
                msg = await asyncio.wait_for(ws.receive_bytes(), timeout=20.0)
                if msg[0] == 1:
                    got.append(np.frombuffer(msg[1:], dtype="<f4"))
            await ws.close()
            return got
        finally:
            await client.close()

    loop = asyncio.new_event_loop()
    try:
        got = loop.run_until_complete(scenario())
    finally:
        loop.close()
    assert len(got) == len(want) == frames - 1
    # (1) against the composed oracles (the session starts from reset states: the warm-up frames leave no trace)
    mcfg = O.MimiConfig()
    with torch.no_grad():
        codes = O.encode(mimi_sd, mcfg, pcm)
        og = L.LMGenOracle({k: v.float() for k, v in lm_sd.items()}, L.LMConfig(**cfg), 1)
        toks = [og.step(codes[:, :, f:f + 1]) for f in range(frames)]
        ref = O.decode(mimi_sd, mcfg, torch.cat([t[:, 1:] for t in toks[1:]], -1))[0, 0].numpy()
    stream = np.concatenate(got)
    assert stream.shape == ref.shape
    err = float(np.abs(stream - ref).max() / np.abs(ref).max())
    assert err < 1e-3, err
    # (2) against the pipeline
    for f, (a, b) in enumerate(zip(got, want)):
        assert a.shape == (FRAME,) and float(np.abs(a - b).max()) <= 1e-5 * max(1.0, float(np.abs(b).max())), f"frame {f}"
