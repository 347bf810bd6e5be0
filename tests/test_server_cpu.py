"""The websocket loop of rstnet_amd/server.py (the role of MLLM_v2/moshi/server.py:44-166) driven on the CPU with stand-in models:
handshake, raw-PCM framing across message boundaries, silent first frames, text pieces, the one-session lock, warm-up."""
import asyncio

import numpy as np
import pytest
import torch

from rstnet_amd import server as S

FRAME = 1920


class _Lm:
    dep_q, num_codebooks = 8, 17


class _FakeMimi:
    sample_rate, frame_rate = 24000, 12.5

    def __init__(self):
        self.encoded, self.resets, self.forever = [], 0, None

    def streaming_forever(self, b):
        self.forever = b

    def reset_streaming(self):
        self.resets += 1

    def encode(self, chunk):
        assert chunk.shape == (1, 1, FRAME)
        self.encoded.append(float(chunk.sum()))
        return torch.full((1, 8, 1), len(self.encoded), dtype=torch.long)

    def decode(self, tokens):
        assert tokens.shape == (1, 8, 1)
        return torch.full((1, 1, FRAME), float(tokens[0, 0, 0]) / 100.0)


class _FakeLMGen:
    """None for the first frame (max_delay = 1), then [text, 8 audio] tokens derived from the frame counter."""
    lm_model = _Lm()

    def __init__(self):
        self.steps, self.resets, self.forever = 0, 0, None

    def streaming_forever(self, b):
        self.forever = b

    def reset_streaming(self):
        self.resets += 1
        self.steps = 0

    def step(self, codes):
        assert codes.shape == (1, 8, 1)
        self.steps += 1
        if self.steps == 1:
            return None
        t = torch.full((1, 9, 1), int(codes[0, 0, 0]), dtype=torch.long)
        t[0, 0, 0] = 3 if self.steps % 2 else 40 + self.steps          # 3 = padding id: no text message
        return t


def _state():
    return S.ServerState(_FakeMimi(), None, "cpu", lm_gen=_FakeLMGen())


def test_pcm_framer_f32_and_s16():
    f = S.PcmFramer(4, "f32")
    x = np.arange(10, dtype=np.float32)
    raw = x.astype("<f4").tobytes()
    f.append_bytes(raw[:7])                       # a partial sample
    assert f.frames() == []
    f.append_bytes(raw[7:])
    fr = f.frames()
    assert len(fr) == 2 and np.array_equal(np.concatenate(fr), x[:8]) and f.frames() == []
    f.append_bytes(np.arange(10, 12, dtype="<f4").tobytes())
    assert np.array_equal(f.frames()[0], x[8:].tolist() + [10.0, 11.0])
    g = S.PcmFramer(3, "s16")
    g.append_bytes(np.array([0, 16384, -32768], dtype="<i2").tobytes())
    assert np.allclose(g.frames()[0], [0.0, 0.5, -1.0])
    assert g.encode(np.array([0.0, 0.5, -1.0, 2.0])) == np.array([0, 16383, -32767, 32767], dtype="<i2").tobytes()
    with pytest.raises(ValueError):
        S.PcmFramer(3, "opus")


def test_warmup_runs_four_silent_frames():
    st = _state()
    assert st.frame_size == FRAME and st.mimi.forever == 1 and st.lm_gen.forever == 1
    st.warmup()
    assert st.mimi.encoded == [0.0] * 4 and st.lm_gen.steps == 4


def test_text_pieces():
    st = _state()
    assert st.text_piece(0) is None and st.text_piece(3) is None and st.text_piece(41) == "<41>"

    class Tok:
        def id_to_piece(self, i):
            return "▁hello"
    st.text_tokenizer = Tok()
    assert st.text_piece(41) == " hello"


def _run(coro):
    loop = asyncio.new_event_loop()
    try:
        return loop.run_until_complete(coro)
    finally:
        loop.close()


def test_websocket_session_frames_text_and_lock():
    from aiohttp import WSMsgType
    from aiohttp.test_utils import TestClient, TestServer

    async def scenario():
        st = _state()
        client = TestClient(TestServer(S.make_app(st)))
        await client.start_server()
        try:
            ws = await client.ws_connect("/api/chat")
            assert (await ws.receive_bytes()) == b"\x00"                 # handshake
            assert st.mimi.resets == 1 and st.lm_gen.resets == 1         # states reset at the start of the session
            # a second client must wait for the lock: no handshake while the first session is alive
            ws2 = await client.ws_connect("/api/chat?pcm=s16")
            with pytest.raises(asyncio.TimeoutError):
                await asyncio.wait_for(ws2.receive_bytes(), timeout=0.1)
            # three frames of audio in pieces that do not line up with frame boundaries
            pcm = (np.arange(3 * FRAME, dtype=np.float32) % 7 - 3) / 10.0
            raw = pcm.astype("<f4").tobytes()
            cuts = [0, 1000, 1000 + 4 * FRAME + 2, len(raw)]
            for a, b in zip(cuts[:-1], cuts[1:]):
                await ws.send_bytes(b"\x01" + raw[a:b])
            await ws.send_bytes(b"\x07junk")                             # unknown kind: ignored
            got = []
            while len([m for m in got if m[0] == 1]) < 2:
                msg = await asyncio.wait_for(ws.receive(), timeout=2.0)
                assert msg.type == WSMsgType.BINARY
                got.append(msg.data)
            # frame 1 is swallowed (LMGen returns None), frames 2 and 3 come back as f32 PCM of value codes / 100
            audio = [np.frombuffer(m[1:], dtype="<f4") for m in got if m[0] == 1]
            assert [a.shape for a in audio] == [(FRAME,), (FRAME,)] and np.allclose(audio[0], 0.02) and np.allclose(audio[1], 0.03)
            text = [m[1:].decode() for m in got if m[0] == 2]
            assert text == ["<42>"]                                      # step 2 -> id 42; step 3 -> padding id 3: silent
            assert st.mimi.encoded == pytest.approx([float(pcm[i * FRAME:(i + 1) * FRAME].sum()) for i in range(3)], abs=1e-3)
            await ws.close()
            # now the second session gets its handshake, with freshly reset states and its own (s16) transport
            assert (await asyncio.wait_for(ws2.receive_bytes(), timeout=2.0)) == b"\x00"
            assert st.mimi.resets == 2 and st.lm_gen.steps == 0
            await ws2.send_bytes(b"\x01" + np.full(2 * FRAME, 8192, dtype="<i2").tobytes())
            msg = await asyncio.wait_for(ws2.receive_bytes(), timeout=2.0)
            assert msg[0] == 1 and len(msg) == 1 + 2 * FRAME             # s16 out as well
            await ws2.close()
        finally:
            await client.close()

    _run(scenario())


def test_opus_framer_seam_with_a_stand_in_sphn(monkeypatch):
    """`?pcm=opus`: the reference's transport (server.py:105-153) at the PcmFramer seam.  `sphn` is absent from the image, so the
    plumbing is exercised with a stand-in module that keeps sphn's call surface (OpusStreamReader.append_bytes / read_pcm,
    OpusStreamWriter.append_pcm / read_bytes) and moves float32 samples unchanged; without any `sphn` the framer refuses loudly."""
    import sys
    import types
    from rstnet_amd import server as S
    monkeypatch.setitem(sys.modules, "sphn", None)          # import sphn -> ImportError
    with pytest.raises(RuntimeError, match="sphn"):
        S.make_framer(4, "opus", 24000)

    class Reader:
        def __init__(self, sr):
            self.sr, self.buf = sr, bytearray()

        def append_bytes(self, b):
            self.buf += b

        def read_pcm(self):
            n = len(self.buf) // 4
            out = np.frombuffer(bytes(self.buf[:4 * n]), dtype="<f4").copy()
            del self.buf[:4 * n]
            return out

    class Writer:
        def __init__(self, sr):
            self.sr, self.out = sr, bytearray()

        def append_pcm(self, pcm):
            self.out += np.asarray(pcm, dtype="<f4").tobytes()

        def read_bytes(self):
            b, self.out = bytes(self.out), bytearray()
            return b
    fake = types.ModuleType("sphn")
    fake.OpusStreamReader, fake.OpusStreamWriter = Reader, Writer
    monkeypatch.setitem(sys.modules, "sphn", fake)
    f = S.make_framer(4, "opus", 24000)
    assert isinstance(f, S.OpusFramer) and isinstance(S.make_framer(4, "s16", 24000), S.PcmFramer)
    x = np.arange(10, dtype=np.float32)
    f.append_bytes(x[:3].tobytes())
    assert f.frames() == []
    f.append_bytes(x[3:].tobytes())
    got = f.frames()
    assert len(got) == 2 and np.array_equal(got[0], x[:4]) and np.array_equal(got[1], x[4:8]) and f.frames() == []
    assert f.encode(x[:4]) == x[:4].tobytes()


def test_bad_transport_is_refused_before_the_upgrade_and_opus_pages_are_paced(monkeypatch):
    """an unknown `?pcm=` value, or `pcm=opus` without `sphn`, is a 400 with the reason -- not an upgraded socket that dies
    with 1011.  With a (stand-in) `sphn` whose writer hands out whole pages only, a frame whose page is not complete yields NO kind-1
    message (server.py:146-150 skips empty payloads); its text message still arrives, and the audio follows with the page."""
    import sys
    import types
    from aiohttp import WSServerHandshakeError
    from aiohttp.test_utils import TestClient, TestServer

    class Reader:
        def __init__(self, sr):
            self.buf = bytearray()

        def append_bytes(self, b):
            self.buf += b

        def read_pcm(self):
            n = len(self.buf) // 4
            out = np.frombuffer(bytes(self.buf[:4 * n]), dtype="<f4").copy()
            del self.buf[:4 * n]
            return out

    class PagedWriter:                       # a page = two frames of samples
        def __init__(self, sr):
            self.out = bytearray()

        def append_pcm(self, pcm):
            self.out += np.asarray(pcm, dtype="<f4").tobytes()

        def read_bytes(self):
            if len(self.out) < 2 * 4 * FRAME:
                return b""
            b, self.out = bytes(self.out[:2 * 4 * FRAME]), self.out[2 * 4 * FRAME:]
            return b

    async def scenario():
        st = _state()
        client = TestClient(TestServer(S.make_app(st)))
        await client.start_server()
        try:
            monkeypatch.setitem(sys.modules, "sphn", None)
            for q, why in (("pcm=mp3", "unknown pcm format"), ("pcm=opus", "sphn")):
                with pytest.raises(WSServerHandshakeError) as e:
                    await client.ws_connect("/api/chat?" + q)
                assert e.value.status == 400
                resp = await client.get("/api/chat?" + q)
                assert resp.status == 400 and why in await resp.text()
            assert st.mimi.resets == 0                                   # no session was started for them
            fake = types.ModuleType("sphn")
            fake.OpusStreamReader, fake.OpusStreamWriter = Reader, PagedWriter
            monkeypatch.setitem(sys.modules, "sphn", fake)
            ws = await client.ws_connect("/api/chat?pcm=opus")
            assert (await ws.receive_bytes()) == b"\x00"
            await ws.send_bytes(b"\x01" + np.zeros(3 * FRAME, dtype="<f4").tobytes())
            got = []
            while not any(m[0] == 1 for m in got):
                got.append(await asyncio.wait_for(ws.receive_bytes(), timeout=2.0))
            # frame 1: LMGen returns None; frame 2: half a page -> text only; frame 3 completes the page -> one kind-1 message of 2 frames
            assert [m[0] for m in got] == [2, 1] and len(got[1]) == 1 + 2 * 4 * FRAME
            await ws.close()
        finally:
            await client.close()

    _run(scenario())


def test_opus_framer_with_the_real_sphn():
    sphn = pytest.importorskip("sphn")           # not in this image: skipped here, runs where a deployment installs it
    from rstnet_amd import server as S
    f = S.make_framer(1920, "opus", 24000)
    w = sphn.OpusStreamWriter(24000)
    w.append_pcm(np.zeros(1920 * 4, dtype=np.float32))
    f.append_bytes(w.read_bytes())
    assert all(fr.shape == (1920,) for fr in f.frames())
