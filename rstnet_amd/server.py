"""Full-duplex websocket loop over the streaming path -- the role of ``MLLM_v2/moshi/server.py:44-166``.

Same structure as the reference: one ``ServerState`` holding the codec and ``LMGen`` in ``streaming_forever(1)`` mode, a warm-up of
four silent frames (``:64-73``), one session at a time behind an ``asyncio.Lock`` (``:59,157``), states reset at the start of every
session (``:160-161``), a ``b"\\x00"`` handshake, then per 1920-sample frame ``mimi.encode -> lm_gen.step -> mimi.decode``
(``:122-136``) with audio going out as kind ``1`` messages and text pieces as kind ``2``.

What differs, and why: the reference moves audio as Opus pages through ``sphn`` (``OpusStreamReader`` / ``OpusStreamWriter``);
``sphn`` / libopus are not part of this image, so the transport here is RAW PCM -- the payload of a kind-1 message is
little-endian mono samples at the codec rate, ``f32`` (default) or ``s16`` (``/api/chat?pcm=s16``), in both directions.  The
message kinds, the handshake, the framing and the locking are the reference's; a client only swaps its Opus codec for a
memcpy.  ``/api/chat?pcm=opus`` selects ``OpusFramer`` -- the reference's ``sphn`` reader / writer at the same seam -- which needs the
``sphn`` package and says so when it is missing.
"""
from __future__ import annotations

import asyncio
import time
from typing import Callable, List, Optional

import numpy as np
import torch

KIND_HANDSHAKE, KIND_AUDIO, KIND_TEXT = 0, 1, 2


def log(level: str, msg: str) -> None:
    print(f"[{level}] {msg}", flush=True)


class PcmFramer:
    """Byte stream -> fixed-size frames of float32 samples (the job ``sphn.OpusStreamReader.read_pcm`` + the ``all_pcm_data``
    accumulation of server.py:113-121 do in the reference).  ``fmt``: ``"f32"`` or ``"s16"`` little-endian mono."""

    def __init__(self, frame_size: int, fmt: str = "f32"):
        if fmt not in ("f32", "s16"):
            raise ValueError(f"unknown pcm format {fmt!r}")
        self.frame_size, self.fmt = frame_size, fmt
        self._width = 4 if fmt == "f32" else 2
        self._bytes = bytearray()

    def append_bytes(self, payload: bytes) -> None:
        self._bytes += payload

    def frames(self) -> List[np.ndarray]:
        """All complete frames received so far (a trailing partial frame -- or partial sample -- waits for more bytes)."""
        n = len(self._bytes) // (self._width * self.frame_size)
        out = []
        for i in range(n):
            raw = bytes(self._bytes[i * self._width * self.frame_size:(i + 1) * self._width * self.frame_size])
            out.append(np.frombuffer(raw, dtype="<f4").copy() if self.fmt == "f32"
                       else np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0)
        del self._bytes[:n * self._width * self.frame_size]
        return out

    def encode(self, pcm: np.ndarray) -> bytes:
        """float32 samples -> payload bytes in this framer's format (the outgoing direction)."""
        pcm = np.asarray(pcm, dtype=np.float32).reshape(-1)
        if self.fmt == "f32":
            return pcm.astype("<f4").tobytes()
        return (np.clip(pcm, -1.0, 1.0) * 32767.0).astype("<i2").tobytes()


class OpusFramer:
    """The reference's own transport at the same seam (server.py:105-153): kind-1 payloads are Opus pages, decoded by
    ``sphn.OpusStreamReader`` and re-encoded by ``sphn.OpusStreamWriter``.  ``sphn`` (Rust + libopus) is not part of this image, so
    the class imports it on construction and raises a clear error when it is absent; with it installed, ``/api/chat?pcm=opus``
    speaks the reference client's wire format."""

    def __init__(self, frame_size: int, sample_rate: int):
        try:
            import sphn
        except ImportError as e:
            raise RuntimeError("the Opus transport needs the `sphn` package (absent from this image); use pcm=f32 or pcm=s16") from e
        self.frame_size = frame_size
        self._reader = sphn.OpusStreamReader(sample_rate)
        self._writer = sphn.OpusStreamWriter(sample_rate)
        self._pcm = np.zeros(0, dtype=np.float32)

    def append_bytes(self, payload: bytes) -> None:
        self._reader.append_bytes(payload)

    def frames(self) -> List[np.ndarray]:
        pcm = self._reader.read_pcm()                       # server.py:113-121
        if pcm.shape[-1]:
            self._pcm = np.concatenate((self._pcm, np.asarray(pcm, dtype=np.float32).reshape(-1)))
        n = self._pcm.shape[0] // self.frame_size
        out = [self._pcm[i * self.frame_size:(i + 1) * self.frame_size].copy() for i in range(n)]
        self._pcm = self._pcm[n * self.frame_size:]
        return out

    def encode(self, pcm: np.ndarray) -> bytes:
        """float32 samples -> the Opus bytes that are ready (possibly empty: the writer emits whole pages), server.py:133,146-150."""
        self._writer.append_pcm(np.asarray(pcm, dtype=np.float32).reshape(-1))
        return bytes(self._writer.read_bytes())


def make_framer(frame_size: int, fmt: str, sample_rate: int):
    return OpusFramer(frame_size, sample_rate) if fmt == "opus" else PcmFramer(frame_size, fmt)


class ServerState:
    """``ServerState`` of server.py:44-166 over ``rstnet_amd``'s ``MimiModel`` / ``LMGen`` (or anything with the same three
    calls -- the CPU test drives it with stand-ins)."""

    def __init__(self, mimi, lm, device, text_tokenizer=None, lm_gen=None, **lm_gen_kwargs):
        from .lm.model import LMGen
        self.mimi, self.text_tokenizer, self.device = mimi, text_tokenizer, device
        self.lm_gen = lm_gen if lm_gen is not None else LMGen(lm, **lm_gen_kwargs)
        self.frame_size = int(round(mimi.sample_rate / mimi.frame_rate)) if hasattr(mimi, "frame_rate") else mimi.frame_hop
        self.lock = asyncio.Lock()
        self.mimi.streaming_forever(1)
        self.lm_gen.streaming_forever(1)

    def _sync(self) -> None:
        if torch.cuda.is_available() and torch.device(self.device).type == "cuda":
            torch.cuda.synchronize(self.device)

    def warmup(self) -> None:
        """server.py:64-73: four silent frames through the whole loop (graph capture happens here, not in the first session)."""
        for _ in range(4):
            chunk = torch.zeros(1, 1, self.frame_size, dtype=torch.float32, device=self.device)
            self.frame(chunk)
        self._sync()

    def frame(self, chunk: torch.Tensor):
        """One 80 ms frame: pcm ``[1, 1, frame_size]`` -> list of (pcm float32 numpy ``[frame_size]``, text token id) -- empty while
        ``LMGen.step`` still returns ``None`` (the first ``max_delay`` frames), server.py:126-136."""
        out = []
        codes = self.mimi.encode(chunk)
        for c in range(codes.shape[-1]):
            tokens = self.lm_gen.step(codes[:, :self._n_user(), c:c + 1].contiguous())
            if tokens is None:
                continue
            assert tokens.shape[1] == self.lm_gen.lm_model.dep_q + 1
            main_pcm = self.mimi.decode(tokens[:, 1:].contiguous())
            out.append((main_pcm[0, 0].detach().float().cpu().numpy(), int(tokens[0, 0, 0].item())))
        return out

    def _n_user(self) -> int:
        lm = self.lm_gen.lm_model
        return lm.num_codebooks - lm.dep_q - 1

    def text_piece(self, token: int) -> Optional[str]:
        """server.py:137-142: padding ids 0 / 3 are silent; other ids become sentencepiece pieces (or ``<id>`` without a tokenizer)."""
        if token in (0, 3):
            return None
        if self.text_tokenizer is None:
            return f"<{token}>"
        return self.text_tokenizer.id_to_piece(token).replace("▁", " ")

    async def handle_chat(self, request):
        from aiohttp import WSMsgType, web
        # the transport is settled BEFORE the upgrade: an unknown ?pcm= value, or pcm=opus without `sphn`, is answered with 400 and the
        # reason (after ws.prepare() the client would only see an abrupt 1011 close)
        try:
            framer = make_framer(self.frame_size, request.query.get("pcm", "f32"), int(getattr(self.mimi, "sample_rate", 24000)))
        except (ValueError, RuntimeError) as e:
            raise web.HTTPBadRequest(text=str(e))
        ws = web.WebSocketResponse()
        await ws.prepare(request)
        close = False
        outbox: asyncio.Queue = asyncio.Queue()

        async def recv_loop():
            nonlocal close
            try:
                async for message in ws:
                    if message.type == WSMsgType.ERROR:
                        log("error", f"{ws.exception()}")
                        break
                    if message.type in (WSMsgType.CLOSED, WSMsgType.CLOSE):
                        break
                    if message.type != WSMsgType.BINARY:
                        log("error", f"unexpected message type {message.type}")
                        continue
                    data = message.data
                    if len(data) == 0:
                        log("warning", "empty message")
                        continue
                    if data[0] == KIND_AUDIO:
                        framer.append_bytes(data[1:])
                    else:
                        log("warning", f"unknown message kind {data[0]}")
            finally:
                close = True
                log("info", "connection closed")

        async def frame_loop():
            while not close:
                await asyncio.sleep(0.001)
                for pcm in framer.frames():
                    be = time.time()
                    chunk = torch.from_numpy(pcm).to(self.device)[None, None]
                    for out_pcm, text_token in self.frame(chunk):
                        payload = framer.encode(out_pcm)
                        if len(payload) > 0:            # (an Opus writer hands out whole pages: nothing yet is not a message)
                            await outbox.put(bytes([KIND_AUDIO]) + payload)
                        piece = self.text_piece(text_token)
                        if piece is not None:
                            await outbox.put(bytes([KIND_TEXT]) + piece.encode("utf8"))
                    log("info", f"frame handled in {1000 * (time.time() - be):.1f}ms")

        async def send_loop():
            while not close or not outbox.empty():
                try:
                    msg = await asyncio.wait_for(outbox.get(), timeout=0.005)
                except asyncio.TimeoutError:
                    continue
                try:
                    await ws.send_bytes(msg)
                except (ConnectionError, RuntimeError):
                    return

        log("info", "accepted connection")
        async with self.lock:       # one session at a time (server.py:157)
            self.mimi.reset_streaming()
            self.lm_gen.reset_streaming()
            await ws.send_bytes(bytes([KIND_HANDSHAKE]))
            await asyncio.gather(frame_loop(), recv_loop(), send_loop())
        log("info", "done with connection")
        return ws


def make_app(state: ServerState):
    from aiohttp import web
    app = web.Application()
    app.router.add_get("/api/chat", state.handle_chat)
    return app


def build_state(args, log_fn: Callable[[str, str], None] = log) -> ServerState:
    """Models for the CLI: checkpoints (``--mimi-weight`` / ``--moshi-weight``, safetensors or torch files) or ``--synthetic``
    seeded random-init weights of the real shapes (there is no hub access in this image)."""
    from . import synth
    from .codec.loaders import get_mimi
    from .lm.model import LMModel
    device = torch.device(args.device)
    if device.type == "cuda":
        torch.cuda.set_device(device)
    cfg = dict(synth.LM_MOSHI_7B if args.lm_config == "moshi7b" else synth.LM_TINY_16Q)
    if args.synthetic:
        mimi = get_mimi(synth.mimi_state_dict(0), device=device)
        sd = synth.lm_state_dict(cfg, seed=0, device=str(device))
    else:
        from .lm.loaders import load_lm_state_dict
        mimi = get_mimi(args.mimi_weight, device=device)
        sd = load_lm_state_dict(args.moshi_weight, device=device)
    lm = LMModel.from_state_dict(sd, cfg)
    tok = None
    if args.tokenizer:
        import sentencepiece
        tok = sentencepiece.SentencePieceProcessor(args.tokenizer)
    log_fn("info", "models loaded")
    return ServerState(mimi, lm, device, text_tokenizer=tok)


def main(argv=None) -> None:
    import argparse
    from aiohttp import web
    p = argparse.ArgumentParser(description="raw-PCM websocket server over the MI355X streaming path (moshi/server.py without Opus)")
    p.add_argument("--host", default="localhost", type=str)
    p.add_argument("--port", default=8998, type=int)
    p.add_argument("--tokenizer", type=str, help="path to a local sentencepiece model (text pieces; ids are sent as <id> without it)")
    p.add_argument("--moshi-weight", type=str, help="path to a local checkpoint file for the LM")
    p.add_argument("--mimi-weight", type=str, help="path to a local checkpoint file for Mimi")
    p.add_argument("--synthetic", action="store_true", help="seeded random-init weights of the real shapes (no checkpoints needed)")
    p.add_argument("--lm-config", choices=["moshi7b", "tiny"], default="moshi7b")
    p.add_argument("--device", type=str, default="cuda")
    args = p.parse_args(argv)
    if not args.synthetic and not (args.moshi_weight and args.mimi_weight):
        p.error("give --moshi-weight and --mimi-weight, or --synthetic")
    torch.manual_seed(42424242)
    state = build_state(args)
    log("info", "warming up the model")
    state.warmup()
    log("info", f"Access the websocket at ws://{args.host}:{args.port}/api/chat  (raw PCM payloads; Opus needs sphn, absent here)")
    with torch.no_grad():
        web.run_app(make_app(state), host=args.host, port=args.port)


if __name__ == "__main__":
    main()
