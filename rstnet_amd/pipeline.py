"""End-to-end streaming loop: PCM in -> Mimi encode -> LMGen.step -> Mimi decode -> PCM out, one 80 ms frame per call.

This is the loop of the reference's ``MLLM_v2/moshi/server.py:122-136`` (``opus_loop``) without the Opus / websocket
transport: ``mimi.encode(chunk)`` -> ``lm_gen.step(codes)`` -> ``mimi.decode(tokens[:, 1:])``; the first ``max_delay``
frames produce no output (``LMGen.step`` returns ``None``).  All state (conv histories, KV rings, token ring
cache) lives in the modules' streaming states on the device; the host only moves 1920 samples in and out per stream.
"""
from __future__ import annotations

from typing import Optional

import torch

from .codec.mimi import MimiCodec
from .lm.model import LMGen


class StreamingPipeline:
    def __init__(self, mimi: MimiCodec, lm_gen: LMGen, batch_size: int):
        assert mimi.quantizer.n_q >= lm_gen.lm_model.dep_q
        self.mimi, self.lm_gen, self.batch_size = mimi, lm_gen, batch_size
        self.frame_size = mimi.frame_hop
        self.n_user = lm_gen.lm_model.num_codebooks - lm_gen.lm_model.dep_q - 1
        self._dec = None
        from . import ops
        self.fuse = bool(ops.PIPELINE_FUSE)     # one graph per frame once the delay line is full (False: the three module calls, as the reference loops)
        self._fused, self._fused_epoch = None, -1
        self.last_codes: Optional[torch.Tensor] = None      # what the encoder emitted for the last frame, [B, K, 1] (valid until the next step)

    def __enter__(self):
        # encode touches only the encoder-side modules' states and decode only the decoder-side ones: one context serves both
        self._mimi_ctx = self.mimi.streaming(self.batch_size)
        self._mimi_ctx.__enter__()
        self._lm_ctx = self.lm_gen.streaming(self.batch_size)
        self._lm_ctx.__enter__()
        self._fused, self._fused_epoch, self.last_codes = None, -1, None     # a graph of an earlier session points at that session's states
        return self

    def __exit__(self, *exc):
        self._fused, self.last_codes = None, None        # (the captured frame keeps the session's state tensors alive)
        self._lm_ctx.__exit__(*exc)
        self._mimi_ctx.__exit__(*exc)
        return False

    def _frame_fn(self, pcm: torch.Tensor) -> torch.Tensor:
        """The three stages of a frame back to back on the device (the function the fused graph captures): no host-side slicing,
        cloning or re-staging of the codes / tokens between them."""
        B = self.batch_size
        codes = self.mimi.quantizer.encode_nlc(self.mimi.encode_latent(pcm))          # [B, K, 1]
        self.last_codes = codes         # (inside the graph: a static buffer every replay refills)
        out, _ = self.lm_gen._frame(codes[:, :self.n_user, 0].contiguous())            # [B, 1 + dep_q]
        return self.mimi._decode(out[:, 1:].reshape(B, -1, 1))

    @torch.no_grad()
    def step(self, pcm: torch.Tensor) -> Optional[torch.Tensor]:
        """pcm fp32 ``[B, 1, 1920]`` -> generated pcm ``[B, 1, 1920]`` (or ``None`` during the first ``max_delay`` frames).

        Once the delay line is full and every module has run its two eager frames (streaming buffers allocated, in-place from then
        on) a frame is ONE captured graph -- encode, LM frame,
        decode -- fed by one copy of the PCM and read by one clone of the result; before that, and on CPU tensors, under
        ``NO_CUDA_GRAPH=1`` or with ``LMGen(check=True)``, the three module calls run as the reference's loop does (server.py:122-136)."""
        assert pcm.shape == (self.batch_size, 1, self.frame_size), tuple(pcm.shape)
        gen = self.lm_gen
        state = gen._streaming_state
        fused_ok = (self.fuse and pcm.is_cuda and state is not None and state.offset >= gen.max_delay + 2 and not gen.check
                    and self.mimi.code_layout == "bkt")
        if fused_ok:
            from . import ops
            from .graphs import Graphed
            if state.offset % 64 == 0:
                ops.persistent_poll(pcm.device)
            epoch = ops.persistent_epoch(pcm.device)
            if self._fused is None or self._fused_epoch != epoch:
                # the first capture follows the modules' own eager frames (the delay line: max_delay + 2 of them).  A RE-capture after the
                # device's persistent launches were retired takes launch-per-op paths that have not run in this session yet: one eager
                # frame first, so that their lazily created state and scratch exist before the capture
                self._fused, self._fused_epoch = Graphed(self._frame_fn, warmup=0 if self._fused is None else 1), epoch
            if not self._fused.disable:
                wav = self._fused(pcm.contiguous())
                state.offset += 1
                return wav.clone()
        codes = self.mimi.encode(pcm)                                   # [B, 8, 1]
        self.last_codes = codes
        tokens = self.lm_gen.step(codes[:, :self.n_user].contiguous())  # [B, 1 + dep_q, 1] or None
        if tokens is None:
            return None
        return self.mimi.decode(tokens[:, 1:].contiguous())
