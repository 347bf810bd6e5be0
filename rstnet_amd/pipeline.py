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

    def __enter__(self):
        # encode touches only the encoder-side modules' states and decode only the decoder-side ones: one context serves both
        self._mimi_ctx = self.mimi.streaming(self.batch_size)
        self._mimi_ctx.__enter__()
        self._lm_ctx = self.lm_gen.streaming(self.batch_size)
        self._lm_ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self._lm_ctx.__exit__(*exc)
        self._mimi_ctx.__exit__(*exc)
        return False

    @torch.no_grad()
    def step(self, pcm: torch.Tensor) -> Optional[torch.Tensor]:
        """pcm fp32 ``[B, 1, 1920]`` -> generated pcm ``[B, 1, 1920]`` (or ``None`` during the first ``max_delay`` frames)."""
        assert pcm.shape == (self.batch_size, 1, self.frame_size), tuple(pcm.shape)
        codes = self.mimi.encode(pcm)                                   # [B, 8, 1]
        tokens = self.lm_gen.step(codes[:, :self.n_user].contiguous())  # [B, 1 + dep_q, 1] or None
        if tokens is None:
            return None
        return self.mimi.decode(tokens[:, 1:].contiguous())
