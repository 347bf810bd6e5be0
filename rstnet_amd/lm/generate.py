"""Streaming generation for the ``GPT`` backbone: the offline loop of ``MLLM_v2/infer_no_streaming.py:168-323``
(``InferenceImp.__call__`` + ``reverse_delay``) with the same prompt handling, sampling rules and stop rule, but O(T):

  * the prompt ``[initial frame | prefix]`` is pushed through the global transformer ONCE (prefill: all positions per launch,
    bf16-MFMA skinny GEMMs + multi-query ring attention), every generated frame afterwards is one graph-replayed T = 1 step
    -- the reference re-runs ``forward_global`` over the whole sequence for every frame (:240);
  * the eight audio tokens of a frame come from eight depth-transformer steps (one captured graph, sampling included) -- the
    reference re-runs the teacher-forced ``forward_local`` over the whole prefix for every codebook (:259).

Both replacements are exact restatements of what the reference computes for the LAST position: its non-streaming passes see
positions 0..t with the causal + context mask and no ring, which a ring of capacity context + 1 (global) / dep_q + 1 (depth)
reproduces -- a ring that size never lets the `delta <= 0` slot-map quirk hide a key the mask would show.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch

from .. import ops
from ..graphs import Graphed as _Graphed
from .gpt import GPT, _GPTState


def reverse_delay(x: torch.Tensor) -> torch.Tensor:
    """infer_no_streaming.py:311-323: codebooks 1..7 are generated one frame late; x ``[8, L]`` (or ``[L, 8]``) -> ``[8, L-1]``."""
    if x.shape[0] != 8:
        x = x.transpose(0, 1)
    out = torch.ones_like(x)
    out[0, :-1] = x[0, :-1]
    out[1:, :-1] = x[1:, 1:]
    return out[:, :-1]


@dataclass
class GenIds:
    """The literal ids of InferenceImp.__init__ (:178-183) and the sampling helpers; parameters here so that small test
    vocabularies can be used."""
    text_pad_token: int = 128003
    text_empty_token: int = 128002
    semantic_pad_token: int = 2049
    acoustic_pad_token: int = 2049
    n_audio_codes: int = 2048
    text_initial_token_id: Optional[int] = None      # None = model.text_initial_token_id (151655)


class InferenceImp:
    """Same constructor and call convention as infer_no_streaming.py:168-190 (``args`` is kept for signature parity and unused).
    ``__call__(seq [K, L], mask)`` returns what the reference returns for task 'TTS' (``reverse_delay`` of the generated
    frames, ``[8, n-1]``); ``generate`` returns every product of the loop for any task."""

    def __init__(self, args, model: GPT, mode: str, temp_text: float, top_k_text: int, temp: float, top_k: int, task_name: str,
                 use_sampling: bool = True, ids: Optional[GenIds] = None,
                 noise: Optional[Callable[[str, int, int], torch.Tensor]] = None):
        self.model, self.args, self.mode, self.task_name = model, args, mode, task_name
        self.n_samples = 1
        self.use_sampling, self.temp_text, self.top_k_text, self.temp, self.top_k = use_sampling, temp_text, top_k_text, temp, top_k
        self.ids = ids or GenIds()
        self.noise = noise           # test hook: Exp(1) draws (kind, g_idx, l_idx) -> [1, k]; disables graph capture
        self._depth: Optional[_Graphed] = None
        self._limits: Optional[torch.Tensor] = None

    # ---- prompt handling (:190-230)
    def split_prompt(self, seq: torch.Tensor):
        ids, task = self.ids, self.task_name
        if task in ("text_only", "word_level_audio_text_alignment", "ASR"):
            pad_len = int(seq[0, 0:1, :].eq(ids.text_pad_token).int().sum())
        elif task in ("audio_only", "TTS"):
            pad_len = int(seq[0, 1:2, :].eq(ids.semantic_pad_token).int().sum())
        else:
            raise NotImplementedError
        seq = seq[:, :, :seq.shape[2] - pad_len]
        if task in ("text_only", "audio_only"):
            n = seq.shape[-1] // 2
            return seq[:, :, :n], n, n
        if task == "TTS":
            n_prefix = seq.shape[2] - int(seq[0, 0, :].eq(ids.text_empty_token).int().sum())
            n = seq.shape[2] - n_prefix
            return seq[:, :, :n_prefix], n, n
        if task == "ASR":
            n_prefix = int(seq[0, 0, :].eq(ids.text_empty_token).int().sum())
            return seq[:, :, :n_prefix + 1], seq.shape[2] - n_prefix + 13, seq.shape[2] - n_prefix - 13
        raise NotImplementedError

    # ---- one frame of the depth transformer: dep_q steps + sampling, tokens [B, dep_q]
    def _exp_noise(self, kind: str, g_idx: int, l_idx: int, B: int, k: int) -> Optional[torch.Tensor]:
        if not self.use_sampling:
            return None
        if self.noise is not None:
            return self.noise(kind, g_idx, l_idx).reshape(B, k).to(self.model.device, torch.float32).contiguous()
        return torch.empty(B, k, device=self.model.device, dtype=torch.float32).exponential_(1)    # utils/sampling.py:44-46

    def _depth_frame(self, text_token: torch.Tensor, h: torch.Tensor, g_idx: int = 0) -> torch.Tensor:
        m = self.model
        dep, cfg = m.codecformer, m.config
        (B,) = text_token.shape
        dep._streaming_state.reset()
        out = torch.empty(B, cfg.dep_q, device=text_token.device, dtype=torch.long)
        prev = text_token
        k_eff = min(self.top_k, cfg.audio_card)
        for l_idx in range(cfg.dep_q):
            y = dep.step(m._codec_in(l_idx, prev, h))
            head = m.audio_linears[l_idx]
            logits = ops.lm_linear(y, head.weight, bias=head.bias_f32())
            prev = ops.lm_sample(logits, use_sampling=self.use_sampling, temp=self.temp, top_k=k_eff,
                                 noise=self._exp_noise("audio", g_idx, l_idx, B, k_eff), limit_dev=self._limits[l_idx:l_idx + 1])
            out[:, l_idx] = prev
        return out

    @torch.no_grad()
    def generate(self, seq: torch.Tensor, mask: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """seq int64 ``[K, L]`` -> {'frames': [n, dep_q], 'text': [n]} (+ 'codes' = reverse_delay(frames) for task 'TTS' with
        8 codebooks)."""
        m, ids = self.model, self.ids
        cfg, dev = m.config, m.device
        seq = seq.to(dev).unsqueeze(0)
        prefix, maxlen, minlen = self.split_prompt(seq)
        B, K = 1, cfg.n_q + 1
        init = m._get_initial_token()
        if ids.text_initial_token_id is not None:
            init[:, 0] = ids.text_initial_token_id
        pre_gen_len = prefix.shape[2]
        n_codes = ids.n_audio_codes
        eager = self.noise is not None or dev.type != "cuda"
        saved = (m.transformer._streaming_state, m._streaming_state, m.codecformer._streaming_state)
        m.transformer._streaming_state = m.transformer._make_state(B, cfg.context + 1)
        m._streaming_state = _GPTState(_Graphed(m._global_step, disable=eager))
        m.codecformer._streaming_state = m.codecformer._init_streaming_state(B, capacity=cfg.dep_q + 1)
        self._limits = torch.zeros(cfg.dep_q, device=dev, dtype=torch.int32)
        depth = _Graphed(self._depth_frame, disable=eager)
        k_text = min(self.top_k_text, cfg.padded_vocab_size)
        frames, texts = [], []
        try:
            h, logits = m.forward_global(torch.cat([init, prefix], dim=-1))      # prefill: positions 0 .. pre_gen_len
            h, logits = h[:, -1], logits[:, -1]
            regime = None
            for g_idx in range(maxlen):
                g_len = pre_gen_len + g_idx
                text_token = ops.lm_sample(logits.contiguous(), use_sampling=self.use_sampling, temp=self.temp_text, top_k=k_text,
                                           noise=self._exp_noise("text", g_idx, 0, B, k_text))
                # blanking regime of this frame (:264-283): first frame -> 2049 everywhere; later l = 0 -> 2048,
                # l > 0 -> 2049 once g_len > minlen else 2048
                new_regime = 0 if g_len == pre_gen_len else (1 if g_len > minlen else 2)
                if new_regime != regime:
                    wide = [True] * cfg.dep_q if new_regime == 0 else [l > 0 and new_regime == 1 for l in range(cfg.dep_q)]
                    self._limits.copy_(torch.tensor([n_codes + 1 if w else n_codes for w in wide], dtype=torch.int32))
                    regime = new_regime
                if eager:
                    audio = self._depth_frame(text_token, h.contiguous(), g_idx)
                else:
                    audio = depth(text_token, h.contiguous())
                audio_host = audio[0].tolist()
                if g_idx > minlen and any(t >= n_codes for t in audio_host[3:]):     # the stop rule of :286-288
                    break
                frames.append(audio[0].clone())
                texts.append(text_token[0].clone())
                if g_idx + 1 == maxlen:
                    break
                col = torch.full((B, K, 1), m.initial_token_id, device=dev, dtype=torch.long)
                col[:, 0, 0] = text_token
                col[:, 1:cfg.dep_q + 1, 0] = audio
                h, logits = m.forward_global(col)
                h, logits = h[:, 0], logits[:, 0]
        finally:
            m.transformer._streaming_state, m._streaming_state, m.codecformer._streaming_state = saved
        out = {"frames": torch.stack(frames) if frames else torch.zeros(0, cfg.dep_q, dtype=torch.long, device=dev),
               "text": torch.stack(texts) if texts else torch.zeros(0, dtype=torch.long, device=dev)}
        if self.task_name == "TTS" and frames and cfg.dep_q == 8:
            out["codes"] = reverse_delay(out["frames"])
        return out

    @torch.no_grad()
    def __call__(self, seq: torch.Tensor, mask: Optional[torch.Tensor] = None):
        if self.mode == "teacher-force":
            raise NotImplementedError("teacher-forced loss evaluation (training-side metric) is outside the generation path")
        out = self.generate(seq, mask)
        if self.task_name == "TTS":
            return out.get("codes", out["frames"])
        return out["frames"]
