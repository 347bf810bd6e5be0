"""CPU oracle of the reference's offline generation loop for the ``GPT`` backbone -- TEST INFRASTRUCTURE ONLY.

Restates ``InferenceImp.__call__`` and ``reverse_delay`` (``/root/reference/MLLM_v2/infer_no_streaming.py:168-323``) on top of
oracle/gpt_oracle.py: every generated frame re-runs the NON-streaming ``forward_global`` over ``[initial frame | prefix]`` and
the non-streaming ``forward_local`` over the whole prefix (O(T^2), exactly what the reference does), the text token is drawn
with ``sample_token`` and the eight audio tokens with ``sample_token_audio`` / ``sample_token_audio_2048``
(``utils/sampling.py:85-158``: probabilities of ids >= 2049 / 2048 blanked after the softmax).

Parity status: pinned.  infer_no_streaming.py cannot be imported here (torchaudio / dataloader dependencies at its top level),
but its ``class InferenceImp`` and ``reverse_delay`` need none of them: tests/golden/make_golden.py takes the two definitions
from the parsed source of the file where it lies and runs them unchanged on the real reference GPT and utils.sampling; the
codes that run returned and the Exp(1) noise it drew are the fixture tests/golden/gpt_generate.npz, which ``generate`` below
reproduces exactly (tests/test_oracle_golden.py).  Building blocks: gpt_tiny.npz, sampling.npz, reverse_delay.npz.
Deviations, on purpose: the reference only returns a result for task 'TTS' (the other tasks end in an unbound
``prompt_audio``, :301-307); this restatement returns the generated frames for every task.  Special ids are parameters
(defaults = the reference's literals).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch

from . import gpt_oracle as Gp


@dataclass
class GenIds:
    text_pad_token: int = 128003
    text_empty_token: int = 128002
    semantic_pad_token: int = 2049
    n_audio_codes: int = 2048          # ids >= this end a generation (:286) and are blanked by sample_token_audio_2048
    text_initial_token_id: int = 151655


def sample_token(logits: torch.Tensor, use_sampling: bool, temp: float, top_k: int, noise: Optional[torch.Tensor], limit: int = 0):
    """sample_token / sample_token_audio(_2048) (utils/sampling.py:85-158) with the Exp(1) draws of `multinomial` passed in."""
    if use_sampling and temp > 0.0:
        probs = torch.softmax(logits / temp, dim=-1)
        if limit:
            probs[..., limit:] = float("-inf")
        p, idx = torch.topk(probs, top_k, dim=-1)
        return idx.gather(-1, (p / noise).argmax(dim=-1, keepdim=True))[..., 0]
    return torch.argmax(logits, dim=-1)


def reverse_delay(x: torch.Tensor) -> torch.Tensor:
    """infer_no_streaming.py:311-323: undo the one-step delay of codebooks 1..7; x [8, L] (or [L, 8]) -> [8, L-1]."""
    if x.shape[0] != 8:
        x = x.transpose(0, 1)
    y = torch.ones_like(x)
    y[0, :-1] = x[0, :-1]
    y[1:, :-1] = x[1:, 1:]
    return y[:, :-1]


def split_prompt(seq: torch.Tensor, task_name: str, ids: GenIds):
    """:190-230: strip the padding, then (prefix, maxlen, minlen) per task; seq [B, K, L]."""
    if task_name in ("text_only", "word_level_audio_text_alignment", "ASR"):
        pad_len = int(seq[0, 0:1, :].eq(ids.text_pad_token).int().sum())
    elif task_name in ("audio_only", "TTS"):
        pad_len = int(seq[0, 1:2, :].eq(ids.semantic_pad_token).int().sum())
    else:
        raise NotImplementedError(task_name)
    seq = seq[:, :, :seq.shape[2] - pad_len]
    if task_name in ("text_only", "audio_only"):
        prefix_len = seq.shape[-1] // 2
        return seq[:, :, :prefix_len], prefix_len, prefix_len
    if task_name == "TTS":
        prefix_len = seq.shape[2] - int(seq[0, 0, :].eq(ids.text_empty_token).int().sum())
        n = seq.shape[2] - prefix_len
        return seq[:, :, :prefix_len], n, n
    if task_name == "ASR":
        prefix_len = int(seq[0, 0, :].eq(ids.text_empty_token).int().sum())
        return seq[:, :, :prefix_len + 1], seq.shape[2] - prefix_len + 13, seq.shape[2] - prefix_len - 13
    raise NotImplementedError(task_name)


@torch.no_grad()
def generate(sd: Gp.SD, cfg: Gp.GPTConfig, seq: torch.Tensor, task_name: str, *, use_sampling: bool = True, temp: float = 0.8,
             top_k: int = 250, temp_text: float = 0.7, top_k_text: int = 25, ids: GenIds = GenIds(),
             noise: Optional[Callable[[str, int, int], torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """seq int64 [K, L] (one utterance, the reference's n_samples = 1) -> {'frames': [n, dep_q], 'text': [n], and for TTS
    'codes': reverse_delay(frames)}.  ``sd`` must be a merged state dict.  ``noise(kind, g_idx, l_idx)`` supplies the Exp(1)
    draws ([1,1,k] for 'text', [1,1,1,k] for 'audio')."""
    seq = seq.unsqueeze(0)
    prefix, maxlen, minlen = split_prompt(seq, task_name, ids)
    prefix = prefix.clone()
    init = torch.full((1, cfg.num_codebooks, 1), cfg.audio_card, dtype=torch.long)
    init[:, 0] = ids.text_initial_token_id
    pre_gen_len = prefix.shape[2]
    frames, texts = [], []
    for g_idx in range(maxlen):
        g_len = prefix.shape[2]
        h, text_logits = Gp.forward_global(sd, cfg, torch.cat([init, prefix], dim=-1), merged=True)
        prefix = torch.cat([prefix, torch.full_like(prefix[:, :, 0:1], cfg.audio_card)], dim=-1)
        nz = noise("text", g_idx, 0) if noise and use_sampling else None
        text_token = sample_token(text_logits[:, -1:, :].float(), use_sampling, temp_text, top_k_text, nz)
        prefix[:, 0, -1] = text_token.squeeze()
        audio, stop = [], False
        for l_idx in range(cfg.dep_q):
            logits = Gp.forward_local(sd, cfg, prefix[:, 0, :], prefix[:, 1:cfg.dep_q + 1, :], h)
            valid = logits[:, -1:, l_idx:l_idx + 1, :].float()
            wide = g_len == pre_gen_len or (l_idx > 0 and g_len > minlen)
            limit = ids.n_audio_codes + 1 if wide else ids.n_audio_codes
            nz = noise("audio", g_idx, l_idx) if noise and use_sampling else None
            tok = sample_token(valid, use_sampling, temp, top_k, nz, limit)
            if g_idx > minlen and l_idx > 2 and int(tok[0, 0]) >= ids.n_audio_codes:
                stop = True
                break
            audio.append(tok.squeeze())
            prefix[:, l_idx + 1, g_len] = tok.squeeze()
        if stop:
            break
        frames.append(torch.stack(audio))
        texts.append(text_token.squeeze())
    out = {"frames": torch.stack(frames) if frames else torch.zeros(0, cfg.dep_q, dtype=torch.long),
           "text": torch.stack(texts) if texts else torch.zeros(0, dtype=torch.long)}
    if task_name == "TTS" and frames and cfg.dep_q == 8:
        out["codes"] = reverse_delay(out["frames"])
    return out
