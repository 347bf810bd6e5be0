"""Band-limited sinc resampling of host waveforms to the codec's 24 kHz -- data preparation in front of the hot path.

The reference hands every waveform whose rate differs from 24 kHz to ``torchaudio.transforms.Resample(sr, 24000)``
(``MLLM_v2/tools/tokenizer/MimiCodec/mimi_tokenizer.py:51-52,66-67``).  torchaudio is a third-party dependency that is neither
vendored under the reference nor installed in this image, so this is a restatement of its published default algorithm
(``torchaudio.functional.resample``: Hann-windowed sinc interpolation, ``lowpass_filter_width=6``, ``rolloff=0.99``; the rates
reduced by their gcd, one polyphase filter per output phase, a strided correlation).  PARITY UNPINNED: no reference test
or fixture holds a resampled waveform and the dependency cannot be run here; the unit tests check the properties that define it
(identity at equal rates, length ``ceil(T * new / orig)``, tone preservation below the cut-off, rejection above it).
Runs on the host (torch CPU) like the reference's call; the codec itself only ever sees 24 kHz.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def sinc_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """-> (filters ``[new, 1, 2 * width + orig]`` fp32, width) for rates already divided by their gcd."""
    base = min(orig_freq, new_freq) * rolloff
    width = math.ceil(lowpass_filter_width * orig_freq / base)
    idx = torch.arange(-width, width + orig_freq, dtype=torch.float64)[None, None] / orig_freq
    t = torch.arange(0, -new_freq, -1, dtype=torch.float64)[:, None, None] / new_freq + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    sinc = torch.where(t == 0, torch.ones_like(t), t.sin() / t)
    return (sinc * window * (base / orig_freq)).float(), width


def resample(wav: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """``[..., T]`` at ``orig_freq`` -> ``[..., ceil(T * new_freq / orig_freq)]`` at ``new_freq``."""
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq <= 0 or new_freq <= 0:
        raise ValueError("sample rates must be positive")
    if orig_freq == new_freq:
        return wav
    g = math.gcd(orig_freq, new_freq)
    o, n = orig_freq // g, new_freq // g
    kernel, width = sinc_kernel(o, n)
    shape = wav.shape
    x = wav.reshape(-1, shape[-1]).float()
    length = x.shape[1]
    x = F.pad(x, (width, width + o))
    y = F.conv1d(x[:, None], kernel, stride=o).transpose(1, 2).reshape(x.shape[0], -1)
    return y[:, :math.ceil(n * length / o)].reshape(*shape[:-1], -1)
