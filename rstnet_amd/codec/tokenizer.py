"""``MimiTokenizer`` -- the tokenizer wrapper of ``MLLM_v2/tools/tokenizer/MimiCodec/mimi_tokenizer.py:14-82`` over the HIP codec,
plus the batched path the reference leaves as a TODO (``egs/pretraining/local/offline_codec_tokenization.py:79``).

Same surface: ``tokenize(wav, sample_rate)`` (other rates are resampled to 24 kHz on the host first, as the reference does with torchaudio;
path strings are not supported here: the image has no general audio-file reader), ``tokenize2``, ``find_length``, ``detokenize``; codes leave as int16 on the host exactly like the reference
(``:72``: "reduce the save space").  ``tokenize_batch`` packs utterances of different lengths into one encode call: the codec
is causal end to end, so only an utterance's LAST, partial frame can see what follows it; there the reference pads every strided
layer's input itself (zeros; replicate in the 25 -> 12.5 Hz down-sampling), which ``MimiCodec.encode(batch, lengths)``
reproduces with ``rst_mask_tail_f32`` -- the per-utterance codes equal those of single-utterance calls
(``tests/test_mimi_gpu.py::test_tokenizer_batch_equals_single``).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from .audio_resample import resample
from .mimi import MimiCodec

FRAME_HOP = 1920


class MimiTokenizer:
    def __init__(self, model: MimiCodec, device: Optional[torch.device] = None):
        self.model = model.eval()
        self.device = torch.device(device) if device is not None else next(model.parameters()).device
        self.sr = 24000

    # ---- reference surface
    def find_length(self, x: torch.Tensor) -> int:
        return x.shape[1]

    def tokenize2(self, token):
        if isinstance(token, torch.Tensor):
            return token.to(torch.int64)
        raise NotImplementedError

    def _to_codec_rate(self, wav: torch.Tensor, sample_rate: int) -> torch.Tensor:
        """``torchaudio.transforms.Resample(sample_rate, self.sr)(wav)`` of mimi_tokenizer.py:66-67 (host side, before the codec)."""
        return wav if sample_rate == self.sr else resample(wav.detach().cpu().float(), sample_rate, self.sr)

    @torch.no_grad()
    def tokenize(self, wav, sample_rate: int = 24000):
        """wav: 1-D tensor = codes that were tokenized offline (returned as is, ``:60-61``); 2-D ``[1, T]`` or 3-D ``[1, 1, T]``
        float waveform -> int16 codes ``[8, ceil(T/1920)]`` on the host."""
        if isinstance(wav, str):
            raise NotImplementedError("audio file paths need an audio reader; pass the waveform tensor")
        if not isinstance(wav, torch.Tensor):
            raise NotImplementedError
        if wav.dim() == 1:
            return wav
        if wav.dim() == 2:
            if wav.numel() == 0:
                return None
            wav = self._to_codec_rate(wav, sample_rate).unsqueeze(1)
        codes = self.model.encode(wav.to(self.device, torch.float32))
        return codes.squeeze(0).detach().cpu().to(torch.int16)

    @torch.no_grad()
    def detokenize(self, codes: torch.Tensor) -> torch.Tensor:
        assert codes.shape[0] == 8
        wav = self.model.decode(codes.unsqueeze(0).to(self.device, torch.int64))
        return wav.squeeze(1).detach().cpu()

    # ---- batched offline tokenization
    @torch.no_grad()
    def tokenize_batch(self, wavs: Sequence[torch.Tensor], sample_rate: int = 24000, max_batch_seconds: float = 1920.0) -> List[torch.Tensor]:
        """Mono waveforms (1-D ``[T_i]`` or ``[1, T_i]``) -> list of int16 codes ``[8, ceil(T_i/1920)]`` in input order.  Utterances are
        sorted by length and packed into zero-padded batches of at most ``max_batch_seconds`` of padded audio."""
        flat = [self._to_codec_rate(w.reshape(-1).float(), sample_rate) for w in wavs]
        order = sorted(range(len(flat)), key=lambda i: flat[i].numel())
        out: List[Optional[torch.Tensor]] = [None] * len(flat)
        budget = int(max_batch_seconds * self.sr)
        i = 0
        while i < len(order):
            j, longest = i, 0
            while j < len(order) and max(longest, flat[order[j]].numel()) * (j - i + 1) <= max(budget, flat[order[j]].numel()):
                longest = max(longest, flat[order[j]].numel())
                j += 1
            batch = torch.zeros(j - i, 1, longest)
            for r, idx in enumerate(order[i:j]):
                batch[r, 0, :flat[idx].numel()] = flat[idx]
            lens = torch.tensor([flat[idx].numel() for idx in order[i:j]], dtype=torch.int32)
            codes = self.model.encode(batch.to(self.device), lengths=lens).cpu()
            for r, idx in enumerate(order[i:j]):
                frames = -(-flat[idx].numel() // FRAME_HOP)
                out[idx] = codes[r, :, :frames].to(torch.int16).contiguous()
            i = j
        return out  # type: ignore[return-value]

    def tokenize_scp(self, items: Dict[str, torch.Tensor], sample_rate: int = 24000, **kw) -> Dict[str, torch.Tensor]:
        """{utterance id: waveform} -> {utterance id: int16 codes}: the ``data_dict`` the reference's offline tokenization
        saves with ``torch.save`` (``offline_codec_tokenization.py:88-101``)."""
        keys = list(items)
        return dict(zip(keys, self.tokenize_batch([items[k] for k in keys], sample_rate, **kw)))
