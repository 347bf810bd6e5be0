"""Host side of offline dataset tokenisation -- what ``MLLM_v2/egs/pretraining/local/offline_codec_tokenization.py:37-120`` does
around ``MimiTokenizer`` (list file ``<utt-id> <path>`` in, ``torch.save``d ``{utt-id: int16 codes [8, F]}`` out, one process per
GPU picked by ``--rank``), with the batching the reference leaves as a TODO (``:79``): utterances are read in chunks, sorted by
length and encoded as zero-padded ragged batches (``MimiTokenizer.tokenize_batch``).

Audio files are read without torchaudio / soundfile: RIFF PCM ``.wav`` through the standard library, ``.npy`` / ``.pt`` arrays
as they are.  Files at other sample rates are resampled to 24 kHz on the host (``codec/audio_resample.py``), as the reference's
``MimiTokenizer`` does with torchaudio."""
from __future__ import annotations

import logging
import os
import wave
from typing import Dict, Iterable, Iterator, List, Sequence, Tuple

import numpy as np
import torch

from .audio_resample import resample

SAMPLE_RATE = 24000


def read_list(path: str) -> List[Tuple[str, str]]:
    """``<utt-id> <path with optional spaces>`` per line (offline_codec_tokenization.py:83-85; also a kaldi ``wav.scp`` whose
    entries are plain paths)."""
    items = []
    with open(path) as f:
        for line in f:
            parts = line.strip().split()
            if len(parts) >= 2:
                items.append((parts[0], " ".join(parts[1:])))
    return items


def device_index(rank_1_based: int, n_devices: int) -> int:
    """run.pl job ids start at 1; jobs beyond the GPU count wrap around (offline_codec_tokenization.py:45-48)."""
    return (rank_1_based - 1) % max(n_devices, 1)


def read_audio(path: str) -> Tuple[torch.Tensor, int]:
    """-> (mono float32 waveform ``[T]`` in [-1, 1], sample rate).  ``.wav``: 8 / 16 / 24 / 32-bit PCM, channels averaged."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        return torch.from_numpy(np.load(path)).float().reshape(-1), SAMPLE_RATE
    if ext in (".pt", ".pth"):
        return torch.load(path, map_location="cpu").float().reshape(-1), SAMPLE_RATE
    if ext != ".wav":
        raise NotImplementedError(f"{path}: only .wav (PCM), .npy and .pt inputs are read by this build")
    with wave.open(path, "rb") as w:
        n_ch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    else:
        raise NotImplementedError(f"{path}: {8 * width}-bit PCM")
    if n_ch > 1:
        x = x.reshape(-1, n_ch).mean(axis=1)
    return torch.from_numpy(np.ascontiguousarray(x)), sr


def chunks(items: Sequence, size: int) -> Iterator[Sequence]:
    for i in range(0, len(items), size):
        yield items[i:i + size]


def tokenize_list(tokenizer, items: Iterable[Tuple[str, str]], chunk_size: int = 256, max_batch_seconds: float = 1920.0,
                  reader=read_audio, skipped: List[str] = None) -> Dict[str, torch.Tensor]:
    """``items``: (utt-id, path) pairs -> ``{utt-id: int16 codes [8, ceil(T/1920)]}`` in input order.  Unreadable / empty files
    are logged and skipped, as the reference does (``:100-101``); a batch whose encode fails (e.g. out of memory on one very long
    utterance) is retried one utterance at a time so that only the offending utterance is lost.  ``skipped`` (optional list)
    receives the ids that produced no codes -- the CLI turns a non-empty list into a non-zero exit status under ``--strict``."""
    out: Dict[str, torch.Tensor] = {}
    skipped = skipped if skipped is not None else []
    for part in chunks(list(items), chunk_size):
        keys, wavs = [], []
        for key, path in part:
            try:
                wav, sr = reader(path)
                if wav.numel() == 0:
                    raise ValueError("empty waveform")
                if sr != SAMPLE_RATE:
                    wav = resample(wav, sr, SAMPLE_RATE)
                keys.append(key)
                wavs.append(wav)
            except Exception as e:      # noqa: BLE001  (one bad file must not stop a shard)
                logging.error(f"an error instance: {key} {path}, {e}")
                skipped.append(key)
        if not wavs:
            continue
        try:
            codes = tokenizer.tokenize_batch(wavs, SAMPLE_RATE, max_batch_seconds=max_batch_seconds)
        except Exception as e:          # noqa: BLE001
            logging.error(f"a batch of {len(wavs)} utterances failed ({e}); retrying them one by one")
            codes = []
            for key, wav in zip(keys, wavs):
                try:
                    codes.append(tokenizer.tokenize(wav.reshape(1, -1), SAMPLE_RATE))
                except Exception as e1:  # noqa: BLE001
                    logging.error(f"an error instance: {key}, {e1}")
                    codes.append(None)
                    skipped.append(key)
        for key, c in zip(keys, codes):
            if c is not None:
                out[key] = c
    return out
