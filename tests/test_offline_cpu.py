"""Host logic of offline tokenisation (rstnet_amd/codec/offline.py): list parsing, rank -> GPU mapping, PCM readers, chunked
batching and error handling with a stand-in tokenizer (the GPU encode itself is covered by tests/test_mimi_gpu.py)."""
import os
import wave

import numpy as np
import pytest
import torch

from rstnet_amd.codec import offline


def _write_wav(path, x, sr=24000, width=2, channels=1):
    with wave.open(path, "wb") as w:
        w.setnchannels(channels)
        w.setsampwidth(width)
        w.setframerate(sr)
        if width == 2:
            w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())
        elif width == 4:
            w.writeframes((np.clip(x, -1, 1) * 2147483647).astype("<i4").tobytes())
        elif width == 3:
            v = (np.clip(x, -1, 1) * 8388607).astype(np.int32)
            w.writeframes(np.stack([v & 255, (v >> 8) & 255, (v >> 16) & 255], -1).astype(np.uint8).tobytes())


def test_read_list_and_device_index(tmp_path):
    p = os.path.join(tmp_path, "wav.scp")
    with open(p, "w") as f:
        f.write("utt1 /data/a.wav\nutt2 /data/dir with space/b.wav\n\nbroken\n")
    assert offline.read_list(p) == [("utt1", "/data/a.wav"), ("utt2", "/data/dir with space/b.wav")]
    assert [offline.device_index(r, 8) for r in (1, 8, 9, 16)] == [0, 7, 0, 7]       # run.pl JOB ids start at 1 and wrap


@pytest.mark.parametrize("width,tol", [(2, 1e-4), (3, 1e-6), (4, 1e-6)])
def test_pcm_wav_reader(tmp_path, width, tol):
    g = np.random.default_rng(width)
    x = (0.5 * g.standard_normal(4801)).clip(-0.99, 0.99).astype(np.float32)
    p = os.path.join(tmp_path, f"a{width}.wav")
    _write_wav(p, x, width=width)
    wav, sr = offline.read_audio(p)
    assert sr == 24000 and wav.dtype == torch.float32 and wav.shape == (4801,)
    assert float((wav - torch.from_numpy(x)).abs().max()) < tol
    stereo = np.stack([x, -x], -1).reshape(-1)
    _write_wav(p, stereo, width=width, channels=2)
    assert float(offline.read_audio(p)[0].abs().max()) < 2 * tol             # channels are averaged


def test_npy_pt_readers_and_unknown_extension(tmp_path):
    x = torch.randn(1, 3000)
    np.save(os.path.join(tmp_path, "a.npy"), x.numpy())
    torch.save(x, os.path.join(tmp_path, "a.pt"))
    for ext in ("npy", "pt"):
        wav, sr = offline.read_audio(os.path.join(tmp_path, f"a.{ext}"))
        assert sr == 24000 and torch.equal(wav, x.reshape(-1))
    with pytest.raises(NotImplementedError):
        offline.read_audio(os.path.join(tmp_path, "a.flac"))


class _FakeTokenizer:
    """Records the batches it is given; 'codes' = one int16 frame count per utterance."""

    def __init__(self):
        self.calls = []

    def tokenize_batch(self, wavs, sample_rate, max_batch_seconds):
        self.calls.append((len(wavs), sample_rate, max_batch_seconds))
        return [torch.full((8, -(-w.numel() // 1920)), w.numel() % 100, dtype=torch.int16) for w in wavs]


def test_tokenize_list_chunks_orders_resamples_and_skips_bad_files(tmp_path):
    lens = [1920, 5000, 700, 24000, 3]
    items = []
    for i, n in enumerate(lens):
        p = os.path.join(tmp_path, f"u{i}.wav")
        _write_wav(p, np.zeros(n, np.float32) + 0.1)
        items.append((f"u{i}", p))
    other_rate = os.path.join(tmp_path, "r.wav")
    _write_wav(other_rate, np.zeros(1000, np.float32), sr=16000)        # resampled to 24 kHz: 1500 samples -> 1 frame
    empty = os.path.join(tmp_path, "e.wav")
    _write_wav(empty, np.zeros(0, np.float32))
    items[2:2] = [("rate", other_rate), ("missing", os.path.join(tmp_path, "nope.wav")), ("empty", empty)]
    tok = _FakeTokenizer()
    skipped = []
    out = offline.tokenize_list(tok, items, chunk_size=3, max_batch_seconds=7.0, skipped=skipped)
    assert list(out) == ["u0", "u1", "rate", "u2", "u3", "u4"]                        # input order, bad entries dropped
    assert skipped == ["missing", "empty"]
    assert [tuple(out[f"u{i}"].shape) for i in range(5)] == [(8, -(-n // 1920)) for n in lens]
    assert tuple(out["rate"].shape) == (8, 1) and int(out["rate"][0, 0]) == 1500 % 100
    assert all(v.dtype == torch.int16 for v in out.values())
    assert sum(c[0] for c in tok.calls) == 6 and all(c[1:] == (24000, 7.0) for c in tok.calls) and len(tok.calls) == 3


class _FailingBatchTokenizer(_FakeTokenizer):
    """Batches that contain the 'poison' length fail as a whole; single utterances of that length fail alone."""

    def tokenize_batch(self, wavs, sample_rate, max_batch_seconds):
        if any(w.numel() == 4444 for w in wavs):
            raise RuntimeError("HIP out of memory (simulated)")
        return super().tokenize_batch(wavs, sample_rate, max_batch_seconds)

    def tokenize(self, wav, sample_rate):
        if wav.numel() == 4444:
            raise RuntimeError("HIP out of memory (simulated)")
        return torch.full((8, -(-wav.numel() // 1920)), wav.numel() % 100, dtype=torch.int16)


def test_a_failing_batch_loses_only_the_offending_utterance(tmp_path):
    """offline_codec_tokenization.py:86-101: the reference's try/except is per utterance."""
    items = []
    for i, n in enumerate([2000, 4444, 3000]):
        p = os.path.join(tmp_path, f"u{i}.wav")
        _write_wav(p, np.zeros(n, np.float32) + 0.1)
        items.append((f"u{i}", p))
    skipped = []
    out = offline.tokenize_list(_FailingBatchTokenizer(), items, chunk_size=8, skipped=skipped)
    assert list(out) == ["u0", "u2"] and skipped == ["u1"]


# ---- host-side resampler (rstnet_amd/codec/audio_resample.py: the role of torchaudio.transforms.Resample in mimi_tokenizer.py:66-67)

def test_resample_identity_and_length():
    from rstnet_amd.codec.audio_resample import resample
    x = torch.randn(2, 1001)
    assert resample(x, 24000, 24000) is x
    for o, n in ((16000, 24000), (44100, 24000), (48000, 24000), (8000, 24000), (22050, 24000)):
        y = resample(x, o, n)
        assert y.shape == (2, -(-1001 * n // o)) and y.dtype == torch.float32


@pytest.mark.parametrize("orig", [16000, 44100, 48000])
def test_resample_preserves_tones_below_the_cutoff_and_rejects_above(orig):
    from rstnet_amd.codec.audio_resample import resample
    t = torch.arange(orig, dtype=torch.float64) / orig               # one second
    f_keep = 1000.0
    y = resample(torch.sin(2 * torch.pi * f_keep * t).float()[None], orig, 24000)[0].double()
    tn = torch.arange(y.numel(), dtype=torch.float64) / 24000
    want = torch.sin(2 * torch.pi * f_keep * tn)
    mid = slice(200, y.numel() - 200)                                 # away from the zero-padded edges
    assert float((y[mid] - want[mid]).abs().max()) < 2e-3
    if orig > 24000:                                                  # a tone above the new Nyquist must not alias through
        f_kill = 15000.0
        z = resample(torch.sin(2 * torch.pi * f_kill * t).float()[None], orig, 24000)[0]
        assert float(z[mid].abs().max()) < 2e-2
