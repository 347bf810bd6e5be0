"""Host-side surface of the `MimiModel` drop-in (moshi/models/compression.py:102-423, loaders.py:105-139) -- no kernels run."""
import os

import numpy as np
import pytest
import torch

from rstnet_amd.codec.loaders import _quantizer_kwargs, _seanet_kwargs, _transformer_kwargs, build_mimi
from rstnet_amd.codec.mimi import MimiModel

G = os.path.join(os.path.dirname(__file__), "golden")


def test_state_dict_keys_and_properties_match_moshi():
    g = np.load(os.path.join(G, "mimi_model.npz"))
    m = build_mimi(n_q=8)
    assert sorted(m.state_dict().keys()) == list(g["keys"])
    props = [m.frame_rate, m.sample_rate, m.channels, m.num_codebooks, m.total_codebooks, m.cardinality]
    assert props == list(g["props"])
    m.set_num_codebooks(4)
    assert m.num_codebooks == 4 and m.total_codebooks == 8
    with pytest.raises(AssertionError):
        m.set_num_codebooks(9)


def test_loader_defaults_are_the_canonical_mimi():
    m = build_mimi()
    assert (m.total_codebooks, m.num_codebooks, m.cardinality, m.dimension, m.frame_hop) == (32, 8, 2048, 512, 1920)
    assert _seanet_kwargs["ratios"] == [8, 6, 5, 4] and _quantizer_kwargs["n_q"] == 32 and _transformer_kwargs["context"] == 250


def test_unsupported_arrangements_fail_loudly():
    m = build_mimi(n_q=8)
    kw = dict(frame_rate=12.5, encoder_frame_rate=25.0, sample_rate=24000, channels=1, causal=True)
    with pytest.raises(NotImplementedError):
        MimiModel(m.encoder, m.decoder, m.quantizer, resample_method="interpolate", encoder_transformer=m.encoder_transformer,
                  decoder_transformer=m.decoder_transformer, **kw)
    with pytest.raises(NotImplementedError):
        MimiModel(m.encoder, m.decoder, m.quantizer, resample_method="conv", **kw)
    with pytest.raises(RuntimeError):      # no CPU fallback behind the module surface
        m.encode(torch.zeros(1, 1, 1920))
