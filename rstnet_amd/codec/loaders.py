"""Construction of the canonical Mimi codec as a ``MimiModel`` -- the role of ``get_mimi`` in
``MLLM_v2/moshi/models/loaders.py:24-66,105-139`` (same hyper-parameters: SEANet 64 filters, ratios [8, 6, 5, 4], two 8-layer
transformers with context 250, split RVQ with 32 x 2048 x 256 codebooks of which 8 are active)."""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Union

import torch

from . import transformer as Stransformer
from .mimi import MimiModel
from .quantization import SplitResidualVectorQuantizer
from .seanet import SEANetDecoder, SEANetEncoder

SAMPLE_RATE = 24000
FRAME_RATE = 12.5

_seanet_kwargs = {
    "channels": 1, "dimension": 512, "causal": True, "n_filters": 64, "n_residual_layers": 1, "activation": "ELU", "compress": 2,
    "dilation_base": 2, "disable_norm_outer_blocks": 0, "kernel_size": 7, "residual_kernel_size": 3, "last_kernel_size": 3,
    "norm": "none", "pad_mode": "constant", "ratios": [8, 6, 5, 4], "true_skip": True,
}
_quantizer_kwargs = {"dimension": 256, "n_q": 32, "bins": 2048, "input_dimension": 512, "output_dimension": 512}
_transformer_kwargs = {
    "d_model": 512, "num_heads": 8, "num_layers": 8, "causal": True, "layer_scale": 0.01, "context": 250, "conv_layout": True,
    "max_period": 10000, "gating": "none", "norm": "layer_norm", "positional_embedding": "rope", "dim_feedforward": 2048,
    "input_dimension": 512, "output_dimensions": [512],
}


def build_mimi(n_q: int = 32, num_codebooks: int = 8) -> MimiModel:
    """The module tree of ``get_mimi`` without weights (``n_q`` trained codebooks, ``num_codebooks`` active)."""
    encoder = SEANetEncoder(**_seanet_kwargs)
    decoder = SEANetDecoder(**_seanet_kwargs)
    model = MimiModel(encoder, decoder, SplitResidualVectorQuantizer(**{**_quantizer_kwargs, "n_q": n_q}), channels=1,
                      sample_rate=SAMPLE_RATE, frame_rate=FRAME_RATE, encoder_frame_rate=SAMPLE_RATE / encoder.hop_length,
                      causal=True, resample_method="conv",
                      encoder_transformer=Stransformer.ProjectedTransformer(**_transformer_kwargs),
                      decoder_transformer=Stransformer.ProjectedTransformer(**_transformer_kwargs))
    model.set_num_codebooks(min(num_codebooks, n_q))
    return model.eval()


def get_mimi(weights: Union[str, Path, Dict[str, torch.Tensor]], device: Union[torch.device, str] = "cuda") -> MimiModel:
    """A ``MimiModel`` with the weights of a checkpoint file (``.safetensors`` or a ``torch.save``d ``{"model": state_dict}``) or of
    a ``state_dict`` given directly; the number of trained codebooks is read off the checkpoint, 8 are active."""
    if isinstance(weights, dict):
        sd = weights
    elif Path(weights).suffix in (".safetensors", ".sft", ".sfts"):
        from safetensors.torch import load_file
        sd = load_file(str(weights))
    else:
        sd = torch.load(weights, map_location="cpu")["model"]
    rest = len({k.split(".")[4] for k in sd if k.startswith("quantizer.rvq_rest.vq.layers.")})
    model = build_mimi(n_q=1 + rest)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    bad = [k for k in list(missing) + list(unexpected) if not k.startswith("semantic_mapping_layer")]
    if bad:
        raise RuntimeError(f"state_dict mismatch: {bad[:6]}")
    return model.to(device)
