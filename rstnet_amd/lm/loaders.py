"""``get_moshi_lm`` -- the role of ``MLLM_v2/moshi/models/loaders.py:68-98,142-159``: the Moshi-7B ``LMModel`` with the weights
of a checkpoint file, bf16 on the device, no second copy of the 15 GB blob (``LMModel.from_state_dict`` adopts the tensors)."""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Optional, Union

import torch

from .model import LMModel

# moshi/models/loaders.py:68-98 (the keys LMModel interprets; the constants it fixes -- rms_norm_f32, SiLU gating, rope,
# multi-linear per-step depth weights -- are what LMModel.from_state_dict passes itself)
_lm_kwargs = dict(dim=4096, text_card=32000, existing_text_padding_id=3, n_q=16, dep_q=8, card=2048, num_heads=32, num_layers=32,
                  hidden_scale=4.125, context=3000, max_period=10000.0, depformer_dim=1024, depformer_dim_feedforward=int(4.125 * 1024),
                  depformer_num_heads=16, depformer_num_layers=6, delays=[0, 0, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1])


def get_moshi_lm(weights: Union[str, Path, Dict[str, torch.Tensor]], device: Union[torch.device, str] = "cuda",
                 lm_kwargs: Optional[dict] = None) -> LMModel:
    """``weights``: a ``.safetensors`` file, a ``torch.save``d package (``pkg["fsdp_best_state"]["model"]``, loaders.py:153-158) or a
    ``state_dict``.  Tensors are moved to ``device`` as bf16 and adopted without copying; ``lm_kwargs`` overrides the Moshi-7B
    hyper-parameters (e.g. a smaller model trained with the same code)."""
    if isinstance(weights, dict):
        sd = weights
    elif Path(weights).suffix in (".safetensors", ".sft", ".sfts"):
        from safetensors.torch import load_file
        sd = load_file(str(weights))
    else:
        sd = torch.load(weights, map_location="cpu")["fsdp_best_state"]["model"]
    sd = {k: v.to(device=device, dtype=torch.bfloat16) for k, v in sd.items()}
    return LMModel.from_state_dict(sd, dict(lm_kwargs or _lm_kwargs))
