"""25 Hz <-> 12.5 Hz learnt resampling (``modules/resample.py`` of the MimiCodec copy): a dense strided
convolution with replicate padding on the way down, a depth-wise transposed convolution on the way up."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .conv import StreamingConv1d, StreamingConvTranspose1d, _to_ncl, _to_nlc


class ConvDownsample1d(nn.Module):
    """``modules/resample.py:14-65``.  Keys: ``conv.conv.conv.weight``."""

    def __init__(self, stride: int, dimension: Optional[int] = None, causal: bool = False, learnt: bool = False,
                 channel_wise: bool = False):
        super().__init__()
        if not learnt or channel_wise:
            raise NotImplementedError("MimiCodec uses the learnt, dense down-sampling convolution")
        assert dimension is not None, "Dimension required for learnt convolutions."
        self.learnt, self.channel_wise = learnt, channel_wise
        self.conv = StreamingConv1d(dimension, dimension, kernel_size=2 * stride, stride=stride, causal=causal, groups=1,
                                    bias=False, pad_mode="replicate")

    def forward_nlc(self, x: torch.Tensor) -> torch.Tensor:
        return self.conv.forward_nlc(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _to_ncl(self.forward_nlc(_to_nlc(x)))


class ConvTrUpsample1d(nn.Module):
    """``modules/resample.py:68-119``.  Keys: ``convtr.convtr.convtr.weight`` ([C, 1, 2*stride])."""

    def __init__(self, stride: int, dimension: Optional[int] = None, causal: bool = False, learnt: bool = False,
                 channel_wise: bool = False):
        super().__init__()
        if not learnt:
            raise NotImplementedError("MimiCodec uses the learnt up-sampling convolution")
        assert dimension is not None, "Dimension required for learnt convolutions."
        self.learnt, self.channel_wise = learnt, channel_wise
        self.convtr = StreamingConvTranspose1d(dimension, dimension, kernel_size=2 * stride, stride=stride, causal=causal,
                                               groups=dimension if channel_wise else 1, bias=False)

    def forward_nlc(self, x: torch.Tensor) -> torch.Tensor:
        return self.convtr.forward_nlc(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _to_ncl(self.forward_nlc(_to_nlc(x)))
