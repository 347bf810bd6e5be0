"""Channels-last functional layer calls of the MimiCodec path, built on ``rstnet_amd.ops`` (HIP only).

Weight repacking (a one-off at load time) uses torch tensor ops on the device; all arithmetic of the forward
path runs in librstnet_hip.so.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from .. import ops


# ----------------------------------------------------------------------------- weight packing (load time)

def pack_conv_weight(w: torch.Tensor, dilation: int = 1) -> torch.Tensor:
    """Conv1d weight ``[Cout, Cin, K]`` -> ``[Cout, K_eff*Cin]`` with index ``tap*Cin + ci`` (dilated taps zero-filled)."""
    cout, cin, k = w.shape
    k_eff = (k - 1) * dilation + 1
    wp = w.new_zeros(cout, k_eff, cin)
    wp[:, ::dilation, :] = w.permute(0, 2, 1)
    return wp.reshape(cout, k_eff * cin).contiguous()


def pack_convtr_weight(w: torch.Tensor, stride: int) -> torch.Tensor:
    """ConvTranspose1d weight ``[Cin, Cout, K]`` -> ``[stride*Cout, q*Cin]``, q = ceil(K/stride):
    ``packed[j*Cout + co][i*Cin + ci] = w[ci][co][j + (q-1-i)*stride]`` (zero where the tap is >= K)."""
    cin, cout, k = w.shape
    q = -(-k // stride)
    wp = w.new_zeros(cin, cout, q * stride)
    wp[:, :, :k] = w
    wp = wp.view(cin, cout, q, stride).flip(2)          # [ci, co, i, j]
    return wp.permute(3, 1, 2, 0).reshape(stride * cout, q * cin).contiguous()


# ----------------------------------------------------------------------------- layers (channels-last)

def conv_out_frames(t_total: int, k_eff: int, stride: int, streaming: bool, pad_total: int) -> int:
    """Number of output frames: ceil-mode with right padding when not streaming (modules/conv.py:50-57),
    floor-mode on the concatenated [previous; x] sequence when streaming (modules/streaming.py:229-231)."""
    if streaming:
        return max(0, (t_total - k_eff) // stride + 1)
    n_frames = (t_total - k_eff + pad_total) / stride + 1
    return max(0, math.ceil(n_frames))


def conv1d(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], *, k_eff: int, stride: int = 1,
           pad_mode: int = ops.PAD_ZERO, act_in: int = ops.ACT_NONE, res: Optional[torch.Tensor] = None,
           hist: Optional[torch.Tensor] = None, act_out: int = ops.ACT_NONE) -> torch.Tensor:
    """Causal conv on ``x [B,T,Cin]``.  Without ``hist``: non-streaming (left pad k_eff-stride, right pad to a full
    last window).  With ``hist [B,P,Cin]``: the window runs over concat(hist, x) and only complete frames are produced."""
    B, T, cin = x.shape
    cout = w_packed.shape[0]
    if hist is None:
        P = k_eff - stride
        t_out = conv_out_frames(T, k_eff, stride, False, P)
    else:
        P = hist.shape[1]
        t_out = conv_out_frames(P + T, k_eff, stride, True, 0)
    return ops.gemm_win(x, w_packed, B=B, T_in=T, T_out=t_out, C_=cin, S=stride, P=P, N=cout, hist=hist if P > 0 else None,
                        bias=bias, res=res, pad_mode=pad_mode, act_in=act_in, act_out=act_out, out_shape=(B, t_out, cout))


def convtr1d(x: torch.Tensor, w_packed: torch.Tensor, bias_tiled: Optional[torch.Tensor], *, kernel: int, stride: int,
             act_in: int = ops.ACT_NONE, hist: Optional[torch.Tensor] = None, act_out: int = ops.ACT_NONE) -> torch.Tensor:
    """Causal transposed conv on ``x [B,T,Cin]`` -> ``[B,T*stride,Cout]``; ``hist [B,q-1,Cin]`` = previous input steps."""
    B, T, cin = x.shape
    q = -(-kernel // stride)
    cout = w_packed.shape[0] // stride
    return ops.gemm_win(x, w_packed, B=B, T_in=T, T_out=T, C_=cin, S=1, P=q - 1, N=stride * cout,
                        hist=hist if q > 1 else None, bias=bias_tiled, act_in=act_in, act_out=act_out,
                        out_shape=(B, T * stride, cout))
