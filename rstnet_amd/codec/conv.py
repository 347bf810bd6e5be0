"""Causal streaming Conv1d / ConvTranspose1d with the reference's module surface.

Mirrors ``modules/conv.py`` + the raw layers of ``modules/streaming.py`` of the MimiCodec tokenizer copy:
same class names, constructor arguments, ``state_dict`` keys (``<name>.conv.conv.{weight,bias}`` /
``<name>.convtr.convtr.{weight,bias}``) and ``[B, C, T]`` tensors at ``forward``.  Internally every layer works
channels-last through ``forward_nlc`` (what the fused SEANet containers call) and runs as a windowed GEMM in
``librstnet_hip.so``; the streaming state is the caller-owned *input history* ``[B, n, C]``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple

import torch
from torch import nn

from .. import ops
from . import functional as RF
from .streaming import StreamingModule

CONV_NORMALIZATIONS = frozenset(["none"])  # weight_norm is a training-time parametrisation: not on the hot path


def _check_norm(norm: str) -> None:
    if norm not in CONV_NORMALIZATIONS:
        raise NotImplementedError(f"norm={norm!r}: only 'none' is supported by the inference path")


def get_extra_padding_for_conv1d(x: torch.Tensor, kernel_size: int, stride: int, padding_total: int = 0) -> int:
    """Right padding that makes the last window full (``modules/conv.py:50-57``)."""
    length = x.shape[-1]
    n_frames = (length - kernel_size + padding_total) / stride + 1
    return (math.ceil(n_frames) - 1) * stride + (kernel_size - padding_total) - length


def _to_nlc(x: torch.Tensor) -> torch.Tensor:
    return ops.transpose12(x.contiguous())


_to_ncl = _to_nlc  # the same kernel: [B, R, C] -> [B, C, R]


def _keep_address(old: Optional[torch.Tensor], new: torch.Tensor) -> torch.Tensor:
    """Streaming state update that keeps the buffer (and so its device address) when the shape is unchanged -- in steady
    state every step sees the same history length, which is what lets a whole codec step be replayed as a HIP graph."""
    if new is old:          # rolled in place (ops.hist_update)
        return old
    if old is not None and old.shape == new.shape:
        old.copy_(new)
        return old
    return new


class _PackedCache:
    """Device-side repacked weights, rebuilt when the parameters they derive from change."""

    def __init__(self) -> None:
        self._key: Optional[Tuple] = None
        self._val: Any = None

    def get(self, params, build):
        key = tuple((p.data_ptr(), p._version, p.device) if p is not None else None for p in params)
        if key != self._key:
            with torch.no_grad():
                self._val = build()
            self._key = key
        return self._val


@dataclass
class _StreamingConvState:
    previous: Optional[torch.Tensor] = None  # [B, n, Cin] channels-last input history

    def reset(self) -> None:
        self.previous = None


class RawStreamingConv1d(StreamingModule[_StreamingConvState]):
    """Parameter holder + kernel call for one Conv1d (``modules/streaming.py:205-244``).  No implicit padding."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, padding: int = 0,
                 dilation: int = 1, groups: int = 1, bias: bool = True, device=None, dtype=None):
        super().__init__()
        assert padding == 0, "Padding should be handled outside."
        assert stride <= kernel_size, "stride must be less than kernel_size."
        if groups != 1:
            raise NotImplementedError("grouped Conv1d is not on the MimiCodec path")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation, self.groups = (kernel_size,), (stride,), (dilation,), groups
        self.padding = (0,)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, kernel_size, device=device, dtype=dtype))
        self.bias = nn.Parameter(torch.empty(out_channels, device=device, dtype=dtype)) if bias else None
        self.reset_parameters()
        self._packed = _PackedCache()
        self._packed_aux = _PackedCache()   # layouts wanted by fused neighbours (first / last SEANet conv)

    def reset_parameters(self) -> None:
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(self.weight.shape[1] * self.weight.shape[2])
            nn.init.uniform_(self.bias, -bound, bound)

    @property
    def k_eff(self) -> int:
        return (self.kernel_size[0] - 1) * self.dilation[0] + 1

    def packed_weight(self) -> torch.Tensor:
        return self._packed.get((self.weight,), lambda: RF.pack_conv_weight(self.weight.detach().float(), self.dilation[0]))

    def _init_streaming_state(self, batch_size: int) -> _StreamingConvState:
        return _StreamingConvState()

    def forward_nlc(self, x: torch.Tensor, *, act_in: int = ops.ACT_NONE, res: Optional[torch.Tensor] = None,
                    pad: Optional[Tuple[int, int]] = None, act_out: int = ops.ACT_NONE) -> torch.Tensor:
        """x ``[B, T, Cin]``.  Not streaming: a plain valid convolution, optionally over a virtual padding
        ``pad = (left, pad_mode)`` (right side completed to a full last window).  Streaming: runs on
        concat(previous, x), emits the complete frames and keeps the rest (``modules/streaming.py:224-236``)."""
        state = self._streaming_state
        B, T, C = x.shape
        w, bias = self.packed_weight(), self.bias
        k, s = self.k_eff, self.stride[0]
        if state is None:
            if pad is None:
                t_out = max(0, (T - k) // s + 1)
                return ops.gemm_win(x, w, B=B, T_in=T, T_out=t_out, C_=C, S=s, P=0, N=self.out_channels, bias=bias, res=res,
                                    act_in=act_in, act_out=act_out, out_shape=(B, t_out, self.out_channels))
            left, pad_mode = pad
            t_out = RF.conv_out_frames(T, k, s, False, left)
            return ops.gemm_win(x, w, B=B, T_in=T, T_out=t_out, C_=C, S=s, P=left, N=self.out_channels, bias=bias, res=res,
                                pad_mode=pad_mode, act_in=act_in, act_out=act_out, out_shape=(B, t_out, self.out_channels))
        prev = state.previous
        n_prev = prev.shape[1] if prev is not None else 0
        total = n_prev + T
        t_out = max(0, (total - k) // s + 1)
        y = ops.gemm_win(x, w, B=B, T_in=T, T_out=t_out, C_=C, S=s, P=n_prev, N=self.out_channels,
                         hist=prev if n_prev > 0 else None, bias=bias, res=res, act_in=act_in, act_out=act_out,
                         out_shape=(B, t_out, self.out_channels))
        state.previous = _keep_address(prev, ops.hist_update(x, prev, total - t_out * s))
        return y

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return _to_ncl(self.forward_nlc(_to_nlc(input)))


class RawStreamingConvTranspose1d(StreamingModule[_StreamingConvState]):
    """One ConvTranspose1d (``modules/streaming.py:255-303``).  Not streaming: the full ``(T-1)*S + K`` output.
    Streaming: ``T*S`` outputs per call; the state is the last ``ceil(K/S) - 1`` input steps instead of the
    reference's ``partial`` output buffer (same values up to fp32 summation order)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, padding: int = 0,
                 output_padding: int = 0, groups: int = 1, bias: bool = True, dilation: int = 1, device=None, dtype=None):
        super().__init__()
        assert padding == 0, "Padding should be handled outside."
        assert dilation == 1, "No dilation for now"
        assert stride <= kernel_size, "stride must be less than kernel_size."
        assert output_padding == 0, "Output padding not supported."
        if groups not in (1, in_channels):
            raise NotImplementedError("ConvTranspose1d supports groups == 1 or depth-wise only")
        if groups != 1 and in_channels != out_channels:
            raise NotImplementedError("depth-wise ConvTranspose1d needs in_channels == out_channels")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation, self.groups = (kernel_size,), (stride,), (1,), groups
        self.padding, self.output_padding = (0,), (0,)
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels // groups, kernel_size, device=device, dtype=dtype))
        self.bias = nn.Parameter(torch.empty(out_channels, device=device, dtype=dtype)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(self.weight.shape[0] * self.weight.shape[2])
            nn.init.uniform_(self.bias, -bound, bound)
        self._packed = _PackedCache()

    @property
    def q(self) -> int:
        return -(-self.kernel_size[0] // self.stride[0])

    def packed(self):
        def build():
            w = self.weight.detach().float()
            if self.groups != 1:
                return w[:, 0].contiguous(), None
            b = self.bias.detach().float().repeat(self.stride[0]).contiguous() if self.bias is not None else None
            return RF.pack_convtr_weight(w, self.stride[0]), b
        return self._packed.get((self.weight, self.bias), build)

    def _init_streaming_state(self, batch_size: int) -> _StreamingConvState:
        return _StreamingConvState()

    def forward_nlc(self, x: torch.Tensor, *, act_in: int = ops.ACT_NONE, trimmed: bool = False,
                    act_out: int = ops.ACT_NONE) -> torch.Tensor:
        """x ``[B, T, Cin]`` -> ``[B, T*S (+ K-S when not streaming and not trimmed), Cout]``."""
        state = self._streaming_state
        B, T, C = x.shape
        K, S, q = self.kernel_size[0], self.stride[0], self.q
        w, bias_t = self.packed()
        hist = None
        if state is not None:
            if state.previous is None:
                state.previous = torch.zeros(B, q - 1, C, device=x.device, dtype=torch.float32)
            hist = state.previous
        elif not trimmed and K > S and T > 0:
            # full output = trimmed output of the input extended by q-1 zero steps, cut to (T-1)*S + K
            x = torch.cat([x, x.new_zeros(B, q - 1, C)], dim=1)
        t_in = x.shape[1]
        if T == 0:
            return x.new_empty(B, 0, self.out_channels)
        if self.groups != 1:
            assert act_in == ops.ACT_NONE and act_out == ops.ACT_NONE
            y = ops.convtr_depthwise(x, w, S, hist=hist)
            if self.bias is not None:
                raise NotImplementedError("depth-wise ConvTranspose1d with bias")
        else:
            y = RF.convtr1d(x, w, bias_t, kernel=K, stride=S, act_in=act_in, hist=hist, act_out=act_out)
        if state is not None:
            state.previous = _keep_address(hist, ops.hist_update(x, hist, q - 1))
        elif not trimmed and K > S:
            y = y[:, : (T - 1) * S + K].contiguous()
        del t_in
        return y

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _to_ncl(self.forward_nlc(_to_nlc(x)))


class NormConv1d(nn.Module):
    def __init__(self, *args, causal: bool = False, norm: str = "none", norm_kwargs: Dict[str, Any] = {}, **kwargs):
        super().__init__()
        _check_norm(norm)
        self.conv = RawStreamingConv1d(*args, **kwargs)
        self.norm_type = norm

    def forward(self, x):
        return self.conv(x)


class NormConvTranspose1d(nn.Module):
    def __init__(self, *args, causal: bool = False, norm: str = "none", norm_kwargs: Dict[str, Any] = {}, **kwargs):
        super().__init__()
        _check_norm(norm)
        self.convtr = RawStreamingConvTranspose1d(*args, **kwargs)
        self.norm_type = norm

    def forward(self, x):
        return self.convtr(x)


@dataclass
class _StreamingConv1dState:
    padding_to_add: int
    original_padding_to_add: int

    def reset(self) -> None:
        self.padding_to_add = self.original_padding_to_add


class StreamingConv1d(StreamingModule[_StreamingConv1dState]):
    """Conv1d with built-in causal padding (``modules/conv.py:168-254``)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, dilation: int = 1,
                 groups: int = 1, bias: bool = True, causal: bool = False, norm: str = "none",
                 norm_kwargs: Dict[str, Any] = {}, pad_mode: str = "reflect"):
        super().__init__()
        self.conv = NormConv1d(in_channels, out_channels, kernel_size, stride, dilation=dilation, groups=groups, bias=bias,
                               causal=causal, norm=norm, norm_kwargs=norm_kwargs)
        self.causal = causal
        self.pad_mode = pad_mode
        if not causal:
            raise NotImplementedError("only causal convolutions are on the MimiCodec path")
        if pad_mode not in ("constant", "replicate"):
            raise NotImplementedError(f"pad_mode={pad_mode!r}: the causal path supports 'constant' and 'replicate'")

    @property
    def _stride(self) -> int:
        return self.conv.conv.stride[0]

    @property
    def _kernel_size(self) -> int:
        return self.conv.conv.kernel_size[0]

    @property
    def _effective_kernel_size(self) -> int:
        return self.conv.conv.k_eff

    @property
    def _padding_total(self) -> int:
        return self._effective_kernel_size - self._stride

    def _init_streaming_state(self, batch_size: int) -> _StreamingConv1dState:
        assert self.causal, "streaming is only supported for causal convs"
        return _StreamingConv1dState(self._padding_total, self._padding_total)

    def forward_nlc(self, x: torch.Tensor, *, act_in: int = ops.ACT_NONE, res: Optional[torch.Tensor] = None,
                    act_out: int = ops.ACT_NONE) -> torch.Tensor:
        raw = self.conv.conv
        state = self._streaming_state
        mode = ops.PAD_REPLICATE if self.pad_mode == "replicate" else ops.PAD_ZERO
        if state is None:
            return raw.forward_nlc(x, act_in=act_in, res=res, pad=(self._padding_total, mode), act_out=act_out)
        if state.padding_to_add > 0 and x.shape[1] > 0:
            # the first chunk is left-padded (modules/conv.py:249-253): seed the raw layer's input history with it
            B, _, C = x.shape
            if mode == ops.PAD_REPLICATE:
                seed = x[:, :1].expand(B, state.padding_to_add, C).contiguous()
            else:
                seed = x.new_zeros(B, state.padding_to_add, C)
            rs = raw._streaming_state
            assert rs is not None, "StreamingConv1d is streaming but its raw conv is not"
            rs.previous = seed if rs.previous is None else torch.cat([rs.previous, seed], dim=1)
            state.padding_to_add = 0
        return raw.forward_nlc(x, act_in=act_in, res=res, act_out=act_out)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _to_ncl(self.forward_nlc(_to_nlc(x)))


@dataclass
class _StreamingConvTr1dState:
    def reset(self) -> None:
        pass


class StreamingConvTranspose1d(StreamingModule[_StreamingConvTr1dState]):
    """ConvTranspose1d with built-in causal trimming (``modules/conv.py:265-329``)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, groups: int = 1,
                 bias: bool = True, causal: bool = False, norm: str = "none", trim_right_ratio: float = 1.0,
                 norm_kwargs: Dict[str, Any] = {}):
        super().__init__()
        self.convtr = NormConvTranspose1d(in_channels, out_channels, kernel_size, stride, groups=groups, bias=bias,
                                          causal=causal, norm=norm, norm_kwargs=norm_kwargs)
        self.causal = causal
        self.trim_right_ratio = trim_right_ratio
        if not causal or trim_right_ratio != 1.0:
            raise NotImplementedError("only causal transposed convolutions with trim_right_ratio=1 are on the MimiCodec path")

    def _init_streaming_state(self, batch_size: int) -> _StreamingConvTr1dState:
        assert self.causal, "streaming is only supported for causal convtrs"
        return _StreamingConvTr1dState()

    def forward_nlc(self, x: torch.Tensor, *, act_in: int = ops.ACT_NONE, act_out: int = ops.ACT_NONE) -> torch.Tensor:
        return self.convtr.convtr.forward_nlc(x, act_in=act_in, trimmed=True, act_out=act_out)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _to_ncl(self.forward_nlc(_to_nlc(x)))
