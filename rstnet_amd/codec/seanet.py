"""SEANet encoder / decoder with the reference's module surface (``modules/seanet.py`` of the MimiCodec copy).

The ``nn.Sequential`` index layout of ``.model`` is kept (ELU modules occupy their slots) so that ``state_dict``
keys such as ``encoder.model.3.conv.conv.weight`` line up with reference checkpoints.  The forward pass is
fused: every ELU is applied by the following convolution's operand load and the residual add lives in the
epilogue of the block's 1x1 convolution -- one kernel per convolution, channels-last throughout.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np
import torch
from torch import nn

from .. import ops
from .conv import StreamingConv1d, StreamingConvTranspose1d, _keep_address, _to_ncl, _to_nlc
from .streaming import StreamingAdd, StreamingContainer


class ELU(nn.Module):
    """Slot-compatible stand-in for ``nn.ELU`` -- fused into the next convolution by the containers below;
    a direct call runs the stand-alone HIP activation kernel."""

    def __init__(self, alpha: float = 1.0):
        super().__init__()
        if alpha != 1.0:
            raise NotImplementedError("ELU alpha != 1")
        self.alpha = alpha

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.activation(x.contiguous(), "elu")


def _activation(name: str, params: dict) -> nn.Module:
    if name != "ELU":
        raise NotImplementedError(f"activation {name!r}: the MimiCodec path uses ELU")
    return ELU(**params)


def _run_fused(model: nn.Sequential, x: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Walk a SEANet ``Sequential`` channels-last with three fusions:
    * an ELU whose input has no other consumer is applied ONCE, in the producer's epilogue (``act_out``), instead of on
      every window tap of the consumer's operand load; otherwise it is folded into the consumer's load (``act_in``);
    * ``lengths`` (int32 ``[B]``, valid rows of ``x`` per batch entry; encoder only): the input of every STRIDED convolution is
      zeroed past the entry's length first -- the zeros the reference's ``pad_for_conv1d`` appends per utterance (ELU(0) = 0,
      so it does not matter on which side of the ELU the mask sits) -- and the lengths shrink by ``ceil(. / stride)``;
    * the skip-add of a residual block lives in the epilogue of its 1x1 convolution (or the whole block is one launch);
    * the first (1 -> C) / last (C -> 1) convolution is folded into the neighbouring residual block."""
    layers = list(model)
    convs = (StreamingConv1d, StreamingConvTranspose1d)
    pending = ops.ACT_NONE      # ELU still to be applied to x by its consumer
    i = 0
    while i < len(layers):
        layer = layers[i]
        nxt = layers[i + 1] if i + 1 < len(layers) else None
        nxt2 = layers[i + 2] if i + 2 < len(layers) else None
        if isinstance(layer, ELU):
            assert pending == ops.ACT_NONE
            pending = ops.ACT_ELU
            i += 1
            continue
        # residual block + ELU + decoder.model.14 (Conv1d C -> 1): only the waveform is written
        if (isinstance(layer, SEANetResnetBlock) and isinstance(nxt, ELU) and isinstance(nxt2, StreamingConv1d)
                and pending == ops.ACT_NONE and layer.can_fuse(post=nxt2)):
            x = layer.forward_nlc(x, post=nxt2)
            i += 3
            continue
        # encoder.model.0 (Conv1d 1 -> C) + residual block: one launch, the conv output never leaves the chip
        pre = None
        if (isinstance(layer, StreamingConv1d) and isinstance(nxt, SEANetResnetBlock) and pending == ops.ACT_NONE
                and nxt.can_fuse(pre=layer)):
            pre, layer = layer, nxt
            i += 1
            nxt = layers[i + 1] if i + 1 < len(layers) else None
            nxt2 = layers[i + 2] if i + 2 < len(layers) else None
        # this layer's output feeds only "ELU -> conv": apply that ELU here, once per element
        elu_out = isinstance(nxt, ELU) and isinstance(nxt2, convs)
        if isinstance(layer, convs):
            stride = layer.conv.conv.stride[0] if isinstance(layer, StreamingConv1d) else 1
            if lengths is not None and stride > 1:
                ops.mask_tail(x, lengths)
                lengths = torch.div(lengths + (stride - 1), stride, rounding_mode="floor").to(torch.int32)
            x = layer.forward_nlc(x, act_in=pending, act_out=ops.ACT_ELU_OUT if elu_out else ops.ACT_NONE)
            pending = ops.ACT_NONE
        elif isinstance(layer, SEANetResnetBlock):
            assert pending == ops.ACT_NONE
            x = layer.forward_nlc(x, pre=pre, elu_out=elu_out)
        else:
            raise NotImplementedError(f"unexpected layer {type(layer).__name__} in SEANet")
        i += 2 if elu_out else 1   # the ELU module after an elu_out producer is already done
    if pending != ops.ACT_NONE:
        x = ops.activation(x, "elu")
    return x


def _plain_causal_conv(conv: StreamingConv1d) -> bool:
    raw = conv.conv.conv
    return (conv.causal and conv.pad_mode == "constant" and raw.stride[0] == 1 and raw.dilation[0] == 1
            and raw.bias is not None and not conv.is_streaming)


class SEANetResnetBlock(StreamingContainer):
    """``x + conv_k1(ELU(conv_k3(ELU(x))))`` (``modules/seanet.py:21-94``)."""

    def __init__(self, dim: int, kernel_sizes: List[int] = [3, 1], dilations: List[int] = [1, 1], activation: str = "ELU",
                 activation_params: dict = {"alpha": 1.0}, norm: str = "none", norm_params: Dict[str, Any] = {},
                 causal: bool = False, pad_mode: str = "reflect", compress: int = 2, true_skip: bool = True):
        super().__init__()
        assert len(kernel_sizes) == len(dilations), "Number of kernel sizes should match number of dilations"
        hidden = dim // compress
        block: List[nn.Module] = []
        for i, (kernel_size, dilation) in enumerate(zip(kernel_sizes, dilations)):
            in_chs = dim if i == 0 else hidden
            out_chs = dim if i == len(kernel_sizes) - 1 else hidden
            block += [_activation(activation, activation_params),
                      StreamingConv1d(in_chs, out_chs, kernel_size=kernel_size, dilation=dilation, norm=norm,
                                      norm_kwargs=norm_params, causal=causal, pad_mode=pad_mode)]
        self.block = nn.Sequential(*block)
        self.add = StreamingAdd()
        self.shortcut: nn.Module
        if true_skip:
            self.shortcut = nn.Identity()
        else:
            self.shortcut = StreamingConv1d(dim, dim, kernel_size=1, norm=norm, norm_kwargs=norm_params, causal=causal,
                                            pad_mode=pad_mode)

    def can_fuse(self, pre: Optional[StreamingConv1d] = None, post: Optional[StreamingConv1d] = None) -> bool:
        """True when rst_seanet_resblock_f32 covers this block (not streaming, identity skip, [k, 1] kernels, dilation 1)
        and, if given, the neighbouring first (1 -> C) / last (C -> 1) convolution."""
        convs = [m for m in self.block if isinstance(m, StreamingConv1d)]
        if self.is_streaming or not isinstance(self.shortcut, nn.Identity) or len(convs) != 2:
            return False
        c1, c2 = convs
        if not (_plain_causal_conv(c1) and _plain_causal_conv(c2) and c2.conv.conv.kernel_size[0] == 1):
            return False
        C, H, Kw = c2.conv.conv.out_channels, c1.conv.conv.out_channels, c1.conv.conv.kernel_size[0]
        if c1.conv.conv.in_channels != C or c2.conv.conv.in_channels != H or c1.conv.conv.weight.dtype != torch.float32:
            return False
        K0 = Kf = 0
        if pre is not None:
            if not (_plain_causal_conv(pre) and pre.conv.conv.in_channels == 1 and pre.conv.conv.out_channels == C):
                return False
            K0 = pre.conv.conv.kernel_size[0]
        if post is not None:
            if not (_plain_causal_conv(post) and post.conv.conv.out_channels == 1 and post.conv.conv.in_channels == C):
                return False
            Kf = post.conv.conv.kernel_size[0]
        return ops.resblock_supported(C, H, Kw, pre is not None, post is not None, K0, Kf)

    def _fused(self, x: torch.Tensor, pre: Optional[StreamingConv1d], post: Optional[StreamingConv1d],
               elu_out: bool = False) -> torch.Tensor:
        c1, c2 = [m.conv.conv for m in self.block if isinstance(m, StreamingConv1d)]
        pre_w = post_w = None
        if pre is not None:
            r = pre.conv.conv
            pre_w = (r._packed_aux.get((r.weight,), lambda: r.weight.detach().float()[:, 0, :].contiguous()), r.bias)
        if post is not None:
            r = post.conv.conv
            post_w = (r._packed_aux.get((r.weight,), lambda: r.weight.detach().float()[0].t().contiguous()), r.bias)
        return ops.seanet_resblock(x, c1.packed_weight(), c1.bias, c2.packed_weight(), c2.bias, Kw=c1.kernel_size[0],
                                   pre=pre_w, post=post_w, elu_out=elu_out)

    def _can_fuse_streaming(self) -> bool:
        """Streaming form of ``can_fuse`` (no neighbouring convolution folded in): the fused kernel takes the k-tap convolution's
        input history ``[B, k - 1, C]`` directly."""
        convs = [m for m in self.block if isinstance(m, StreamingConv1d)]
        if not self.is_streaming or not isinstance(self.shortcut, nn.Identity) or len(convs) != 2:
            return False
        c1, c2 = convs
        for c in (c1, c2):
            raw = c.conv.conv
            if not (c.causal and c.pad_mode == "constant" and raw.stride[0] == 1 and raw.dilation[0] == 1 and raw.bias is not None
                    and c.is_streaming and raw.is_streaming):
                return False
        C, H, Kw = c2.conv.conv.out_channels, c1.conv.conv.out_channels, c1.conv.conv.kernel_size[0]
        if c2.conv.conv.kernel_size[0] != 1 or c1.conv.conv.in_channels != C or c2.conv.conv.in_channels != H or Kw < 2:
            return False
        return c1.conv.conv.weight.dtype == torch.float32 and ops.resblock_supported(C, H, Kw)

    def _fused_streaming(self, x: torch.Tensor, elu_out: bool) -> torch.Tensor:
        """One launch per block and frame: the first convolution's history (its last ``k - 1`` input steps, zeros before the first
        chunk: modules/conv.py:249-253, streaming.py:224-236) goes straight into the kernel and is rolled afterwards."""
        c1, c2 = [m for m in self.block if isinstance(m, StreamingConv1d)]
        raw1, raw2 = c1.conv.conv, c2.conv.conv
        st, rs = c1._streaming_state, raw1._streaming_state
        B, T, C = x.shape
        Kw = raw1.kernel_size[0]
        if st.padding_to_add > 0 and T > 0:
            rs.previous = x.new_zeros(B, st.padding_to_add, C)
            st.padding_to_add = 0
        hist = rs.previous
        assert hist is not None and hist.shape[1] == Kw - 1, "fused streaming residual block: unexpected history length"
        y = ops.seanet_resblock(x, raw1.packed_weight(), raw1.bias, raw2.packed_weight(), raw2.bias, Kw=Kw, hist=hist, elu_out=elu_out)
        rs.previous = _keep_address(hist, ops.hist_update(x, hist, Kw - 1))
        if c2._streaming_state is not None:
            c2._streaming_state.padding_to_add = 0
        return y

    def forward_nlc(self, x: torch.Tensor, pre: Optional[StreamingConv1d] = None,
                    post: Optional[StreamingConv1d] = None, elu_out: bool = False) -> torch.Tensor:
        """``elu_out``: return ELU(block(x)) (the caller's next layer is ``ELU -> conv``)."""
        if self.can_fuse(pre, post):
            return self._fused(x, pre, post, elu_out)
        assert pre is None and post is None
        if x.shape[1] > 0 and self._can_fuse_streaming():
            return self._fused_streaming(x, elu_out)
        u = x if isinstance(self.shortcut, nn.Identity) else self.shortcut.forward_nlc(x)
        convs = [m for m in self.block if isinstance(m, StreamingConv1d)]
        h = x
        # x also feeds the skip, so its ELU is applied on the first conv's operand load; a hidden activation whose only consumer
        # is a kernel-1 conv (no streaming history of it is kept) gets its ELU once, in the producer's epilogue
        pending = ops.ACT_ELU
        for i, conv in enumerate(convs[:-1]):
            nxt = convs[i + 1]
            in_producer = nxt._effective_kernel_size == 1 and nxt._stride == 1
            h = conv.forward_nlc(h, act_in=pending, act_out=ops.ACT_ELU_OUT if in_producer else ops.ACT_NONE)
            pending = ops.ACT_NONE if in_producer else ops.ACT_ELU
        last = convs[-1]
        if h.shape[1] == u.shape[1] and last._stride == 1 and last._effective_kernel_size == 1:
            # skip-add (and the caller's ELU) fused into the epilogue
            return last.forward_nlc(h, act_in=pending, res=u, act_out=ops.ACT_ELU_OUT if elu_out else ops.ACT_NONE)
        v = last.forward_nlc(h, act_in=pending)
        y = _to_nlc(self.add(_to_ncl(u), _to_ncl(v)))
        return ops.activation(y, "elu") if elu_out else y

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _to_ncl(self.forward_nlc(_to_nlc(x)))


class SEANetEncoder(StreamingContainer):
    """``modules/seanet.py:97-241``; ``forward`` takes ``[B, channels, T]`` and returns ``[B, dimension, T/hop]``."""

    def __init__(self, channels: int = 1, dimension: int = 128, n_filters: int = 32, n_residual_layers: int = 3,
                 ratios: List[int] = [8, 5, 4, 2], activation: str = "ELU", activation_params: dict = {"alpha": 1.0},
                 norm: str = "none", norm_params: Dict[str, Any] = {}, kernel_size: int = 7, last_kernel_size: int = 7,
                 residual_kernel_size: int = 3, dilation_base: int = 2, causal: bool = False, pad_mode: str = "reflect",
                 true_skip: bool = True, compress: int = 2, disable_norm_outer_blocks: int = 0,
                 mask_fn: Optional[nn.Module] = None, mask_position: Optional[int] = None):
        super().__init__()
        if mask_fn is not None:
            raise NotImplementedError("mask_fn is a training-time feature")
        self.channels, self.dimension, self.n_filters = channels, dimension, n_filters
        self.ratios = list(reversed(ratios))
        self.n_residual_layers = n_residual_layers
        self.hop_length = int(np.prod(self.ratios))
        self.n_blocks = len(self.ratios) + 2
        self.disable_norm_outer_blocks = disable_norm_outer_blocks
        assert 0 <= disable_norm_outer_blocks <= self.n_blocks
        mult = 1
        model: List[nn.Module] = [StreamingConv1d(channels, mult * n_filters, kernel_size, norm=norm, norm_kwargs=norm_params,
                                                  causal=causal, pad_mode=pad_mode)]
        for ratio in self.ratios:
            for j in range(n_residual_layers):
                model += [SEANetResnetBlock(mult * n_filters, kernel_sizes=[residual_kernel_size, 1],
                                            dilations=[dilation_base ** j, 1], norm=norm, norm_params=norm_params,
                                            activation=activation, activation_params=activation_params, causal=causal,
                                            pad_mode=pad_mode, compress=compress, true_skip=true_skip)]
            model += [_activation(activation, activation_params),
                      StreamingConv1d(mult * n_filters, mult * n_filters * 2, kernel_size=ratio * 2, stride=ratio, norm=norm,
                                      norm_kwargs=norm_params, causal=causal, pad_mode=pad_mode)]
            mult *= 2
        model += [_activation(activation, activation_params),
                  StreamingConv1d(mult * n_filters, dimension, last_kernel_size, norm=norm, norm_kwargs=norm_params,
                                  causal=causal, pad_mode=pad_mode)]
        self.model = nn.Sequential(*model)

    def forward_nlc(self, x: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        return _run_fused(self.model, x, lengths)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _to_ncl(self.forward_nlc(_to_nlc(x)))


class SEANetDecoder(StreamingContainer):
    """``modules/seanet.py:244-395``; ``forward`` takes ``[B, dimension, T]`` and returns ``[B, channels, T*hop]``."""

    def __init__(self, channels: int = 1, dimension: int = 128, n_filters: int = 32, n_residual_layers: int = 3,
                 ratios: List[int] = [8, 5, 4, 2], activation: str = "ELU", activation_params: dict = {"alpha": 1.0},
                 final_activation: Optional[str] = None, final_activation_params: Optional[dict] = None, norm: str = "none",
                 norm_params: Dict[str, Any] = {}, kernel_size: int = 7, last_kernel_size: int = 7,
                 residual_kernel_size: int = 3, dilation_base: int = 2, causal: bool = False, pad_mode: str = "reflect",
                 true_skip: bool = True, compress: int = 2, disable_norm_outer_blocks: int = 0,
                 trim_right_ratio: float = 1.0):
        super().__init__()
        if final_activation is not None:
            raise NotImplementedError("final_activation is not used by MimiCodec")
        self.dimension, self.channels, self.n_filters = dimension, channels, n_filters
        self.ratios = list(ratios)
        self.n_residual_layers = n_residual_layers
        self.hop_length = int(np.prod(self.ratios))
        self.n_blocks = len(self.ratios) + 2
        self.disable_norm_outer_blocks = disable_norm_outer_blocks
        assert 0 <= disable_norm_outer_blocks <= self.n_blocks
        mult = int(2 ** len(self.ratios))
        model: List[nn.Module] = [StreamingConv1d(dimension, mult * n_filters, kernel_size, norm=norm, norm_kwargs=norm_params,
                                                  causal=causal, pad_mode=pad_mode)]
        for ratio in self.ratios:
            model += [_activation(activation, activation_params),
                      StreamingConvTranspose1d(mult * n_filters, mult * n_filters // 2, kernel_size=ratio * 2, stride=ratio,
                                               norm=norm, norm_kwargs=norm_params, causal=causal,
                                               trim_right_ratio=trim_right_ratio)]
            for j in range(n_residual_layers):
                model += [SEANetResnetBlock(mult * n_filters // 2, kernel_sizes=[residual_kernel_size, 1],
                                            dilations=[dilation_base ** j, 1], activation=activation,
                                            activation_params=activation_params, norm=norm, norm_params=norm_params,
                                            causal=causal, pad_mode=pad_mode, compress=compress, true_skip=true_skip)]
            mult //= 2
        model += [_activation(activation, activation_params),
                  StreamingConv1d(n_filters, channels, last_kernel_size, norm=norm, norm_kwargs=norm_params, causal=causal,
                                  pad_mode=pad_mode)]
        self.model = nn.Sequential(*model)

    def forward_nlc(self, z: torch.Tensor) -> torch.Tensor:
        return _run_fused(self.model, z)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        return _to_ncl(self.forward_nlc(_to_nlc(z)))
