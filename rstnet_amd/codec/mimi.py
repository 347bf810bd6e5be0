"""``MimiCodec`` -- drop-in for the reference's ``tools/tokenizer/MimiCodec/model/models/MimiCodec.py`` (and its
``AudioCodec/MimiCodec`` twin): same constructor keywords, same ``state_dict`` keys, same
``encode(audio[B,1,T]) -> codes[B,K,ceil(T/hop)]`` / ``decode(codes) -> wav[B,1,F*hop]`` contract.

The whole path stays channels-last on the device: audio ``[B,1,T]`` *is* ``[B,T,1]``, the RVQ kernel writes the
``[B,K,F]`` code tensor directly and the last convolution writes ``[B,T,1]`` = ``[B,1,T]`` -- no layout
conversion kernel runs in ``encode`` / ``decode``.
"""
from __future__ import annotations

import json
from typing import Dict, List, Optional

import torch
from torch import nn

from . import transformer as Stransformer
from .quantization import SplitResidualVectorQuantizer
from .resample import ConvDownsample1d, ConvTrUpsample1d
from .seanet import SEANetDecoder, SEANetEncoder
from dataclasses import dataclass, field

from .. import ops
from ..graphs import Graphed
from .streaming import StreamingModule


@dataclass
class _MimiState:
    """Per-streaming-session graph wrappers (one per input shape) of the encode / decode step."""
    enc: dict = field(default_factory=dict)
    dec: dict = field(default_factory=dict)
    calls: int = 0

    def reset(self) -> None:
        self.enc.clear()
        self.dec.clear()


class MimiCodec(StreamingModule[_MimiState]):
    code_layout = "bkt"      # class default (also for subclasses with their own constructor, e.g. MimiModel)

    def __init__(self, sample_rate: int = 24000, n_filters: int = 64, encoder_rates: List[int] = [4, 5, 6, 8],
                 compress: int = 2, causal: bool = True, latent_dim: int = 512, codebook_size: int = 4096,
                 codebook_dim: int = 32, rvq_layers: int = 8, num_heads: int = 8, num_layers: int = 8,
                 layer_scale: float = 0.01, context: int = 250, dim_feedforward: int = 2048,
                 semantic_feature_dim: int = 1024, target_frame_rate: float = 12.5, code_layout: str = "bkt"):
        super().__init__()
        if code_layout not in ("bkt", "btk"):
            raise ValueError(f"code_layout must be 'bkt' or 'btk', got {code_layout!r}")
        # "bkt": codes [B, K, F] -- the tokenizer copy (MLLM_v2/tools/tokenizer/MimiCodec/model/models/MimiCodec.py:93-110) and moshi.
        # "btk": codes [B, F, K] -- the AudioCodec twin (AudioCodec/MimiCodec/models/MimiCodec.py:94-111 over
        #        quantization/vq_dc.py:148-162, whose third-party ResidualVQ concatenates the per-level indices on the LAST axis).
        self.code_layout = code_layout
        self.sample_rate = sample_rate
        seanet_kwargs = dict(channels=1, dimension=latent_dim, causal=causal, n_filters=n_filters, n_residual_layers=1,
                             activation="ELU", compress=compress, dilation_base=2, disable_norm_outer_blocks=0, kernel_size=7,
                             residual_kernel_size=3, last_kernel_size=3, norm="none", pad_mode="constant",
                             ratios=list(encoder_rates), true_skip=True)
        quantizer_kwargs = dict(dimension=codebook_dim, n_q=rvq_layers, bins=codebook_size, input_dimension=latent_dim,
                                output_dimension=latent_dim)
        transformer_kwargs = dict(d_model=latent_dim, num_heads=num_heads, num_layers=num_layers, causal=causal,
                                  layer_scale=layer_scale, context=context, conv_layout=True, max_period=10000,
                                  gating="none", norm="layer_norm", positional_embedding="rope",
                                  dim_feedforward=dim_feedforward, input_dimension=latent_dim,
                                  output_dimensions=[latent_dim])
        self.encoder = SEANetEncoder(**seanet_kwargs)
        self.decoder = SEANetDecoder(**seanet_kwargs)
        self.hop_length = 1
        for r in encoder_rates:
            self.hop_length *= r
        self.encoder_frame_rate = sample_rate / self.hop_length
        self.target_frame_rate = target_frame_rate
        self.learnt = True
        stride = int(self.encoder_frame_rate / self.target_frame_rate)
        self.downsample = ConvDownsample1d(stride, dimension=latent_dim, learnt=True, causal=causal)
        self.upsample = ConvTrUpsample1d(stride, dimension=latent_dim, learnt=True, causal=causal, channel_wise=True)
        self.encoder_transformer = Stransformer.ProjectedTransformer(**transformer_kwargs)
        self.decoder_transformer = Stransformer.ProjectedTransformer(**transformer_kwargs)
        self.quantizer = SplitResidualVectorQuantizer(**quantizer_kwargs)
        self.frame_hop = self.hop_length * stride  # samples per code frame (1920 for Mimi)

    def _init_streaming_state(self, batch_size: int) -> _MimiState:
        return _MimiState()

    # ------------------------------------------------------------------ reference API
    def forward(self, audio_data: torch.Tensor, semantic_features: torch.Tensor):
        raise NotImplementedError("the GAN / distillation training forward is out of scope of the inference hot path")

    def encode_latent(self, audio_data: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        """audio ``[B, 1, T]`` -> un-quantised 12.5 Hz latent, channels-last ``[B, F, latent_dim]``.  ``lengths`` (valid samples
        per batch entry of a zero-padded ragged batch): every entry's frames equal those of that utterance encoded alone."""
        B, C, T = audio_data.shape
        assert C == 1, "MimiCodec is mono"
        x = audio_data.contiguous().view(B, T, 1)
        if lengths is not None:
            assert not self.is_streaming, "ragged batches are an offline (non-streaming) facility"
            lengths = torch.as_tensor(lengths, dtype=torch.int32, device=x.device).contiguous()
        if self.is_streaming:
            # a frame step: the history rolls of all convolutions run as one launch at the end of the step
            with ops.hist_batch():
                z = self.encoder.forward_nlc(x, lengths)
                z = self.encoder_transformer.forward_nlc(z)[0]
                return self.downsample.forward_nlc(z)
        z = self.encoder.forward_nlc(x, lengths)
        z = self.encoder_transformer.forward_nlc(z)[0]
        if lengths is not None:
            hop = self.hop_length
            z = ops.mask_tail(z.contiguous(), torch.div(lengths + (hop - 1), hop, rounding_mode="floor").to(torch.int32), replicate=True)
        return self.downsample.forward_nlc(z)

    @torch.no_grad()
    def encode(self, audio_data: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``[B, 1, T]`` fp32 -> ``[B, K, ceil(T / 1920)]`` int64 (streaming: floor, remainder kept in the conv states).
        ``lengths``: see ``encode_latent`` (frames past ``ceil(lengths[b] / 1920)`` of entry b are meaningless)."""
        codes = self._encode_bkt(audio_data, lengths)
        return codes.transpose(1, 2).contiguous() if self.code_layout == "btk" else codes

    def _encode_bkt(self, audio_data: torch.Tensor, lengths: Optional[torch.Tensor]) -> torch.Tensor:
        state = self._streaming_state
        if lengths is not None:
            return self.quantizer.encode_nlc(self.encode_latent(audio_data, lengths))
        if state is None or not audio_data.is_cuda:
            return self.quantizer.encode_nlc(self.encode_latent(audio_data))
        # streaming: after two eager frames every step has the same shapes and buffer addresses -> replay a HIP graph
        # (keyed by ops.persistent_epoch too: a device whose persistent transformer launches needed repairs -- polled here every 64
        # steps, without synchronising -- moves to the layer loop, which is another graph)
        if state.calls % 64 == 0:
            ops.persistent_poll(audio_data.device)
        state.calls += 1
        key = tuple(audio_data.shape) + (ops.persistent_epoch(audio_data.device),)
        g = state.enc.get(key)
        if g is None:
            g = state.enc[key] = Graphed(lambda a: self.quantizer.encode_nlc(self.encode_latent(a)), warmup=2)
        return g(audio_data.contiguous()).clone()

    @torch.no_grad()
    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        """``[B, K, F]`` int64 (``[B, F, K]`` with ``code_layout="btk"``) -> ``[B, 1, F * 1920]`` fp32 (not trimmed, as in the reference)."""
        if self.code_layout == "btk":
            codes = codes.transpose(1, 2)
        state = self._streaming_state
        if state is None or not codes.is_cuda:
            return self._decode(codes)
        key = tuple(codes.shape) + (ops.persistent_epoch(codes.device),)
        g = state.dec.get(key)
        if g is None:
            g = state.dec[key] = Graphed(self._decode, warmup=2)
        return g(codes.contiguous()).clone()

    def _decode(self, codes: torch.Tensor) -> torch.Tensor:
        z = self.quantizer.decode_nlc(codes.contiguous())
        if self.is_streaming:
            with ops.hist_batch():       # see encode_latent
                z = self.upsample.forward_nlc(z)
                z = self.decoder_transformer.forward_nlc(z)[0]
                y = self.decoder.forward_nlc(z)
            return y.view(y.shape[0], 1, y.shape[1])
        z = self.upsample.forward_nlc(z)
        z = self.decoder_transformer.forward_nlc(z)[0]
        y = self.decoder.forward_nlc(z)
        return y.view(y.shape[0], 1, y.shape[1])

    @classmethod
    def from_config(cls, config_path: str) -> "MimiCodec":
        with open(config_path, "r") as f:
            return cls(**json.load(f))

    # ------------------------------------------------------------------ conveniences
    @classmethod
    def from_state_dict(cls, sd: Dict[str, torch.Tensor], **overrides) -> "MimiCodec":
        """Build a model whose hyper-parameters are read off the tensor shapes of a reference ``state_dict``."""
        n_filters = sd["encoder.model.0.conv.conv.weight"].shape[0]
        rates, i = [], 3
        while f"encoder.model.{i}.conv.conv.weight" in sd and sd[f"encoder.model.{i}.conv.conv.weight"].shape[2] > 3:
            rates.append(sd[f"encoder.model.{i}.conv.conv.weight"].shape[2] // 2)
            i += 3
        n_layers = len({k.split(".")[3] for k in sd if k.startswith("encoder_transformer.transformer.layers.")})
        emb = sd["quantizer.rvq_first.vq.layers.0._codebook.embedding_sum"]
        rest = len({k.split(".")[4] for k in sd if k.startswith("quantizer.rvq_rest.vq.layers.")})
        kw = dict(n_filters=n_filters, encoder_rates=list(reversed(rates)), latent_dim=sd["downsample.conv.conv.conv.weight"].shape[0],
                  codebook_size=emb.shape[0], codebook_dim=emb.shape[1], rvq_layers=1 + rest, num_layers=n_layers,
                  dim_feedforward=sd["encoder_transformer.transformer.layers.0.linear1.weight"].shape[0])
        kw.update(overrides)      # e.g. code_layout="btk" for the AudioCodec twin
        model = cls(**kw)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        missing = [k for k in missing if not k.startswith("semantic_mapping_layer")]
        unexpected = [k for k in unexpected if not k.startswith("semantic_mapping_layer")]
        if missing or unexpected:
            raise RuntimeError(f"state_dict mismatch: missing={missing[:5]} unexpected={unexpected[:5]}")
        return model.eval()


class MimiModel(MimiCodec):
    """Drop-in for ``MLLM_v2/moshi/models/compression.py:102-423`` -- the composition form of the same codec that the Moshi
    pipeline uses (``moshi/models/loaders.py:105-139`` builds it): the constructor takes the sub-modules instead of
    hyper-parameters, the ``state_dict`` keys, ``encode`` / ``decode`` and the streaming protocol are those of ``MimiCodec``
    (same kernels, same graphs).  Supported: the causal ``resample_method="conv"`` arrangement with both transformers (what the
    loader builds); ``forward`` (training) is out of scope."""

    def __init__(self, encoder: nn.Module, decoder: nn.Module, quantizer: nn.Module, frame_rate: float, encoder_frame_rate: float,
                 sample_rate: int, channels: int, causal: bool = False, encoder_transformer: Optional[nn.Module] = None,
                 decoder_transformer: Optional[nn.Module] = None, resample_method: str = "interpolate",
                 upsample_channel_wise_bug: bool = True, freeze_encoder: bool = False, freeze_quantizer: bool = False,
                 freeze_quantizer_level: int = -1, torch_compile_encoder_decoder: bool = False):
        StreamingModule.__init__(self)
        assert resample_method in ["interpolate", "conv", "avg_pool"], f"Invalid resample_method {resample_method}"
        if resample_method != "conv" or encoder_transformer is None or decoder_transformer is None or channels != 1:
            raise NotImplementedError("MimiModel: the decode path implements mono audio, resample_method='conv' and both transformers "
                                      "(the arrangement of moshi.models.loaders.get_mimi)")
        assert encoder_frame_rate > frame_rate, "Cannot upsample with conv."
        stride = encoder_frame_rate / frame_rate
        assert stride == int(stride), f"Only integer strides are supported, got {stride}"
        self.encoder, self.decoder, self.quantizer = encoder, decoder, quantizer
        self.encoder_transformer, self.decoder_transformer = encoder_transformer, decoder_transformer
        self._frame_rate, self._sample_rate, self._channels = frame_rate, sample_rate, channels
        self.encoder_frame_rate, self.target_frame_rate = encoder_frame_rate, frame_rate
        self.resample_method, self.torch_compile_encoder_decoder = resample_method, torch_compile_encoder_decoder
        self.freeze_quantizer = freeze_quantizer
        self.freeze_quantizer_level = freeze_quantizer_level if freeze_quantizer_level > 0 else quantizer.num_codebooks
        dimension = encoder.dimension
        assert isinstance(dimension, int), f"Dimension should be int, got {dimension} of type {type(dimension)}."
        self.dimension = dimension
        self.hop_length = encoder.hop_length
        self.learnt = True
        self.downsample = ConvDownsample1d(int(stride), dimension=dimension, learnt=True, causal=causal)
        self.upsample = ConvTrUpsample1d(int(stride), dimension=dimension, learnt=True, causal=causal,
                                         channel_wise=upsample_channel_wise_bug)
        self.frame_hop = self.hop_length * int(stride)

    @property
    def channels(self) -> int:
        return self._channels

    @property
    def frame_rate(self) -> float:
        return self._frame_rate

    @property
    def sample_rate(self) -> int:
        return self._sample_rate

    @property
    def total_codebooks(self) -> int:
        """Total number of quantizer codebooks available."""
        return self.quantizer.total_codebooks

    @property
    def num_codebooks(self) -> int:
        """Active number of codebooks used by the quantizer."""
        return self.quantizer.num_codebooks

    def set_num_codebooks(self, n: int) -> None:
        """Set the active number of codebooks used by the quantizer."""
        self.quantizer.set_num_codebooks(n)

    @property
    def cardinality(self) -> int:
        """Cardinality of each codebook."""
        return self.quantizer.cardinality

    def decode_latent(self, codes: torch.Tensor) -> torch.Tensor:
        """codes ``[B, K, F]`` -> quantised latent ``[B, dimension, F]`` (compression.py:421-423)."""
        return self.quantizer.decode(codes)

    def encode_to_latent(self, x: torch.Tensor, quantize: bool = True) -> torch.Tensor:
        """audio ``[B, 1, T]`` -> 12.5 Hz latent ``[B, dimension, F]``, quantised (the default) or not (compression.py:383-398)."""
        z = self.encode_latent(x)
        if quantize:
            z = self.quantizer.decode_nlc(self.quantizer.encode_nlc(z))
        return z.transpose(1, 2).contiguous()

