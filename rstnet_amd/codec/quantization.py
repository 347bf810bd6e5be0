"""Split residual vector quantiser (inference half) with the reference's module surface
(``quantization/vq.py`` + ``quantization/core_vq.py`` of the MimiCodec tokenizer copy).

``state_dict`` keys: ``rvq_{first,rest}.{input_proj,output_proj}.weight`` and
``rvq_*.vq.layers.{j}._codebook.{embedding_sum, cluster_usage, _initialized}`` (legacy names remapped on load).
The fused hot path (``SplitResidualVectorQuantizer.encode_nlc / decode_nlc``) is three launches each way:
one GEMM for both input projections, one search kernel running all 8 levels (residual in LDS), and the
mirror gather + GEMM on decode.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import nn

from .. import ops
from .conv import _PackedCache, _to_ncl, _to_nlc


class EuclideanCodebook(nn.Module):
    """Buffers + nearest-centroid lookup of ``core_vq.py:73-219`` (training-time EMA logic is out of scope)."""

    def __init__(self, dim: int, codebook_size: int, decay: float = 0.99, epsilon: float = 1e-5,
                 threshold_usage_ratio: float = 0.1, replaced_usage_ratio: float = 1.0, check_unused_every: int = 5):
        super().__init__()
        self.dim, self.codebook_size, self.epsilon, self.decay = dim, codebook_size, epsilon, decay
        self.register_buffer("_initialized", torch.tensor([False], dtype=torch.float))
        self.register_buffer("cluster_usage", torch.ones(codebook_size))
        self.register_buffer("embedding_sum", torch.zeros(codebook_size, dim))
        self._emb = _PackedCache()
        self._emb_tables = _PackedCache()

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs) -> None:
        for old, new in (("inited", "_initialized"), ("cluster_size", "cluster_usage"), ("embed_avg", "embedding_sum"),
                         ("embed_sum", "embedding_sum")):
            if prefix + old in state_dict:
                state_dict[prefix + new] = state_dict.pop(prefix + old)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    @property
    def embedding(self) -> torch.Tensor:
        """``embedding_sum / clamp(cluster_usage, eps)`` (core_vq.py:142-150), computed once per weight version."""
        return self._emb.get((self.embedding_sum, self.cluster_usage),
                             lambda: (self.embedding_sum / self.cluster_usage.clamp(min=self.epsilon)[:, None]).contiguous())

    def _tables(self):
        emb = self.embedding
        return self._emb_tables.get((emb,), lambda: (emb[None].contiguous(),) + ops.rvq_pack(emb[None].contiguous()))

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """``[*, D]`` floats -> ``[*]`` int64 indices of the nearest centroid."""
        assert x.dtype.is_floating_point, f"Input should be floats, got {x.dtype}"
        emb, packed, e2 = self._tables()
        flat = x.reshape(-1, x.shape[-1]).contiguous()
        codes = ops.rvq_search(flat, emb, packed, e2, 1, flat.shape[0], [(0, 1)])
        return codes.view(x.shape[:-1])

    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        assert not codes.dtype.is_floating_point, f"Codes should be integers, got {codes.dtype}"
        emb = self.embedding
        out = ops.rvq_gather(codes.reshape(1, 1, -1).contiguous(), emb[None].contiguous(), [(0, 1)])
        return out.view(*codes.shape, self.dim)


class VectorQuantization(nn.Module):
    """``core_vq.py:222-310`` with ``codebook_dim == dim`` (identity projections)."""

    def __init__(self, dim: int, codebook_size: int, codebook_dim: Optional[int] = None, decay: float = 0.99,
                 epsilon: float = 1e-5, threshold_usage_ratio: float = 0.1, **kwargs):
        super().__init__()
        if codebook_dim is not None and codebook_dim != dim:
            raise NotImplementedError("codebook_dim != dim")
        self.project_in, self.project_out = nn.Identity(), nn.Identity()
        self.epsilon = epsilon
        self._codebook = EuclideanCodebook(dim=dim, codebook_size=codebook_size, decay=decay, epsilon=epsilon,
                                           threshold_usage_ratio=threshold_usage_ratio, **kwargs)
        self.codebook_size = codebook_size

    @property
    def embedding(self) -> torch.Tensor:
        return self._codebook.embedding

    def encode(self, x: torch.Tensor) -> torch.Tensor:       # [B, D, N] -> [B, N]
        return self._codebook.encode(_to_nlc(x))

    def decode(self, codes: torch.Tensor) -> torch.Tensor:   # [B, N] -> [B, D, N]
        return _to_ncl(self._codebook.decode(codes))


class ResidualVectorQuantization(nn.Module):
    """``core_vq.py:313-384``: the sequential residual loop, fused into one kernel launch."""

    def __init__(self, *, num_quantizers: int, codebook_offset: int = 0, **kwargs):
        super().__init__()
        self.layers = nn.ModuleList([VectorQuantization(**kwargs) for _ in range(num_quantizers)])
        self.codebook_offset = codebook_offset
        self._tables_cache = _PackedCache()

    def tables(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        embs = [l.embedding for l in self.layers]

        def build():
            emb = torch.stack(embs).contiguous()
            return (emb,) + ops.rvq_pack(emb)
        return self._tables_cache.get(tuple(embs), build)

    def encode(self, x: torch.Tensor, n_q: Optional[int] = None) -> torch.Tensor:   # [B, D, T] -> [n_q, B, T]
        n_q = n_q or len(self.layers)
        emb, packed, e2 = self.tables()
        B, D, T = x.shape
        codes = ops.rvq_search(_to_nlc(x).reshape(B * T, D), emb, packed, e2, B, T, [(0, n_q)])
        return codes[:, :n_q].transpose(0, 1).contiguous()

    def decode(self, codes: torch.Tensor) -> torch.Tensor:                          # [K, B, T] -> [B, D, T]
        emb, _, _ = self.tables()
        K, B, T = codes.shape
        q = ops.rvq_gather(codes.transpose(0, 1).contiguous(), emb, [(0, K)])
        return _to_ncl(q.view(B, T, -1))


class _Proj(nn.Module):
    """1x1 ``Conv1d`` without bias used as input / output projection (``vq.py:88-96``): weight ``[out, in, 1]``."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 1))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)

    def forward(self, x: torch.Tensor) -> torch.Tensor:   # [B, C, T]
        return _to_ncl(ops.linear(_to_nlc(x), self.weight[:, :, 0].contiguous()))


class ResidualVectorQuantizer(nn.Module):
    """``vq.py:19-176`` (inference methods)."""

    def __init__(self, dimension: int = 128, input_dimension: Optional[int] = None, output_dimension: Optional[int] = None,
                 n_q: int = 8, q_dropout: bool = False, q_first_only_proba: float = 0.0, no_quantization_rate: float = 0.0,
                 bins: int = 1024, decay: float = 0.99, threshold_usage_ratio: float = 0.1, replaced_usage_ratio: float = 1.0,
                 codebook_offset: int = 0, force_projection: bool = False, generator_seed: Optional[int] = None):
        super().__init__()
        self.max_n_q = self.n_q = n_q
        self.dimension = dimension
        self.input_dimension = input_dimension or dimension
        self.output_dimension = output_dimension or dimension
        self.bins = bins
        self.input_proj: nn.Module = (nn.Identity() if self.input_dimension == dimension and not force_projection
                                      else _Proj(self.input_dimension, dimension))
        self.output_proj: nn.Module = (nn.Identity() if self.output_dimension == dimension and not force_projection
                                       else _Proj(dimension, self.output_dimension))
        self.vq = ResidualVectorQuantization(dim=dimension, codebook_size=bins, num_quantizers=n_q, decay=decay,
                                             threshold_usage_ratio=threshold_usage_ratio,
                                             replaced_usage_ratio=replaced_usage_ratio, codebook_offset=codebook_offset)

    def forward(self, x: torch.Tensor, frame_rate: int):
        raise NotImplementedError("training forward (commitment loss, dropout) is out of scope; use encode / decode")

    def encode(self, x: torch.Tensor) -> torch.Tensor:    # [B, C, T] -> [B, K, T]
        if x.shape[-1] == 0:
            return torch.empty((x.shape[0], self.n_q, 0), device=x.device, dtype=torch.int64)
        return self.vq.encode(self.input_proj(x), n_q=self.n_q).transpose(0, 1).contiguous()

    def decode(self, codes: torch.Tensor) -> torch.Tensor:  # [B, K, T] -> [B, C, T]
        return self.output_proj(self.vq.decode(codes.transpose(0, 1).contiguous()))

    @property
    def total_codebooks(self) -> int:
        return self.max_n_q

    @property
    def num_codebooks(self) -> int:
        return self.n_q

    def set_num_codebooks(self, n: int) -> None:
        assert 0 <= n <= self.max_n_q
        self.n_q = n

    @property
    def cardinality(self) -> int:
        return self.bins


class SplitResidualVectorQuantizer(nn.Module):
    """``vq.py:179-358``: one semantic level + (n_q - 1) acoustic levels, each group with its own projections,
    both quantising the same latent."""

    def __init__(self, *, n_q: int = 8, no_quantization_rate: float = 0.0, no_quantization_mode: str = "same",
                 n_q_semantic: int = 1, **kwargs):
        super().__init__()
        assert n_q > n_q_semantic, f"Number of quantizers {n_q} must be larger than the number of semantic quantizers {n_q_semantic}."
        self.max_n_q = n_q
        self.n_q_semantic = n_q_semantic
        self.n_q_acoustic = n_q - n_q_semantic
        kwargs.pop("q_dropout", None)
        kwargs.pop("generator_seed", None)
        self.rvq_first = ResidualVectorQuantizer(n_q=n_q_semantic, force_projection=True, q_dropout=False, **kwargs)
        self.rvq_rest = ResidualVectorQuantizer(n_q=n_q - n_q_semantic, codebook_offset=1, force_projection=True, **kwargs)
        self._fused = _PackedCache()

    # ---- fused channels-last hot path -------------------------------------------------------------------
    def _fused_tables(self):
        f, r = self.rvq_first, self.rvq_rest
        embs = [l.embedding for l in f.vq.layers] + [l.embedding for l in r.vq.layers]
        params = (f.input_proj.weight, r.input_proj.weight, f.output_proj.weight, r.output_proj.weight) + tuple(embs)

        def build():
            w_in = torch.cat([f.input_proj.weight[:, :, 0], r.input_proj.weight[:, :, 0]], 0).float().contiguous()
            w_out = torch.cat([f.output_proj.weight[:, :, 0], r.output_proj.weight[:, :, 0]], 1).float().contiguous()
            emb = torch.stack(embs).contiguous()
            packed, e2 = ops.rvq_pack(emb)
            return w_in, w_out, emb, packed, e2
        return self._fused.get(params, build)

    def _groups(self, n_total: Optional[int] = None) -> List[Tuple[int, int]]:
        ns = self.rvq_first.n_q
        nr = self.rvq_rest.n_q if n_total is None else n_total - ns
        return [(0, ns), (self.n_q_semantic, nr)]

    def encode_nlc(self, z: torch.Tensor) -> torch.Tensor:
        """z ``[B, F, C]`` channels-last latent -> codes ``[B, K, F]`` int64."""
        B, F, C = z.shape
        K = self.rvq_first.n_q + self.rvq_rest.n_q
        if F == 0:
            return torch.empty((B, K, 0), device=z.device, dtype=torch.int64)
        w_in, _, emb, packed, e2 = self._fused_tables()
        x = ops.linear(z.reshape(B * F, C), w_in)           # both 1x1 input projections in one GEMM
        codes = ops.rvq_search(x, emb, packed, e2, B, F, self._groups())
        return codes if K == codes.shape[1] else codes[:, :K].contiguous()

    def decode_nlc(self, codes: torch.Tensor) -> torch.Tensor:
        """codes ``[B, K, F]`` -> quantised latent ``[B, F, C]`` (sum of both groups' output projections)."""
        B, K, F = codes.shape
        _, w_out, emb, _, _ = self._fused_tables()
        if K < self.max_n_q:  # fewer codebooks than trained: the gather only reads levels < K
            pad = torch.zeros(B, self.max_n_q - K, F, device=codes.device, dtype=torch.int64)
            codes = torch.cat([codes, pad], 1)
        q = ops.rvq_gather(codes.contiguous(), emb, self._groups(K))
        return ops.linear(q, w_out).view(B, F, -1)

    # ---- reference surface ([B, C, T]) -----------------------------------------------------------------------
    def forward(self, x: torch.Tensor, frame_rate: int, semantic_features: Optional[torch.Tensor] = None):
        raise NotImplementedError("training forward (losses, distillation) is out of scope; use encode / decode")

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        return self.encode_nlc(_to_nlc(x))

    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        return _to_ncl(self.decode_nlc(codes))

    @property
    def total_codebooks(self) -> int:
        return self.rvq_first.max_n_q + self.rvq_rest.max_n_q

    @property
    def num_codebooks(self) -> int:
        return self.rvq_first.num_codebooks + self.rvq_rest.num_codebooks

    @property
    def n_q(self) -> int:
        return self.rvq_first.n_q + self.rvq_rest.n_q

    @property
    def dimension(self) -> int:
        return self.rvq_first.dimension

    @property
    def semantic_quantizer(self) -> ResidualVectorQuantizer:
        return self.rvq_first

    @property
    def acoustic_quantizer(self) -> ResidualVectorQuantizer:
        return self.rvq_rest

    def set_num_codebooks(self, n: int) -> None:
        assert self.n_q_semantic <= n <= self.total_codebooks
        self.rvq_rest.set_num_codebooks(n - self.n_q_semantic)

    @property
    def cardinality(self) -> int:
        assert self.rvq_rest.cardinality == self.rvq_first.cardinality
        return self.rvq_first.cardinality
