"""Tensor-level front of the C ABI: PyTorch is used for device memory and streams only.

Every function takes / returns CUDA (HIP) fp32 tensors in the CHANNELS-LAST layout ``[B, T, C]`` unless
stated otherwise, launches on the current stream OF THE DEVICE THE TENSORS LIVE ON (every public function runs under a
device guard for its first tensor argument, so a process may drive ``cuda:N`` without making it the current device) and raises
if handed CPU tensors -- there is no host fallback.

Scratch (split-K partials + arrival counters, attention split workspaces, packed operand buffers, RVQ atomic keys) is cached
per (device, STREAM, shape): launches on one stream are ordered, so layers of equal shape share a buffer; work issued on another
stream gets its own.  A captured graph owns the buffers of its capture stream -- replays of graphs captured on the same stream
must not overlap each other (the server and the pipelines run one session at a time, as the reference's lock does).
"""
from __future__ import annotations

import ctypes as C
import functools
import math
from typing import NamedTuple, Optional, Sequence, Tuple

import torch

from . import _lib

ACT_NONE, ACT_ELU, ACT_GELU = 0, 1, 1   # act_in: 1 = ELU ; act_out: 1 = GELU
ACT_ELU_OUT = 2                          # act_out: ELU applied last (after the residual)
PAD_ZERO, PAD_REPLICATE = 0, 1


# Optional per-launch instrumentation (bench.py's roofline leg): a list that receives
# (kernel name, start event, end event, algorithmic flops, algorithmic bytes) for every GEMM launch.
PROFILE = None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _first_tensor(args):
    for a in args:
        if isinstance(a, torch.Tensor):
            return a
        if isinstance(a, (tuple, list)) and a and isinstance(a[0], torch.Tensor):
            return a[0]
    return None


def _on_tensor_device(fn):
    """Device guard: run ``fn`` with the device of its first tensor argument current, so that ``_stream()``, the scratch
    allocations and the launch itself all address that device (a tool driving ``cuda:3`` need not call ``set_device``)."""
    @functools.wraps(fn)
    def guarded(*args, **kwargs):
        t = _first_tensor(args)
        if t is not None and t.is_cuda and t.device.index != torch.cuda.current_device():
            with torch.cuda.device(t.device):
                return fn(*args, **kwargs)
        return fn(*args, **kwargs)
    return guarded


class _PackedWeights:
    """Cache of device-side re-packed copies of weight tensors, keyed by (device, address, shape).  An entry dies with the
    STORAGE of its source (weak reference): per-step slices of a parameter are fresh tensor objects on every call but share
    the parameter's storage, and an address that the allocator hands out again after a free can never hit a stale entry.
    A changed ``_version`` (in-place update of the weights) re-packs."""

    def __init__(self):
        self._d: dict = {}

    def get(self, w: torch.Tensor, build):
        from torch.multiprocessing.reductions import StorageWeakRef
        key = (w.device, w.data_ptr(), tuple(w.shape), w.dtype)
        hit = self._d.get(key)
        if hit is not None and hit[1] == w._version and not hit[2].expired():
            return hit[0]
        if len(self._d) > 64:
            for k in [k for k, v in self._d.items() if v[2].expired()]:
                del self._d[k]
        packed = build()
        self._d[key] = (packed, w._version, StorageWeakRef(w.untyped_storage()))
        return packed

    def clear(self) -> None:
        self._d.clear()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr() if t.numel() > 0 else None


def _chk(t: Optional[torch.Tensor], name: str, dtype=torch.float32, contiguous: bool = True) -> None:
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"rstnet_amd.ops: `{name}` is on {t.device}; the HIP path needs a CUDA/HIP tensor (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"rstnet_amd.ops: `{name}` must be {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"rstnet_amd.ops: `{name}` must be contiguous")


_gemm_scratch: dict = {}


def _scratch(cache: dict, device, shape_key: tuple, build):
    """Per-(device, stream, shape) scratch with a twin for graph capture.  A buffer that is first allocated INSIDE a capture is
    zero-filled by a node of the graph, i.e. by one more launch on every replay (measured: 19 such fills per codec frame); so the
    first eager use of a shape also creates the entry captures will look up (key "graph": all captured graphs of a device share
    it -- their replays must not overlap, see the module docstring)."""
    capturing = torch.cuda.is_current_stream_capturing()
    key = (device, "graph" if capturing else _stream()) + shape_key
    sc = cache.get(key)
    if sc is None:
        sc = cache[key] = build()
        if not capturing and (device, "graph") + shape_key not in cache:
            cache[(device, "graph") + shape_key] = build()
    return sc


# K split of the medium-row GEMMs across workgroups: None = rst_gemm_win_split_plan; an int overrides it (probes / A/B runs only)
GEMM_WIN_SPLIT = None


def _gemm_split_scratch(device, M: int, N: int, K: int):
    """Split-K plan + scratch of the few- / medium-row (streaming step) GEMMs, cached per (stream, shape); launches on one stream
    are ordered and the counters re-arm themselves, so layers of equal shape share the buffers."""
    if M > 4096 or M == 0:
        return 1, None, None

    def build():
        sk = int(_lib.lib().rst_gemm_win_split_plan(M, N, K)) if GEMM_WIN_SPLIT is None else int(GEMM_WIN_SPLIT)
        if sk > 1:
            return (sk, torch.empty(sk, M, N, device=device, dtype=torch.float32),
                    torch.zeros(int(_lib.lib().rst_gemm_win_split_tiles(M, N)), device=device, dtype=torch.int32))
        return (1, None, None)
    return _scratch(_gemm_scratch, device, ("gemm_win", M, N, K, GEMM_WIN_SPLIT), build)


_skinny_f32_weights = _PackedWeights()
SKINNY_F32_MAX_ROWS = 128

# Large launches (more than 4096 rows, N > 64, K % 64 == 0 -- the batched encode / decode of utterances) run on the bf16 matrix
# instruction with every fp32 operand split into three bf16 planes (rst_gemm_win_b3_f32: six products per fp32 product, fp32
# accuracy, 2.7x the f32 instruction's rate).  False: the f32 matrix instruction everywhere (the A/B switch of tools/ab.py).
GEMM_B3 = True
# End-to-end streaming (pipeline.StreamingPipeline): encode -> LM frame -> decode as ONE captured graph per frame once the delay line is
# full.  False: the three module calls with their own graphs (the A/B switch of tools/ab.py).
PIPELINE_FUSE = True
_b3_weights = _PackedWeights()


def gemm_win_b3_pack_weight(w: torch.Tensor) -> torch.Tensor:
    """fp32 ``[N, K]`` -> its three bf16 planes in the operand order of the large-M kernel (rst_gemm_win_b3_pack_weight), cached per
    storage / version like the other packed copies."""
    _chk(w, "w")
    N, K = w.shape

    def build():
        n = int(_lib.lib().rst_gemm_win_b3_weight_elems(N, K))
        if n <= 0:
            raise ValueError(f"rstnet_amd.ops: no three-plane form for a [{N}, {K}] weight")
        w3 = torch.empty(n, device=w.device, dtype=torch.int16)
        _lib.check(_lib.lib().rst_gemm_win_b3_pack_weight(_ptr(w), _ptr(w3), N, K, _stream()))
        return w3
    return _b3_weights.get(w, build)


def _b3_shape(M: int, N: int, K: int) -> bool:
    return GEMM_B3 and M > 4096 and N > 64 and K % 64 == 0


@functools.lru_cache(maxsize=1024)
def _b3_supported(B: int, T_in: int, T_out: int, C_: int, K: int, N: int, S: int, P: int, pad_mode: int, x_bstride: int, has_hist: bool) -> bool:
    return bool(_lib.lib().rst_gemm_win_b3_supported(B, T_in, T_out, C_, K, N, S, P, pad_mode, x_bstride, int(has_hist)))


def _b3_route(x, hist, w, B: int, T_in: int, T_out: int, C_: int, K: int, N: int, S: int, P: int, pad_mode: int) -> bool:
    """True when this launch runs on the three-plane bf16 kernel: the LIBRARY's predicate (rst_gemm_win_b3_supported: shape, padding,
    4 GB buffer-offset reach) plus 16-byte-aligned operands -- rst_gemm_win_b3_f32 refuses everything else, so a profile row labelled
    `gemm_win_b3` names the kernel that ran."""
    if not GEMM_B3 or not _b3_shape(B * T_out, N, K):
        return False
    if x.data_ptr() % 16 or w.data_ptr() % 16:
        return False
    return _b3_supported(B, T_in, T_out, C_, K, N, S, P, pad_mode, T_in * C_, hist is not None)


def skinny_f32_pack_weight(w: torch.Tensor) -> torch.Tensor:
    """fp32 ``[N, K]`` -> MFMA-ordered copy ``[ceil(N/32)*32, ceil(K/8)*8]`` (rst_skinny_f32_pack_weight), cached per storage."""
    _chk(w, "w")
    N, K = w.shape

    def build():
        wp = torch.empty((N + 31) // 32 * 32, (K + 7) // 8 * 8, device=w.device, dtype=torch.float32)
        _lib.check(_lib.lib().rst_skinny_f32_pack_weight(_ptr(w), _ptr(wp), N, K, _stream()))
        return wp
    return _skinny_f32_weights.get(w, build)


# K split of the few-row GEMMs across workgroups: None = rst_skinny_f32_split_plan; an int overrides it (probes / A/B runs only)
SKINNY_F32_SPLIT = None

# Plain few-row linears WITHOUT a LayerNorm in front (no window) read their rows row-major inside the GEMM (rst_linear_few_rows_f32): no
# packing launch, the same bits.  False: pack, then the GEMM on the packed operand (the A/B switch of tools/probes/few_row_linear_probe.py).
# (With a LayerNorm the packing launch stays: applying it inside the GEMM was built twice and measured slower, csrc/skinny_f32.hip.)
SKINNY_F32_ROWS = True

# LayerNorm in front of a few-row linear (streamed transformer layers at more than two streams): applied by the packing launch
# (rst_skinny_f32_pack_ln) instead of a launch of its own.  False: LayerNorm, then pack (the A/B switch of tools/ab.py).
SKINNY_F32_PACK_LN = True


class PackedRows:
    """The result of a few-row linear kept in the packed operand order of the NEXT few-row linear (``ops.linear(..., out_packed=True)``):
    ``xp [32 | 64 | 128, N]`` fp32 as rst_skinny_f32_pack_win lays rows out, ``shape`` the logical ``[..., N]``.  Only ``ops.linear``
    consumes it."""

    def __init__(self, xp: torch.Tensor, shape: tuple):
        self.xp, self.shape = xp, tuple(shape)

    def contiguous(self):
        return self


# linear1 -> GELU -> linear2 of a streamed transformer layer (few-row route): linear1 writes the operand of linear2 in packed order,
# no packing launch between them.  False: row-major result + pack (the A/B switch of tools/ab.py).
SKINNY_F32_CHAIN = True


def linear_chains(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor) -> bool:
    """True when ``ops.linear(x, w1, out_packed=True)`` may feed ``ops.linear(., w2)``: both on the few-row skinny route."""
    K = x.shape[-1]
    M = x.numel() // K if K else 0
    return bool(SKINNY_F32_CHAIN and x.is_cuda and M > 4 and _few_rows(M, w1.shape[0], K) and _few_rows(M, w2.shape[0], w1.shape[0])
                and w1.shape[0] % 8 == 0 and w2.shape[1] == w1.shape[0])


def _gemm_few_rows(x, hist, w, bias, res, scale, out, B, T_in, T_out, C_, K, N, S, P, pad_mode, act_in, act_out, ln=None,
                   out_packed: bool = False) -> None:
    """The streaming-step route of gemm_win / linear: gather + pack the activation windows, then the few-row fp32 skinny GEMM.
    ``ln = (gamma, beta, eps)`` (plain linears only): the LayerNorm of the rows, applied while they are packed.  ``x`` may be a
    ``PackedRows`` (no packing at all); ``out_packed``: ``out`` is the ``xp`` of a ``PackedRows``."""
    M = B * T_out
    wp = skinny_f32_pack_weight(w)
    dev = x.xp.device if isinstance(x, PackedRows) else x.device

    def build():
        sk = int(_lib.lib().rst_skinny_f32_split_plan(M, N, K)) if SKINNY_F32_SPLIT is None else int(SKINNY_F32_SPLIT)
        return (sk, torch.empty(sk, M, N, device=dev, dtype=torch.float32),
                torch.zeros(((M + 31) // 32) * ((N + 31) // 32), device=dev, dtype=torch.int32)) if sk > 1 else (1, None, None)
    sc = _scratch(_gemm_scratch, dev, ("skinny", M, N, K, SKINNY_F32_SPLIT), build)
    plain = hist is None and S == 1 and P == 0 and T_in == T_out and C_ == K and act_in == ACT_NONE
    if SKINNY_F32_ROWS and plain and ln is None and not isinstance(x, PackedRows) and K % 8 == 0 and x.data_ptr() % 16 == 0:
        # the rows as they are: one launch
        _lib.check(_lib.lib().rst_linear_few_rows_f32(_ptr(x), K, _ptr(wp), _ptr(bias), _ptr(res), _ptr(scale), _ptr(out), M, N, K, N, act_out,
                                                     sc[0], _ptr(sc[1]), _ptr(sc[2]), int(out_packed), _stream()))
        return
    if isinstance(x, PackedRows):
        xp = x.xp
        if not (xp.shape[1] == wp.shape[1] and ln is None and hist is None):
            raise ValueError(f"rstnet_amd.ops: a PackedRows operand of {tuple(xp.shape)} does not fit packed weights {tuple(wp.shape)} / takes no LayerNorm or history")
    else:
        xp = torch.empty(32 if M <= 32 else (64 if M <= 64 else 128), wp.shape[1], device=x.device, dtype=torch.float32)
    if isinstance(x, PackedRows):
        pass
    elif ln is not None:
        if not (hist is None and S == 1 and P == 0 and T_in == T_out and C_ == K and act_in == ACT_NONE):
            raise ValueError("rstnet_amd.ops: a LayerNorm can only be folded into the packing launch of a plain linear")
        _chk(ln[0], "ln gamma")
        _chk(ln[1], "ln beta")
        _lib.check(_lib.lib().rst_skinny_f32_pack_ln(_ptr(x), _ptr(ln[0]), _ptr(ln[1]), float(ln[2]), _ptr(xp), M, K, _stream()))
    else:
        _lib.check(_lib.lib().rst_skinny_f32_pack_win(_ptr(x), _ptr(hist), _ptr(xp), B, T_in, T_out, C_, K, S, P, pad_mode, T_in * C_, act_in,
                                                     _stream()))
    _lib.check(_lib.lib().rst_gemm_skinny_f32(_ptr(xp), _ptr(wp), _ptr(bias), _ptr(res), _ptr(scale), _ptr(out), M, N, K, N, act_out,
                                             sc[0], _ptr(sc[1]), _ptr(sc[2]), int(out_packed), _stream()))


def _few_rows(M: int, N: int, K: int) -> bool:
    # weight-bandwidth-bound shapes only: a handful of rows against at least 256 KB of weights
    return 1 <= M <= SKINNY_F32_MAX_ROWS and N * K >= 65536


def gemm_win(x: torch.Tensor, w: torch.Tensor, *, B: int, T_in: int, T_out: int, C_: int, S: int, P: int, N: int,
             hist: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
             res: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None, pad_mode: int = PAD_ZERO,
             act_in: int = ACT_NONE, act_out: int = ACT_NONE, out: Optional[torch.Tensor] = None,
             out_shape: Optional[Tuple[int, ...]] = None) -> torch.Tensor:
    """rst_gemm_win_f32.  ``w`` is ``[N, K]``; the output is ``[B*T_out, N]`` reshaped to ``out_shape``."""
    for t, n in ((x, "x"), (w, "w"), (hist, "hist"), (bias, "bias"), (res, "res"), (scale, "scale")):
        _chk(t, n)
    K = w.shape[1]
    assert w.shape[0] == N
    if out is None:
        out = torch.empty(out_shape if out_shape is not None else (B, T_out, N), device=x.device, dtype=torch.float32)
    else:
        _chk(out, "out")
    assert out.numel() == B * T_out * N, (tuple(out.shape), B, T_out, N)
    if res is not None:
        assert res.numel() == out.numel()
    if hist is not None:
        assert hist.numel() == B * P * C_, (tuple(hist.shape), B, P, C_)
    prof = PROFILE
    if _few_rows(B * T_out, N, K):        # the same route with and without instrumentation: profiles describe the shipped path
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _gemm_few_rows(x, hist, w, bias, res, scale, out, B, T_in, T_out, C_, K, N, S, P, pad_mode, act_in, act_out)
        if prof is not None:
            e1.record()
            prof.append(("gemm_skinny_f32", e0, e1, 2.0 * B * T_out * N * K, 4 * (w.numel() + x.numel() + out.numel()), (B * T_out, N, K)))
        return out
    split_k, ws, cnt = _gemm_split_scratch(x.device, B * T_out, N, K)
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    # (zero padding only: a history buffer / replicate padding keeps the launch on the f32 instruction)
    b3 = split_k <= 1 and _b3_route(x, hist, w, B, T_in, T_out, C_, K, N, S, P, pad_mode)
    if b3:
        _lib.check(_lib.lib().rst_gemm_win_b3_f32(_ptr(x), _ptr(hist), _ptr(w), _ptr(gemm_win_b3_pack_weight(w)), _ptr(bias), _ptr(res),
                                                  _ptr(scale), _ptr(out), B, T_in, T_out, C_, K, N, S, P, pad_mode, T_in * C_, N, act_in,
                                                  act_out, _stream()))
    else:
        _lib.check(_lib.lib().rst_gemm_win_f32(_ptr(x), _ptr(hist), _ptr(w), _ptr(bias), _ptr(res), _ptr(scale), _ptr(out),
                                               B, T_in, T_out, C_, K, N, S, P, pad_mode, T_in * C_, N, act_in, act_out,
                                               split_k, _ptr(ws), _ptr(cnt), _stream()))
    if prof is not None:
        e1.record()
        nbytes = 4 * (x.numel() + w.numel() + out.numel() + (res.numel() if res is not None else 0))
        prof.append(("gemm_win_b3" if b3 else "gemm_win", e0, e1, 2.0 * B * T_out * N * K, nbytes, (B * T_out, N, K)))
    return out


def linear(x, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, res: Optional[torch.Tensor] = None,
           scale: Optional[torch.Tensor] = None, act_out: int = ACT_NONE,
           ln: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None, out_packed: bool = False):
    """``y = epi(LN(x) @ w.T + bias)`` over the last dim of ``x`` (rst_linear_f32).  ``ln = (gamma, beta, eps)``: the LayerNorm in
    front of the linear; it is the prologue of the launch on the one/two-position GEMV route, part of the packing launch on the
    few-row route and a separate launch otherwise.  ``out_packed`` (only where ``linear_chains`` says so): returns a ``PackedRows`` --
    the operand of the next few-row linear in its packed order; ``x`` may be one."""
    for t, n in ((w, "w"), (bias, "bias"), (res, "res"), (scale, "scale")):
        _chk(t, n)
    N = w.shape[0]
    if isinstance(x, PackedRows) or out_packed:
        K = x.shape[-1]
        M = 1
        for d in x.shape[:-1]:
            M *= d
        assert w.shape[1] == K and M > 4 and _few_rows(M, N, K), "packed rows travel between few-row linears only"
        dev = x.xp.device if isinstance(x, PackedRows) else x.device
        if not isinstance(x, PackedRows):
            _chk(x, "x")
        if out_packed:
            assert res is None and N % 8 == 0
            out = PackedRows(torch.empty(32 if M <= 32 else (64 if M <= 64 else 128), N, device=dev, dtype=torch.float32), (*x.shape[:-1], N))
        else:
            out = torch.empty(*x.shape[:-1], N, device=dev, dtype=torch.float32)
        fold = ln is not None and SKINNY_F32_PACK_LN and K % 4 == 0
        xin = x
        if ln is not None and not fold:
            xin = layernorm(x, ln[0], ln[1], ln[2])
        prof = PROFILE
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _gemm_few_rows(xin, None, w, bias, res, scale, out.xp if out_packed else out, 1, M, M, K, K, N, 1, 0, 0, ACT_NONE, act_out,
                       ln=ln if fold else None, out_packed=out_packed)
        if prof is not None:
            e1.record()
            prof.append(("gemm_skinny_f32", e0, e1, 2.0 * M * N * K, 4 * (w.numel() + M * K + M * N), (M, N, K)))
        return out
    _chk(x, "x")
    K = x.shape[-1]
    assert w.shape[1] == K
    M = x.numel() // K if K else 0
    out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
    if 1 <= M <= 4 and K % 8 == 0 and K * M <= 32768 and act_out in (ACT_NONE, ACT_GELU):
        # a streaming step of one or two positions: weight-streaming GEMV (every CU pulls rows of w; no split-K hand-off)
        g, b, eps = ln if ln is not None else (None, None, 0.0)
        _chk(g, "ln gamma"); _chk(b, "ln beta")
        _lib.check(_lib.lib().rst_gemv_f32(_ptr(x), _ptr(g), _ptr(b), float(eps), _ptr(w), _ptr(bias), _ptr(res), _ptr(scale), _ptr(out),
                                           M, N, K, act_out, _stream()))
        return out
    fold_ln = ln is not None and SKINNY_F32_PACK_LN and _few_rows(M, N, K) and K % 4 == 0
    if ln is not None and not fold_ln:
        x = layernorm(x, ln[0], ln[1], ln[2])
    prof = PROFILE
    if _few_rows(M, N, K):
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _gemm_few_rows(x, None, w, bias, res, scale, out, 1, M, M, K, K, N, 1, 0, 0, ACT_NONE, act_out, ln=ln if fold_ln else None)
        if prof is not None:
            e1.record()
            prof.append(("gemm_skinny_f32", e0, e1, 2.0 * M * N * K, 4 * (w.numel() + x.numel() + out.numel()), (M, N, K)))
        return out
    split_k, ws, cnt = _gemm_split_scratch(x.device, M, N, K)
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    b3 = split_k <= 1 and _b3_route(x, None, w, 1, M, M, K, K, N, 1, 0, PAD_ZERO)
    if split_k > 1:
        _lib.check(_lib.lib().rst_gemm_win_f32(_ptr(x), None, _ptr(w), _ptr(bias), _ptr(res), _ptr(scale), _ptr(out), 1, M, M, K, K, N,
                                               1, 0, 0, M * K, N, 0, act_out, split_k, _ptr(ws), _ptr(cnt), _stream()))
    elif b3:
        _lib.check(_lib.lib().rst_gemm_win_b3_f32(_ptr(x), None, _ptr(w), _ptr(gemm_win_b3_pack_weight(w)), _ptr(bias), _ptr(res), _ptr(scale),
                                                  _ptr(out), 1, M, M, K, K, N, 1, 0, 0, M * K, N, 0, act_out, _stream()))
    else:
        _lib.check(_lib.lib().rst_linear_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(res), _ptr(scale), _ptr(out), M, K, N,
                                             act_out, _stream()))
    if prof is not None:
        e1.record()
        nbytes = 4 * (x.numel() + w.numel() + out.numel() + (res.numel() if res is not None else 0))
        prof.append(("gemm_win_b3" if b3 else "gemm_win", e0, e1, 2.0 * M * N * K, nbytes, (M, N, K)))
    return out


def resblock_supported(C: int, H: int, Kw: int, pre: bool = False, post: bool = False, K0: int = 0, Kf: int = 0) -> bool:
    return bool(_lib.lib().rst_seanet_resblock_supported(C, H, Kw, int(pre), int(post), K0, Kf))


def seanet_resblock(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, *, Kw: int,
                    hist: Optional[torch.Tensor] = None, pre: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                    post: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, elu_out: bool = False) -> torch.Tensor:
    """Fused SEANet residual block (rst_seanet_resblock_f32).  ``x [B,T,C]`` -> ``[B,T,C]``;
    ``pre=(w0 [C,K0], b0 [C])``: x is the mono audio ``[B,T,1]``; ``post=(wf [Kf,C], bf [1])``: returns ``[B,T,1]``."""
    C, H = w2.shape
    B, T = x.shape[0], x.shape[1]
    tensors = [(x, "x"), (w1, "w1"), (b1, "b1"), (w2, "w2"), (b2, "b2"), (hist, "hist")]
    w0 = b0 = wf = bf = None
    K0 = Kf = 0
    if pre is not None:
        w0, b0 = pre
        K0 = w0.shape[1]
        tensors += [(w0, "w0"), (b0, "b0")]
        assert x.shape[2] == 1
    else:
        assert x.shape[2] == C
    if post is not None:
        wf, bf = post
        Kf = wf.shape[0]
        tensors += [(wf, "wf"), (bf, "bf")]
    for t, n in tensors:
        _chk(t, n)
    out = torch.empty(B, T, 1 if post is not None else C, device=x.device, dtype=torch.float32)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    # whole-utterance launches (more than 4096 rows, no streaming history): the three-plane bf16 form, as the large GEMMs
    b3 = (GEMM_B3 and hist is None and B * T > 4096 and x.data_ptr() % 16 == 0
          and _resblock_b3_supported(B, T, C, H, Kw, pre is not None, post is not None, K0, Kf))
    if b3:
        wp = resblock_b3_pack_weights(w0, w1, w2, Kw)
        _lib.check(_lib.lib().rst_seanet_resblock_b3_f32(_ptr(x), _ptr(wp), _ptr(b0), _ptr(b1), _ptr(b2), _ptr(wf), _ptr(bf), _ptr(out),
                                                         B, T, C, H, Kw, K0, Kf, int(elu_out), _stream()))
    else:
        _lib.check(_lib.lib().rst_seanet_resblock_f32(_ptr(x), _ptr(hist), _ptr(w0), _ptr(b0), _ptr(w1), _ptr(b1), _ptr(w2),
                                                      _ptr(b2), _ptr(wf), _ptr(bf), _ptr(out), B, T, C, H, Kw, K0, Kf, int(elu_out),
                                                      _stream()))
    if prof is not None:
        e1.record()
        flops = 2.0 * B * T * (Kw * C * H + H * C) + (2.0 * B * T * C * K0) + (2.0 * B * T * C * Kf)
        prof.append(("resblock_b3" if b3 else "resblock", e0, e1, flops, 4 * (x.numel() + out.numel()), (B * T, C, Kw * C)))
    return out


@functools.lru_cache(maxsize=256)
def _resblock_b3_supported(B: int, T: int, C: int, H: int, Kw: int, pre: bool, post: bool, K0: int, Kf: int) -> bool:
    return bool(_lib.lib().rst_seanet_resblock_b3_supported(B, T, C, H, Kw, int(pre), int(post), K0, Kf))


_rb3_weights = _PackedWeights()


def resblock_b3_pack_weights(w0: Optional[torch.Tensor], w1: torch.Tensor, w2: torch.Tensor, Kw: int) -> torch.Tensor:
    """The block's matrices as three bf16 planes in matrix-instruction operand order (rst_seanet_resblock_b3_pack), cached per
    storage / version of ``w1`` (a block's three weights change together: they are re-packed from one state dict)."""
    C, H = w2.shape

    def build():
        n = int(_lib.lib().rst_seanet_resblock_b3_weight_elems(C))
        if n <= 0:
            raise ValueError(f"rstnet_amd.ops: no three-plane residual block for C = {C}")
        wp = torch.empty(n, device=w1.device, dtype=torch.int16)
        _lib.check(_lib.lib().rst_seanet_resblock_b3_pack(_ptr(w0), _ptr(w1), _ptr(w2), _ptr(wp), C, H, Kw, w0.shape[1] if w0 is not None else 0,
                                                          _stream()))
        return (wp, w2._version, None if w0 is None else w0._version)
    hit = _rb3_weights.get(w1, build)
    if hit[1] != w2._version or hit[2] != (None if w0 is None else w0._version):       # the companions changed in place: re-pack
        _rb3_weights._d.pop((w1.device, w1.data_ptr(), tuple(w1.shape), w1.dtype), None)
        hit = _rb3_weights.get(w1, build)
    return hit[0]


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> torch.Tensor:
    for t, n in ((x, "x"), (gamma, "gamma"), (beta, "beta")):
        _chk(t, n)
    D = x.shape[-1]
    out = torch.empty_like(x)
    _lib.check(_lib.lib().rst_layernorm_f32(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(out), x.numel() // D, D, eps, _stream()))
    return out


def rope_coef(max_period: float, D: int) -> float:
    """The fp32 scalar multiplying ``arange(D/2)`` in the reference (modules/rope.py:37-38)."""
    return float(torch.tensor(-math.log(max_period) * 2 / D, dtype=torch.float32))


def rope_split(qkv: torch.Tensor, H: int, *, q: Optional[torch.Tensor] = None, k: Optional[torch.Tensor] = None,
               v: Optional[torch.Tensor] = None, pos0: int = 0, pos_dev: Optional[torch.Tensor] = None, ring: bool = False,
               rope: bool = True, max_period: float = 10000.0):
    """qkv ``[B,T,3*H*D]`` -> q ``[B,H,T,D]`` and k/v written into ``[B,H,cap,D]`` buffers (allocated if None)."""
    _chk(qkv, "qkv")
    B, T, E3 = qkv.shape
    D = E3 // (3 * H)
    if q is None:
        q = torch.empty(B, H, T, D, device=qkv.device, dtype=torch.float32)
    if k is None:
        assert not ring
        k = torch.empty(B, H, T, D, device=qkv.device, dtype=torch.float32)
        v = torch.empty(B, H, T, D, device=qkv.device, dtype=torch.float32)
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk(t, n)
    if pos_dev is not None:
        _chk(pos_dev, "pos_dev", torch.int64)
    cap = k.shape[2]
    _lib.check(_lib.lib().rst_rope_split_f32(_ptr(qkv), _ptr(q), _ptr(k), _ptr(v), _ptr(pos_dev), pos0, B, T, H, D, cap,
                                             int(ring), int(rope), rope_coef(max_period, D), _stream()))
    return q, k, v


_attn_scratch: dict = {}


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, pos0: int = 0, pos_dev: Optional[torch.Tensor] = None,
              ring: bool = False, context: Optional[int] = None) -> torch.Tensor:
    """q ``[B,H,T,D]``, k/v ``[B,G,cap,D]`` (G = H, or fewer key/value heads on the few-query ring path) -> ``[B,T,H*D]``."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk(t, n)
    B, H, T, D = q.shape
    G, cap = k.shape[1], k.shape[2]
    out = torch.empty(B, T, H * D, device=q.device, dtype=torch.float32)
    if pos_dev is not None:
        _chk(pos_dev, "pos_dev", torch.int64)
    if G != H and not (ring and pos_dev is not None and D in (64, 128)):
        raise NotImplementedError("grouped key/value heads are served by the ring decode path only (head dim 64 / 128)")
    if ring and pos_dev is not None and (T <= 8 or G != H) and D in (64, 128):
        # streaming step with a handful of new queries: split every query over the occupied ring slots instead of walking
        # the ring tile by tile with one wave per head
        # (at most ~1024 workgroups: four of them fit a CU -- attn_decode_dense_kernel -- and a fifth would wait for a second round)
        splits = max(1, min(4, cap // 64, 1024 // max(1, B * T * H)))
        sc = _scratch(_attn_scratch, q.device, (B * T, H, splits, D),
                      lambda: (torch.empty(B * T, H, splits, D + 2, device=q.device, dtype=torch.float32),
                               torch.zeros(B * T, H, device=q.device, dtype=torch.int32)))
        _lib.check(_lib.lib().rst_attn_decode_multi_f32(_ptr(q), _ptr(k), _ptr(v), _ptr(sc[0]), _ptr(sc[1]), _ptr(out), _ptr(pos_dev),
                                                       B, T, H, D, cap, int(context) if context else 0, splits, G, _stream()))
        return out
    _lib.check(_lib.lib().rst_attention_f32(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(pos_dev), pos0, B, T, H, D, cap,
                                            int(ring), int(context) if context else 0, _stream()))
    return out


# A streaming step of a codec transformer layer's attention (a few new positions of many streams) as ONE launch: split + RoPE + ring
# append + the queries against the ring (rst_attention_step_f32).  False: rope_split + attention, two launches (the A/B switch).
ATTENTION_STEP = True
ATTENTION_STEP_PACKED = True       # its result straight in the out-projection's packed operand order on the few-row route (False: row-major)


def attention_step_supported(qkv: torch.Tensor, H: int, cap: int) -> bool:
    B, T, E3 = qkv.shape
    return bool(ATTENTION_STEP and qkv.is_cuda and E3 % (3 * H) == 0 and qkv.data_ptr() % 16 == 0 and
                _lib.lib().rst_attention_step_supported(T, E3 // (3 * H), cap))


def attention_step(qkv: torch.Tensor, H: int, k: torch.Tensor, v: torch.Tensor, pos_dev: torch.Tensor, *, context: Optional[int] = None,
                   rope: bool = True, max_period: float = 10000.0, out_packed: bool = False):
    """qkv ``[B,T,3*H*D]`` (the in-projection of the T new steps) -> ``[B,T,H*D]``; k / v ``[B,H,cap,D]`` rings, appended in place at
    slots ``(pos_dev + t) % cap``.  ``out_packed``: the result as a ``PackedRows`` -- the operand of the out-projection's few-row GEMM."""
    for t, n in ((qkv, "qkv"), (k, "k"), (v, "v")):
        _chk(t, n)
    _chk(pos_dev, "pos_dev", torch.int64)
    B, T, E3 = qkv.shape
    D = E3 // (3 * H)
    if k.shape != v.shape or k.shape[0] != B or k.shape[1] != H or k.shape[3] != D:
        raise ValueError(f"rstnet_amd.ops: rings {tuple(k.shape)} / {tuple(v.shape)} do not belong to qkv {tuple(qkv.shape)} with {H} heads")
    M = B * T
    rows = (32 if M <= 32 else (64 if M <= 64 else 128)) if out_packed else 0
    if out_packed and (M > SKINNY_F32_MAX_ROWS or (H * D) % 8):
        raise ValueError(f"rstnet_amd.ops: a packed attention result needs <= {SKINNY_F32_MAX_ROWS} rows and H * D % 8 == 0 (rows {M}, H * D {H * D})")
    out = torch.empty((rows, H * D) if out_packed else (B, T, H * D), device=qkv.device, dtype=torch.float32)
    _lib.check(_lib.lib().rst_attention_step_f32(_ptr(qkv), _ptr(k), _ptr(v), _ptr(out), _ptr(pos_dev), B, T, H, D, k.shape[2],
                                                 int(context) if context else 0, int(rope), rope_coef(max_period, D), rows, _stream()))
    return PackedRows(out, (B, T, H * D)) if out_packed else out


_rope_tables: dict = {}
# Whole-utterance passes of the codec transformers read q / k / v in place from the in-projection's output and rotate them on load
# (rst_attention_qkv_f32).  False: rope_split + attention as two launches (the A/B switch of tools/ab.py; the streaming steps always do).
ATTENTION_FUSED_QKV = True


def rope_table(T: int, D: int, max_period: float, device) -> torch.Tensor:
    """``[T, D]`` fp32: (cos, sin) of pair i of position t at ``[t, 2i]``, ``[t, 2i + 1]`` (rst_rope_table_f32; modules/rope.py:37-62).
    ONE table per (D, max_period, device), grown to the longest pass seen (rounded up to 256 positions) and read by shorter passes as a
    prefix -- variable-length utterances launch nothing for their rotation once the longest has passed.  A table is built on the stream
    that first needs it; readers on other streams wait for the build's event, and a table replaced by a longer one stays alive with the
    tensors that reference it."""
    device = torch.device(device)
    key = (D, float(max_period), device)
    hit = _rope_tables.get(key)
    if hit is None or hit[0].shape[0] < T:
        Tp = (T + 255) // 256 * 256
        tab = torch.empty(Tp, D, device=device, dtype=torch.float32)
        with torch.cuda.device(device):
            _lib.check(_lib.lib().rst_rope_table_f32(_ptr(tab), Tp, D, rope_coef(max_period, D), 0, _stream()))
            ev = torch.cuda.Event()
            ev.record()
            hit = _rope_tables[key] = (tab, ev, torch.cuda.current_stream())
    tab, ev, built_on = hit
    if torch.cuda.current_stream(device) != built_on and not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream(device).wait_event(ev)
    return tab[:T]


def attention_qkv(qkv: torch.Tensor, H: int, *, rope: bool = True, max_period: float = 10000.0, context: Optional[int] = None) -> torch.Tensor:
    """qkv ``[B, T, 3*H*D]`` (the in-projection's output, "b t (p h d)") -> ``[B, T, H*D]``: causal (+ ``context``) attention over
    positions 0 .. T-1 with interleaved RoPE applied as q / k are loaded -- the whole-utterance pass of
    ``StreamingMultiheadAttention`` (modules/transformer.py:376-423) without the split / rotate launch and its q, k, v copies."""
    _chk(qkv, "qkv")
    B, T, E3 = qkv.shape
    D = E3 // (3 * H)
    out = torch.empty(B, T, H * D, device=qkv.device, dtype=torch.float32)
    tab = rope_table(T, D, max_period, qkv.device) if rope else None
    _lib.check(_lib.lib().rst_attention_qkv_f32(_ptr(qkv), _ptr(tab), _ptr(out), B, T, H, D, int(context) if context else 0, _stream()))
    return out


def rvq_pack(emb: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """emb ``[L, n_codes, D]`` -> (packed ``[L, D/8, n_codes, 2, 4]``, e2 ``[L, n_codes]``)."""
    _chk(emb, "emb")
    L, n_codes, D = emb.shape
    packed = torch.empty(L, D // 8, n_codes, 2, 4, device=emb.device, dtype=torch.float32)
    e2 = torch.empty(L, n_codes, device=emb.device, dtype=torch.float32)
    for l in range(L):
        _lib.check(_lib.lib().rst_rvq_pack_f32(_ptr(emb[l]), _ptr(packed[l]), _ptr(e2[l]), n_codes, D, _stream()))
    return packed, e2


def _int_array(vals: Sequence[int]):
    return (C.c_int * len(vals))(*vals)


_rvq_keys: dict = {}


# Streaming RVQ (<= 64 frames per call): all residual levels in ONE launch with in-kernel hand-offs between the code slices'
# workgroups (rst_rvq_search_chain_f32).  False: one launch per level (the A/B switch of tools/ab.py).
RVQ_CHAIN = True
_rvq_slots: dict = {}


@functools.lru_cache(maxsize=64)
def _rvq_chain_supported(dev: int, M: int, n_codes: int, L: int, D: int, n_groups: int) -> bool:
    # the library's own answer (slices x groups x frame tiles within the CUs, LDS footprint): other shapes take the per-level launches
    with torch.cuda.device(dev):
        return bool(_lib.lib().rst_rvq_chain_supported(M, n_codes, L, D, n_groups))


def rvq_search(x: torch.Tensor, emb: torch.Tensor, packed: torch.Tensor, e2: torch.Tensor, B: int, F: int,
               groups: Sequence[Tuple[int, int]], return_dist: bool = False):
    """x ``[B*F, n_groups*D]`` projected latents -> codes ``[B, L, F]`` int64 (levels outside ``groups`` untouched)."""
    for t, n in ((x, "x"), (emb, "emb"), (packed, "packed"), (e2, "e2")):
        _chk(t, n)
    L, n_codes, D = emb.shape
    M = B * F
    assert x.shape == (M, len(groups) * D), (tuple(x.shape), M, len(groups), D)
    covered = sorted(l for g0, n in groups for l in range(g0, g0 + n)) == list(range(L))
    codes = (torch.empty if covered else torch.zeros)(B, L, F, device=x.device, dtype=torch.int64)
    dist = torch.zeros(L, M, device=x.device, dtype=torch.float32) if return_dist else None
    keys = None
    if 0 < M <= 64 and RVQ_CHAIN and n_codes % 128 == 0 and depth_frame_enabled(x.device) and \
            _rvq_chain_supported(_device_index(x.device), M, n_codes, L, D, len(groups)):
        # streaming step, ONE launch for all levels (+ its one-workgroup finish launch): the slices' workgroups hand every level's
        # decision over in-kernel (rst_rvq_search_chain_f32); a device that had to repair falls back to the launch-per-level form below
        n_slots = int(_lib.lib().rst_rvq_chain_slot_elems(M, n_codes, L))
        sc = _scratch(_rvq_slots, x.device, (L, M, n_codes),
                      lambda: (torch.full((n_slots,), -1, device=x.device, dtype=torch.int64), new_persistent_status(x.device)))
        _lib.check(_lib.lib().rst_rvq_search_chain_f32(_ptr(x), _ptr(emb), _ptr(packed), _ptr(e2), _ptr(codes), _ptr(dist), _ptr(sc[0]), _ptr(sc[1]),
                                                       M, max(F, 1), x.shape[1], D, n_codes, L, len(groups), _int_array([g[0] for g in groups]),
                                                       _int_array([g[1] for g in groups]), _stream()))
        return (codes, dist) if return_dist else codes
    if 0 < M <= 64:     # streaming step: the few-frame form (codes spread over workgroups)
        keys = _scratch(_rvq_keys, x.device, (L, M), lambda: torch.full((L, M), -1, device=x.device, dtype=torch.int64))
    _lib.check(_lib.lib().rst_rvq_search_f32(_ptr(x), _ptr(emb), _ptr(packed), _ptr(e2), _ptr(codes), _ptr(dist), _ptr(keys), M, max(F, 1),
                                             x.shape[1], D, n_codes, L, len(groups), _int_array([g[0] for g in groups]),
                                             _int_array([g[1] for g in groups]), _stream()))
    return (codes, dist) if return_dist else codes


def rvq_gather(codes: torch.Tensor, emb: torch.Tensor, groups: Sequence[Tuple[int, int]]) -> torch.Tensor:
    """codes ``[B, L, F]`` int64 -> ``[B*F, n_groups*D]`` sums of codebook rows per group."""
    _chk(codes, "codes", torch.int64)
    _chk(emb, "emb")
    B, L, F = codes.shape
    _, n_codes, D = emb.shape
    out = torch.empty(B * F, len(groups) * D, device=codes.device, dtype=torch.float32)
    _lib.check(_lib.lib().rst_rvq_gather_f32(_ptr(codes), _ptr(emb), _ptr(out), B * F, max(F, 1), D, n_codes, L, len(groups),
                                             _int_array([g[0] for g in groups]), _int_array([g[1] for g in groups]),
                                             _stream()))
    return out


def convtr_depthwise(x: torch.Tensor, w: torch.Tensor, stride: int, hist: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x ``[B,T,C]``, w ``[C,Kw]`` -> ``[B,T*stride,C]``."""
    for t, n in ((x, "x"), (w, "w"), (hist, "hist")):
        _chk(t, n)
    B, T, Cc = x.shape
    out = torch.empty(B, T * stride, Cc, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().rst_convtr_depthwise_f32(_ptr(x), _ptr(hist), _ptr(w), _ptr(out), B, T, Cc, w.shape[1], stride,
                                                   _stream()))
    return out


def activation(x: torch.Tensor, act: str) -> torch.Tensor:
    """Stand-alone ELU / GELU (only used when an activation module is called outside a fused container)."""
    _chk(x, "x")
    out = torch.empty_like(x)
    _lib.check(_lib.lib().rst_act_f32(_ptr(x), _ptr(out), x.numel(), {"elu": 1, "gelu": 2}[act], _stream()))
    return out


def transpose12(x: torch.Tensor) -> torch.Tensor:
    """``[B, R, C] -> [B, C, R]`` (layout adapter between the reference's [B,C,T] and channels-last)."""
    _chk(x, "x")
    B, R, Cc = x.shape
    if R == 1 or Cc == 1:
        return x.reshape(B, Cc, R)
    out = torch.empty(B, Cc, R, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().rst_transpose_f32(_ptr(x), _ptr(out), B, R, Cc, _stream()))
    return out


def mask_tail(x: torch.Tensor, lengths: torch.Tensor, replicate: bool = False) -> torch.Tensor:
    """In place: rows ``t >= lengths[b]`` of ``x [B,T,C]`` become zeros (or copies of the last valid row)."""
    _chk(x, "x")
    _chk(lengths, "lengths", torch.int32)
    B, T, Cc = x.shape
    _lib.check(_lib.lib().rst_mask_tail_f32(_ptr(x), _ptr(lengths), B, T, Cc, int(replicate), _stream()))
    return x


_hist_pending: Optional[list] = None      # deferred in-place rolls of the enclosing `hist_batch()` block
HIST_BATCH_MAX = 32


class hist_batch:
    """``with ops.hist_batch():`` -- the steady-state (in-place) history rolls requested inside the block are deferred and run as
    ONE launch at its end (rst_hist_update_batch_f32).  Valid because a layer's history is read only by that layer's own
    convolution, which has run by then; the block keeps the layer inputs alive until the roll."""

    def __enter__(self):
        global _hist_pending
        self._outer = _hist_pending
        _hist_pending = []
        return self

    def __exit__(self, *exc):
        global _hist_pending
        pending, _hist_pending = _hist_pending, self._outer
        if exc[0] is None:
            flush_hist_updates(pending)
        return False


def flush_hist_updates(pending: list) -> None:
    for i in range(0, len(pending), HIST_BATCH_MAX):
        part = pending[i:i + HIST_BATCH_MAX]
        n, B = len(part), part[0][0].shape[0]
        xs = (C.c_void_p * n)(*[x.data_ptr() for x, _ in part])
        hs = (C.c_void_p * n)(*[h.data_ptr() for _, h in part])
        ti, pp, cc = _int_array([x.shape[1] for x, _ in part]), _int_array([h.shape[1] for _, h in part]), _int_array([h.shape[2] for _, h in part])
        with torch.cuda.device(part[0][0].device):
            _lib.check(_lib.lib().rst_hist_update_batch_f32(xs, hs, ti, pp, cc, n, B, _stream()))


def hist_update(x: torch.Tensor, hist_in: Optional[torch.Tensor], P_out: int) -> torch.Tensor:
    """Last ``P_out`` steps of concat(hist_in, x) along time; x ``[B,T,C]``, hist ``[B,P,C]``.  In steady state (same history
    length, at most 16384 elements per stream) the roll happens IN PLACE and ``hist_in`` itself is returned: the state keeps
    its address, which is what lets a whole codec step be replayed as a HIP graph."""
    _chk(x, "x")
    _chk(hist_in, "hist_in")
    B, T, Cc = x.shape
    P_in = hist_in.shape[1] if hist_in is not None else 0
    if hist_in is not None and P_in == P_out and P_out * Cc <= 16384 and P_out > 0:
        if _hist_pending is not None and (not _hist_pending or _hist_pending[0][0].shape[0] == B):
            _hist_pending.append((x, hist_in))       # rolled by the enclosing hist_batch() block, in one launch with its peers
            return hist_in
        _lib.check(_lib.lib().rst_hist_update_f32(_ptr(x), _ptr(hist_in), _ptr(hist_in), B, T, P_in, P_out, Cc, _stream()))
        return hist_in
    out = torch.empty(B, P_out, Cc, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().rst_hist_update_f32(_ptr(x), _ptr(hist_in), _ptr(out), B, T, P_in, P_out, Cc, _stream()))
    return out


# ----------------------------------------------------------------------------------------------------------------------
# RQ-Transformer decode step (T = 1, small batch): bf16 weights, fp32 activations
# ----------------------------------------------------------------------------------------------------------------------
PROLOGUE_NONE, PROLOGUE_RMSNORM, PROLOGUE_SILU_GATE = 0, 1, 2


def gemv_bf16(x: torch.Tensor, w: torch.Tensor, *, prologue: int = PROLOGUE_NONE, alpha: Optional[torch.Tensor] = None,
              eps: float = 1e-8, res: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None, gate_out: bool = False) -> torch.Tensor:
    """``y[B,N] = (res +) (bias +) P(x) @ w.T`` with ``w`` bf16 ``[N,K]`` (rst_gemv_bf16_f32).  ``x`` is fp32 ``[B,K]``
    (``[B,2K]`` for the SiLU-gate prologue).  ``gate_out`` (``w = [W_u ; W_v]``): returns ``silu(u) * v`` of shape ``[B, N/2]``."""
    _chk(x, "x")
    _chk(w, "w", torch.bfloat16)
    _chk(alpha, "alpha")
    _chk(res, "res")
    _chk(bias, "bias")
    B = x.shape[0]
    N, K = w.shape
    assert x.shape[1] == (2 * K if prologue == PROLOGUE_SILU_GATE else K), (tuple(x.shape), N, K, prologue)
    No = N // 2 if gate_out else N
    if out is None:
        out = torch.empty(B, No, device=x.device, dtype=torch.float32)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.lib().rst_gemv_bf16_f32(_ptr(x), _ptr(alpha), _ptr(w), _ptr(res), _ptr(bias), _ptr(out), B, N, K, x.shape[1], No,
                                           prologue, eps, int(gate_out), _stream()))
    if prof is not None:
        e1.record()
        prof.append(("gemv_bf16", e0, e1, 2.0 * B * N * K, 2 * N * K + 4 * (x.numel() + out.numel()), (B, N, K)))
    return out


# Measured on MI355X (tools/bench_depth.py, graph-replayed chains at the depth transformer's shape): the out-projection with the
# attention as its prologue costs 8.8 us per launch against 4.0 us (plain out-projection) + 3.2 us (attn_small) for the two
# launches it replaces -- the prologue's dependent chain (qkv row -> 16 K/V rows -> scores -> softmax -> LDS) is longer than a
# kernel boundary.  The fused form stays available (and tested) but is not the default.
FUSE_SHORT_RING_ATTENTION = False


def gemv_attn_supported(B: int, H: int, D: int, cap: int, rope: bool, G: Optional[int] = None) -> bool:
    """Whether ``gemv_attn`` serves this attention AND is wanted: batch <= 2, a ring of <= 8 slots, no rotary embedding, one kv
    head per query head and a power-of-two head dim -- the depth transformer of both LM families."""
    return FUSE_SHORT_RING_ATTENTION and B <= 2 and 1 <= cap <= 8 and not rope and (G is None or G == H) and 4 <= D <= 256 and D & (D - 1) == 0 and B * H * D <= 32768


def gemv_attn(qkv: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, pos_dev: torch.Tensor, w: torch.Tensor, *,
              context: Optional[int] = None, res: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``res + W_out attention(qkv)`` in ONE launch (rst_gemv_attn_bf16_f32): ``qkv [B, 3*H*D]`` un-rotated, the ring
    ``[B, H, cap <= 8, D]`` (the new step is appended by the launch), ``w`` bf16 ``[N, H*D]``."""
    _chk(qkv, "qkv"); _chk(k_cache, "k_cache"); _chk(v_cache, "v_cache"); _chk(res, "res"); _chk(bias, "bias")
    _chk(pos_dev, "pos_dev", torch.int64)
    _chk(w, "w", torch.bfloat16)
    B, H, cap, D = k_cache.shape
    N = w.shape[0]
    assert qkv.shape == (B, 3 * H * D) and w.shape[1] == H * D, (tuple(qkv.shape), tuple(w.shape), H, D)
    out = torch.empty(B, N, device=qkv.device, dtype=torch.float32)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.lib().rst_gemv_attn_bf16_f32(_ptr(qkv), _ptr(k_cache), _ptr(v_cache), _ptr(pos_dev), _ptr(w), _ptr(res), _ptr(bias),
                                                _ptr(out), B, N, H, D, cap, int(context) if context else 0, qkv.shape[1], N, _stream()))
    if prof is not None:
        e1.record()
        prof.append(("gemv_bf16", e0, e1, 2.0 * B * N * H * D, 2 * N * H * D + 4 * (qkv.numel() + out.numel()), (B, N, H * D)))
    return out


def gemv_embed(add: torch.Tensor, table: torch.Tensor, tokens: torch.Tensor, col: int, w: torch.Tensor, *, alpha: torch.Tensor,
               eps: float = 1e-8, bias: Optional[torch.Tensor] = None):
    """First GEMV of a depth step (rst_gemv_embed_bf16_f32): ``x = add + table[tokens[:, col]]`` (``add`` fp32 ``[B, K]``, possibly a
    column block of a wider row-major buffer: unit column stride, any row stride), ``y = RMSNorm(x) @ w.T``.  Returns ``(y, x)``."""
    _chk(tokens, "tokens", torch.int64)
    _chk(table, "table", torch.bfloat16)
    _chk(w, "w", torch.bfloat16)
    _chk(alpha, "alpha"); _chk(bias, "bias")
    if not add.is_cuda or add.dtype != torch.float32 or add.dim() != 2 or add.stride(1) != 1:
        raise ValueError("rstnet_amd.ops: `add` must be a float32 CUDA/HIP matrix with unit column stride")
    B, K = add.shape
    N = w.shape[0]
    assert w.shape[1] == K and table.shape[1] == K and tokens.shape[0] == B and 0 <= col < tokens.shape[1]
    y = torch.empty(B, N, device=add.device, dtype=torch.float32)
    x = torch.empty(B, K, device=add.device, dtype=torch.float32)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.lib().rst_gemv_embed_bf16_f32(_ptr(add), _ptr(table), _ptr(tokens), _ptr(x), _ptr(alpha), _ptr(w), _ptr(bias), _ptr(y),
                                                 B, N, K, add.stride(0) if B > 1 else K, N, tokens.shape[1], col, table.shape[0], float(eps),
                                                 _stream()))
    if prof is not None:
        e1.record()
        prof.append(("gemv_bf16", e0, e1, 2.0 * B * N * K, 2 * N * K + 4 * (2 * B * K + y.numel()), (B, N, K)))
    return y, x


_skinny_weights = _PackedWeights()
_skinny_weights_gated = _PackedWeights()


def skinny_pack_weight(w: torch.Tensor, interleave_halves: bool = False) -> torch.Tensor:
    """bf16 ``[N, K]`` -> the MFMA-ordered copy ``[ceil(N/32)*32, K]`` of rst_skinny_pack_weight_bf16, cached per weight
    storage (the row-major original stays: the batch <= 2 GEMV streams that one).  ``interleave_halves``: the layout of gated
    layers (``w = [W_u ; W_v]``) whose GEMM applies ``silu(u) * v`` in its epilogue."""
    _chk(w, "w", torch.bfloat16)
    N, K = w.shape

    def build():
        wp = torch.empty((N + 31) // 32 * 32, K, device=w.device, dtype=torch.bfloat16)
        _lib.check(_lib.lib().rst_skinny_pack_weight_bf16(_ptr(w), _ptr(wp), N, K, int(interleave_halves), _stream()))
        return wp
    return (_skinny_weights_gated if interleave_halves else _skinny_weights).get(w, build)


def skinny_pack_act(x: torch.Tensor, *, prologue: int = PROLOGUE_NONE, alpha: Optional[torch.Tensor] = None,
                    eps: float = 1e-8) -> torch.Tensor:
    """fp32 ``[B, K]`` (``[B, 2K]`` for the SiLU gate) -> packed bf16 hi / lo planes ``[2, ceil(B/32)*32, K]`` of P(x)."""
    _chk(x, "x")
    _chk(alpha, "alpha")
    B = x.shape[0]
    K = x.shape[1] // 2 if prologue == PROLOGUE_SILU_GATE else x.shape[1]
    xp = torch.empty(2, (B + 31) // 32 * 32, K, device=x.device, dtype=torch.bfloat16)
    _lib.check(_lib.lib().rst_skinny_pack_act_f32(_ptr(x), _ptr(alpha), _ptr(xp), B, K, x.shape[1], prologue, eps, _stream()))
    return xp


class PackedAct(NamedTuple):
    """Activations already in the skinny GEMM's operand form (bf16 hi / lo planes ``[2, ceil(B/32)*32, K]``, rows past B zero),
    as emitted directly by a producer kernel (``lm_attn_decode(..., packed=True)``)."""
    xp: torch.Tensor
    B: int
    K: int


_packed_out: dict = {}


def _packed_buffer(device, B: int, K: int, role: str = "in") -> torch.Tensor:
    """Persistent, zero-initialised operand buffer per (stream, shape): producers write rows < B only, so the pad rows stay zero;
    layers of equal shape share it (launches on one stream are ordered).  ``role`` keeps the output of a GEMM that emits a packed
    operand apart from the operand it reads."""
    return _scratch(_packed_out, device, (B, K, role), lambda: torch.zeros(2, (B + 31) // 32 * 32, K, device=device, dtype=torch.bfloat16))


# K up to which a plain / RMSNorm linear of 2 < B <= 64 rows takes the fp32 activations directly (rst_gemm_skinny_x32_bf16_f32: the
# operand is formed inside the GEMM, no packing launch); above it the packing launch + packed GEMM (+ K split) stay
SKINNY_X32_MAX_K = 2048


def gemm_skinny(x, w: torch.Tensor, *, prologue: int = PROLOGUE_NONE, alpha: Optional[torch.Tensor] = None,
                eps: float = 1e-8, res: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None, gate_out: bool = False):
    """``y[B,N] = (res +) (bias +) P(x) @ w.T`` for 2 < B <= 64 on the bf16 matrix cores: prologue + hi/lo split + packing of
    the activations (one small launch; skipped when ``x`` is a ``PackedAct``), then rst_gemm_skinny_bf16_f32 against the packed
    copy of ``w``."""
    _chk(res, "res")
    _chk(bias, "bias")
    N, K = w.shape
    gate_out = gate_out and N % 32 == 0 and res is None
    wp = skinny_pack_weight(w, interleave_halves=gate_out)
    if not isinstance(x, PackedAct) and prologue in (PROLOGUE_NONE, PROLOGUE_RMSNORM) and K <= SKINNY_X32_MAX_K and K % 256 == 0 and x.dim() == 2 and \
            x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0:      # (K this short is never split: split plan >= 4096)
        # fp32 rows straight into the GEMM (one launch instead of two)
        _chk(alpha, "alpha")
        if not x.is_cuda or x.dtype != torch.float32:
            raise TypeError("rstnet_amd.ops: `x` must be a float32 CUDA/HIP matrix")
        assert x.shape[1] == K, (tuple(x.shape), N, K, prologue)
        B = x.shape[0]
        out = None if gate_out else torch.empty(B, N, device=x.device, dtype=torch.float32)
        gp = _packed_buffer(x.device, B, N // 2, "gate") if gate_out else None
        prof = PROFILE
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(_lib.lib().rst_gemm_skinny_x32_bf16_f32(_ptr(x), _ptr(alpha) if prologue == PROLOGUE_RMSNORM else None, float(eps),
                                                          1 if prologue == PROLOGUE_RMSNORM else 0, x.stride(0) if B > 1 else K, _ptr(wp), _ptr(res),
                                                          _ptr(bias), _ptr(out), B, N, K, N, _ptr(gp), _stream()))
        if prof is not None:
            e1.record()
            prof.append(("gemm_skinny", e0, e1, 2.0 * B * N * K, 2 * N * K + 4 * (x.numel() + B * N), (B, N, K)))
        return PackedAct(gp, B, N // 2) if gate_out else out
    if isinstance(x, PackedAct):
        assert prologue == PROLOGUE_NONE and x.K == K
        xp, B = x.xp, x.B
        x = xp
    else:
        B = x.shape[0]
        xp = skinny_pack_act(x, prologue=prologue, alpha=alpha, eps=eps)
    assert xp.shape[2] == K, (tuple(x.shape), N, K, prologue)
    # gate_out: w is a stacked [W_u ; W_v]; the epilogue emits silu(u) * v as the packed operand of the next GEMM
    out = None if gate_out else torch.empty(B, N, device=x.device, dtype=torch.float32)
    gp = _packed_buffer(x.device, B, N // 2, "gate") if gate_out else None
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    def build():         # split-K scratch of this shape: partial tiles + arrival counters (zero once, the kernel re-arms them)
        sk = int(_lib.lib().rst_skinny_bf16_split_plan(B, N, K))
        return (sk, torch.empty(sk, (B + 31) // 32 * 32, N, device=xp.device, dtype=torch.float32),
                torch.zeros((N + 31) // 32, device=xp.device, dtype=torch.int32)) if sk > 1 else (1, None, None)
    sc = _scratch(_gemm_scratch, xp.device, ("skinny_bf16", B, N, K), build)
    _lib.check(_lib.lib().rst_gemm_skinny_bf16_f32(_ptr(xp), _ptr(wp), _ptr(res), _ptr(bias), _ptr(out), B, N, K, N, _ptr(gp),
                                                  sc[0], _ptr(sc[1]), _ptr(sc[2]), _stream()))
    if prof is not None:
        e1.record()
        prof.append(("gemm_skinny", e0, e1, 2.0 * B * N * K, 2 * N * K + 4 * (x.numel() + B * N), (B, N, K)))
    return PackedAct(gp, B, N // 2) if gate_out else out


_skinny_weights_fp8 = _PackedWeights()


def skinny_pack_weight_fp8(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """bf16 ``[N, K]`` -> (fp8 e4m3 copy in MFMA order ``[ceil(N/32)*32, K]`` uint8, per-row scales fp32), cached per weight."""
    _chk(w, "w", torch.bfloat16)
    N, K = w.shape

    def build():
        n32 = (N + 31) // 32 * 32
        wp = torch.empty(n32, K, device=w.device, dtype=torch.uint8)
        sc = torch.empty(n32, device=w.device, dtype=torch.float32)
        _lib.check(_lib.lib().rst_skinny_pack_weight_fp8(_ptr(w), _ptr(wp), _ptr(sc), N, K, _stream()))
        return wp, sc
    return _skinny_weights_fp8.get(w, build)


def gemm_skinny_fp8(x: torch.Tensor, w: torch.Tensor, *, prologue: int = PROLOGUE_NONE, alpha: Optional[torch.Tensor] = None,
                    eps: float = 1e-8, res: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The fp8 (e4m3, per-row scales) form of ``gemm_skinny``: ``y = (res +) (bias +) Q(P(x)) @ Q(w).T`` for 1 <= B <= 64."""
    _chk(x, "x")
    _chk(alpha, "alpha")
    _chk(res, "res")
    _chk(bias, "bias")
    B = x.shape[0]
    N, K = w.shape
    assert x.shape[1] == (2 * K if prologue == PROLOGUE_SILU_GATE else K)
    wp, wsc = skinny_pack_weight_fp8(w)
    b32 = (B + 31) // 32 * 32
    xp = torch.empty(b32, K, device=x.device, dtype=torch.uint8)
    xsc = torch.empty(b32, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().rst_skinny_pack_act_fp8(_ptr(x), _ptr(alpha), _ptr(xp), _ptr(xsc), B, K, x.shape[1], prologue, eps, _stream()))
    out = torch.empty(B, N, device=x.device, dtype=torch.float32)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.lib().rst_gemm_skinny_fp8_f32(_ptr(xp), _ptr(xsc), _ptr(wp), _ptr(wsc), _ptr(res), _ptr(bias), _ptr(out), B, N, K, N,
                                                 _stream()))
    if prof is not None:
        e1.record()
        prof.append(("gemm_skinny_fp8", e0, e1, 2.0 * B * N * K, N * K + 4 * (x.numel() + out.numel()), (B, N, K)))
    return out


def lm_gated_pair(x: torch.Tensor, w_in: torch.Tensor, w_out: torch.Tensor, *, alpha: torch.Tensor, eps: float, res: torch.Tensor,
                  bias_in: Optional[torch.Tensor] = None, bias_out: Optional[torch.Tensor] = None, fp8: bool = False) -> torch.Tensor:
    """The gated MLP of a decode step: ``res + W_out (silu(u) * v)``, ``[u ; v] = W_in rmsnorm(x)`` (modules/gating.py:12-51,
    lit_model.py:399-403).  Batch <= 2: two GEMVs, the gate in the epilogue of the first (row pairs per wave).  Above: the first skinny GEMM applies
    the gate in its epilogue and hands the packed operand straight to the second -- the gated activation never exists in fp32."""
    B = x.shape[0]
    if not fp8 and B <= 2 and B * max(w_out.shape[1], w_in.shape[1]) <= 32768 and w_in.shape[0] % 2 == 0:
        # GEMV pair: every wave of the first owns a (u, v) row pair and writes silu(u) * v; the second is a plain GEMV
        g = gemv_bf16(x, w_in, prologue=PROLOGUE_RMSNORM, alpha=alpha, eps=eps, bias=bias_in, gate_out=True)
        return gemv_bf16(g, w_out, res=res, bias=bias_out)
    if fp8 or B <= 2 or B > 64 or w_in.shape[0] % 32:
        u = lm_linear(x, w_in, prologue=PROLOGUE_RMSNORM, alpha=alpha, eps=eps, bias=bias_in, fp8=fp8)
        return lm_linear(u, w_out, prologue=PROLOGUE_SILU_GATE, res=res, bias=bias_out, fp8=fp8)
    g = gemm_skinny(x, w_in, prologue=PROLOGUE_RMSNORM, alpha=alpha, eps=eps, bias=bias_in, gate_out=True)
    return gemm_skinny(g, w_out, res=res, bias=bias_out)


def lm_linear(x: torch.Tensor, w: torch.Tensor, *, prologue: int = PROLOGUE_NONE, alpha: Optional[torch.Tensor] = None,
              eps: float = 1e-8, res: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None, fp8: bool = False) -> torch.Tensor:
    """Batch-size dispatch of one decode-step linear: weight-streaming GEMV for B <= 2, bf16-MFMA skinny GEMM above (the
    prologue then runs inside the activation-packing launch).  ``fp8``: the opt-in e4m3 path (any batch <= 64)."""
    if isinstance(x, PackedAct):
        return gemm_skinny(x, w, prologue=prologue, res=res, bias=bias)
    if fp8 and x.shape[0] <= 64 and w.shape[1] % 32 == 0 and w.shape[1] <= 16384:
        return gemm_skinny_fp8(x, w, prologue=prologue, alpha=alpha, eps=eps, res=res, bias=bias)
    # the GEMV stages B x K fp32 activations in LDS: beyond two rows that footprint costs occupancy (fewer weight loads in
    # flight) and the matrix-core path is as fast or faster (measured: 4096 x 4096 at B = 3: 16.4 vs 16.5 us, B = 4: 20.9 vs 16.5)
    if x.shape[0] <= 2 and x.shape[0] * w.shape[1] <= 32768:
        return gemv_bf16(x, w, prologue=prologue, alpha=alpha, eps=eps, res=res, bias=bias)
    if x.shape[0] <= 64:
        return gemm_skinny(x, w, prologue=prologue, alpha=alpha, eps=eps, res=res, bias=bias)
    # more rows than one skinny tile set (prompt prefill): chunks of 64 rows, each streaming the weights once
    return torch.cat([gemm_skinny(x[i:i + 64], w, prologue=prologue, alpha=alpha, eps=eps,
                                  res=None if res is None else res[i:i + 64], bias=bias) for i in range(0, x.shape[0], 64)])


def embed_sum(tokens: torch.Tensor, tables: Sequence[torch.Tensor], tok_index: Sequence[int],
              add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out[b] = (add[b] +) sum_i tables[i][tokens[b, tok_index[i]]]``; tokens int64 ``[B, n]`` (rows contiguous, any row stride), tables
    bf16 ``[rows, D]``; ``add`` fp32 ``[B, D]`` with unit column stride and any row stride (a column block of a wider matrix is read in place)."""
    _chk(tokens, "tokens", torch.int64, contiguous=False)
    if add is not None:
        _chk(add, "add", contiguous=False)
        assert add.dim() == 2 and add.stride(1) == 1, "embed_sum: `add` needs unit column stride"
    assert tokens.dim() == 2 and tokens.stride(1) == 1, "embed_sum: `tokens` needs unit column stride"
    for t in tables:
        _chk(t, "table", torch.bfloat16)
    B, D = tokens.shape[0], tables[0].shape[1]
    out = torch.empty(B, D, device=tokens.device, dtype=torch.float32)
    tabs = (C.c_void_p * len(tables))(*[t.data_ptr() for t in tables])
    _lib.check(_lib.lib().rst_embed_sum_bf16(_ptr(tokens), tabs, _int_array(list(tok_index)), _int_array([t.shape[0] for t in tables]),
                                            len(tables), _ptr(add), _ptr(out), B, D, tokens.stride(0), add.stride(0) if add is not None else D,
                                            _stream()))
    return out


def rmsnorm(x: torch.Tensor, alpha: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    _chk(x, "x")
    _chk(alpha, "alpha")
    out = torch.empty_like(x)
    _lib.check(_lib.lib().rst_rmsnorm_f32(_ptr(x), _ptr(alpha), _ptr(out), x.numel() // x.shape[-1], x.shape[-1], eps, _stream()))
    return out


# Workgroups a (stream, head) pair's ring is split over in the single-step decode attention (rst_lm_attn_decode_f32): at most this many,
# at least 128 slots each, ~1024 workgroups per launch (the A/B switch of tools/ab.py).
LM_ATTN_MAX_SPLITS = 16


def lm_attn_splits(cap: int, pairs: int) -> int:
    return max(1, min(int(LM_ATTN_MAX_SPLITS), cap // 128, 1024 // max(1, pairs)))


def lm_rope_append(qkv: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, pos_dev: torch.Tensor, *, heads: int, rope: bool,
                   max_period: float = 10000.0, rope_dims: int = 0) -> torch.Tensor:
    """qkv ``[B, T, (H+2G)*D]`` (T new steps) -> rotated q ``[B, H, T, D]``; k/v appended to ring slots ``(pos+t) % cap`` of
    ``[B,G,cap,D]``.  ``rope_dims``: leading head dims that rotate (0 = all)."""
    for t, n in ((qkv, "qkv"), (k_cache, "k_cache"), (v_cache, "v_cache")):
        _chk(t, n)
    _chk(pos_dev, "pos_dev", torch.int64)
    B, G, cap, D = k_cache.shape
    T = qkv.shape[1]
    assert qkv.dim() == 3 and qkv.shape[2] == (heads + 2 * G) * D, (tuple(qkv.shape), heads, G, D)
    q = torch.empty(B, heads, T, D, device=qkv.device, dtype=torch.float32)
    _lib.check(_lib.lib().rst_lm_rope_append_f32(_ptr(qkv), _ptr(q), _ptr(k_cache), _ptr(v_cache), _ptr(pos_dev), B, T, heads, G, D, cap,
                                                qkv.shape[2], int(rope), rope_coef(max_period, rope_dims or D), rope_dims, _stream()))
    return q


def lm_rope_table(pos_dev: torch.Tensor, D: int, *, max_period: float = 10000.0, rope_dims: int = 0) -> torch.Tensor:
    """The rotation of ONE decode step as ``[D/2, 2]`` (cos, sin) pairs (rst_lm_rope_table_f32; identity beyond ``rope_dims / 2``): computed
    once per frame and handed to every layer's ``lm_attn_decode(rope_table=...)`` -- the values the attention launch would compute itself."""
    _chk(pos_dev, "pos_dev", torch.int64)
    out = torch.empty(D // 2, 2, device=pos_dev.device, dtype=torch.float32)
    _lib.check(_lib.lib().rst_lm_rope_table_f32(_ptr(pos_dev), _ptr(out), D, rope_dims, rope_coef(max_period, rope_dims or D), _stream()))
    return out


def lm_attn_decode(qkv: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, pos_dev: torch.Tensor, *, rope: bool,
                   context: Optional[int], max_period: float = 10000.0, splits: Optional[int] = None,
                   scratch: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, heads: Optional[int] = None,
                   rope_dims: int = 0, packed: bool = False, rope_table: Optional[torch.Tensor] = None):
    """Single-query attention of the new step given its qkv row ``[B, (H+2G)*D]`` (RoPE, ring append, attention and the
    reduction over slot splits in ONE launch) -> ``[B, H*D]``; the ring is ``[B,G,cap,D]`` (``heads`` = H when G < H).
    ``scratch = (ws [B,H,splits,D+2] fp32, counters [B,H] int32 zeros)`` may be passed to reuse buffers (the counters re-arm
    themselves).  ``rope_dims``: leading head dims that rotate (0 = all; the frequencies then span ``rope_dims``)."""
    _chk(qkv, "qkv")
    kv16 = k_cache.dtype == torch.bfloat16          # bf16 rings: the reference's cache precision (long-ring form)
    _chk(k_cache, "k_cache", k_cache.dtype if kv16 else torch.float32)
    _chk(v_cache, "v_cache", k_cache.dtype)
    _chk(pos_dev, "pos_dev", torch.int64)
    _chk(rope_table, "rope_table")
    B, G, cap, D = k_cache.shape
    H = heads or G
    assert qkv.shape[1] == (H + 2 * G) * D, (tuple(qkv.shape), H, G, D)
    assert rope_table is None or rope_table.numel() == D, "rope_table: [D/2, 2] of lm_rope_table"
    if kv16 and cap <= 64:
        raise NotImplementedError("bf16 KV rings are served by the long-ring attention (capacity > 64)")
    if splits is None:
        splits = 1 if cap <= 64 else lm_attn_splits(cap, B * H)
    ws = counters = None
    if splits > 1:
        if scratch is None:
            scratch = (torch.empty(B, H, splits, D + 2, device=qkv.device, dtype=torch.float32),
                       torch.zeros(B, H, device=qkv.device, dtype=torch.int32))
        ws, counters = scratch
        _chk(ws, "ws"); _chk(counters, "counters", torch.int32)
        assert ws.numel() >= B * H * splits * (D + 2) and counters.numel() >= B * H
    # packed=True (head dim 64 / 128): the result leaves as the bf16 hi / lo operand of the out-projection's skinny GEMM
    packed = packed and D in (64, 128)
    out = None if packed else torch.empty(B, H * D, device=qkv.device, dtype=torch.float32)
    xp = _packed_buffer(qkv.device, B, H * D) if packed else None
    _lib.check(_lib.lib().rst_lm_attn_decode_f32(_ptr(qkv), _ptr(k_cache), _ptr(v_cache), _ptr(ws), _ptr(counters), _ptr(out),
                                                _ptr(pos_dev), B, H, D, cap, int(context) if context else 0, splits, qkv.shape[1],
                                                int(rope), rope_coef(max_period, rope_dims or D), G, rope_dims, _ptr(xp), int(kv16),
                                                _ptr(rope_table) if rope else None, _stream()))
    return PackedAct(xp, B, H * D) if packed else out


_sample_ws: dict = {}


def lm_sample(logits: torch.Tensor, *, use_sampling: bool, temp: float, top_k: int, noise: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None, limit: int = 0, limit_dev: Optional[torch.Tensor] = None, top_p: float = 0.0,
              two_level: bool = True) -> torch.Tensor:
    """logits fp32 ``[B, V]`` -> tokens int64 ``[B]`` (greedy, or top-k sampling with Exp(1) ``noise [B, top_k]``).  ``limit``
    (or the int32 device scalar ``limit_dev``): ids >= limit are never drawn when sampling.  ``top_p > 0``: nucleus sampling
    (utils/sampling.py:66-82) instead of top-k; ``noise`` is then ``[B, V]`` (one draw per sorted position).  ``two_level=False``
    keeps the one-workgroup-per-row kernel for vocabularies above 32768 (A/B measurements, tests)."""
    _chk(logits, "logits")
    _chk(limit_dev, "limit_dev", torch.int32)
    B, V = logits.shape
    sampling = bool(use_sampling and temp > 0)
    nucleus = sampling and top_p > 0.0
    if noise is not None:       # a [B, >= top_k] column slice of a wider noise buffer is fine (row stride is passed on)
        if not noise.is_cuda or noise.dtype != torch.float32 or noise.dim() != 2 or noise.stride(1) != 1 or noise.shape[0] != B:
            raise ValueError("rstnet_amd.ops: `noise` must be a float32 CUDA/HIP tensor [B, k] with unit column stride")
        if nucleus and noise.shape[1] < V:
            raise ValueError(f"rstnet_amd.ops: top_p sampling draws one noise value per vocabulary entry: noise [B, >= {V}] needed")
    if out is None:
        out = torch.empty(B, device=logits.device, dtype=torch.int64)
    elif not out.is_cuda or out.dtype != torch.int64 or out.dim() != 1 or out.shape[0] != B:   # may be a column of a [B, n] buffer
        raise ValueError("rstnet_amd.ops: `out` must be an int64 CUDA/HIP vector of length B")
    ws, nbytes = None, 0
    if nucleus or (two_level and V > 32768):
        nbytes = int(_lib.lib().rst_lm_sample_workspace_bytes(B, V, top_k if sampling else 1, int(nucleus)))
        if nbytes:
            ws = _scratch(_sample_ws, logits.device, ("sample", nbytes), lambda: torch.empty((nbytes + 7) // 8, device=logits.device, dtype=torch.int64))
    _lib.check(_lib.lib().rst_lm_sample_f32(_ptr(logits), _ptr(noise), _ptr(out), B, V, V, top_k, noise.stride(0) if noise is not None else 0,
                                           out.stride(0) if B > 1 else 1, int(use_sampling), float(temp), int(limit), _ptr(limit_dev),
                                           float(top_p) if nucleus else 0.0, _ptr(ws), nbytes, _stream()))
    return out


def lm_ring_begin(cache: torch.Tensor, user_tokens: torch.Tensor, initial: torch.Tensor, delays: torch.Tensor,
                  offset_dev: torch.Tensor, first_user_k: int) -> torch.Tensor:
    """Start of an ``LMGen.step`` frame on the device (models/model.py:506-521): user streams into the token ring
    ``cache [B, K, CT]`` at their delayed columns, initial tokens while ``offset <= delay``; returns the model input ``[B, K]``."""
    for t, n in ((cache, "cache"), (user_tokens, "user_tokens"), (initial, "initial"), (offset_dev, "offset_dev")):
        _chk(t, n, torch.int64)
    _chk(delays, "delays", torch.int32)
    B, K, CT = cache.shape
    Ki = user_tokens.shape[1]
    assert user_tokens.shape[0] == B and initial.numel() == K and delays.numel() == K
    out = torch.empty(B, K, device=cache.device, dtype=torch.int64)
    _lib.check(_lib.lib().rst_lm_ring_begin_i64(_ptr(cache), _ptr(user_tokens), _ptr(initial), _ptr(delays), _ptr(offset_dev), _ptr(out),
                                               B, K, CT, Ki, first_user_k, _stream()))
    return out


def lm_ring_commit(cache: torch.Tensor, tokens: torch.Tensor, delays: torch.Tensor, offset_dev: torch.Tensor, max_delay: int) -> torch.Tensor:
    """End of the frame (models/model.py:545-562): ``offset_dev += 1``, generated ``tokens [B, n]`` into the ring, returns the
    delay-aligned gather ``[B, n]`` (meaningful once ``offset > max_delay``)."""
    for t, n in ((cache, "cache"), (tokens, "tokens"), (offset_dev, "offset_dev")):
        _chk(t, n, torch.int64)
    _chk(delays, "delays", torch.int32)
    B, K, CT = cache.shape
    n_out = tokens.shape[1]
    out = torch.empty(B, n_out, device=cache.device, dtype=torch.int64)
    _lib.check(_lib.lib().rst_lm_ring_commit_i64(_ptr(cache), _ptr(tokens), _ptr(delays), _ptr(offset_dev), _ptr(out), B, K, CT, n_out,
                                                int(max_delay), _stream()))
    return out




# ----------------------------------------------------------------------------------------------------------------------
# The depth phase of a frame as ONE persistent launch (rst_depth_decode_frame)
# ----------------------------------------------------------------------------------------------------------------------
DEPTH_FRAME_MAX_L, DEPTH_FRAME_MAX_Q = 8, 8
_depth_ws: dict = {}

# ---- health of the persistent launches (csrc/persist.h).  A launch whose workgroups are not all resident (device shared with other
# work) times out and is repaired in-stream by its one-workgroup twin: outputs stay right, but a repaired frame costs >= 0.1 s.  Every
# launcher's status words are registered here; `persistent_poll` reads their repair counters WITHOUT synchronising (async copy into
# pinned memory, evaluated on the next poll) and retires the persistent path on a device that had to repair -- sessions then
# re-capture their frame graphs on the launch-per-op chain (`persistent_epoch` changes).
PERSISTENT_MAX_REPAIRS = 0          # repairs tolerated per device before the launch-per-op chain takes over
_persist_status: dict = {}          # device -> list of weakrefs of status tensors (int32 [4])
_persist_off: dict = {}             # device -> reason string
_persist_epoch: dict = {}           # device -> int, bumped when the device's persistent path is retired
_persist_pending: dict = {}         # device -> (event, [pinned int32 [4] copies])
_persist_seen: dict = {}            # device -> repairs counted by the last completed poll


def new_persistent_status(device) -> torch.Tensor:
    """The 4 status words of a persistent launcher (time-out codes in flight | frames repaired | OR of repaired codes | reserved),
    registered for `persistent_poll`."""
    import weakref
    device = torch.device(device)
    st = torch.zeros(4, device=device, dtype=torch.int32)
    _persist_status.setdefault(device, []).append(weakref.ref(st))
    return st


def persistent_epoch(device) -> int:
    return _persist_epoch.get(torch.device(device), 0)


def persistent_repairs(device, synchronize: bool = True) -> int:
    """Frames / codec steps of `device` that the repair launches had to recompute so far.  ``synchronize=True`` reads the device words
    now (a device-to-host copy: waits for the stream); ``False`` returns the count seen by the last completed `persistent_poll` copy
    without touching the device (0 before the first one)."""
    device = torch.device(device)
    if not synchronize:
        pend = _persist_pending.get(device)
        if pend is not None and pend[0].query():
            _persist_seen[device] = int(sum(int(c[1]) for c in pend[1]))
        return _persist_seen.get(device, 0)
    alive = [r() for r in _persist_status.get(device, [])]
    n = int(sum(int(t[1].item()) for t in alive if t is not None))
    _persist_seen[device] = n
    return n


def persistent_rearm(device) -> None:
    """Gives `device` its persistent frame launches back after `persistent_poll` retired them (e.g. the process that shared the GPU
    has gone): clears the repair counters, bumps the epoch so that sessions re-capture their graphs.  The in-stream repair launch
    keeps every frame correct either way; this is about speed only.

    Call it with NO frame in flight on `device`: the status words are shared with captured graphs and per-session streams, and a
    repair launch's atomics racing with the reset would lose or mis-count a repair.  The whole device is therefore drained first
    (every stream, not only the current one), and the counters are zeroed on a quiet device."""
    device = torch.device(device)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    for r in _persist_status.get(device, []):
        t = r()
        if t is not None:
            t.zero_()
    _persist_pending.pop(device, None)
    _persist_seen.pop(device, None)
    if _persist_off.pop(device, None) is not None:
        _persist_epoch[device] = _persist_epoch.get(device, 0) + 1


def _retire_persistent(device, reason: str) -> None:
    import warnings
    if device not in _persist_off:
        _persist_off[device] = reason
        _persist_epoch[device] = _persist_epoch.get(device, 0) + 1
        warnings.warn(f"rstnet_amd: persistent frame launches retired on {device}: {reason}; the launch-per-op chain takes over "
                      "(outputs were repaired in-stream, nothing wrong left the device)", RuntimeWarning)


def persistent_poll(device, synchronize: bool = False) -> None:
    """Looks at the repair counters of `device` (the copy requested by the PREVIOUS call unless `synchronize`), retires the persistent
    path beyond PERSISTENT_MAX_REPAIRS repairs and requests the next copy.  Cheap enough for once every few dozen frames; never
    called inside a graph capture."""
    device = torch.device(device)
    if device.type != "cuda" or device in _persist_off or torch.cuda.is_current_stream_capturing():
        return
    if synchronize:
        n = persistent_repairs(device)
        if n > PERSISTENT_MAX_REPAIRS:
            _retire_persistent(device, f"{n} frame(s) needed the one-workgroup repair launch (hand-offs timed out: not all workgroups were resident)")
        return
    pend = _persist_pending.get(device)
    if pend is not None:
        ev, copies = pend
        if not ev.query():
            return
        n = int(sum(int(c[1]) for c in copies))
        _persist_seen[device] = n
        _persist_pending.pop(device, None)
        if n > PERSISTENT_MAX_REPAIRS:
            _retire_persistent(device, f"{n} frame(s) needed the one-workgroup repair launch (hand-offs timed out: not all workgroups were resident)")
            return
    refs = _persist_status.get(device, [])
    alive = [t for t in (r() for r in refs) if t is not None]
    _persist_status[device] = [r for r in refs if r() is not None]
    if not alive:
        return
    with torch.cuda.device(device):
        copies = [torch.empty(4, dtype=torch.int32).pin_memory() for _ in alive]
        for c, t in zip(copies, alive):
            c.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
    _persist_pending[device] = (ev, copies)


def depth_frame_enabled(device=None) -> bool:
    """RST_DEPTH_FRAME=0 keeps the launch-per-op depth phase / codec transformer layers (A/B measurements, or a device known to be
    shared with other work); a device whose persistent launches needed repairs is retired automatically (`persistent_poll`)."""
    import os
    if os.environ.get("RST_DEPTH_FRAME", "1") in ("0", ""):
        return False
    return device is None or torch.device(device) not in _persist_off


@functools.lru_cache(maxsize=256)
def _depth_frame_grid(dev: int, B: int, E: int, H: int, Hd: int, card: int, dep_q: int, L: int, top_k: int) -> int:
    # keyed by the device ordinal: the answer embeds that device's CU count and its occupancy query
    with torch.cuda.device(dev):
        return int(_lib.lib().rst_depth_frame_supported(B, E, H, Hd, card, dep_q, L, top_k))


def _device_index(device=None) -> int:
    """Ordinal of ``device`` (a torch device / int / None = the current device): the key of every per-device cache."""
    if device is None:
        return torch.cuda.current_device()
    d = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
    return d.index if d.index is not None else torch.cuda.current_device()


def depth_frame_supported(B: int, E: int, H: int, Hd: int, card: int, dep_q: int, L: int, top_k: int, device=None) -> bool:
    """Shapes ``rst_depth_decode_frame`` serves -- the library's own answer (rst_depth_frame_supported: batch 1 / 2, E and Hd multiples
    of 8, card <= 4096, at most 8 layers and 8 steps, the LDS footprint, a grid in which every workgroup owns rows of every
    all-to-all op and that still has a workgroup per head, and the occupancy query), so that callers can pick the per-op path
    instead of catching an error."""
    if not (1 <= B <= 2 and 1 <= dep_q <= DEPTH_FRAME_MAX_Q and 1 <= L <= DEPTH_FRAME_MAX_L and H >= 1 and E % max(H, 1) == 0):
        return False
    return _depth_frame_grid(_device_index(device), B, E, H, Hd, card, dep_q, L, top_k if 0 < top_k < card else card) > 0


def depth_decode_frame(tables, h_all: torch.Tensor, tokens: torch.Tensor, noise: Optional[torch.Tensor], *, use_sampling: bool,
                       temp: float, top_k: int, eps: float, context: Optional[int] = None, limits: Optional[torch.Tensor] = None,
                       ring_cap: Optional[int] = None) -> None:
    """``tables``: a ``lm.depth_frame.DepthFrameTables``; ``h_all`` fp32 ``[B, dep_q * E]``; ``tokens`` int64 ``[B, >= dep_q + 1]`` with the
    text token in column 0 -- columns 1 .. dep_q are written; ``noise`` fp32 ``[B, >= dep_q * top_k]`` (unit column stride);
    ``limits`` int32 ``[dep_q]`` on the device (optional id blanking per step); ``ring_cap``: capacity of the depth transformer's KV
    ring (default dep_q, the LMGen setup; the slot -> position map of RingKVCache.complete depends on it)."""
    _chk(h_all, "h_all")
    _chk(limits, "limits", torch.int32)
    if not tokens.is_cuda or tokens.dtype != torch.int64 or tokens.dim() != 2 or tokens.stride(1) != 1:
        raise ValueError("rstnet_amd.ops: `tokens` must be an int64 CUDA/HIP matrix with unit column stride")
    B = h_all.shape[0]
    t = tables
    assert h_all.shape[1] == t.dep_q * t.E and tokens.shape[0] == B and tokens.shape[1] >= t.dep_q + 1
    sampling = bool(use_sampling and temp > 0)
    if sampling:
        if noise is None or not noise.is_cuda or noise.dtype != torch.float32 or noise.dim() != 2 or noise.stride(1) != 1 or \
                noise.shape[0] != B or noise.shape[1] < t.dep_q * top_k:
            raise ValueError("rstnet_amd.ops: `noise` must be float32 CUDA/HIP [B, >= dep_q * top_k] with unit column stride")
    ws = _scratch(_depth_ws, h_all.device, (B, t.E, t.Hd, t.card),
                  lambda: torch.zeros(int(_lib.lib().rst_depth_frame_workspace_bytes(B, t.E, t.Hd, t.card)) // 8, device=h_all.device,
                                      dtype=torch.int64))
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.lib().rst_depth_decode_frame(
        t.in_proj, t.out_proj, t.norm1, t.norm2, t.gate_in, t.gate_out, t.heads, t.head_bias, t.emb, t.emb_rows,
        _ptr(h_all), _ptr(tokens), _ptr(noise) if sampling else None, _ptr(limits), _ptr(ws), _ptr(t.status),
        B, t.E, t.H, t.Hd, t.card, t.dep_q, t.L, h_all.stride(0) if B > 1 else h_all.shape[1], tokens.stride(0) if B > 1 else tokens.shape[1],
        noise.stride(0) if (sampling and B > 1) else (noise.shape[1] if sampling else 0), int(top_k), int(sampling), float(temp), float(eps),
        int(context) if context else 0, int(ring_cap) if ring_cap else t.dep_q, _stream()))
    if prof is not None:
        e1.record()
        n_w = t.dep_q * (t.L * (3 * t.E * t.E + t.E * t.E + 3 * t.Hd * t.E) + t.card * t.E)      # weight elements read once per frame
        prof.append(("depth_frame", e0, e1, 2.0 * B * n_w, 2 * n_w + 4 * h_all.numel(), (B, t.dep_q, t.L)))


# ----------------------------------------------------------------------------------------------------------------------
# The temporal transformer of a batch-1 LM step as ONE persistent launch (rst_temporal_decode_frame)
# ----------------------------------------------------------------------------------------------------------------------
# "auto": the persistent launch once the temporal rings hold at least TEMPORAL_FRAME_AUTO_POS steps.  Measured per frame (lm_b1, launch per
# op vs persistent, profiles/r06_temporal_persistent.txt): ring offset 0: 3.78 / 3.78 ms, 500: 3.90 / 3.95, 1000: 4.02 / 3.99, 2000: 4.08 /
# 4.08, 3000 (full): 4.20 / 4.14 -- the ring rows are part of its weight stream, so it gains where they are many.  True / False: always /
# never (tools/ab.py TEMPORAL_FRAME=True).
TEMPORAL_FRAME = "auto"
TEMPORAL_FRAME_AUTO_POS = 2048


def temporal_frame_wanted(host_pos: int) -> bool:
    """Whether a batch-1 temporal step at (host-side) position ``host_pos`` should take the persistent launch."""
    if TEMPORAL_FRAME == "auto":
        return host_pos >= TEMPORAL_FRAME_AUTO_POS
    return bool(TEMPORAL_FRAME)


TEMPORAL_FRAME_MAX_L = 40
_temporal_ws: dict = {}


@functools.lru_cache(maxsize=64)
def _temporal_frame_grid(dev: int, E: int, H: int, Hd: int, L: int, cap: int, kv16: int) -> int:
    with torch.cuda.device(dev):
        return int(_lib.lib().rst_temporal_frame_supported(E, H, Hd, L, cap, kv16))


def temporal_frame_supported(B: int, E: int, H: int, Hd: int, L: int, cap: int, kv_bf16: bool, device=None) -> bool:
    """Shapes ``rst_temporal_decode_frame`` serves -- the library's own answer (batch 1, E <= 4096, head dim 64 / 128, <= 40 layers, the
    LDS footprint, a workgroup per head, the occupancy query) -- on a device whose persistent launches have not been retired."""
    if not (TEMPORAL_FRAME and depth_frame_enabled(device) and B == 1 and 1 <= L <= TEMPORAL_FRAME_MAX_L and H >= 1 and E % max(H, 1) == 0):
        return False
    return _temporal_frame_grid(_device_index(device), E, H, Hd, L, cap, int(kv_bf16)) > 0


class TemporalFrameTables:
    """Host arrays of device pointers of ``rst_temporal_decode_frame`` for one streaming session: the layers' weights (bf16), norm gains
    (fp32) and KV rings.  Keeps the tensors alive (a captured graph embeds the pointers)."""

    def __init__(self, layers: Sequence[dict], *, H: int, context: Optional[int], eps: float):
        """``layers``: dicts of device tensors ``in_proj [3E, E], out_proj [E, E], gate_in [2 Hd, E], gate_out [E, Hd]`` (bf16),
        ``norm1, norm2 [E]`` (fp32), ``k_cache, v_cache [1, H, cap, D]`` (bf16 or fp32)."""
        L = len(layers)
        keep = []

        def column(name, dtype):
            out = []
            for ly in layers:
                t = ly[name]
                if not (t.is_cuda and t.is_contiguous() and t.dtype == dtype):
                    raise ValueError(f"rstnet_amd.ops: temporal-frame table {name!r} needs contiguous {dtype} device tensors")
                keep.append(t)
                out.append(t.data_ptr())
            return out
        E = layers[0]["out_proj"].shape[0]
        self.L, self.E, self.H, self.Hd = L, E, H, layers[0]["gate_out"].shape[1]
        kvd = layers[0]["k_cache"].dtype
        self.kv_bf16 = kvd == torch.bfloat16
        self.cap = layers[0]["k_cache"].shape[2]
        for ly in layers:
            assert tuple(ly["in_proj"].shape) == (3 * E, E) and tuple(ly["out_proj"].shape) == (E, E)
            assert tuple(ly["gate_in"].shape) == (2 * self.Hd, E) and tuple(ly["gate_out"].shape) == (E, self.Hd)
            assert tuple(ly["k_cache"].shape) == (1, H, self.cap, E // H) and ly["v_cache"].shape == ly["k_cache"].shape
        rows = [column("in_proj", torch.bfloat16), column("out_proj", torch.bfloat16), column("gate_in", torch.bfloat16),
                column("gate_out", torch.bfloat16), column("norm1", torch.float32), column("norm2", torch.float32), column("k_cache", kvd),
                column("v_cache", kvd)]
        # the [8][L] pointer table lives in device memory (addresses fit int64: the top bit of a device pointer is clear)
        self.dev_tables = torch.tensor(rows, dtype=torch.int64, device=layers[0]["in_proj"].device)
        self.context, self.eps = context, float(eps)
        self.status = new_persistent_status(layers[0]["in_proj"].device)
        self._keep = keep


def temporal_decode_frame(tables: "TemporalFrameTables", x: torch.Tensor, pos_dev: torch.Tensor, rope_table: Optional[torch.Tensor]) -> torch.Tensor:
    """x fp32 ``[1, E]`` -> ``[1, E]`` through every layer of ``tables`` (one new step at position ``*pos_dev``, appended to the rings);
    ``rope_table``: fp32 ``[D/2, 2]`` of `lm_rope_table` or None (no rotation)."""
    _chk(x, "x")
    _chk(pos_dev, "pos_dev", torch.int64)
    _chk(rope_table, "rope_table")
    t = tables
    assert x.shape == (1, t.E)
    y = torch.empty_like(x)
    ws = _scratch(_temporal_ws, x.device, (t.E, t.Hd, t.H),
                  lambda: torch.zeros(int(_lib.lib().rst_temporal_frame_workspace_bytes(t.E, t.Hd, t.H)) // 8, device=x.device, dtype=torch.int64))
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.lib().rst_temporal_decode_frame(
        _ptr(t.dev_tables), _ptr(x), _ptr(y), _ptr(pos_dev), _ptr(rope_table), _ptr(ws), _ptr(t.status), t.E, t.H, t.Hd, t.L, t.cap,
        int(t.context) if t.context else 0, int(t.kv_bf16), t.eps, _stream()))
    if prof is not None:
        e1.record()
        n_w = t.L * (4 * t.E * t.E + 3 * t.Hd * t.E)
        prof.append(("temporal_frame", e0, e1, 2.0 * n_w, 2 * n_w, (1, t.L, t.E)))
    return y


# ----------------------------------------------------------------------------------------------------------------------
# One streaming step of a Mimi transformer as ONE persistent launch (rst_codec_transformer_frame)
# ----------------------------------------------------------------------------------------------------------------------
_ctr_ws: dict = {}
_ctr_status: dict = {}


@functools.lru_cache(maxsize=256)
def _codec_tr_grid(dev: int, B: int, T: int, E: int, H: int, F: int, L: int, cap: int) -> int:
    with torch.cuda.device(dev):
        return int(_lib.lib().rst_codec_transformer_supported(B, T, E, H, F, L, cap))


def codec_transformer_frame_supported(B: int, T: int, E: int, H: int, F: int, L: int, cap: int, device=None) -> bool:
    """Shapes the persistent launch serves (else the launch-per-op layer loop runs) -- the library's own answer
    (rst_codec_transformer_supported: at most 4 rows and 4 new positions, <= 8 layers, head dim a multiple of 16 that divides 1024, a
    grid with rows for every workgroup and a workgroup per (stream, head), the occupancy query)."""
    if not (depth_frame_enabled(device) and B >= 1 and 1 <= T <= 4 and B * T <= 4 and H >= 1 and E % max(H, 1) == 0):
        return False
    return _codec_tr_grid(_device_index(device), B, T, E, H, F, L, cap) > 0


def codec_transformer_status(device) -> torch.Tensor:
    """The 4 device status words of the persistent codec-transformer launches of ``device`` (persist.h: [0] time-out codes of the step in
    flight, [1] steps repaired by the one-workgroup launch, [2] OR of the repaired codes)."""
    device = torch.device(device)
    st = _ctr_status.get(device)
    if st is None:
        st = _ctr_status[device] = new_persistent_status(device)
    return st


def codec_transformer_frame(x: torch.Tensor, layers: Sequence[dict], pos_dev: torch.Tensor, *, H: int, context: Optional[int], rope: bool,
                            max_period: float, eps: float) -> torch.Tensor:
    """x fp32 ``[B, T, E]`` (the new positions of every stream) -> ``[B, T, E]`` through all ``layers``: each a dict of fp32 device
    tensors ``in_proj [3E, E], out_proj [E, E], linear1 [F, E], linear2 [E, F], norm1_w/b, norm2_w/b [E], scale1/scale2 [E] or None,
    k_cache / v_cache [B, H, cap, D]`` (rings: the new steps are appended).  ``pos_dev``: int64 device scalar, position of x[:, 0]."""
    _chk(x, "x")
    _chk(pos_dev, "pos_dev", torch.int64)
    B, T, E = x.shape
    L = len(layers)
    F_ = layers[0]["linear1"].shape[0]
    cap = layers[0]["k_cache"].shape[2]
    for ly in layers:
        for name in ("in_proj", "out_proj", "linear1", "linear2", "norm1_w", "norm1_b", "norm2_w", "norm2_b", "k_cache", "v_cache"):
            _chk(ly[name], name)
        _chk(ly.get("scale1"), "scale1"); _chk(ly.get("scale2"), "scale2")

    def table(name):
        return (C.c_void_p * L)(*[_ptr(ly.get(name)) for ly in layers])
    has_scale = layers[0].get("scale1") is not None
    y = torch.empty_like(x)
    ws = _scratch(_ctr_ws, x.device, (B * T, E, F_),
                  lambda: torch.zeros(int(_lib.lib().rst_codec_transformer_workspace_bytes(B * T, E, F_)) // 8, device=x.device, dtype=torch.int64))
    D = E // H
    _lib.check(_lib.lib().rst_codec_transformer_frame(
        table("in_proj"), table("out_proj"), table("linear1"), table("linear2"), table("norm1_w"), table("norm1_b"), table("norm2_w"),
        table("norm2_b"), table("scale1") if has_scale else None, table("scale2") if has_scale else None, table("k_cache"), table("v_cache"),
        _ptr(x), _ptr(y), _ptr(pos_dev), _ptr(ws), _ptr(codec_transformer_status(x.device)), B, T, E, H, F_, L, cap,
        int(context) if context else 0, int(rope), rope_coef(max_period, D), float(eps), _stream()))
    return y


# every public entry point runs under the device guard of its first tensor argument
for _name in ("skinny_f32_pack_weight", "gemm_win", "linear", "seanet_resblock", "layernorm", "rope_split", "attention", "rvq_pack",
              "rvq_search", "rvq_gather", "convtr_depthwise", "activation", "transpose12", "mask_tail", "hist_update", "gemv_bf16",
              "gemv_attn", "gemv_embed", "skinny_pack_weight", "skinny_pack_act", "gemm_skinny", "skinny_pack_weight_fp8", "gemm_skinny_fp8", "lm_gated_pair",
              "lm_linear", "embed_sum", "rmsnorm", "lm_rope_append", "lm_rope_table", "lm_attn_decode", "lm_sample", "lm_ring_begin", "lm_ring_commit",
              "depth_decode_frame", "temporal_decode_frame", "codec_transformer_frame"):
    globals()[_name] = _on_tensor_device(globals()[_name])
del _name

