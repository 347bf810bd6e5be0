"""ctypes binding of ``librstnet_hip.so`` (see ``include/rstnet_hip.h``).

There is deliberately NO fallback: if the shared library is missing or an entry point is absent the
import of any op raises.  The product path never computes on the CPU and never touches ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librstnet_hip.so")

_p = C.c_void_p
_i = C.c_int
_l = C.c_int64
_f = C.c_float

# name -> argtypes (all functions return int); mirrors include/rstnet_hip.h one to one
SIGNATURES = {
    "rst_build_id": [_p, _i],
    "rst_gemm_win_split_plan": [_l, _i, _i],
    "rst_gemm_win_split_tiles": [_l, _i],
    "rst_gemm_win_f32": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _l, _i, _i, _i, _i, _p, _p, _p],
    "rst_gemm_win_b3_supported": [_i, _i, _i, _i, _i, _i, _i, _i, _i, _l, _i],
    "rst_gemm_win_b3_weight_elems": [_i, _i],
    "rst_gemm_win_b3_pack_weight": [_p, _p, _i, _i, _p],
    "rst_gemm_win_b3_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _l, _i, _i, _i, _p],
    "rst_conv1d_causal_f32": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "rst_convtr1d_causal_f32": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "rst_seanet_resblock_supported": [_i, _i, _i, _i, _i, _i, _i],
    "rst_seanet_resblock_f32": [_p] * 11 + [_i] * 8 + [_p],
    "rst_seanet_resblock_b3_supported": [_i] * 9,
    "rst_seanet_resblock_b3_weight_elems": [_i],
    "rst_seanet_resblock_b3_pack": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "rst_seanet_resblock_b3_f32": [_p] * 8 + [_i] * 8 + [_p],
    "rst_linear_f32": [_p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _p],
    "rst_layernorm_f32": [_p, _p, _p, _p, _l, _i, _f, _p],
    "rst_rope_split_f32": [_p, _p, _p, _p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _f, _p],
    "rst_attention_f32": [_p, _p, _p, _p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _p],
    "rst_rope_table_f32": [_p, _i, _i, _f, _l, _p],
    "rst_attention_qkv_f32": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "rst_rvq_pack_f32": [_p, _p, _p, _i, _i, _p],
    "rst_rvq_search_f32": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), _p],
    "rst_rvq_chain_slot_elems": [_i, _i, _i],
    "rst_rvq_chain_supported": [_i, _i, _i, _i, _i],
    "rst_rvq_search_chain_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), _p],
    "rst_rvq_gather_f32": [_p, _p, _p, _i, _i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), _p],
    "rst_convtr_depthwise_f32": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "rst_act_f32": [_p, _p, _l, _i, _p],
    "rst_transpose_f32": [_p, _p, _i, _i, _i, _p],
    "rst_hist_update_f32": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "rst_hist_update_batch_f32": [C.POINTER(_p), C.POINTER(_p), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), _i, _i, _p],
    "rst_skinny_f32_pack_weight": [_p, _p, _i, _i, _p],
    "rst_skinny_f32_pack_win": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _l, _i, _p],
    "rst_skinny_f32_pack_ln": [_p, _p, _p, _f, _p, _i, _i, _p],
    "rst_skinny_f32_split_plan": [_i, _i, _i],
    "rst_gemm_skinny_f32": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p],
    "rst_linear_few_rows_f32": [_p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p],
    "rst_mask_tail_f32": [_p, _p, _i, _i, _i, _i, _p],
    "rst_codec_transformer_workspace_bytes": [_i, _i, _i],
    "rst_codec_transformer_supported": [_i, _i, _i, _i, _i, _i, _i],
    "rst_codec_transformer_frame": [C.POINTER(_p)] * 12 + [_p, _p, _p, _p, _p] + [_i] * 9 + [_f, _f, _p],
    "rst_gemv_f32": [_p, _p, _p, _f, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "rst_gemv_bf16_f32": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p],
    "rst_gemv_attn_bf16_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "rst_gemv_embed_bf16_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p],
    "rst_depth_frame_workspace_bytes": [_i, _i, _i, _i],
    "rst_depth_frame_supported": [_i, _i, _i, _i, _i, _i, _i, _i],
    "rst_depth_decode_frame": [C.POINTER(_p)] * 9 + [C.POINTER(_i), _p, _p, _p, _p, _p, _p] + [_i] * 12 + [_f, _f, _i, _i, _p],
    "rst_temporal_frame_workspace_bytes": [_i, _i, _i],
    "rst_temporal_frame_supported": [_i, _i, _i, _i, _i, _i],
    "rst_temporal_decode_frame": [_p] * 7 + [_i] * 7 + [_f, _p],
    "rst_skinny_pack_weight_bf16": [_p, _p, _i, _i, _i, _p],
    "rst_skinny_pack_act_f32": [_p, _p, _p, _i, _i, _i, _i, _f, _p],
    "rst_skinny_pack_weight_fp8": [_p, _p, _p, _i, _i, _p],
    "rst_skinny_pack_act_fp8": [_p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "rst_gemm_skinny_fp8_f32": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "rst_gemm_skinny_bf16_f32": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p, _p, _p],
    "rst_gemm_skinny_x32_bf16_f32": [_p, _p, _f, _i, _i, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    "rst_skinny_bf16_split_plan": [_i, _i, _i],
    "rst_embed_sum_bf16": [_p, C.POINTER(_p), C.POINTER(_i), C.POINTER(_i), _i, _p, _p, _i, _i, _i, _i, _p],
    "rst_rmsnorm_f32": [_p, _p, _p, _l, _i, _f, _p],
    "rst_lm_rope_append_f32": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _p],
    "rst_lm_rope_table_f32": [_p, _p, _i, _i, _f, _p],
    "rst_lm_attn_decode_f32": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _p, _i, _p, _p],
    "rst_attn_decode_multi_f32": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "rst_attention_step_supported": [_i, _i, _i],
    "rst_attention_step_f32": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i, _p],
    "rst_lm_sample_workspace_bytes": [_i, _i, _i, _i],
    "rst_lm_sample_f32": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i, _p, _f, _p, _l, _p],
    "rst_lm_ring_begin_i64": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "rst_lm_ring_commit_i64": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
}

_lib: Optional[C.CDLL] = None


class RstError(RuntimeError):
    """An entry point of librstnet_hip.so returned a non-zero status."""


def lib() -> C.CDLL:
    """Loads the library once; raises (never falls back) if it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C rstnet_amd/csrc`).  rstnet_amd has no CPU / PyTorch fallback.")
    # torch first: it ships its own libamdhip64 and owns the device memory / streams this library is handed.  Loading the
    # system HIP runtime before it would put two runtimes in the process (the second one finds "no ROCm-capable device").
    import torch  # noqa: F401
    handle = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = _i
    handle.rst_version.restype = _i
    handle.rst_version.argtypes = []
    handle.rst_last_error.restype = C.c_char_p
    handle.rst_last_error.argtypes = []
    _lib = handle
    return handle


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().rst_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"librstnet_hip: {msg}")
        raise RstError(f"librstnet_hip error {rc}: {msg}")


def build_id() -> str:
    """The loaded library's build id (csrc/Makefile: hash of its sources)."""
    buf = C.create_string_buffer(64)
    n = lib().rst_build_id(buf, 64)
    return buf.value.decode() if n > 0 else "unknown"
