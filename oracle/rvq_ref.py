"""ctypes front of oracle/rvq_ref.c -- TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

from . import build as _build

_lib = None


def _load():
    global _lib
    if _lib is None:
        path = _build.LIB if os.path.exists(_build.LIB) else _build.build()
        _lib = C.CDLL(path)
        _lib.rvq_search_ref.restype = C.c_int
        _lib.rvq_search_ref.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return _lib


def rvq_search(x: np.ndarray, emb: np.ndarray):
    """x [M, D] fp32, emb [L, n_codes, D] fp32 -> (codes [L, M] int64, score [L, M] fp32) with the fixed fp32
    evaluation order documented in rvq_ref.c."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    emb = np.ascontiguousarray(emb, dtype=np.float32)
    M, D = x.shape
    L, n_codes, D2 = emb.shape
    assert D == D2
    codes = np.zeros((L, M), dtype=np.int64)
    dist = np.zeros((L, M), dtype=np.float32)
    rc = _load().rvq_search_ref(x.ctypes.data, emb.ctypes.data, M, D, n_codes, L, codes.ctypes.data, dist.ctypes.data)
    assert rc == 0
    return codes, dist
