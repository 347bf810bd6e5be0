"""CPU checks of the C-ABI boundary: the shared library loads and exports every symbol include/rstnet_hip.h
declares (no kernels are launched), and the ctypes table mirrors the header."""
import ctypes
import os
import re

import pytest

from rstnet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "rstnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rst_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return ctypes.CDLL(_lib.LIB_PATH)


def test_header_declares_functions():
    names = header_functions()
    assert "rst_gemm_win_f32" in names and "rst_rvq_search_f32" in names and len(names) >= 14


def test_library_exports_every_declared_symbol(built):
    for name in header_functions():
        assert hasattr(built, name), f"{name} declared in include/rstnet_hip.h but not exported"


def test_ctypes_table_matches_header(built):
    declared = set(header_functions()) - {"rst_version", "rst_last_error"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    text = open(os.path.join(ROOT, "include", "rstnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, argtypes in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\((.*?)\)\s*;" % name, text, flags=re.S)
        n_args = len([a for a in m.group(1).split(",") if a.strip()])
        assert n_args == len(argtypes), (name, n_args, len(argtypes))


def test_version_and_error_string(built):
    built.rst_version.restype = ctypes.c_int
    built.rst_last_error.restype = ctypes.c_char_p
    assert built.rst_version() >= 100
    assert isinstance(built.rst_last_error(), bytes)


def test_no_cpu_fallback():
    """The product path must refuse CPU tensors instead of silently computing elsewhere."""
    import torch
    from rstnet_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm(torch.zeros(2, 8), torch.ones(8), torch.zeros(8), 1e-5)
