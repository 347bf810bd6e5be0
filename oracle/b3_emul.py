"""TEST INFRASTRUCTURE ONLY (imported by tests/ alone): a CPU restatement of the arithmetic of the three-plane bf16 GEMM
(rstnet_amd/csrc/gemm_win.hip, gemm_win_b3_stream_kernel) -- not of anything in the reference, whose convolutions are plain fp32
(`modules/conv.py:178-329` -> `F.conv1d`); it documents why that kernel may stand in for them.

    x = hi + mid + lo   with  hi = bf16_rne(x),  mid = bf16_rne(x - hi),  lo = bf16_rne(x - hi - mid)        (exact in fp32)
    x * w ~= lo*hi' + hi*lo' + mid*mid' + mid*hi' + hi*mid' + hi*hi'                                        (six of nine products)

Every kept product is exact in fp32 (8 x 8 significand bits); the three dropped ones are bounded by 2^-23 |x||w|."""
from typing import Tuple

import torch


def split3(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """fp32 -> (hi, mid, lo), each a bf16-representable fp32 tensor; round to nearest even at every level."""
    x = x.float()
    hi = x.bfloat16().float()
    r = x - hi
    mid = r.bfloat16().float()
    lo = (r - mid).bfloat16().float()
    return hi, mid, lo


def matmul6(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """``x [M,K] @ w[N,K].T`` through the six kept plane products, accumulated in float64 (isolates the error of the scheme from the
    accumulation order of any particular kernel)."""
    xh, xm, xl = (t.double() for t in split3(x))
    wh, wm, wl = (t.double() for t in split3(w))
    return xl @ wh.T + xh @ wl.T + xm @ wm.T + xm @ wh.T + xh @ wm.T + xh @ wh.T
