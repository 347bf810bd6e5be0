"""Numerics of the three-plane bf16 GEMM against the f32 matrix instruction (fp64 reference): backward error in units of 2^-24 and the
signed bias on all-positive operands (a truncating accumulator shows up as a negative bias that grows with K)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rstnet_amd import ops  # noqa: E402

DEV = "cuda:0"
U = 2.0 ** -24
for K in (96, 512, 2048, 8192):
    for positive in (False, True):
        g = torch.Generator().manual_seed(K)
        M, N = 8192, 256
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        if positive:
            x, w = x.abs(), w.abs()
        ref = F.linear(x.double(), w.double())
        bound = F.linear(x.double().abs(), w.double().abs())
        out = {}
        for name, flag in (("b3", True), ("f32", False)):
            ops.GEMM_B3 = flag
            y = ops.linear(x.to(DEV), w.to(DEV)).double().cpu()
            d = y - ref
            out[name] = (float((d.abs() / bound).max()) / U, float((d / bound).mean()) / U, float(d.abs().max() / ref.abs().max()))
        print(f"K={K:5d} positive={positive!s:5s} " + "  ".join(f"{n}: max {v[0]:7.2f} U  mean signed {v[1]:+7.3f} U  rel-to-max {v[2]:.2e}" for n, v in out.items()), flush=True)
