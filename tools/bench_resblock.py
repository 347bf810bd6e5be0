"""Times the fused residual-block kernels alone at the shapes of the headline step (B = 64 x 10 s): C = 64 with the first convolution,
C = 64 with the last one, C = 128 plain -- three-plane bf16 form and (--f32) the f32-instruction kernels.

    python tools/bench_resblock.py [--iters 10] [--f32] [--only pre|post|c128|plain]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rstnet_amd import ops, synth                     # noqa: E402
from rstnet_amd.codec import functional as RF         # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--f32", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--lib", default="", help="another build of the library (A/B measurements)")
    a = ap.parse_args()
    if a.lib:
        from rstnet_amd import _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
    ops.GEMM_B3 = not a.f32
    g = torch.Generator().manual_seed(0)
    B = a.batch
    cases = [("pre", 64, 240000), ("post", 64, 240000), ("plain", 64, 240000), ("c128", 128, 60000)]
    for name, C, T in cases:
        if a.only and a.only != name:
            continue
        w1 = RF.pack_conv_weight(synth._xavier(g, C // 2, C, 3)).to(DEV)
        w2 = RF.pack_conv_weight(synth._xavier(g, C, C // 2, 1)).to(DEV)
        b1, b2 = (0.1 * torch.randn(C // 2, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
        kw = {}
        if name == "pre":
            x = (torch.rand(B, T, 1, generator=g) * 2 - 1).to(DEV)
            kw["pre"] = (synth._xavier(g, C, 1, 7)[:, 0].contiguous().to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV))
            kw["elu_out"] = True
        else:
            x = torch.empty(B, T, C, device=DEV).uniform_(-2, 2)
            if name == "post":
                kw["post"] = (synth._xavier(g, 1, C, 3)[0].t().contiguous().to(DEV), torch.zeros(1, device=DEV))
        run = lambda: ops.seanet_resblock(x, w1, b1, w2, b2, Kw=3, **kw)      # noqa: E731
        y = run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            y = None                  # (the caching allocator then hands the same block out again: no 4 GB allocation inside the timed loop)
            y = run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        flops = 2.0 * B * T * (3 * C * (C // 2) + (C // 2) * C) + (2.0 * B * T * C * 7 if name == "pre" else 0)
        nbytes = 4.0 * (x.numel() + y.numel())
        print(f"{name:6s} C={C:3d} rows={B * T:9d}: {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s (fp32-equivalent)  {nbytes / ms / 1e6:7.1f} GB/s (x in + y out)"
              f"  [{'f32 instruction' if a.f32 else 'three-plane bf16'}]")
        del x, y


if __name__ == "__main__":
    main()
