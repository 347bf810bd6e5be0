"""gemm_win_b3 (fp32 activations, split inside the launch) vs gemm_p3 (activations pre-split into planes, both operands by LDS-DMA) at
the GEMM shapes of the headline step.  Prints ms and TFLOP/s (fp32-equivalent) per shape and route."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rstnet_amd import ops                     # noqa: E402

DEV = "cuda:0"
# (name, B, T_in, C, S, P, K, N): A(b, t, k) = x_b[(t * S - P) * C + k]
SHAPES = [
    ("enc conv k8s4 64->128", 64, 240000, 64, 4, 4, 512, 128),
    ("enc conv k10s5 128->256", 64, 60000, 128, 5, 5, 1280, 256),
    ("enc conv k12s6 256->512", 64, 12000, 256, 6, 6, 3072, 512),
    ("enc conv k16s8 512->1024", 64, 2000, 512, 8, 8, 8192, 1024),
    ("rb256 conv3", 64, 12000, 256, 1, 2, 768, 128),
    ("rb256 conv1", 64, 12000, 128, 1, 0, 128, 256),
    ("tr in_proj", 1, 16000, 512, 1, 0, 512, 1536),
    ("tr out_proj", 1, 16000, 512, 1, 0, 512, 512),
    ("tr linear1", 1, 16000, 512, 1, 0, 512, 2048),
    ("tr linear2", 1, 16000, 2048, 1, 0, 2048, 512),
    ("dec convtr 1024->512 k16s8", 64, 250, 1024, 1, 1, 2048, 4096),
    ("dec convtr 256->128 k10s5", 64, 12000, 256, 1, 1, 512, 640),
    ("dec convtr 128->64 k8s4", 64, 60000, 128, 1, 1, 256, 256),
]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    tot = {"b3": 0.0, "p3": 0.0, "p3->planes": 0.0}
    for name, B, T, C, S, P, K, N in SHAPES:
        if a.only and a.only not in name:
            continue
        T_out = -(-T // S)
        x = torch.empty(B, T, C, device=DEV).uniform_(-1, 1)
        w = torch.empty(N, K, device=DEV).uniform_(-1, 1) / K ** 0.5
        fl = 2.0 * B * T_out * N * K
        ms_b3 = timeit(lambda: ops.gemm_win(x, w, B=B, T_in=T, T_out=T_out, C_=C, S=S, P=P, N=N), a.iters)
        xp = ops.p3_split(x)
        ms_p3 = timeit(lambda: ops.gemm_p3(xp, w, T_out=T_out, S=S, P=P, N=N), a.iters)
        ms_pp = timeit(lambda: ops.gemm_p3(xp, w, T_out=T_out, S=S, P=P, N=N, out_f32=False, out_p3=True), a.iters)
        ms_split = timeit(lambda: ops.p3_split(x), a.iters)
        y3 = ops.gemm_win(x, w, B=B, T_in=T, T_out=T_out, C_=C, S=S, P=P, N=N)
        yp, _ = ops.gemm_p3(xp, w, T_out=T_out, S=S, P=P, N=N)
        diff = float((y3.view(-1) - yp.view(-1)).abs().max() / y3.abs().max())
        tot["b3"] += ms_b3; tot["p3"] += ms_p3; tot["p3->planes"] += ms_pp
        print(f"{name:28s} M={B * T_out:8d} N={N:5d} K={K:5d}  b3 {ms_b3:7.3f} ms {fl / ms_b3 / 1e9:6.1f} TF/s | p3 {ms_p3:7.3f} ms {fl / ms_p3 / 1e9:6.1f} TF/s"
              f" | p3 -> planes {ms_pp:7.3f} ms {fl / ms_pp / 1e9:6.1f} TF/s | split alone {ms_split:6.3f} ms | max rel diff {diff:.1e}")
        del x, xp, y3, yp
    print("total ms:", {k: round(v, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
