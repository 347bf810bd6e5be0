"""us per few-row linear (M rows against [N, K] fp32 weights), a graph of 32 back-to-back calls replayed: the packed route (packing launch +
GEMM) against the in-place route (rst_linear_few_rows_f32), with and without the LayerNorm in front.

    python tools/probes/few_row_linear_probe.py [--rows 64]

Then the two launches of the packed route on their own, and the sweep behind rst_skinny_f32_split_plan (rows x shapes x splits).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from rstnet_amd import ops  # noqa: E402


def graph_time(fn, reps=32, iters=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            y = fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    del y
    return e0.elapsed_time(e1) / iters / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=64)
    a = ap.parse_args()
    dev = "cuda:0"
    M = a.rows
    print(f"{M} rows; us per linear (32 calls per graph)")
    print(f"  {'N x K':14s} {'LayerNorm':10s} {'packed route':>14s} {'in place':>10s}")
    for N, K in ((1536, 512), (512, 512), (2048, 512), (512, 2048)):
        x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        ln = (torch.ones(K, device=dev), torch.zeros(K, device=dev), 1e-5)
        for use_ln in (False, True):
            t = {}
            for rows in (False, True):
                ops.SKINNY_F32_ROWS = rows
                t[rows] = graph_time(lambda: ops.linear(x, w, ln=ln if use_ln else None))
            # (with a LayerNorm the packing launch stays whatever the switch says: both columns time the same launches)
            print(f"  {f'{N} x {K}':14s} {str(use_ln):10s} {t[False]:14.2f} " + (f"{t[True]:10.2f}" if not use_ln else f"{'(packed)':>10s}"))


def parts(M):
    """the two launches of the packed route on their own"""
    dev = "cuda:0"
    print(f"{M} rows; us per launch: GEMM on an already packed operand | LayerNorm + pack alone | plain pack alone")
    for N, K in ((1536, 512), (512, 512), (2048, 512), (512, 2048)):
        x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        w0 = torch.randn(K, K, device=dev)
        ln = (torch.ones(K, device=dev), torch.zeros(K, device=dev), 1e-5)
        ops.SKINNY_F32_ROWS = False
        h = ops.linear(x, w0, out_packed=True)          # [M, K] in packed order
        t_gemm = graph_time(lambda: ops.linear(h, w))
        xp = torch.empty(64 if M <= 64 else 128, K, device=dev)
        from rstnet_amd import _lib
        L = _lib.lib()
        st = torch.cuda.current_stream
        t_ln = graph_time(lambda: (_lib.check(L.rst_skinny_f32_pack_ln(x.data_ptr(), ln[0].data_ptr(), ln[1].data_ptr(), 1e-5, xp.data_ptr(), M, K, st().cuda_stream)), xp)[1])
        t_pk = graph_time(lambda: (_lib.check(L.rst_skinny_f32_pack_win(x.data_ptr(), None, xp.data_ptr(), 1, M, M, K, K, 1, 0, 0, M * K, 0, st().cuda_stream)), xp)[1])
        print(f"  {f'{N} x {K}':14s} {t_gemm:8.2f} {t_ln:8.2f} {t_pk:8.2f}")


def grid():
    """the sweep behind rst_skinny_f32_split_plan: 16 - 128 rows x 11 shapes x 6 splits"""
    dev = "cuda:0"
    from rstnet_amd import _lib
    for M in (16, 32, 64, 128):
        print(f"{M} rows; pack + GEMM us by K split across workgroups (* = rst_skinny_f32_split_plan)")
        for N, K in ((512, 512), (1536, 512), (2048, 512), (512, 2048), (1024, 8192), (512, 3072), (1024, 512), (1024, 1536), (4096, 2048), (256, 512), (512, 1024)):
            x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
            plan = int(_lib.lib().rst_skinny_f32_split_plan(M, N, K))
            row = []
            for sk in (1, 2, 4, 8, 16, 32):
                ops.SKINNY_F32_SPLIT = sk
                ops.SKINNY_F32_ROWS = False
                row.append(f"{sk}{'*' if sk == plan else ''}: {graph_time(lambda: ops.linear(x, w)):6.2f}")
            ops.SKINNY_F32_SPLIT = None
            print(f"  {f'{N} x {K}':14s} " + "   ".join(row))


if __name__ == "__main__":
    main()
    parts(64)
    grid()
