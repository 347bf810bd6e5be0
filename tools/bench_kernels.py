"""Micro-benchmarks of individual C-ABI kernels (HIP-event timing); also the workload for rocprofv3 --pmc passes.

    python tools/bench_kernels.py [case ...]        cases: res64 res64pre res64post res128 gemm:<M>x<N>x<K>[:elu]
                                                           conv:<B>x<T>x<Cin>x<N>x<Kw>x<S>[:elu][:res] ...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rstnet_amd import ops  # noqa: E402
from rstnet_amd.codec import functional as RF  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def timeit_graph(fn, iters=50, warm=3):
    """Device time per call with the host launch cost taken out: `iters` calls captured in one HIP graph, replayed."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def res_case(C, B, T, pre=False, post=False):
    g = torch.Generator().manual_seed(0)
    H = C // 2
    w1 = (torch.randn(H, 3 * C, generator=g) * 0.05).to(DEV)
    w2 = (torch.randn(C, H, generator=g) * 0.05).to(DEV)
    b1, b2 = torch.zeros(H, device=DEV), torch.zeros(C, device=DEV)
    x = torch.randn(B, T, 1 if pre else C, generator=g).to(DEV)
    kw = {}
    if pre:
        kw["pre"] = ((torch.randn(C, 7, generator=g) * 0.3).to(DEV), torch.zeros(C, device=DEV))
    if post:
        kw["post"] = ((torch.randn(3, C, generator=g) * 0.1).to(DEV), torch.zeros(1, device=DEV))
    ms = timeit(lambda: ops.seanet_resblock(x, w1, b1, w2, b2, Kw=3, **kw))
    fl = 2.0 * B * T * (3 * C * H + H * C)
    return ms, fl


def gemm_case(M, N, K, elu=False, res=False):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(DEV)
    r = torch.randn(M, N, generator=g).to(DEV) if res else None
    ms = timeit(lambda: ops.gemm_win(x, w, B=1, T_in=M, T_out=M, C_=K, S=1, P=0, N=N, res=r, act_in=int(elu)))
    return ms, 2.0 * M * N * K


def conv_case(B, T, Cin, N, Kw, S, elu=False, res=False):
    """gemm_win in its window form: conv Cin -> N, kernel Kw, stride S over [B, T, Cin] (a transposed conv k = q*S of Cout
    channels is conv:B x T x Cin x (S*Cout) x q x 1).  Causal left padding Kw - S, zeros."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, Cin, generator=g).to(DEV)
    w = (torch.randn(N, Kw * Cin, generator=g) * 0.05).to(DEV)
    T_out = T // S
    r = torch.randn(B, T_out, N, generator=g).to(DEV) if res else None
    bias = torch.zeros(N, device=DEV)
    out = torch.empty(B, T_out, N, device=DEV)
    ms = timeit(lambda: ops.gemm_win(x, w, B=B, T_in=T, T_out=T_out, C_=Cin, S=S, P=Kw - S, N=N, bias=bias, res=r,
                                     act_in=int(elu), out=out), iters=8, warm=2)
    return ms, 2.0 * B * T_out * N * Kw * Cin


def sample_case(V, k, sampling=True):
    g = torch.Generator().manual_seed(0)
    logits = (torch.randn(1, V, generator=g) * 3).to(DEV)
    noise = torch.empty(1, k).exponential_(1, generator=g).to(DEV)
    out = torch.empty(1, dtype=torch.long, device=DEV)
    ms = timeit_graph(lambda: ops.lm_sample(logits, use_sampling=sampling, temp=0.8, top_k=k, noise=noise, out=out))
    return ms, 1.0


def skinny_case(B, N, K, gate=False):
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV)
    x = torch.randn(B, 2 * K if gate else K, generator=g).to(DEV)
    r = torch.randn(B, N, generator=g).to(DEV)
    fn = (lambda: ops.gemv_bf16(x, w, res=r, prologue=2 if gate else 0)) if B <= 4 else \
         (lambda: ops.gemm_skinny(x, w, res=r, prologue=2 if gate else 0))
    ms = timeit_graph(fn)
    return ms, 2.0 * N * K


def gemv_case(B, N, K, mode="plain", copies=None):
    """Batch <= 2 GEMV over ROTATING weight copies (> 1 GB in total, so no launch finds its weights in the 256 MB MALL)."""
    g = torch.Generator().manual_seed(0)
    copies = copies or max(2, int(1.5e9 // (2 * N * K)) + 1)
    ws = [(torch.randn(N, K, device=DEV) * 0.05).bfloat16()]
    ws += [ws[0].clone() for _ in range(copies - 1)]
    x = torch.randn(B, 2 * K if mode == "gate" else K, generator=g).to(DEV)
    r = torch.randn(B, N, generator=g).to(DEV)
    alpha = torch.ones(K, device=DEV)
    state = {"i": 0}

    def fn():
        w = ws[state["i"] % copies]
        state["i"] += 1
        if mode == "norm":
            ops.gemv_bf16(x, w, prologue=ops.PROLOGUE_RMSNORM, alpha=alpha, eps=1e-8)
        elif mode == "normgate":
            ops.gemv_bf16(x, w, prologue=ops.PROLOGUE_RMSNORM, alpha=alpha, eps=1e-8, gate_out=True)
        elif mode == "gate":
            ops.gemv_bf16(x, w, res=r, prologue=ops.PROLOGUE_SILU_GATE)
        else:
            ops.gemv_bf16(x, w, res=r)
    return fn, max(copies, 48), 2.0 * N * K


def gemv_cases(cases):
    """python tools/bench_kernels.py gemv:BxNxK[:norm|normgate|gate] ...   -- graph-timed, weights rotating through > 1 GB."""
    for c in cases:
        parts = c.split(":")
        Bq, N, K = [int(v) for v in parts[1].split("x")]
        mode = parts[2] if len(parts) > 2 else "plain"
        fn, iters, nbytes = gemv_case(Bq, N, K, mode)
        ms = timeit_graph(fn, iters=iters)
        print(f"{c:28s} {ms * 1e3:6.1f} us {nbytes / ms / 1e9:5.2f} TB/s", flush=True)


def main():
    if len(sys.argv) > 1 and all(a.startswith("gemv:") for a in sys.argv[1:]):
        return gemv_cases(sys.argv[1:])
    cases = sys.argv[1:] or ["res64", "res64pre", "res64post", "res128", "gemm:3840000x128x512:elu", "gemm:16000x1024x8192",
                             "gemm:128000x512x3072", "gemm:16000x512x512"]
    B, T = 16, 240000
    for c in cases:
        if c.startswith("sample"):
            parts = c.split(":")
            ms, fl = sample_case(int(parts[1]), int(parts[2]), sampling="greedy" not in parts)
            print(f"{c:32s} {ms * 1e3:8.1f} us", flush=True)
            continue
        if c.startswith("skinny"):
            parts = c.split(":")
            Bq, N, K = [int(v) for v in parts[1].split("x")]
            ms, nbytes = skinny_case(Bq, N, K, gate="gate" in parts[2:])
            print(f"{c:32s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:8.1f} GB/s", flush=True)
            continue
        if c.startswith("conv:"):
            parts = c.split(":")
            ms, fl = conv_case(*[int(v) for v in parts[1].split("x")], elu="elu" in parts[2:], res="res" in parts[2:])
            print(f"{c:40s} {ms:8.3f} ms  {fl / ms / 1e9:7.2f} TFLOP/s", flush=True)
            continue
        if c.startswith("res"):
            C = 128 if "128" in c else 64
            ms, fl = res_case(C, B, T // (4 if C == 128 else 1), pre="pre" in c, post="post" in c)
        else:
            parts = c.split(":")
            M, N, K = [int(v) for v in parts[1].split("x")]
            ms, fl = gemm_case(M, N, K, elu="elu" in parts[2:], res="res" in parts[2:])
        print(f"{c:32s} {ms:8.3f} ms  {fl / ms / 1e9:7.2f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
