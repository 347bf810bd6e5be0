"""A/B of the temporal transformer of a batch-1 LM step: ONE persistent launch (csrc/lm_temporal.hip, ops.TEMPORAL_FRAME) against the
five launches per layer, same weights, same rings -- outputs compared step by step (short context, across the split thresholds, across
the ring wrap), then both timed as captured graphs.

    python tools/probes/temporal_frame_probe.py [--layers 32] [--kv f32] [--steps 140] [--time 60]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from rstnet_amd import ops  # noqa: E402
from rstnet_amd.lm.model import StreamingTransformer  # noqa: E402


def run_steps(tr, xs, persistent, pos0=0, rings=None):
    ops.TEMPORAL_FRAME = persistent
    outs = []
    with tr.streaming(1):
        st = tr._streaming_state
        if rings is not None:
            for l in range(len(st.k)):
                st.k[l].copy_(rings[0][l]); st.v[l].copy_(rings[1][l])
        st.pos.fill_(pos0)
        for x in xs:
            outs.append(tr.step(x).clone())
        torch.cuda.synchronize()
        status = st.tables.status.tolist() if st.tables is not None else None
    return outs, status


def compare(a, b, what):
    worst = 0.0
    for i, (u, v) in enumerate(zip(a, b)):
        err = ((u - v).abs().max() / v.abs().max().clamp_min(1e-20)).item()
        if not (err == err):
            print(f"  {what}: step {i}: NaN"); worst = float("nan"); break
        worst = max(worst, err)
    print(f"  {what}: {len(a)} steps, max |persistent - per-op| / max |per-op| = {worst:.3e}")
    return worst


def time_graph(tr, x, persistent, pos0, iters):
    ops.TEMPORAL_FRAME = persistent
    with tr.streaming(1):
        st = tr._streaming_state
        st.pos.fill_(pos0)
        for _ in range(3):
            tr.step(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = tr.step(x)
        st.pos.fill_(pos0)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        st.pos.fill_(pos0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        status = st.tables.status.tolist() if st.tables is not None else None
        del y
    return e0.elapsed_time(e1) / iters, status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--kv", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--steps", type=int, default=140)
    ap.add_argument("--time", type=int, default=60)
    ap.add_argument("--dim", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--cap", type=int, default=3000)
    a = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    kvd = torch.bfloat16 if a.kv == "bf16" else torch.float32
    ff = int(4.125 * a.dim)
    tr = StreamingTransformer(a.dim, a.heads, a.layers, ff, context=a.cap, positional_embedding="rope", device=dev, dtype=torch.bfloat16,
                              kv_dtype=kvd)
    print(f"layers {a.layers} dim {a.dim} heads {a.heads} hidden {tr.layers[0].gating.linear_out.weight.shape[1]} cap {a.cap} kv {a.kv}; "
          f"persistent supported: {ops.temporal_frame_supported(1, a.dim, a.heads, tr.layers[0].gating.linear_out.weight.shape[1], a.layers, a.cap, kvd == torch.bfloat16, dev)}")
    g = torch.Generator(device=dev).manual_seed(1)
    xs = [torch.randn(1, a.dim, device=dev, generator=g) for _ in range(a.steps)]
    t0 = time.time()
    ref, _ = run_steps(tr, xs, False)
    got, status = run_steps(tr, xs, True)
    print(f"short context (positions 0..{a.steps - 1}), status {status}, {time.time() - t0:.1f} s")
    bad = compare(got, ref, "from an empty ring")
    # across the wrap of a full ring: random history, positions cap - 8 .. cap + 12
    H, D = a.heads, a.dim // a.heads
    rings = ([(0.5 * torch.randn(1, H, a.cap, D, device=dev, generator=g)).to(kvd) for _ in range(a.layers)],
             [(0.5 * torch.randn(1, H, a.cap, D, device=dev, generator=g)).to(kvd) for _ in range(a.layers)])
    ref2, _ = run_steps(tr, xs[:20], False, pos0=a.cap - 8, rings=rings)
    got2, status2 = run_steps(tr, xs[:20], True, pos0=a.cap - 8, rings=rings)
    print(f"full ring across the wrap, status {status2}")
    bad = max(bad, compare(got2, ref2, "full ring"))
    if a.time > 0:
        for pos0, name in ((10, "short context"), (a.cap + 100, "full ring")):
            t_op, _ = time_graph(tr, xs[0], False, pos0, a.time)
            t_ps, st = time_graph(tr, xs[0], True, pos0, a.time)
            print(f"timing, {name}: per-op chain {t_op * 1e3:.1f} us / step, persistent {t_ps * 1e3:.1f} us / step ({t_ps / t_op:.3f}x), status {st}")
    print("PARITY", "ok" if bad < 2e-4 else "FAILED", bad)


if __name__ == "__main__":
    main()
