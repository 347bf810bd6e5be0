"""The depth phase of one LM frame on its own (BASELINE configs[2] shape: 8 steps x 6 layers x 1024, batch 1): `LMGen._depth`
captured as a HIP graph and replayed -- device time without the temporal stack around it.  Also times chains of single ops at
the depth transformer's shapes (graph-replayed, weights rotating) so that per-launch costs can be read off directly.

    python tools/bench_depth.py [--batch 1] [--greedy]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rstnet_amd import ops, synth  # noqa: E402
from rstnet_amd.lm.model import LMGen, LMModel  # noqa: E402

DEV = "cuda:0"


def graph_time(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--greedy", action="store_true")
    args = ap.parse_args()
    B = args.batch
    cfg = dict(synth.LM_MOSHI_7B, num_layers=1)          # the temporal stack is not what is measured here
    model = LMModel.from_state_dict(synth.lm_state_dict(cfg, seed=0, device=DEV), cfg)
    gen = LMGen(model, use_sampling=not args.greedy)
    tokens = torch.zeros(B, cfg["dep_q"] + 1, dtype=torch.long, device=DEV)
    h_t = torch.randn(B, cfg["dim"], device=DEV)
    noise = None if args.greedy else torch.empty(B, cfg["dep_q"] * gen.top_k, device=DEV).exponential_(1)
    for mode, name in (("1", "ONE persistent launch (rst_depth_decode_frame)"), ("0", "launch per op")):
        os.environ["RST_DEPTH_FRAME"] = mode
        with gen.streaming(B):
            ms = graph_time(lambda: gen._depth(tokens, h_t, noise))
        print(f"depth phase (8 steps x 6 layers, batch {B}, {'greedy' if args.greedy else 'sampling'}), {name}: {ms * 1e3:8.1f} us per frame",
              flush=True)
    model.depth_frame_tables().check()
    if os.environ.get("BENCH_DEPTH_CHAINS", "1") == "0":
        return

    # single-op chains at the depth shapes: N launches of the same op in one graph (dependent through the stream), rotating weights
    E, Hd = 1024, 2816

    def chain(name, make, n=96):
        ws = make()
        state = {"i": 0}

        def one():
            ws[state["i"] % len(ws)]()
            state["i"] += 1

        def run():
            for _ in range(n):
                one()
        t = graph_time(run, iters=10) / n
        print(f"  {name:44s} {t * 1e3:6.2f} us per launch", flush=True)

    x = torch.randn(B, E, device=DEV)
    alpha = torch.ones(E, device=DEV)
    res = torch.randn(B, E, device=DEV)

    def weights(n, k, copies=24):
        return [(torch.randn(n, k, device=DEV) * 0.03).bfloat16() for _ in range(copies)]
    chain("qkv: 3072x1024 rmsnorm", lambda: [(lambda w=w: ops.gemv_bf16(x, w, prologue=ops.PROLOGUE_RMSNORM, alpha=alpha)) for w in weights(3 * E, E)])
    chain("out-proj: 1024x1024 + res", lambda: [(lambda w=w: ops.gemv_bf16(x, w, res=res)) for w in weights(E, E)])
    chain("out-proj: 1024x1024 no res", lambda: [(lambda w=w: ops.gemv_bf16(x, w)) for w in weights(E, E)])
    chain("ffn-in: 5632x1024 rmsnorm + gate", lambda: [(lambda w=w: ops.gemv_bf16(x, w, prologue=ops.PROLOGUE_RMSNORM, alpha=alpha, gate_out=True))
                                                   for w in weights(2 * Hd, E)])
    g_ = torch.randn(B, Hd, device=DEV)
    chain("ffn-out: 1024x2816 + res", lambda: [(lambda w=w: ops.gemv_bf16(g_, w, res=res)) for w in weights(E, Hd)])
    kc, vc = torch.zeros(B, 16, 8, 64, device=DEV), torch.zeros(B, 16, 8, 64, device=DEV)
    pos = torch.full((1,), 5, dtype=torch.long, device=DEV)
    qkv = torch.randn(B, 3 * E, device=DEV)
    chain("out-proj with attention prologue + res", lambda: [(lambda w=w: ops.gemv_attn(qkv, kc, vc, pos, w, res=res)) for w in weights(E, E)])
    chain("attn_small alone", lambda: [lambda: ops.lm_attn_decode(qkv, kc, vc, pos, rope=False, context=None)])
    logits = torch.randn(B, 2048, device=DEV) * 3
    nz = torch.empty(B, 250, device=DEV).exponential_(1)
    out = torch.empty(B, dtype=torch.long, device=DEV)
    chain("sample 2048 top-250", lambda: [lambda: ops.lm_sample(logits, use_sampling=True, temp=0.8, top_k=250, noise=nz, out=out)])
    chain("head: 2048x1024", lambda: [(lambda w=w: ops.gemv_bf16(x, w)) for w in weights(2048, E)])
    y = torch.empty(256, device=DEV)
    chain("empty-ish kernel (rmsnorm 1x256)", lambda: [lambda: ops.rmsnorm(y.view(1, 256), y)])


if __name__ == "__main__":
    main()
