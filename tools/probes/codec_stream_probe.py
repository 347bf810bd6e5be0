"""Where a streamed codec frame goes at N concurrent streams: the stages of `MimiCodec.encode` / `decode` of one 80 ms frame, each
captured as a HIP graph of its own and replayed (us per replay), and the whole encode / decode steps the same way.

    python tools/probes/codec_stream_probe.py [--streams 32] [--iters 200]

Module-level switches of rstnet_amd.ops can be set from the command line for an A/B inside one call: --set NAME=VALUE (repeatable),
e.g. --set SKINNY_F32_ROWS=False --set ATTENTION_STEP=False.
"""
import argparse
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from rstnet_amd import ops, synth  # noqa: E402
from rstnet_amd.codec.mimi import MimiCodec  # noqa: E402


def graph_time(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    del out
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=32)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--set", action="append", default=[])
    a = ap.parse_args()
    for kv in a.set:
        k, v = kv.split("=", 1)
        if not hasattr(ops, k):
            raise SystemExit(f"rstnet_amd.ops has no switch {k}")
        setattr(ops, k, ast.literal_eval(v))
    dev = "cuda:0"
    B = a.streams
    mimi = MimiCodec.from_state_dict(synth.mimi_state_dict(0)).to(dev)
    pcm = synth.synth_audio(B, 1920 * 4, seed=7).to(dev)
    sw = " ".join(a.set) or "(defaults)"
    print(f"{B} streams, one 80 ms frame per replay, us per replay; switches: {sw}")
    with torch.no_grad(), mimi.streaming(B):
        # two eager frames first: conv histories, rings and scratch exist before anything is captured
        for s in range(2):
            codes = mimi.quantizer.encode_nlc(mimi.encode_latent(pcm[:, :, s * 1920:(s + 1) * 1920].contiguous()))
            mimi._decode(codes)
        x = pcm[:, :, :1920].contiguous().view(B, 1920, 1)

        def enc_seanet():
            with ops.hist_batch():
                return mimi.encoder.forward_nlc(x, None)
        z = enc_seanet()

        def enc_tr():
            return mimi.encoder_transformer.forward_nlc(z)[0]
        zt = enc_tr()

        def enc_down():
            with ops.hist_batch():
                return mimi.downsample.forward_nlc(zt)
        lat = enc_down()

        def enc_rvq():
            return mimi.quantizer.encode_nlc(lat)
        codes = enc_rvq()

        def dec_rvq():
            return mimi.quantizer.decode_nlc(codes)
        q = dec_rvq()

        def dec_up():
            with ops.hist_batch():
                return mimi.upsample.forward_nlc(q)
        u = dec_up()

        def dec_tr():
            return mimi.decoder_transformer.forward_nlc(u)[0]
        ut = dec_tr()

        def dec_seanet():
            with ops.hist_batch():
                return mimi.decoder.forward_nlc(ut)

        rows = [("encoder SEANet", enc_seanet), ("encoder transformer (8 layers)", enc_tr), ("downsample", enc_down), ("RVQ encode", enc_rvq),
                ("RVQ decode", dec_rvq), ("upsample", dec_up), ("decoder transformer (8 layers)", dec_tr), ("decoder SEANet", dec_seanet)]
        total = 0.0
        for name, fn in rows:
            t = graph_time(fn, a.iters)
            total += t
            print(f"  {name:34s} {t:8.1f}")
        print(f"  {'sum of the stages':34s} {total:8.1f}")
        frame = pcm[:, :, :1920].contiguous()
        t_enc = graph_time(lambda: mimi.quantizer.encode_nlc(mimi.encode_latent(frame)), a.iters)
        t_dec = graph_time(lambda: mimi._decode(codes), a.iters)
        print(f"  {'encode step (one graph)':34s} {t_enc:8.1f}")
        print(f"  {'decode step (one graph)':34s} {t_dec:8.1f}")
        print(f"  {'encode + decode':34s} {t_enc + t_dec:8.1f}")


if __name__ == "__main__":
    main()
