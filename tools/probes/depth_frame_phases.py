"""Where the time of one persistent depth-frame launch goes (tools build: the stamps compiled in under RST_ABLATION).

    python tools/probes/depth_frame_phases.py [batch]

Moshi-7B's depth transformer (8 steps x 6 layers x 1024, 2048-way heads) through `LMGen._depth`; reads the 100 MHz stamps of
workgroup 0 (owner of head 0 and of the sampler) and prints the mean time per op boundary.  A stamp costs ~0.2 us itself."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rstnet_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "rstnet_amd", "librstnet_hip_ablation.so")
from rstnet_amd import synth  # noqa: E402
from rstnet_amd.lm.model import LMGen, LMModel  # noqa: E402

LAYER = ["rmsnorm 1", "in-proj rows", "gather qkv", "history write + barrier", "attention (wave 0)", "gather att", "out-proj rows",
         "gather x", "rmsnorm 2", "ffn-in rows", "gather h", "ffn-out rows", "gather x"]
MAXL, MAXQ, SL = 8, 8, 13


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = "cuda:0"
    os.environ["RST_DEPTH_FRAME"] = "1"
    cfg = dict(synth.LM_MOSHI_7B, num_layers=1)
    model = LMModel.from_state_dict(synth.lm_state_dict(cfg, seed=0, device=dev), cfg)
    gen = LMGen(model, use_sampling=True)
    Q, L = cfg["dep_q"], cfg["depformer_num_layers"]
    tokens = torch.zeros(B, Q + 1, dtype=torch.long, device=dev)
    lib = _lib.lib()
    lib.rst_debug_depth_frame_stamps.argtypes = [C.c_void_p, C.c_int]
    lib.rst_debug_depth_frame_stamps.restype = C.c_int
    step = SL * MAXL + 4 + 8
    n = step * MAXQ + 2
    lay = np.zeros(SL)
    samp = np.zeros(8)
    per_step = np.zeros((Q, 4))           # embed | layers | head rows | sampler (gather + sample)
    total, frames = 0.0, 0
    g = torch.Generator(device=dev).manual_seed(3)
    with gen.streaming(B):
        for it in range(24):
            h_t = torch.randn(B, cfg["dim"], device=dev, generator=g)
            noise = torch.empty(B, Q * gen.top_k, device=dev).exponential_(1, generator=g)
            gen._depth(tokens, h_t, noise)
            torch.cuda.synchronize()
            if it < 4:
                continue
            buf = (C.c_ulonglong * n)()
            assert lib.rst_debug_depth_frame_stamps(buf, n) == n
            raw = np.array(buf, dtype=np.float64)
            prev = raw[0]
            for k in range(Q):
                s = raw[1 + k * step:1 + (k + 1) * step]
                per_step[k, 0] += s[0] - prev
                prev = s[0]
                for l in range(L):
                    a = s[1 + l * SL:1 + (l + 1) * SL]
                    lay += np.diff(np.concatenate([[prev], a]))
                    prev = a[-1]
                per_step[k, 1] += prev - s[0]
                tail = s[1 + SL * MAXL:]
                samp += np.diff(np.concatenate([tail[1:2], tail[3:10], tail[2:3]]))
                per_step[k, 2] += tail[0] - prev
                per_step[k, 3] += tail[2] - tail[0]
                prev = tail[2]
            total += prev - raw[0]
            frames += 1
    model.depth_frame_tables().check()
    print(f"batch {B}: {frames} launches, stamps span {total / frames / 100:.1f} us per frame ({Q} steps x {L} layers)")
    lay /= frames * Q * L * 100.0
    for nm, v in zip(LAYER, lay):
        print(f"  {nm:26s} {v:6.2f} us")
    print(f"  {'layer':26s} {lay.sum():6.2f} us   x {Q * L} = {lay.sum() * Q * L:.1f} us")
    per_step /= frames * 100.0
    print("  per step (us): embed %.2f | layers %.2f | head rows %.2f | logits gather + sampler %.2f" % tuple(per_step.mean(0)))
    samp /= frames * Q * 100.0
    print("  sampler (us): keys + max %.2f | denominator %.2f | k-th key search %.2f | ties %.2f | compaction %.2f | ranks + draw %.2f | "
          "winner %.2f | token out %.2f" % tuple(samp))
    print("  sampler by step:", " ".join(f"{v:.1f}" for v in per_step[:, 3]))


if __name__ == "__main__":
    main()
