"""Where the time of one persistent codec-transformer launch goes (tools build: the stamps compiled in under RST_ABLATION).

    python tools/probes/codec_tr_phases.py [streams] [chunk]

Streams a Mimi encoder_transformer (8 layers, E=512, F=2048, ring of 250) past the ring wrap, then reads the 100 MHz stamps of
workgroup 0 (owner of a head) and of the last workgroup (no head) for a number of frames and prints the mean time per op boundary.
A stamp costs ~0.2 us itself (s_memrealtime + an LDS store by one lane): 20 of them stretch a layer by ~4 us."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rstnet_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "rstnet_amd", "librstnet_hip_ablation.so")
from rstnet_amd import ops, synth  # noqa: E402
from rstnet_amd.codec.mimi import MimiCodec  # noqa: E402

NAMES = ["LN1", "in-proj rows", "gather qkv", "rope + ring write", "attention", "gather att", "out-proj rows", "gather x", "LN2",
         "linear1 rows", "gather h", "linear2 rows", "gather x"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = "cuda:0"
    sd = synth.mimi_state_dict(1234)
    model = MimiCodec.from_state_dict(sd).to(dev)
    tr = model.encoder_transformer
    g = torch.Generator().manual_seed(1)
    lib = _lib.lib()
    lib.rst_debug_codec_tr_stamps.argtypes = [C.c_void_p, C.c_int]
    lib.rst_debug_codec_tr_stamps.restype = C.c_int
    SP = 20                        # stamps per layer: 13 op boundaries (in order) + 7 inside the attention
    n = 2 * (SP * 8 + 2)
    acc = np.zeros((2, 13 * 8 + 1))
    att = np.zeros(8)
    total, frames = 0.0, 0
    with tr.streaming(B):
        for i in range(400 // chunk):
            x = torch.randn(B, 512, chunk, generator=g).to(dev)
            tr(x)
            if i * chunk < 300:
                continue
            torch.cuda.synchronize()
            buf = (C.c_ulonglong * n)()
            assert lib.rst_debug_codec_tr_stamps(buf, n) == n
            raw = np.array(buf, dtype=np.float64).reshape(2, SP * 8 + 2)
            lay = raw[:, 1:1 + SP * 8].reshape(2, 8, SP)
            st = np.concatenate([raw[:, :1], lay[:, :, :13].reshape(2, -1)], axis=1)
            a = lay[0]                                     # workgroup 0: rope/ring write -> sub-stamps 13..19 -> attention done
            seq = np.concatenate([a[:, 3:4], a[:, 13:20], a[:, 4:5]], axis=1)
            att += np.diff(seq, axis=1).mean(0)
            st = np.maximum.accumulate(st, axis=1)         # boundaries a workgroup does not pass (no head) keep the previous stamp
            acc += np.diff(st, axis=1, prepend=st[:, :1])
            total += st[0, -1] - st[0, 0]
            frames += 1
    acc /= frames
    print(f"streams {B}, chunk {chunk}: {frames} launches, stamps span {total / frames / 100:.1f} us per launch (8 layers)")
    per = acc[:, 1:].reshape(2, 8, 13).mean(1) / 100.0        # us per boundary, mean over layers
    print(f"{'op':22s} {'wg 0':>8s} {'last wg':>8s}   (us, mean over the 8 layers)")
    for j, nm in enumerate(NAMES):
        print(f"{nm:22s} {per[0, j]:8.2f} {per[1, j]:8.2f}")
    print(f"{'layer':22s} {per[0].sum():8.2f} {per[1].sum():8.2f}")
    att /= frames * 100.0
    for nm, v in zip(["K / V loads issued", "(A) q.k -> LDS", "mask, row max, barrier", "numerators, row sum, barrier", "(B) p.V", "partials -> LDS", "barrier", "combine + publish"], att):
        print(f"  attention: {nm:30s} {v:6.2f}")


if __name__ == "__main__":
    main()
