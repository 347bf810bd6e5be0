"""Where the time of one persistent temporal-frame launch goes (tools build: `make -C rstnet_amd/csrc ablation`, stamps under
RST_ABLATION).

    python tools/probes/temporal_frame_phases.py [--layers 32] [--pos 10] [--wg 0]

Reads the 100 MHz stamps of the first comm thread of workgroup `--wg` at the ten hand-off boundaries of every layer and prints the mean
duration of each span (a stamp costs ~0.2 us itself)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rstnet_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "rstnet_amd", "librstnet_hip_ablation.so")
from rstnet_amd import ops  # noqa: E402
from rstnet_amd.lm.model import StreamingTransformer  # noqa: E402

SPANS = ["gather x (ffn-out rows of the previous layer run meanwhile)", "norm1 (2 barriers)", "gather q/k/v of the head + rope (qkv rows run meanwhile)",
         "ring walk (2 barriers)", "merge + gather attention output", "barrier -> out-proj rows; gather x", "norm2 (2 barriers)",
         "gather gated activation (ffn-in rows run meanwhile)", "barrier"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--pos", type=int, default=10)
    ap.add_argument("--wg", type=int, default=0)
    ap.add_argument("--iters", type=int, default=12)
    a = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    dim, heads, cap = 4096, 32, 3000
    tr = StreamingTransformer(dim, heads, a.layers, int(4.125 * dim), context=cap, positional_embedding="rope", device=dev, dtype=torch.bfloat16,
                              kv_dtype=torch.bfloat16)
    ops.TEMPORAL_FRAME = True
    lib = _lib.lib()
    lib.rst_debug_temporal_frame_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.rst_debug_temporal_frame_stamps.restype = C.c_int
    SL = 16
    n = a.layers * SL
    x = torch.randn(1, dim, device=dev)
    acc = np.zeros((a.layers, 9))
    own = np.zeros((a.layers, 5))
    tot, cnt = 0.0, 0
    with tr.streaming(1):
        st = tr._streaming_state
        st.pos.fill_(a.pos)
        buf = (C.c_ulonglong * n)()
        lib.rst_debug_temporal_frame_stamps(buf, n, a.wg)        # selects the workgroup for the next launches
        for it in range(a.iters):
            tr.step(x)
            torch.cuda.synchronize()
            assert lib.rst_debug_temporal_frame_stamps(buf, n, a.wg) == n
            if it < 3:
                continue
            raw = np.array(buf, dtype=np.float64).reshape(a.layers, SL)
            us = raw[:, :10] / 100.0
            acc += np.diff(us, axis=1)
            tot += us[-1, -1] - us[0, 0]
            # own rows published -> hand-off complete, and the sweeps that took
            own[:, 0] += us[:, 1] - raw[:, 10] / 100.0; own[:, 1] += raw[:, 11]
            own[:, 2] += us[:, 6] - raw[:, 12] / 100.0; own[:, 3] += raw[:, 13]
            own[:, 4] += us[:, 8] - raw[:, 14] / 100.0
            cnt += 1
    wb = (C.c_ulonglong * 256)()
    lib.rst_debug_temporal_frame_wstamps.argtypes = [C.c_void_p]
    lib.rst_debug_temporal_frame_wstamps.restype = C.c_int
    if lib.rst_debug_temporal_frame_wstamps(wb) == 256 and a.layers > 1:
        w = np.array(wb, dtype=np.float64).reshape(64, 4)
        names = {0: "A", 1: "KV", 2: "B", 3: "C", 4: "D"}
        print("weight wave 0, layer 1, last launch: block type | wait before (us) | multiply + publish (us) | gap to the next block's entry (next + issue)")
        for k in range(64):
            if w[k, 1] == 0:
                break
            gap = (w[k + 1, 1] - w[k, 3]) / 100.0 if k + 1 < 64 and w[k + 1, 1] else float("nan")
            print(f"   {names.get(int(w[k, 0]), '?'):2s}  {(w[k, 2] - w[k, 1]) / 100.0:7.2f}  {(w[k, 3] - w[k, 2]) / 100.0:7.2f}  {gap:7.2f}")
    acc /= cnt
    print(f"workgroup {a.wg}, position {a.pos}: launch (first stamp -> last stamp) {tot / cnt:.1f} us over {a.layers} layers = {tot / cnt / a.layers:.2f} us per layer")
    mean = acc[1:].mean(axis=0) if a.layers > 1 else acc[0]
    for name, v in zip(SPANS, mean):
        print(f"  {v:7.2f} us  {name}")
    print(f"  {mean.sum():7.2f} us  sum (layers 1..)")
    print("first layer:", " ".join(f"{v:.2f}" for v in acc[0]))
    o = own[1:].mean(axis=0) / cnt
    print(f"  own rows published -> vector gathered: x (layer input) {o[0]:.2f} us in {o[1]:.1f} sweeps; x (after attention) {o[2]:.2f} us in {o[3]:.1f} sweeps; "
          f"gated activation {o[4]:.2f} us")


if __name__ == "__main__":
    main()
