"""Where does the three-plane residual block differ from the f32-instruction kernel?  (debug probe)"""
import sys
import torch
sys.path.insert(0, ".")
from rstnet_amd import ops, synth
from rstnet_amd.codec import functional as RF
DEV = "cuda:0"
for (C, B, T) in ((64, 1, 4100), (64, 3, 20011), (64, 9, 3000)):
    g = torch.Generator().manual_seed(1)
    w1 = synth._xavier(g, C // 2, C, 3); w2 = synth._xavier(g, C, C // 2, 1)
    b1, b2 = 0.1 * torch.randn(C // 2, generator=g), 0.1 * torch.randn(C, generator=g)
    x = (torch.rand(B, T, C, generator=g) * 4 - 2).to(DEV)
    args = (RF.pack_conv_weight(w1).to(DEV), b1.to(DEV), RF.pack_conv_weight(w2).to(DEV), b2.to(DEV))
    y3 = ops.seanet_resblock(x, *args, Kw=3)
    ops.GEMM_B3 = False
    y1 = ops.seanet_resblock(x, *args, Kw=3)
    ops.GEMM_B3 = True
    d = (y3 - y1).abs()
    bad = (d > 1e-3).nonzero()
    print(C, B, T, "max diff", float(d.max()), "bad elements", bad.shape[0])
    if bad.shape[0]:
        rows = torch.unique(bad[:, 0] * T + bad[:, 1])
        print("  bad rows (b*T+t):", rows[:40].tolist(), "...", rows[-5:].tolist(), "count", rows.numel())
        print("  bad t mod 32:", torch.unique(bad[:, 1] % 32).tolist())
        print("  bad channels:", torch.unique(bad[:, 2]).tolist()[:70])
        r = bad[0]
        print("  first bad", r.tolist(), float(y3[r[0], r[1], r[2]]), float(y1[r[0], r[1], r[2]]), "x", float(x[r[0], r[1], r[2]]))
