"""The fused SEANet residual block on the bf16 matrix instruction (rst_seanet_resblock_b3_f32, csrc/resblock_b3.hip: three-plane operands,
six products per fp32 product, the hidden activation split in the first GEMM's epilogue) against fp64 references of
modules/seanet.py:21-94 -- C = 64 plain / with the first convolution ("pre", :184-193) / with the last one ("post", :368-379), C = 128
plain -- at whole-utterance sizes with utterance edges, ragged last tiles and more tiles than resident waves; and next to the f32-instruction
kernel of the same block (`ops.GEMM_B3 = False`), whose error it must not exceed."""
import pytest
import torch
import torch.nn.functional as F

from rstnet_amd import _lib, ops, synth
from rstnet_amd.codec import functional as RF

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _weights(C, seed, K0=7, Kf=3):
    g = torch.Generator().manual_seed(seed)
    w1 = synth._xavier(g, C // 2, C, 3)
    w2 = synth._xavier(g, C, C // 2, 1)
    b1, b2 = 0.1 * torch.randn(C // 2, generator=g), 0.1 * torch.randn(C, generator=g)
    w0 = synth._xavier(g, C, 1, K0)
    b0 = 0.1 * torch.randn(C, generator=g)
    wf = synth._xavier(g, 1, C, Kf)
    bf = 0.1 * torch.randn(1, generator=g)
    return g, w0, b0, w1, b1, w2, b2, wf, bf


def _block64(x, w1, b1, w2, b2):
    """fp64 y = x + conv_k1(ELU(conv_k3(ELU(x)))) on [B, C, T] (causal, zero padding)."""
    h = F.conv1d(F.pad(F.elu(x), (2, 0)), w1.double(), b1.double())
    return x + F.conv1d(F.elu(h), w2.double(), b2.double())


def _run(x_nlc, w1, b1, w2, b2, **kw):
    ops.PROFILE = []
    y = ops.seanet_resblock(x_nlc.to(DEV), RF.pack_conv_weight(w1).to(DEV), b1.to(DEV), RF.pack_conv_weight(w2).to(DEV), b2.to(DEV), Kw=3, **kw)
    names, ops.PROFILE = [r[0] for r in ops.PROFILE], None
    return y.cpu(), names


def _err(y, ref):
    return float((y.double() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("C,B,T,elu_out", [(64, 3, 20011, False), (64, 1, 4100, True), (64, 9, 3000, False), (128, 3, 20011, True), (128, 1, 4097, False),
                                           (128, 40, 700, False)])
def test_plain_block(C, B, T, elu_out, monkeypatch):
    g, w0, b0, w1, b1, w2, b2, wf, bf = _weights(C, 100 + C + T)
    x = torch.rand(B, C, T, generator=g) * 4 - 2
    ref = _block64(x.double(), w1, b1, w2, b2)
    if elu_out:
        ref = F.elu(ref)
    ref = ref.transpose(1, 2)
    x_nlc = x.transpose(1, 2).contiguous()
    y3, names = _run(x_nlc, w1, b1, w2, b2, elu_out=elu_out)
    assert names == ["resblock_b3"]
    monkeypatch.setattr(ops, "GEMM_B3", False)
    y1, names1 = _run(x_nlc, w1, b1, w2, b2, elu_out=elu_out)
    assert names1 == ["resblock"]
    e3, e1 = _err(y3, ref), _err(y1, ref)
    print(f"C={C} B={B} T={T}: max error / output scale: three-plane {e3:.2e}, f32 instruction {e1:.2e}")
    assert e3 < 2e-6 and e3 < 1.5 * e1 + 1e-7


@pytest.mark.parametrize("B,T,K0", [(3, 20011, 7), (1, 4099, 7), (70, 131, 7), (2, 9000, 8), (2, 9000, 3)])
def test_block_with_the_first_convolution(B, T, K0, monkeypatch):
    C = 64
    g, w0, b0, w1, b1, w2, b2, wf, bf = _weights(C, 200 + T + K0, K0=K0)
    a = torch.rand(B, 1, T, generator=g) * 2 - 1
    x = F.conv1d(F.pad(a.double(), (K0 - 1, 0)), w0.double(), b0.double())
    ref = _block64(x, w1, b1, w2, b2).transpose(1, 2)
    pre = (w0[:, 0].contiguous().to(DEV), b0.to(DEV))
    y3, names = _run(a.view(B, T, 1), w1, b1, w2, b2, pre=pre)
    assert names == ["resblock_b3"]
    monkeypatch.setattr(ops, "GEMM_B3", False)
    y1, _ = _run(a.view(B, T, 1), w1, b1, w2, b2, pre=pre)
    e3, e1 = _err(y3, ref), _err(y1, ref)
    print(f"pre B={B} T={T} K0={K0}: three-plane {e3:.2e}, f32 instruction {e1:.2e}")
    assert e3 < 2e-6 and e3 < 1.5 * e1 + 1e-7


@pytest.mark.parametrize("B,T,Kf", [(3, 20011, 3), (1, 4099, 3), (70, 131, 3), (2, 9000, 4), (2, 9000, 1)])
def test_block_with_the_last_convolution(B, T, Kf, monkeypatch):
    C = 64
    g, w0, b0, w1, b1, w2, b2, wf, bf = _weights(C, 300 + T + Kf, Kf=Kf)
    x = torch.rand(B, C, T, generator=g) * 4 - 2
    yb = _block64(x.double(), w1, b1, w2, b2)
    ref = F.conv1d(F.pad(F.elu(yb), (Kf - 1, 0)), wf.double(), bf.double()).transpose(1, 2)
    post = (wf[0].t().contiguous().to(DEV), bf.to(DEV))
    x_nlc = x.transpose(1, 2).contiguous()
    y3, names = _run(x_nlc, w1, b1, w2, b2, post=post)
    assert names == ["resblock_b3"] and y3.shape == (B, T, 1)
    monkeypatch.setattr(ops, "GEMM_B3", False)
    y1, _ = _run(x_nlc, w1, b1, w2, b2, post=post)
    e3, e1 = _err(y3, ref), _err(y1, ref)
    print(f"post B={B} T={T} Kf={Kf}: three-plane {e3:.2e}, f32 instruction {e1:.2e}")
    assert e3 < 2e-6 and e3 < 1.5 * e1 + 1e-7


def test_routing_and_refusals():
    L = _lib.lib()
    assert L.rst_seanet_resblock_b3_supported(64, 240000, 64, 32, 3, 0, 0, 0, 0) == 1
    assert L.rst_seanet_resblock_b3_supported(64, 240000, 64, 32, 3, 1, 0, 7, 0) == 1
    assert L.rst_seanet_resblock_b3_supported(64, 240000, 64, 32, 3, 1, 1, 7, 3) == 0          # both ends folded in: the f32 kernel
    assert L.rst_seanet_resblock_b3_supported(64, 60000, 128, 64, 3, 0, 0, 0, 0) == 1
    assert L.rst_seanet_resblock_b3_supported(64, 60000, 128, 64, 3, 0, 1, 0, 3) == 0
    assert L.rst_seanet_resblock_b3_supported(64, 12000, 256, 128, 3, 0, 0, 0, 0) == 0
    assert L.rst_seanet_resblock_b3_supported(1, 20_000_000, 64, 32, 3, 0, 0, 0, 0) == 0       # an utterance beyond 4 GB
    assert L.rst_seanet_resblock_b3_weight_elems(64) > 0 and L.rst_seanet_resblock_b3_weight_elems(128) > 0 and L.rst_seanet_resblock_b3_weight_elems(96) < 0
    # few rows and streaming chunks stay on the f32-instruction kernels
    g, w0, b0, w1, b1, w2, b2, wf, bf = _weights(64, 7)
    x = torch.rand(2, 300, 64, generator=g)
    _, names = _run(x, w1, b1, w2, b2)
    assert names == ["resblock"]
    hist = torch.zeros(2, 2, 64)
    _, names = _run(torch.rand(2, 5000, 64, generator=g), w1, b1, w2, b2, hist=hist.to(DEV))
    assert names == ["resblock"]
