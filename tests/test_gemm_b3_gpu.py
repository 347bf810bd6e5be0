"""The three-plane bf16 form of the large windowed GEMMs (rst_gemm_win_b3_f32, gemm_win.hip: every fp32 operand = hi + mid + lo in
bf16, six of the nine cross products on the bf16 matrix instruction, fp32 accumulate) against fp64 references: the split is exact,
and the results carry fp32 accuracy -- the same error scale as the f32 matrix instruction's, which stays the route of every other
shape (and of `ops.GEMM_B3 = False`)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import mimi_oracle as O
from rstnet_amd import _lib, ops, synth
from rstnet_amd.codec import functional as RF

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
U = 2.0 ** -24


def unpack_b3(w3: torch.Tensor, N: int, K: int) -> torch.Tensor:
    """[ceil(N/256)*8 blocks of 32 rows][K/16][3 planes][64 lanes][8] int16 (matrix-instruction operand order: lane (n % 32) + 32 *
    (k % 16 // 8) of a block holds k % 8 .. of row n) -> the three planes as fp32 ``[3, ceil(N/256)*256, K]``."""
    nb = w3.numel() // (3 * 32 * K)                  # 32-row blocks, a multiple of 8 (whole 256-row tiles)
    t = w3.view(nb, K // 16, 3, 2, 32, 8)            # [block][k-tile][plane][k half][row in block][k % 8]
    t = t.permute(2, 0, 4, 1, 3, 5).reshape(3, nb * 32, K)
    return (t.to(torch.int32) << 16).view(torch.float32)


@pytest.mark.parametrize("N,K", [(128, 32), (200, 64), (513, 2048), (64, 16)])   # (the packing itself only needs K % 16 == 0)
def test_split_weights_are_exact(N, K):
    g = torch.Generator().manual_seed(N + K)
    w = (torch.randn(N, K, generator=g) * torch.exp(4 * torch.randn(N, K, generator=g))).to(DEV)      # wide dynamic range
    w3 = ops.gemm_win_b3_pack_weight(w)
    assert w3.numel() == _lib.lib().rst_gemm_win_b3_weight_elems(N, K) == -(-N // 256) * 256 * K * 3
    planes = unpack_b3(w3, N, K)
    assert not planes[:, N:].any()                                # rows past N are zero
    hi, mid, lo = planes[:, :N]
    assert torch.equal((hi + mid) + lo, w)                       # three bf16 numbers, exactly the fp32 value
    assert float((mid.abs() / hi.abs().clamp_min(1e-30)).max()) <= 2.0 ** -8 * 1.01
    assert float((lo.abs() / hi.abs().clamp_min(1e-30)).max()) <= 2.0 ** -16 * 1.01


def _backward_error(y, ref, bound):
    return float(((y.double().cpu() - ref).abs() / bound).max())


@pytest.mark.parametrize("M,N,K", [(16640, 512, 192), (100000, 200, 64), (8200, 1024, 2048), (5000, 128, 64), (4100, 128, 4096)])
def test_b3_linear_carries_fp32_accuracy(M, N, K, monkeypatch):
    g = torch.Generator().manual_seed(M % 1000 + N + K)
    x = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K, generator=g))
    w = torch.randn(N, K, generator=g) / K ** 0.5
    ref = F.linear(x.double(), w.double())
    bound = F.linear(x.double().abs(), w.double().abs())          # sum_k |x_k| |w_k|: the scale every rounding error is relative to
    monkeypatch.setattr(ops, "GEMM_B3", True)
    ops.PROFILE = []
    y3 = ops.linear(x.to(DEV), w.to(DEV))
    names, ops.PROFILE = [r[0] for r in ops.PROFILE], None
    assert names == ["gemm_win_b3"]
    monkeypatch.setattr(ops, "GEMM_B3", False)
    y1 = ops.linear(x.to(DEV), w.to(DEV))
    e3, e1 = _backward_error(y3, ref, bound), _backward_error(y1, ref, bound)
    print(f"M={M} N={N} K={K}: backward error / 2^-24: three-plane {e3 / U:.2f}, f32 instruction {e1 / U:.2f}")
    # measured (tools/probes/b3_numerics.py): 4.6 - 5.0 against the f32 instruction's 5.3 - 6.1 on random-sign operands for K = 96 ..
    # 8192 (66 against 105 on all-positive ones at K = 8192): the dropped cross terms (<= 2^-23 |x||w|) cost less than the f32
    # chain's own K roundings
    assert e3 < 1.25 * e1 + U
    assert e3 < 16 * U


def test_b3_conv_with_elu_residual_and_utterance_edges():
    """k3 s1 64 -> 256 over 40 utterances of 700 steps with ELU on load, residual and ELU-out: interior tiles on the bf16 instruction,
    tiles that touch an utterance edge on the f32 one, alternating inside every resident workgroup; and the strided SEANet shape."""
    g = torch.Generator().manual_seed(35)
    B, cin, cout, T = 40, 64, 256, 700
    x = torch.rand(B, cin, T, generator=g) * 4 - 2
    w = synth._xavier(g, cout, cin, 3)
    b = 0.1 * torch.randn(cout, generator=g)
    res = torch.randn(B, cout, T, generator=g)
    nlc = lambda t: t.transpose(1, 2).contiguous().to(DEV)
    y = RF.conv1d(nlc(x), RF.pack_conv_weight(w).to(DEV), b.to(DEV), k_eff=3, act_in=ops.ACT_ELU, res=nlc(res), act_out=ops.ACT_ELU_OUT)
    ref = F.elu(res.double() + F.conv1d(F.pad(F.elu(x.double()), (2, 0)), w.double(), b.double()))
    err = float((y.transpose(1, 2).double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err
    B, cin, cout, K, S, T = 3, 64, 128, 8, 4, 131073
    x = torch.rand(B, cin, T, generator=g) * 2 - 1
    w = synth._xavier(g, cout, cin, K)
    y = RF.conv1d(nlc(x), RF.pack_conv_weight(w).to(DEV), None, k_eff=K, stride=S)
    ref = O.causal_conv1d(x.double(), w.double(), None, stride=S)
    err = float((y.transpose(1, 2).double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err


def _names_of(fn):
    ops.PROFILE = []
    out = fn()
    names, ops.PROFILE = [r[0] for r in ops.PROFILE], None
    return out, names


def test_b3_wide_tiles_lean_and_masked():
    """The 128 x 256 tile form (512 threads; chosen when N >= 256 and the wide tiles outnumber the CUs two to one): a linear whose tiles
    are all full (lean body), the same with ragged rows and columns (masked body, partial epilogue), and a k3 convolution over three
    utterances with ELU on load + residual + ELU out (masked rows at every utterance start, M not a multiple of 128)."""
    g = torch.Generator().manual_seed(41)
    for M, N, K in ((516 * 128, 256, 128), (66100, 520, 64), (40000, 1024, 320)):
        x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
        b, res, scale = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.rand(N, generator=g)
        ref = F.linear(x.double(), w.double(), b.double())
        y, names = _names_of(lambda: ops.linear(x.to(DEV), w.to(DEV), b.to(DEV)))
        assert names == ["gemm_win_b3"]
        assert float((y.double().cpu() - ref).abs().max() / ref.abs().max()) < 2e-6, (M, N, K)
        y = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), res=res.to(DEV), scale=scale.to(DEV), act_out=ops.ACT_GELU)
        want = res.double() + scale.double() * F.gelu(ref)
        assert float((y.double().cpu() - want).abs().max() / want.abs().max()) < 2e-6, (M, N, K)
    B, cin, cout, T = 3, 64, 512, 22100
    x = torch.rand(B, cin, T, generator=g) * 4 - 2
    w = synth._xavier(g, cout, cin, 3)
    b = 0.1 * torch.randn(cout, generator=g)
    res = torch.randn(B, cout, T, generator=g)
    nlc = lambda t: t.transpose(1, 2).contiguous().to(DEV)
    y, names = _names_of(lambda: RF.conv1d(nlc(x), RF.pack_conv_weight(w).to(DEV), b.to(DEV), k_eff=3, act_in=ops.ACT_ELU, res=nlc(res),
                                           act_out=ops.ACT_ELU_OUT))
    assert names == ["gemm_win_b3"]
    ref = F.elu(res.double() + F.conv1d(F.pad(F.elu(x.double()), (2, 0)), w.double(), b.double()))
    assert float((y.transpose(1, 2).double().cpu() - ref).abs().max() / ref.abs().max()) < 2e-6


def test_b3_shapes_that_stay_on_the_f32_instruction():
    """A history buffer (streaming chunk of > 4096 rows), replicate padding and K % 64 != 0 keep the f32 kernels -- same results as ever."""
    g = torch.Generator().manual_seed(42)
    B, cin, cout, K, S, T = 2, 64, 128, 4, 2, 12000
    full = torch.rand(B, cin, T + K - S, generator=g) * 2 - 1
    w = synth._xavier(g, cout, cin, K)
    nlc = lambda t: t.transpose(1, 2).contiguous().to(DEV)
    hist, x = full[:, :, :K - S], full[:, :, K - S:]
    y, names = _names_of(lambda: RF.conv1d(nlc(x), RF.pack_conv_weight(w).to(DEV), None, k_eff=K, stride=S, hist=nlc(hist)))
    assert names == ["gemm_win"]
    ref = F.conv1d(full.double(), w.double(), None, stride=S)
    assert float((y.transpose(1, 2).double().cpu() - ref).abs().max() / ref.abs().max()) < 2e-6
    y, names = _names_of(lambda: RF.conv1d(nlc(x), RF.pack_conv_weight(w).to(DEV), None, k_eff=K, stride=S, pad_mode=ops.PAD_REPLICATE))
    assert names == ["gemm_win"]
    ref = O.causal_conv1d(x.double(), w.double(), None, stride=S, pad_mode="replicate")
    assert float((y.transpose(1, 2).double().cpu() - ref).abs().max() / ref.abs().max()) < 2e-6


def test_b3_pack_rejects_bad_shapes():
    w = torch.randn(64, 24, device=DEV)
    with pytest.raises(ValueError):
        ops.gemm_win_b3_pack_weight(w)        # K % 16 != 0
    assert not ops._b3_shape(5000, 128, 48) and not ops._b3_shape(4096, 128, 64) and not ops._b3_shape(5000, 64, 64)


def test_b3_refuses_what_it_does_not_run():
    """rst_gemm_win_b3_f32 never changes instruction silently: a call outside rst_gemm_win_b3_supported fails, and the Python layer
    routes by that predicate (so `gemm_win_b3` in a profile names the kernel that ran)."""
    L = _lib.lib()
    assert L.rst_gemm_win_b3_supported(1, 5000, 5000, 128, 128, 128, 1, 0, 0, 5000 * 128, 0) == 1
    assert L.rst_gemm_win_b3_supported(1, 5000, 5000, 128, 128, 128, 1, 0, 1, 5000 * 128, 0) == 0      # replicate padding
    assert L.rst_gemm_win_b3_supported(1, 5000, 5000, 128, 128, 128, 1, 0, 0, 5000 * 128, 1) == 0      # history buffer
    assert L.rst_gemm_win_b3_supported(1, 5000, 5000, 96, 96, 128, 1, 0, 0, 5000 * 96, 0) == 0         # K % 64
    assert L.rst_gemm_win_b3_supported(1, 4096, 4096, 128, 128, 128, 1, 0, 0, 4096 * 128, 0) == 0      # not a large launch
    assert L.rst_gemm_win_b3_supported(1, 5000, 5000, 128, 128, 64, 1, 0, 0, 5000 * 128, 0) == 0       # N <= 64
    assert L.rst_gemm_win_b3_supported(70, 240000, 240000, 64, 64, 128, 1, 0, 0, 240000 * 64, 0) == 0  # activations beyond 4 GB
    M, N, K = 5000, 128, 128
    x, w = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV)
    w3 = ops.gemm_win_b3_pack_weight(w)
    y = torch.empty(M, N, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    args = lambda xx, pad: (xx.data_ptr(), None, w.data_ptr(), w3.data_ptr(), None, None, None, y.data_ptr(), 1, M, M, K, K, N, 1, 0, pad, M * K, N, 0, 0, st)
    assert L.rst_gemm_win_b3_f32(*args(x, 0)) == 0
    assert L.rst_gemm_win_b3_f32(*args(x, 1)) < 0 and b"gemm_win_b3" in L.rst_last_error()
    x_off = torch.randn(M * K + 4, device=DEV)[1:1 + M * K].view(M, K)          # 4-byte aligned only
    assert x_off.data_ptr() % 16 != 0
    assert L.rst_gemm_win_b3_f32(*args(x_off, 0)) < 0
    out, names = _names_of(lambda: ops.linear(x_off, w))
    assert names == ["gemm_win"]                                                 # routed to the f32 instruction by the caller
    assert float((out - x_off @ w.t()).abs().max()) < 1e-3


def test_b3_tiny_magnitudes_degrade_like_flush_to_zero():
    """Operands whose lo (then mid) plane leaves bf16's normal range: |x| ~ 2^-100 .. 2^-124.  fp32 still carries 24 bits there; the
    three-plane form keeps fp32 accuracy down to |x| ~ 2^-110 and below that stays within 2 * 2^-126 * sum_k |w_k| absolute of the exact
    result (the planes that fall out of range are at most that large) -- the statement of include/rstnet_hip.h."""
    g = torch.Generator().manual_seed(5)
    M, N, K = 4224, 128, 256
    w = torch.randn(N, K, generator=g) / K ** 0.5
    for e in (-100, -112, -118, -124):
        x = torch.randn(M, K, generator=g) * 2.0 ** e
        ref = F.linear(x.double(), w.double())
        bound = F.linear(x.double().abs(), w.double().abs())
        y3, names = _names_of(lambda: ops.linear(x.to(DEV), w.to(DEV)))
        assert names == ["gemm_win_b3"]
        err = (y3.double().cpu() - ref).abs()
        floor = 2 * 2.0 ** -126 * w.double().abs().sum(1)                        # per output column
        excess = float(((err - floor[None, :]).clamp_min(0) / bound).max())
        print(f"|x| ~ 2^{e}: max backward error {float((err / bound).max()) / U:.2f} x 2^-24, beyond the 2^-126 floor {excess / U:.2f} x 2^-24")
        assert excess < 16 * U
        if e >= -108:
            assert float((err / bound).max()) < 16 * U
    # and tiny WEIGHTS against normal activations (the split is symmetric)
    x = torch.randn(M, K, generator=g)
    w_t = w * 2.0 ** -118
    ref = F.linear(x.double(), w_t.double())
    bound = F.linear(x.double().abs(), w_t.double().abs())
    y3 = ops.linear(x.to(DEV), w_t.to(DEV))
    err = (y3.double().cpu() - ref).abs()
    floor = 2 * 2.0 ** -126 * x.double().abs().sum(1)
    assert float(((err - floor[:, None]).clamp_min(0) / bound).max()) < 16 * U


def test_b3_non_finite_operands_stay_non_finite():
    """+-Inf / NaN activations (and finite ones beyond the largest bf16, whose hi plane rounds to Inf): every output of the rows they
    sit in is non-finite on both routes -- the f32 instruction gives +-Inf or NaN, the three-plane form NaN (x - hi = Inf - Inf) -- and
    every other row is untouched, bit for bit.  Same for a non-finite weight and its output column."""
    g = torch.Generator().manual_seed(6)
    M, N, K = 4224, 192, 128
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    clean3 = ops.linear(x.to(DEV), w.to(DEV)).cpu()
    bad = {7: float("inf"), 1000: float("-inf"), 2049: float("nan"), 4223: 3.4e38}
    xb = x.clone()
    for r, v in bad.items():
        xb[r, (r * 7) % K] = v
    y3, names = _names_of(lambda: ops.linear(xb.to(DEV), w.to(DEV)))
    assert names == ["gemm_win_b3"]
    y3 = y3.cpu()
    ops.GEMM_B3 = False
    try:
        y1 = ops.linear(xb.to(DEV), w.to(DEV)).cpu()
    finally:
        ops.GEMM_B3 = True
    rows = torch.tensor(sorted(bad))
    assert not torch.isfinite(y3[rows[:3]]).any() and not torch.isfinite(y1[rows[:3]]).any()       # Inf, -Inf, NaN rows: both non-finite
    assert not torch.isfinite(y3[4223]).any() and torch.isfinite(y1[4223]).all()                   # 3.4e38: finite fp32, beyond bf16 (documented)
    keep = torch.ones(M, dtype=torch.bool)
    keep[rows] = False
    assert torch.equal(y3[keep], clean3[keep])                                                     # nothing leaks across rows
    wb = w.clone()
    wb[5, 3] = float("inf")
    y3 = ops.linear(x.to(DEV), wb.to(DEV)).cpu()
    assert not torch.isfinite(y3[:, 5]).any()
    colkeep = torch.ones(N, dtype=torch.bool)
    colkeep[5] = False
    assert torch.equal(y3[:, colkeep], clean3[:, colkeep])
