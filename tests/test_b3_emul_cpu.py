"""The arithmetic the three-plane bf16 GEMM rests on (oracle/b3_emul.py), checked on the host: the split is exact, each level is at
most 2^-8 of the one before, and the six kept products reproduce the fp32 product to 2^-23 |x||w| per term -- the bound DESIGN.md
3.1b quotes and tests/test_gemm_b3_gpu.py checks on the kernel itself."""
import torch

from oracle import b3_emul as E


def _operands(seed, M=64, N=48, K=512):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g) * torch.exp(3 * torch.randn(M, K, generator=g))      # wide dynamic range
    w = torch.randn(N, K, generator=g) / K ** 0.5
    return x, w


def test_split_is_exact_and_geometric():
    x, _ = _operands(0)
    hi, mid, lo = E.split3(x)
    assert torch.equal((hi + mid) + lo, x)                      # three bf16 numbers, exactly the fp32 value
    for t in (hi, mid, lo):
        assert torch.equal(t.bfloat16().float(), t)             # each plane IS a bf16 number
    nz = hi != 0
    assert float((mid[nz].abs() / hi[nz].abs()).max()) <= 2.0 ** -8
    assert float((lo[nz].abs() / hi[nz].abs()).max()) <= 2.0 ** -16
    # values that are bf16 numbers already have empty lower planes
    hb, mb, lb = E.split3(x.bfloat16().float())
    assert not mb.any() and not lb.any()


def test_six_products_carry_fp32_accuracy():
    for seed in range(3):
        x, w = _operands(seed)
        ref = x.double() @ w.double().T
        bound = x.double().abs() @ w.double().abs().T            # sum_k |x_k||w_k|
        err = ((E.matmul6(x, w) - ref).abs() / bound).max()
        assert float(err) <= 2.0 ** -23, float(err)              # the three dropped cross terms: 2 * 2^-8 * 2^-16 + 2^-32 per product
        # for scale: rounding the OPERANDS to bf16 (one plane) is four orders of magnitude worse
        one = ((x.bfloat16().double() @ w.bfloat16().double().T - ref).abs() / bound).max()
        assert float(one) > 1e3 * float(err)


def test_kept_products_are_exact_in_fp32():
    """8-bit x 8-bit significands: a plane product has at most 16 significant bits, so fp32 holds it exactly -- the matrix core's
    fp32 accumulator sees the true products."""
    x, w = _operands(5, M=32, N=32, K=64)
    for a in E.split3(x):
        for b in E.split3(w):
            p32 = a[:, None, :] * b[None, :, :]
            assert torch.equal(p32.double(), a[:, None, :].double() * b[None, :, :].double())
