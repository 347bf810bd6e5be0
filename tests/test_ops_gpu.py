"""GPU parity of every C-ABI kernel against the CPU oracle and the reference-generated golden fixtures.

Tolerances: integer outputs (RVQ codes) bit-exact; fp32 outputs <= 1e-4 relative to the tensor's max (the HIP
kernels accumulate in a different order than oneDNN/MKL; the north-star budget is 1e-3).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import mimi_oracle as O
from oracle import rvq_ref
from rstnet_amd import ops, synth
from rstnet_amd.codec import functional as RF
from tests.golden import cases

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
TOL = 1e-4


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def nlc(x):  # [B,C,T] cpu -> [B,T,C] gpu
    return x.transpose(1, 2).contiguous().to(DEV)


def ncl(y):  # [B,T,C] gpu -> [B,C,T] cpu
    return y.transpose(1, 2).contiguous().cpu()


@pytest.mark.parametrize("name", list(cases.CONV_CASES))
def test_conv1d(name):
    B, cin, cout, T, K, S = cases.CONV_CASES[name]
    w, b, x = cases.layer_tensors(name, (cout, cin, K), cout, (B, cin, T))
    gold = torch.from_numpy(np.load(os.path.join(G, "layers.npz"))[f"conv.{name}"])
    y = RF.conv1d(nlc(x), RF.pack_conv_weight(w).to(DEV), b.to(DEV), k_eff=K, stride=S)
    assert ncl(y).shape == gold.shape
    assert rel_err(ncl(y), gold) < TOL
    assert rel_err(ncl(y), O.causal_conv1d(x, w, b, stride=S)) < TOL


@pytest.mark.parametrize("dilation,pad_mode", [(2, "constant"), (1, "replicate"), (3, "replicate")])
def test_conv1d_dilation_and_replicate(dilation, pad_mode):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 37, generator=g)
    w = torch.randn(24, 16, 3, generator=g) * 0.2
    y = RF.conv1d(nlc(x), RF.pack_conv_weight(w, dilation).to(DEV), None, k_eff=(3 - 1) * dilation + 1, stride=1,
                  pad_mode=ops.PAD_REPLICATE if pad_mode == "replicate" else ops.PAD_ZERO)
    assert rel_err(ncl(y), O.causal_conv1d(x, w, None, dilation=dilation, pad_mode=pad_mode)) < TOL


def test_conv1d_downsample_replicate():
    """ConvDownsample1d: dense 512->512 k4 s2, replicate padding, no bias."""
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 512, 25, generator=g)
    w = synth._xavier(g, 512, 512, 4)
    y = RF.conv1d(nlc(x), RF.pack_conv_weight(w).to(DEV), None, k_eff=4, stride=2, pad_mode=ops.PAD_REPLICATE)
    ref = O.causal_conv1d(x, w, None, stride=2, pad_mode="replicate")
    assert ncl(y).shape == ref.shape == (2, 512, 13)
    assert rel_err(ncl(y), ref) < TOL


@pytest.mark.parametrize("name", list(cases.CONVTR_CASES))
def test_convtr1d(name):
    B, cin, cout, T, K, S = cases.CONVTR_CASES[name]
    w, b, x = cases.layer_tensors(name, (cin, cout, K), cout, (B, cin, T))
    gold = torch.from_numpy(np.load(os.path.join(G, "layers.npz"))[f"convtr.{name}"])
    y = RF.convtr1d(nlc(x), RF.pack_convtr_weight(w, S).to(DEV), b.repeat(S).to(DEV), kernel=K, stride=S)
    assert ncl(y).shape == gold.shape
    assert rel_err(ncl(y), gold) < TOL


@pytest.mark.parametrize("name", list(cases.RESBLOCK_CASES))
def test_resblock(name):
    B, dim, T = cases.RESBLOCK_CASES[name]
    w1, b1, x = cases.layer_tensors(name + ".1", (dim // 2, dim, 3), dim // 2, (B, dim, T))
    w2, b2, _ = cases.layer_tensors(name + ".3", (dim, dim // 2, 1), dim, (1, 1, 1))
    gold = torch.from_numpy(np.load(os.path.join(G, "layers.npz"))[f"resblock.{name}"])
    xg = nlc(x)
    h = RF.conv1d(xg, RF.pack_conv_weight(w1).to(DEV), b1.to(DEV), k_eff=3, act_in=ops.ACT_ELU)
    y = RF.conv1d(h, RF.pack_conv_weight(w2).to(DEV), b2.to(DEV), k_eff=1, act_in=ops.ACT_ELU, res=xg)
    assert rel_err(ncl(y), gold) < TOL


# (64, 5, 20011): 785 tiles -- more than the resident workgroups of the tile-streaming C = 64 kernel, so every one walks several
@pytest.mark.parametrize("C,B,T", [(64, 2, 300), (64, 1, 128), (64, 3, 129), (64, 5, 20011), (128, 2, 200), (128, 1, 63)])
def test_fused_resblock(C, B, T):
    """rst_seanet_resblock_f32 (plain) == the oracle's resnet_block."""
    assert ops.resblock_supported(C, C // 2, 3)
    name = f"fused{C}_{T}"
    w1, b1, x = cases.layer_tensors(name + ".1", (C // 2, C, 3), C // 2, (B, C, T))
    w2, b2, _ = cases.layer_tensors(name + ".3", (C, C // 2, 1), C, (1, 1, 1))
    x = x * 2 - 1
    sd = {"p.block.1.conv.conv.weight": w1, "p.block.1.conv.conv.bias": b1,
          "p.block.3.conv.conv.weight": w2, "p.block.3.conv.conv.bias": b2}
    y = ops.seanet_resblock(nlc(x), RF.pack_conv_weight(w1).to(DEV), b1.to(DEV), RF.pack_conv_weight(w2).to(DEV), b2.to(DEV), Kw=3)
    assert rel_err(ncl(y), O.resnet_block(sd, "p", x)) < TOL


@pytest.mark.parametrize("B,T", [(2, 500), (1, 126), (1, 127), (3, 253), (5, 20011)])
def test_fused_resblock_with_first_and_last_conv(B, T):
    """pre: encoder.model.0 (Conv1d 1->64 k7) folded in;  post: ELU + decoder.model.14 (Conv1d 64->1 k3) folded in."""
    C = 64
    name = f"fusedpp_{T}"
    w1, b1, _ = cases.layer_tensors(name + ".1", (C // 2, C, 3), C // 2, (1, 1, 1))
    w2, b2, _ = cases.layer_tensors(name + ".3", (C, C // 2, 1), C, (1, 1, 1))
    w0, b0, a = cases.layer_tensors(name + ".0", (C, 1, 7), C, (B, 1, T))
    wf, bf, xin = cases.layer_tensors(name + ".f", (1, C, 3), 1, (B, C, T))
    a, xin = a * 2 - 1, xin * 2 - 1
    sd = {"p.block.1.conv.conv.weight": w1, "p.block.1.conv.conv.bias": b1,
          "p.block.3.conv.conv.weight": w2, "p.block.3.conv.conv.bias": b2}
    args = (RF.pack_conv_weight(w1).to(DEV), b1.to(DEV), RF.pack_conv_weight(w2).to(DEV), b2.to(DEV))
    # pre
    ref = O.resnet_block(sd, "p", O.causal_conv1d(a, w0, b0))
    y = ops.seanet_resblock(a.view(B, T, 1).to(DEV), *args, Kw=3, pre=(w0[:, 0].contiguous().to(DEV), b0.to(DEV)))
    assert rel_err(ncl(y), ref) < TOL
    # post
    ref = O.causal_conv1d(F.elu(O.resnet_block(sd, "p", xin)), wf, bf)
    y = ops.seanet_resblock(nlc(xin), *args, Kw=3, post=(wf[0].t().contiguous().to(DEV), bf.to(DEV)))
    assert y.shape == (B, T, 1)
    assert rel_err(y.view(B, 1, T), ref) < TOL
    # both
    ref = O.causal_conv1d(F.elu(O.resnet_block(sd, "p", O.causal_conv1d(a, w0, b0))), wf, bf)
    y = ops.seanet_resblock(a.view(B, T, 1).to(DEV), *args, Kw=3, pre=(w0[:, 0].contiguous().to(DEV), b0.to(DEV)),
                            post=(wf[0].t().contiguous().to(DEV), bf.to(DEV)))
    assert rel_err(y.view(B, 1, T), ref) < TOL


@pytest.mark.parametrize("chunk", [1, 3, 7])
@pytest.mark.parametrize("K,S", [(7, 1), (3, 1), (8, 4), (10, 5), (4, 2)])
def test_conv1d_streaming_equals_batch(K, S, chunk):
    """Mirror of conv_test.py:85-109 / streaming.py:306-358: chunked calls with a history buffer == one call."""
    g = torch.Generator().manual_seed(K * 10 + S)
    B, cin, cout, T = 2, 8, 12, 41
    x = torch.rand(B, cin, T, generator=g)
    w = synth._xavier(g, cout, cin, K)
    b = 0.1 * torch.randn(cout, generator=g)
    wp, bd = RF.pack_conv_weight(w).to(DEV), b.to(DEV)
    full = O.causal_conv1d(x, w, b, stride=S)
    xg = nlc(x)
    hist = torch.zeros(B, K - S, cin, device=DEV)
    outs = []
    for s in range(0, T, chunk):
        xc = xg[:, s:s + chunk].contiguous()
        y = RF.conv1d(xc, wp, bd, k_eff=K, stride=S, hist=hist)
        consumed = y.shape[1] * S
        hist = ops.hist_update(xc, hist, hist.shape[1] + xc.shape[1] - consumed)
        outs.append(y)
    y = torch.cat(outs, 1)
    n = y.shape[1]
    assert n == T // S  # floor-mode frame count
    assert rel_err(ncl(y), full[..., :n]) < TOL


@pytest.mark.parametrize("chunk", [1, 2, 5])
@pytest.mark.parametrize("K,S", [(16, 8), (10, 5), (7, 2), (4, 3), (6, 1)])
def test_convtr1d_streaming_equals_batch(K, S, chunk):
    g = torch.Generator().manual_seed(K * 10 + S)
    B, cin, cout, T = 2, 8, 6, 11
    x = torch.rand(B, cin, T, generator=g)
    w = synth._xavier(g, cin, cout, K)
    b = 0.1 * torch.randn(cout, generator=g)
    wp, bt = RF.pack_convtr_weight(w, S).to(DEV), b.repeat(S).to(DEV)
    full = O.causal_convtr1d(x, w, b, stride=S)
    q = -(-K // S)
    xg = nlc(x)
    hist = torch.zeros(B, q - 1, cin, device=DEV)
    outs = []
    for s in range(0, T, chunk):
        xc = xg[:, s:s + chunk].contiguous()
        outs.append(RF.convtr1d(xc, wp, bt, kernel=K, stride=S, hist=hist))
        hist = ops.hist_update(xc, hist, q - 1)
    assert rel_err(ncl(torch.cat(outs, 1)), full) < TOL


@pytest.mark.parametrize("rows,D", [(5, 512), (1000, 512), (7, 48), (3, 30)])
def test_layernorm(rows, D):
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, D, generator=g) * 3 + 1
    gamma, beta = torch.randn(D, generator=g), torch.randn(D, generator=g)
    y = ops.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5)
    assert rel_err(y, F.layer_norm(x, (D,), gamma, beta, 1e-5)) < 1e-5


@pytest.mark.parametrize("B,M,K,N,act", [(1, 300, 512, 1536, 0), (1, 77, 512, 2048, 1), (1, 1, 2048, 512, 0), (1, 130, 36, 20, 0)])
def test_linear_epilogues(B, M, K, N, act):
    g = torch.Generator().manual_seed(M)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    res, scale = torch.randn(M, N, generator=g), torch.rand(N, generator=g)
    ref = F.linear(x, w)
    if act:
        ref = F.gelu(ref)
    y = ops.linear(x.to(DEV), w.to(DEV), act_out=act)
    assert rel_err(y, ref) < TOL
    y2 = ops.linear(x.to(DEV), w.to(DEV), res=res.to(DEV), scale=scale.to(DEV), act_out=act)
    assert rel_err(y2, res + scale * ref) < TOL


@pytest.mark.parametrize("T,context,H,D", [(250, 250, 8, 64), (300, 250, 8, 64), (37, None, 2, 64), (100, 10, 4, 32), (65, 33, 2, 128)])
def test_rope_attention_batch(T, context, H, D):
    g = torch.Generator().manual_seed(T)
    B = 2
    qkv = torch.randn(B, T, 3 * H * D, generator=g)
    q, k, v = qkv.view(B, T, 3, H, D).permute(2, 0, 3, 1, 4)
    qr, kr = O.rope_interleaved(q.contiguous(), k.contiguous(), 0, 10000.0)
    ref = F.scaled_dot_product_attention(qr, kr, v, O.attention_mask(T, context)).permute(0, 2, 1, 3).reshape(B, T, H * D)
    qg, kg, vg = ops.rope_split(qkv.to(DEV), H)
    assert rel_err(qg, qr) < 1e-5 and rel_err(kg, kr) < 1e-5 and torch.equal(vg.cpu(), v.contiguous())
    out = ops.attention(qg, kg, vg, context=context)
    assert rel_err(out, ref) < TOL


@pytest.mark.parametrize("T,context,H,D,rope", [(250, 250, 8, 64, True), (77, 20, 2, 32, True), (300, 250, 4, 128, True), (130, None, 3, 64, False),
                                               (1, 250, 8, 64, True), (513, 100, 2, 64, True)])
def test_attention_fused_qkv_reads_the_projection_in_place(T, context, H, D, rope):
    """rst_attention_qkv_f32 (round 5): the whole-utterance pass of StreamingMultiheadAttention (modules/transformer.py:376-423) with
    q / k / v read in place from the in-projection's [B, T, 3*H*D] output and rotated on load, against the oracle's
    rope + scaled_dot_product_attention and against the two-launch form (rope_split + attention)."""
    g = torch.Generator().manual_seed(T + D)
    B = 3
    qkv = torch.randn(B, T, 3 * H * D, generator=g)
    q, k, v = qkv.view(B, T, 3, H, D).permute(2, 0, 3, 1, 4)
    qr, kr = (O.rope_interleaved(q.contiguous(), k.contiguous(), 0, 10000.0) if rope else (q.contiguous(), k.contiguous()))
    ref = F.scaled_dot_product_attention(qr, kr, v, O.attention_mask(T, context)).permute(0, 2, 1, 3).reshape(B, T, H * D)
    out = ops.attention_qkv(qkv.to(DEV), H, rope=rope, context=context)
    assert rel_err(out, ref) < TOL
    two = ops.attention(*ops.rope_split(qkv.to(DEV), H, rope=rope), context=context)
    assert rel_err(out, two) < 2e-6
    tab = ops.rope_table(T, D, 10000.0, DEV)                          # the table holds what rope_split evaluates per element
    ang = torch.exp(torch.arange(D // 2, dtype=torch.float32) * ops.rope_coef(10000.0, D))[None] * torch.arange(T, dtype=torch.float32)[:, None]
    # (an ulp of the frequency times positions up to T: ~T * 1e-7 on the angle)
    assert rel_err(tab[:, 0::2], torch.cos(ang)) < 2e-4 and rel_err(tab[:, 1::2], torch.sin(ang)) < 2e-4


def test_rvq_search_bit_exact_vs_c_oracle_and_reference():
    sd = synth.mimi_state_dict(cases.MIMI_SEED)
    cfg = O.MimiConfig()
    z = cases.rvq_latent(sd)  # [4,512,250]
    B, _, Fr = z.shape
    gold = torch.from_numpy(np.load(os.path.join(G, "rvq.npz"))["codes"]).long()
    xf = F.conv1d(z, sd["quantizer.rvq_first.input_proj.weight"]).transpose(1, 2).reshape(-1, 256)
    xr = F.conv1d(z, sd["quantizer.rvq_rest.input_proj.weight"]).transpose(1, 2).reshape(-1, 256)
    emb = torch.stack([O.codebook(sd, "quantizer.rvq_first.vq.layers.0")] +
                      [O.codebook(sd, f"quantizer.rvq_rest.vq.layers.{j}") for j in range(7)])
    x = torch.cat([xf, xr], 1).contiguous().to(DEV)
    embg = emb.to(DEV)
    packed, e2 = ops.rvq_pack(embg)
    codes, dist = ops.rvq_search(x, embg, packed, e2, B, Fr, [(0, 1), (1, 7)], return_dist=True)
    torch.cuda.synchronize()
    c_first, d_first = rvq_ref.rvq_search(xf.numpy(), emb[:1].numpy())
    c_rest, d_rest = rvq_ref.rvq_search(xr.numpy(), emb[1:].numpy())
    c_ref = torch.from_numpy(np.concatenate([c_first, c_rest])).view(8, B, Fr).transpose(0, 1)
    d_ref = torch.from_numpy(np.concatenate([d_first, d_rest]))
    assert torch.equal(codes.cpu(), c_ref), "HIP codes differ from the C oracle"
    assert torch.equal(dist.cpu(), d_ref), "HIP winning scores are not bit-identical to the C oracle's fmaf chain"
    assert torch.equal(codes.cpu(), gold), "HIP codes differ from the reference-generated fixture"
    # decode side
    zq = ops.rvq_gather(codes, embg, [(0, 1), (1, 7)])
    ref_first = O.rvq_levels_decode(sd, "quantizer.rvq_first", gold[:, :1].transpose(0, 1)).transpose(1, 2).reshape(-1, 256)
    ref_rest = O.rvq_levels_decode(sd, "quantizer.rvq_rest", gold[:, 1:].transpose(0, 1)).transpose(1, 2).reshape(-1, 256)
    assert torch.equal(zq.cpu(), torch.cat([ref_first, ref_rest], 1))


@pytest.mark.parametrize("M,F_", [(1, 1), (33, 33), (64, 32), (95, 19)])
def test_rvq_search_ragged_sizes(M, F_):
    g = torch.Generator().manual_seed(M)
    emb = torch.randn(3, 64, 16, generator=g)
    x = torch.randn(M, 16, generator=g)
    embg = emb.to(DEV)
    packed, e2 = ops.rvq_pack(embg)
    codes, dist = ops.rvq_search(x.to(DEV), embg, packed, e2, M // F_, F_, [(0, 3)], return_dist=True)
    c_ref, d_ref = rvq_ref.rvq_search(x.numpy(), emb.numpy())
    assert torch.equal(codes.cpu(), torch.from_numpy(c_ref).view(3, M // F_, F_).transpose(0, 1))
    assert torch.equal(dist.cpu(), torch.from_numpy(d_ref))


@pytest.mark.parametrize("M,F_,n_codes,D,groups", [(1, 1, 2048, 256, [(0, 1), (1, 7)]), (2, 2, 2048, 256, [(0, 1), (1, 7)]), (32, 1, 2048, 256, [(0, 1), (1, 7)]),
                                                   (33, 33, 256, 16, [(0, 3)]), (64, 2, 2048, 256, [(0, 1), (1, 7)]), (5, 5, 128, 32, [(0, 4)])])
def test_rvq_chain_one_launch_for_all_levels(M, F_, n_codes, D, groups, monkeypatch):
    """rst_rvq_search_chain_f32 (round 5): the streaming form with ALL residual levels in one launch (in-kernel hand-offs between the code
    slices' workgroups) against the C oracle and against the launch-per-level form -- codes AND winning scores bit for bit; then the
    repair path: a time-out code planted in the status word makes the finish launch recompute everything alone, same bits, counted."""
    g = torch.Generator().manual_seed(M + n_codes)
    L = sum(n for _, n in groups)
    emb = torch.randn(L, n_codes, D, generator=g)
    x = torch.randn(M, len(groups) * D, generator=g)
    embg = emb.to(DEV)
    packed, e2 = ops.rvq_pack(embg)
    monkeypatch.setattr(ops, "RVQ_CHAIN", True)
    codes, dist = ops.rvq_search(x.to(DEV), embg, packed, e2, M // F_, F_, groups, return_dist=True)
    key = next(k for k in ops._rvq_slots if k[-3:] == (L, M, n_codes) and k[1] != "graph")
    slots, status = ops._rvq_slots[key]
    torch.cuda.synchronize()
    assert status.tolist() == [0, 0, 0, 0] and bool((slots == -1).all())          # nothing timed out; the slots are re-armed
    refs = []
    for gi, (g0, n) in enumerate(groups):
        c_ref, d_ref = rvq_ref.rvq_search(x[:, gi * D:(gi + 1) * D].contiguous().numpy(), emb[g0:g0 + n].numpy())
        refs.append((g0, n, torch.from_numpy(c_ref), torch.from_numpy(d_ref)))
        assert torch.equal(codes[:, g0:g0 + n].cpu(), torch.from_numpy(c_ref).view(n, M // F_, F_).transpose(0, 1))
        assert torch.equal(dist[g0:g0 + n].cpu(), torch.from_numpy(d_ref))
    monkeypatch.setattr(ops, "RVQ_CHAIN", False)
    codes2, dist2 = ops.rvq_search(x.to(DEV), embg, packed, e2, M // F_, F_, groups, return_dist=True)
    assert torch.equal(codes, codes2) and torch.equal(dist, dist2)
    # a second call reuses the re-armed slots; then the planted time-out
    monkeypatch.setattr(ops, "RVQ_CHAIN", True)
    codes3 = ops.rvq_search(x.to(DEV), embg, packed, e2, M // F_, F_, groups)
    assert torch.equal(codes3, codes)
    status[0] = 1
    codes4, dist4 = ops.rvq_search(x.to(DEV), embg, packed, e2, M // F_, F_, groups, return_dist=True)
    torch.cuda.synchronize()
    assert torch.equal(codes4, codes) and torch.equal(dist4, dist)
    assert status.tolist() == [0, 1, 1, 0] and bool((slots == -1).all())
    status.zero_()


@pytest.mark.parametrize("M,K,N", [(6, 512, 1536), (33, 512, 2048), (64, 512, 1536), (128, 256, 512), (5, 520, 512)])
def test_few_row_linear_applies_layernorm_while_packing(M, K, N, monkeypatch):
    """rst_skinny_f32_pack_ln (round 5): nn.LayerNorm in front of a few-row linear (modules/transformer.py:595-650 for a streamed frame of
    several streams) applied by the packing launch -- bit-identical to LayerNorm followed by the pack, and equal to torch's
    layer_norm + linear within the GEMM's tolerance."""
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g) * 2 + 0.3
    w = torch.randn(N, K, generator=g) / K ** 0.5
    gamma, beta = 1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    ref = F.linear(F.layer_norm(x, (K,), gamma, beta, 1e-5), w)
    ln = (gamma.to(DEV), beta.to(DEV), 1e-5)
    monkeypatch.setattr(ops, "SKINNY_F32_ROWS", False)       # (the packed-operand route: round 6 reads plain rows inside the GEMM, below)
    monkeypatch.setattr(ops, "SKINNY_F32_PACK_LN", True)
    ops.PROFILE = []
    y = ops.linear(x.to(DEV), w.to(DEV), ln=ln)
    names, ops.PROFILE = [r[0] for r in ops.PROFILE], None
    assert names == ["gemm_skinny_f32"]
    monkeypatch.setattr(ops, "SKINNY_F32_PACK_LN", False)
    y2 = ops.linear(x.to(DEV), w.to(DEV), ln=ln)
    assert torch.equal(y, y2)
    assert rel_err(y, ref) < TOL


@pytest.mark.parametrize("M", [5, 32, 64, 70, 128])
def test_few_row_linears_chain_through_the_packed_operand(M, monkeypatch):
    """linear1 -> GELU -> linear2 of a streamed transformer layer on the few-row route (round 5): linear1 writes its result in the
    packed operand order of linear2 (`out_packed`, `ops.PackedRows`), pad rows of the last batch tile as zeros -- the same bits as the
    row-major result followed by the packing launch, with and without the LayerNorm in front."""
    g = torch.Generator().manual_seed(M)
    K, Hd = 512, 2048
    x = torch.randn(M, K, generator=g)
    w1, w2 = torch.randn(Hd, K, generator=g) / K ** 0.5, torch.randn(K, Hd, generator=g) / Hd ** 0.5
    gamma, beta, scale = 1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g), 0.3 * torch.randn(K, generator=g)
    xg, w1g, w2g, sg = x.to(DEV), w1.to(DEV), w2.to(DEV), scale.to(DEV)
    ln = (gamma.to(DEV), beta.to(DEV), 1e-5)
    assert ops.linear_chains(xg, w1g, w2g)
    h = ops.linear(xg, w1g, act_out=ops.ACT_GELU, ln=ln, out_packed=True)
    assert isinstance(h, ops.PackedRows) and h.shape == (M, Hd)
    y = ops.linear(h, w2g, res=xg, scale=sg)
    y2 = ops.linear(ops.linear(xg, w1g, act_out=ops.ACT_GELU, ln=ln), w2g, res=xg, scale=sg)
    assert torch.equal(y, y2)
    ref = x + scale * F.linear(F.gelu(F.linear(F.layer_norm(x, (K,), gamma, beta, 1e-5), w1)), w2)
    assert rel_err(y, ref) < TOL
    monkeypatch.setattr(ops, "SKINNY_F32_CHAIN", False)
    assert not ops.linear_chains(xg, w1g, w2g)


@pytest.mark.parametrize("M", [5, 33, 64, 100, 128])
@pytest.mark.parametrize("N,K", [(1536, 512), (512, 512), (2048, 512), (512, 2048), (96, 1280)])
def test_few_row_linear_reads_its_rows_in_place(M, N, K, monkeypatch):
    """rst_linear_few_rows_f32 (round 6): a plain few-row linear reads its rows row-major inside the GEMM -- with the residual / LayerScale /
    GELU epilogue and the packed output for the next linear -- and gives the bits of the packing launch followed by the GEMM on the packed
    operand (modules/transformer.py:395-423,540-569 at more than two streams per step); a LayerNorm in front keeps its packing launch."""
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * 2 + 0.3).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    res, scale = torch.randn(M, N, generator=g).to(DEV), torch.rand(N, generator=g).to(DEV)
    ln = ((1 + 0.1 * torch.randn(K, generator=g)).to(DEV), (0.1 * torch.randn(K, generator=g)).to(DEV), 1e-5)
    cases = [dict(), dict(ln=ln), dict(res=res, scale=scale), dict(ln=ln, act_out=ops.ACT_GELU), dict(ln=ln, res=res, scale=scale, act_out=ops.ACT_GELU)]
    got, want = [], []
    for rows, dst in ((True, got), (False, want)):
        monkeypatch.setattr(ops, "SKINNY_F32_ROWS", rows)
        for kw in cases:
            for _ in range(2):         # twice: the split-K arrival counters re-arm
                y = ops.linear(x, w, **kw)
            dst.append(y)
        if N % 8 == 0 and M > 4:
            h = ops.linear(x, w, ln=ln, act_out=ops.ACT_GELU, out_packed=True)
            assert isinstance(h, ops.PackedRows)
            dst.append(h.xp)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    xc, wc = x.cpu(), w.cpu()
    ref = res.cpu() + scale.cpu() * F.gelu(F.linear(F.layer_norm(xc, (K,), ln[0].cpu(), ln[1].cpu(), 1e-5), wc))
    assert rel_err(got[4], ref) < TOL
    assert rel_err(got[0], F.linear(xc, wc)) < TOL


@pytest.mark.parametrize("T,D,H,cap,context", [(2, 64, 8, 250, 250), (1, 64, 8, 250, 250), (4, 32, 4, 64, 40), (3, 128, 2, 300, None), (2, 64, 8, 250, 100),
                                              (2, 64, 2, 3000, 750)])      # (the last: 62 KB of scores in LDS, two batches of ring slots per class)
def test_attention_step_is_rope_split_plus_ring_attention(T, D, H, cap, context, monkeypatch):
    """rst_attention_step_f32 (round 6): split of the in-projection's output, interleaved RoPE, ring append and the T new queries against the
    ring in ONE launch vs rst_rope_split_f32 + rst_attn_decode_multi_f32 -- same ring contents, same outputs (to the summation order of the
    softmax), from an empty ring, across the wrap and deep into steady state (modules/transformer.py:376-416, RingKVCache.complete)."""
    B = 5
    g = torch.Generator().manual_seed(T * 100 + D)
    E = H * D
    for pos0 in (0, cap - 2 * T - 1, 7 * cap + 3):
        k1, v1 = (torch.randn(B, H, cap, D, generator=g).to(DEV) for _ in range(2))
        k2, v2 = k1.clone(), v1.clone()
        if pos0 == 0:
            k1.fill_(float("nan")); v1.fill_(float("nan"))      # an empty ring is never read by the one-launch step
            k2.zero_(); v2.zero_()
        pos1, pos2 = (torch.tensor([pos0], device=DEV, dtype=torch.long) for _ in range(2))
        off = pos0
        for step in range(6):
            qkv = torch.randn(B, T, 3 * E, generator=g).to(DEV)
            assert ops.attention_step_supported(qkv, H, cap)
            a = ops.attention_step(qkv, H, k1, v1, pos1, context=context, rope=True, max_period=10000.0)
            q, kk, vv = ops.rope_split(qkv, H, k=k2, v=v2, pos0=off, pos_dev=pos2, ring=True, rope=True, max_period=10000.0)
            b = ops.attention(q, kk, vv, pos0=off, pos_dev=pos2, ring=True, context=context)
            pos1.add_(T); pos2.add_(T); off += T
            used = min(cap, off)
            if pos0 == 0:
                sl = torch.arange(used, device=DEV)
                assert torch.allclose(k1[:, :, sl], k2[:, :, sl], rtol=0, atol=1e-6) and torch.equal(v1[:, :, sl], v2[:, :, sl])
            else:
                assert torch.allclose(k1, k2, rtol=0, atol=1e-6) and torch.equal(v1, v2)
            assert torch.isfinite(a).all()
            assert rel_err(a, b.cpu()) < 2e-6, (pos0, step)
    # the same step written as the out-projection's packed operand (few-row route): the GEMM on it gives the bits of pack + GEMM
    if (B * T) > 4 and E % 8 == 0 and ops._few_rows(B * T, E, E):
        w = torch.randn(E, E, generator=g).to(DEV) / E ** 0.5
        k3, v3, pos3 = k1.clone(), v1.clone(), pos1.clone()
        a_rows = ops.attention_step(qkv, H, k1, v1, pos1, context=context)
        a_pack = ops.attention_step(qkv, H, k3, v3, pos3, context=context, out_packed=True)
        assert isinstance(a_pack, ops.PackedRows) and a_pack.shape == (B, T, E) and a_pack.xp.shape[0] % 32 == 0
        monkeypatch.setattr(ops, "SKINNY_F32_ROWS", False)
        assert torch.equal(ops.linear(a_pack, w), ops.linear(a_rows, w))
    monkeypatch.setattr(ops, "ATTENTION_STEP", False)
    assert not ops.attention_step_supported(qkv, H, cap)


def test_rvq_tie_takes_lowest_index():
    emb = torch.randn(1, 64, 16)
    emb[0, 40] = emb[0, 7]  # exact duplicate rows -> exact tie
    x = emb[0, 40:41].clone()
    embg = emb.to(DEV)
    packed, e2 = ops.rvq_pack(embg)
    codes = ops.rvq_search(x.to(DEV), embg, packed, e2, 1, 1, [(0, 1)])
    assert codes.item() == 7


def test_convtr_depthwise_upsample():
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 512, 13, generator=g)
    w = synth._xavier(g, 512, 1, 4)
    y = ops.convtr_depthwise(nlc(x), w[:, 0].contiguous().to(DEV), 2)
    assert rel_err(ncl(y), O.causal_convtr1d(x, w, None, stride=2, groups=512)) < 1e-6


def test_transpose_and_hist_update():
    x = torch.randn(3, 37, 70)
    assert torch.equal(ops.transpose12(x.to(DEV)).cpu(), x.transpose(1, 2).contiguous())
    h = torch.randn(3, 5, 70)
    for P_out in (0, 3, 5, 20, 42):
        ref = torch.cat([h, x], 1)[:, 42 - P_out:]
        assert torch.equal(ops.hist_update(x.to(DEV), h.to(DEV), P_out).cpu(), ref)


def test_empty_inputs():
    w = torch.randn(8, 4 * 3).to(DEV)
    y = RF.conv1d(torch.zeros(2, 0, 4, device=DEV), w, None, k_eff=3)
    assert y.shape == (2, 0, 8)
    assert ops.linear(torch.zeros(0, 12, device=DEV), w).shape == (0, 8)


# ---- the gemm_win instance that carries the headline step (VERDICT r1 "what's weak" #2): launches with >= 768 tiles of
# 128 x 128 take 16-wide k-chunks (gemm_win.hip: launch_cfg<2, 2, 2, 2, 16>), which none of the small cases above reaches

def _tiles(M, N):
    return -(-M // 128) * -(-N // 128)


def test_gemm_win_kb16_strided_conv():
    """SEANet encoder shape 64 -> 128, k8 s4 at B * T_out = 98 307 rows (769 tiles): interior tiles, utterance-boundary tiles
    (3 utterances whose row counts are not multiples of 128) and the ceil-mode right padding."""
    g = torch.Generator().manual_seed(31)
    B, cin, cout, K, S, T = 3, 64, 128, 8, 4, 131073
    x = torch.rand(B, cin, T, generator=g) * 2 - 1
    w = synth._xavier(g, cout, cin, K)
    b = 0.1 * torch.randn(cout, generator=g)
    y = RF.conv1d(nlc(x), RF.pack_conv_weight(w).to(DEV), b.to(DEV), k_eff=K, stride=S)
    assert _tiles(y.shape[0] * y.shape[1], cout) >= 768
    ref = O.causal_conv1d(x, w, b, stride=S)
    assert ncl(y).shape == ref.shape and rel_err(ncl(y), ref) < TOL


def test_gemm_win_kb16_elu_on_load_with_residual():
    """k3 s1 32 -> 256 with the ELU applied on operand load and a fused residual + ELU-out epilogue (the res-block form),
    49 200 rows x 256 columns = 770 tiles."""
    g = torch.Generator().manual_seed(32)
    B, cin, cout, T = 2, 32, 256, 24600
    x = torch.rand(B, cin, T, generator=g) * 4 - 2
    w = synth._xavier(g, cout, cin, 3)
    b = 0.1 * torch.randn(cout, generator=g)
    res = torch.randn(B, cout, T, generator=g)
    y = RF.conv1d(nlc(x), RF.pack_conv_weight(w).to(DEV), b.to(DEV), k_eff=3, act_in=ops.ACT_ELU, res=nlc(res), act_out=ops.ACT_ELU_OUT)
    assert _tiles(B * T, cout) >= 768
    ref = F.elu(res + O.causal_conv1d(F.elu(x), w, b))
    assert rel_err(ncl(y), ref) < TOL


def test_gemm_win_kb16_convtr():
    """SEANet decoder shape 128 -> 64, k8 s4 (GEMM N = S * Cout = 256, window of q = 2 input steps), 49 160 input steps = 770 tiles."""
    g = torch.Generator().manual_seed(33)
    B, cin, cout, K, S, T = 2, 128, 64, 8, 4, 24580
    x = torch.rand(B, cin, T, generator=g) * 2 - 1
    w = synth._xavier(g, cin, cout, K)
    b = 0.1 * torch.randn(cout, generator=g)
    y = RF.convtr1d(nlc(x), RF.pack_convtr_weight(w, S).to(DEV), b.repeat(S).to(DEV), kernel=K, stride=S)
    assert _tiles(B * T, S * cout) >= 768
    ref = O.causal_convtr1d(x, w, b, stride=S)
    assert ncl(y).shape == ref.shape and rel_err(ncl(y), ref) < TOL


def test_gemm_win_kb16_linear_epilogues():
    """Linear 12 295 x 512 -> 1024 (776 tiles) with GELU, LayerScale and residual in the epilogue (the transformer FFN form)."""
    g = torch.Generator().manual_seed(34)
    M, K, N = 12295, 512, 1024
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    res, scale = torch.randn(M, N, generator=g), torch.rand(N, generator=g)
    assert _tiles(M, N) >= 768
    ref = F.linear(x, w)
    assert rel_err(ops.linear(x.to(DEV), w.to(DEV)), ref) < TOL
    y = ops.linear(x.to(DEV), w.to(DEV), res=res.to(DEV), scale=scale.to(DEV), act_out=ops.ACT_GELU)
    assert rel_err(y, res + scale * F.gelu(ref)) < TOL


# ---- the tile-streaming form of the 128 x 128 configuration (gemm_win_stream_kernel): resident workgroups walk several tiles each,
# interior tiles take the streamed K loop, edge tiles the general routine -- every way the two can alternate inside one workgroup

@pytest.mark.parametrize("M,N,K", [
    (16640, 512, 96),      # 520 tiles of 32-wide k-chunks: more than the resident workgroups (2 per CU), 3 k-tiles per tile
    (16640, 512, 32),      # one k-tile per tile: no second k-tile to keep in registers -> general routine for every tile
    (16640, 512, 40),      # K not a multiple of the k-chunk -> general routine
    (100000, 200, 64),     # >= 768 tiles of 16-wide chunks; the second column tile is ragged (N = 200): fast / general alternate
    (131073 + 5, 128, 48), # ragged last row tile at the end of a workgroup's run
    (4100, 128, 4096),     # barely past the medium-M split route: 33 tiles, fewer than the resident workgroups
])
def test_gemm_win_stream_linear_shapes(M, N, K):
    g = torch.Generator().manual_seed(M % 1000 + N + K)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    b, res, scale = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.rand(N, generator=g)
    ref = F.linear(x.double(), w.double(), b.double())
    assert rel_err(ops.linear(x.to(DEV), w.to(DEV), b.to(DEV)), ref.float()) < TOL
    y = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), res=res.to(DEV), scale=scale.to(DEV), act_out=ops.ACT_GELU)
    assert rel_err(y, (res.double() + scale.double() * F.gelu(ref)).float()) < TOL


def test_gemm_win_stream_short_utterances():
    """Conv k3 s1 over 40 utterances of 700 steps (T_out = 700: every row tile of 128 straddles or touches an utterance edge somewhere
    in the run, windows reach 2 steps back): interior and edge tiles interleave tile by tile inside each resident workgroup."""
    g = torch.Generator().manual_seed(35)
    B, cin, cout, T = 40, 64, 256, 700
    x = torch.rand(B, cin, T, generator=g) * 2 - 1
    w = synth._xavier(g, cout, cin, 3)
    b = 0.1 * torch.randn(cout, generator=g)
    y = RF.conv1d(nlc(x), RF.pack_conv_weight(w).to(DEV), b.to(DEV), k_eff=3, act_in=ops.ACT_ELU)
    ref = O.causal_conv1d(F.elu(x), w, b)
    assert rel_err(ncl(y), ref) < TOL


def test_gemm_win_deep_split():
    """160 rows x 128 columns x K = 4096: five 32-row tiles, K split 32 ways -- the last arriver reads the partials four splits at a
    time (eight batches), summed in split order."""
    from rstnet_amd import _lib
    M, N, K = 160, 128, 4096
    assert _lib.lib().rst_gemm_win_split_plan(M, N, K) >= 16
    g = torch.Generator().manual_seed(36)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    ref = F.linear(x.double(), w.double()).float()
    for _ in range(2):          # the arrival counters re-arm
        assert rel_err(ops.linear(x.to(DEV), w.to(DEV)), ref) < TOL


# ---- streaming-step shapes: split-K across workgroups (medium-M gemm_win, few-row skinny GEMM)

@pytest.mark.parametrize("B,cin,cout,K,S,T,elu", [(1, 64, 128, 8, 4, 1920, True), (1, 128, 64, 3, 1, 480, True), (2, 64, 32, 3, 1, 1920, True),
                                                  (1, 256, 512, 12, 6, 96, False), (3, 128, 256, 10, 5, 480, True)])
def test_conv_streaming_step_shapes_with_history(B, cin, cout, K, S, T, elu):
    """One 80 ms frame of the SEANet encoder layers (rows = 1920 / 480 / 96 / 16 per stream) with a history buffer in front: the
    medium-M launches split K over workgroups (rst_gemm_win_split_plan > 1), the few-row ones go through the skinny GEMM with its
    own split; both against the oracle convolution of [history ; chunk]."""
    g = torch.Generator().manual_seed(T + cout)
    P = K - S
    full = torch.rand(B, cin, P + T, generator=g) * 2 - 1
    w = synth._xavier(g, cout, cin, K)
    b = 0.1 * torch.randn(cout, generator=g)
    hist, x = full[:, :, :P], full[:, :, P:]
    xin = F.elu(full) if elu else full
    ref = F.conv1d(xin, w, b, stride=S)
    y = RF.conv1d(nlc(x), RF.pack_conv_weight(w).to(DEV), b.to(DEV), k_eff=K, stride=S, hist=nlc(hist), act_in=ops.ACT_ELU if elu else ops.ACT_NONE)
    assert ncl(y).shape == ref.shape and rel_err(ncl(y), ref) < TOL


def test_split_plans_cover_the_streaming_shapes():
    from rstnet_amd import _lib
    L = _lib.lib()
    assert L.rst_gemm_win_split_plan(480, 128, 512) > 1 and L.rst_gemm_win_split_plan(1920, 64, 192) >= 1
    assert L.rst_gemm_win_split_plan(640000, 128, 512) == 1                   # the batch path is untouched
    assert L.rst_skinny_f32_split_plan(2, 1024, 8192) >= 4 and L.rst_skinny_f32_split_plan(16, 512, 3072) >= 4
    assert L.rst_skinny_f32_split_plan(2, 32768, 512) == 1


@pytest.mark.parametrize("M,N,K", [(2, 1024, 8192), (16, 512, 3072), (96, 256, 1280), (1, 512, 2048), (64, 4096, 2048), (33, 100, 1000)])
def test_few_row_linear_split_k(M, N, K):
    """The few-row skinny GEMM with K split across workgroups (the 33 MB 512 -> 1024 k16 layer and friends) vs F.linear, with the
    residual / LayerScale / GELU epilogue; twice in a row (the arrival counters must re-arm)."""
    g = torch.Generator().manual_seed(M + N)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    res, scale = torch.randn(M, N, generator=g), torch.rand(N, generator=g)
    ref = F.linear(x, w)
    for _ in range(2):
        assert rel_err(ops.linear(x.to(DEV), w.to(DEV)), ref) < TOL
        y = ops.linear(x.to(DEV), w.to(DEV), res=res.to(DEV), scale=scale.to(DEV), act_out=ops.ACT_GELU)
        assert rel_err(y, res + scale * F.gelu(ref)) < TOL
