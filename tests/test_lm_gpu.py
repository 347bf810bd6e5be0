"""GPU parity of the RQ-Transformer decode step (csrc/lm_*.hip + rstnet_amd/lm) against the CPU oracle
(oracle/lm_oracle.py) and the reference-generated fixture tests/golden/lm_tiny.npz.

bf16 weights are shared by both sides (the oracle up-casts them), activations are fp32 on both sides, so logits must
agree to fp32 summation-order noise: tolerance 1e-3 relative (north star), observed ~1e-6.  Greedy tokens: exact."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import lm_oracle as L
from rstnet_amd import ops, synth
from rstnet_amd.lm.model import LMGen, LMModel
from tests.golden import cases

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("B,N,K", [(1, 12288, 4096), (1, 4096, 11264), (2, 1000, 1024), (3, 37, 2816), (4, 2048, 1024), (1, 5, 8)])
def test_gemv_bf16_prologues(B, N, K):
    g = torch.Generator().manual_seed(N + K)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    x = torch.randn(B, K, generator=g)
    res = torch.randn(B, N, generator=g)
    alpha = 1 + 0.1 * torch.randn(K, generator=g)
    wf = w.float()
    y = ops.gemv_bf16(x.to(DEV), w.to(DEV))
    assert rel_err(y, x @ wf.t()) < 1e-5
    y = ops.gemv_bf16(x.to(DEV), w.to(DEV), res=res.to(DEV))
    assert rel_err(y, res + x @ wf.t()) < 1e-5
    y = ops.gemv_bf16(x.to(DEV), w.to(DEV), prologue=ops.PROLOGUE_RMSNORM, alpha=alpha.to(DEV), eps=1e-8)
    assert rel_err(y, L.rms_norm(x, alpha) @ wf.t()) < 1e-5
    u = torch.randn(B, 2 * K, generator=g)
    y = ops.gemv_bf16(u.to(DEV), w.to(DEV), prologue=ops.PROLOGUE_SILU_GATE)
    assert rel_err(y, (F.silu(u[:, :K]) * u[:, K:]) @ wf.t()) < 1e-5
    if N % 2 == 0:      # the gate in the producer: rows (q, N/2 + q) per wave
        bias = torch.randn(N, generator=g)
        for pro, xin in ((ops.PROLOGUE_NONE, x), (ops.PROLOGUE_RMSNORM, L.rms_norm(x, alpha))):
            h = xin @ wf.t() + bias
            y = ops.gemv_bf16(x.to(DEV), w.to(DEV), prologue=pro, alpha=alpha.to(DEV), eps=1e-8, bias=bias.to(DEV), gate_out=True)
            assert y.shape == (B, N // 2) and rel_err(y, F.silu(h[:, :N // 2]) * h[:, N // 2:]) < 2e-5


# K >= 8192 (>= 4096 above 32 rows): K is also split across workgroups (rst_skinny_bf16_split_plan > 1), four / two column tiles per
# workgroup; N = 1001: ragged last tile and the 4-byte partial stores (N % 4 != 0)
@pytest.mark.parametrize("B,N,K", [(32, 4096, 4096), (5, 1000, 1024), (17, 37, 2816), (64, 2048, 1024), (33, 12288, 4096), (8, 96, 64),
                                   (32, 1000, 8192), (7, 1001, 8192), (3, 4096, 11264), (48, 200, 4096),
                                   # column tiles per workgroup = ceil(tiles / CUs) without a K split (round 4): 2, 3 (ragged last tile) and 4
                                   (32, 12288, 1024), (32, 22500, 1024), (16, 32000, 256)])
def test_gemm_skinny_bf16(B, N, K):
    """bf16-MFMA skinny GEMM with hi/lo-split activations: fp32-class accuracy against the fp32 oracle product."""
    g = torch.Generator().manual_seed(B + N + K)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    x = torch.randn(B, K, generator=g)
    res = torch.randn(B, N, generator=g)
    wf = w.float()
    ref = x.double() @ wf.double().t()
    y = ops.gemm_skinny(x.to(DEV), w.to(DEV))
    assert rel_err(y, ref) < 5e-5
    y2 = ops.gemm_skinny(x.to(DEV), w.to(DEV))
    assert torch.equal(y, y2), "split-K reduction must be deterministic"
    y = ops.gemm_skinny(x.to(DEV), w.to(DEV), res=res.to(DEV))
    assert rel_err(y, res.double() + ref) < 5e-5
    u = torch.randn(B, 2 * K, generator=g)
    y = ops.gemm_skinny(u.to(DEV), w.to(DEV), prologue=ops.PROLOGUE_SILU_GATE)
    assert rel_err(y, (F.silu(u[:, :K]) * u[:, K:]).double() @ wf.double().t()) < 5e-5


@pytest.mark.parametrize("B,I,K,bias", [(1, 11264, 4096, False), (2, 2816, 1024, True), (1, 24, 16, True), (3, 1408, 512, False),
                                        (32, 11264, 4096, False), (33, 4864, 896, True), (64, 96, 64, True),
                                        (40, 2816, 4096, True)])      # the gated epilogue behind a K split (two batch tiles)
def test_gated_pair_epilogue_fusion(B, I, K, bias):
    """Batch > 2: the first GEMM of the gated MLP applies silu(u) * v in its epilogue and hands the packed hi/lo operand
    to the second (weights packed with the two halves interleaved).  fp32-class accuracy against the fp64 product."""
    g = torch.Generator().manual_seed(B + I + K)
    w_in = (torch.randn(2 * I, K, generator=g) / K ** 0.5).bfloat16()
    w_out = (torch.randn(K, I, generator=g) / I ** 0.5).bfloat16()
    b_in = torch.randn(2 * I, generator=g) if bias else None
    b_out = torch.randn(K, generator=g) if bias else None
    x = torch.randn(B, K, generator=g)
    alpha = 1 + 0.1 * torch.randn(K, generator=g)
    h = L.rms_norm(x, alpha).double() @ w_in.double().t()
    if bias:
        h = h + b_in.double()
    ref = x.double() + (F.silu(h[:, :I]) * h[:, I:]) @ w_out.double().t()
    if bias:
        ref = ref + b_out.double()
    dev = lambda t: None if t is None else t.to(DEV)
    y = ops.lm_gated_pair(x.to(DEV), w_in.to(DEV), w_out.to(DEV), alpha=alpha.to(DEV), eps=1e-8, res=x.to(DEV), bias_in=dev(b_in),
                          bias_out=dev(b_out))
    assert rel_err(y, ref) < 5e-5
    if B > 2:
        packed = ops.gemm_skinny(x.to(DEV), w_in.to(DEV), prologue=ops.PROLOGUE_RMSNORM, alpha=alpha.to(DEV), eps=1e-8, bias=dev(b_in),
                                 gate_out=True)
        assert isinstance(packed, ops.PackedAct) and packed.K == I


def test_lm_batch8_matches_oracle():
    """The B > 4 path (rmsnorm + skinny GEMM) through forward_text / forward_depformer on the tiny config."""
    cfg = dict(synth.LM_TINY)
    sd = synth.lm_state_dict(cfg, cases.LM_SEED)
    model = LMModel.from_state_dict({k: v.to(DEV) for k, v in sd.items()}, cfg)
    ocfg = L.LMConfig(**cfg)
    sdf = {k: v.float() for k, v in sd.items()}
    B = 8
    gt = torch.Generator().manual_seed(6)
    st = L.new_transformer_state(B, ocfg.num_layers, ocfg.num_heads, ocfg.dim // ocfg.num_heads, ocfg.context)
    with model.streaming(B):
        for s in range(12):
            toks = torch.randint(0, cfg["card"], (B, cfg["n_q"] + 1, 1), generator=gt)
            ref_out, ref_logits = L.forward_text(sdf, ocfg, toks, st)
            out, logits = model.forward_text(toks.to(DEV))
            assert rel_err(out, ref_out) < 1e-3 and rel_err(logits, ref_logits) < 1e-3, f"step {s}"
        dst = L.new_transformer_state(B, ocfg.depformer_num_layers, ocfg.depformer_num_heads,
                                      ocfg.depformer_dim // ocfg.depformer_num_heads, ocfg.dep_q)
        model.depformer._streaming_state = model.depformer._init_streaming_state(B)
        prev = torch.randint(0, cfg["text_card"], (B, 1, 1), generator=gt)
        for cb in range(cfg["dep_q"]):
            rl = L.forward_depformer(sdf, ocfg, cb, prev, ref_out, dst)
            gl = model.forward_depformer(cb, prev.to(DEV), out)
            assert rel_err(gl, rl) < 1e-3
            prev = torch.randint(0, cfg["card"], (B, 1, 1), generator=gt)


def test_embed_sum_and_rmsnorm():
    g = torch.Generator().manual_seed(1)
    tabs = [(0.5 * torch.randn(33, 256, generator=g)).bfloat16() for _ in range(5)]
    toks = torch.tensor([[3, -1, 32, 0, 7], [-1, -1, 5, 32, 1]])
    out = ops.embed_sum(toks.to(DEV), [t.to(DEV) for t in tabs], [1, 2, 3, 4, 0])
    ref = None
    for i, col in enumerate([1, 2, 3, 4, 0]):
        e = L.scaled_embedding(tabs[i], toks[:, col])
        ref = e if ref is None else ref + e
    assert torch.equal(out.cpu(), ref)
    # ids outside a table never read out of bounds: >= rows -> the last row, other negative ids -> row 0 (the reference raises)
    bad = torch.tensor([[33, -5, 1000000, 0, 7], [2, 2, 2, 2, 2]])
    out = ops.embed_sum(bad.to(DEV), [t.to(DEV) for t in tabs], [1, 2, 3, 4, 0])
    clamped = torch.tensor([[32, 0, 32, 0, 7], [2, 2, 2, 2, 2]])
    ref = None
    for i, col in enumerate([1, 2, 3, 4, 0]):
        e = L.scaled_embedding(tabs[i], clamped[:, col])
        ref = e if ref is None else ref + e
    assert torch.equal(out.cpu(), ref)
    x = torch.randn(3, 1000, generator=g)
    a = 1 + 0.1 * torch.randn(1000, generator=g)
    assert rel_err(ops.rmsnorm(x.to(DEV), a.to(DEV), 1e-8), L.rms_norm(x, a)) < 1e-6


# (2, 64, 2048, 2048, 600): 16 slot splits, more than 8 of them holding keys from step 512 on (the combine reads them 8 at a time)
@pytest.mark.parametrize("H,D,cap,context,steps,rope", [(4, 64, 8, None, 8, False), (2, 128, 10, 10, 25, True), (32, 128, 300, 300, 40, True),
                                                        (4, 64, 300, 250, 320, True), (16, 64, 8, None, 8, False),
                                                        (2, 64, 2048, 2048, 600, True)])
def test_rope_append_and_ring_attention(H, D, cap, context, steps, rope):
    _ring_attention_case(H, D, cap, context, steps, rope, torch.float32)


@pytest.mark.parametrize("H,D,cap,context,steps", [(32, 128, 300, 300, 40), (4, 64, 300, 250, 320), (8, 128, 3000, 3000, 12),
                                                   (2, 128, 2048, 2048, 600)])
def test_ring_attention_bf16_kv(H, D, cap, context, steps):
    """bf16 KV rings (the reference's cache precision, modules/transformer.py:228): against the oracle ring that rounds what it
    stores to bf16 -- same rounded keys / values on both sides, so the fp32 attention over them must agree like the fp32 ring does;
    the ring contents must be bit-identical bf16."""
    _ring_attention_case(H, D, cap, context, steps, True, torch.bfloat16)


@pytest.mark.parametrize("H,D,cap,context,steps,kv_dtype", [(32, 128, 300, 300, 30, torch.bfloat16), (32, 64, 70, 70, 90, torch.float32),
                                                            (32, 128, 70, 60, 80, torch.bfloat16)])
def test_ring_attention_many_streams(H, D, cap, context, steps, kv_dtype):
    """26 streams x 32 heads = 832 workgroups: more than three per CU, i.e. the launch takes `attn_decode_dense_kernel` (one slot per
    lane group in flight, four workgroups per CU) -- same slot order per lane group, so the same results as the two-slot form."""
    _ring_attention_case(H, D, cap, context, steps, True, kv_dtype, B=26)


def _ring_attention_case(H, D, cap, context, steps, rope, kv_dtype, B=2):
    """Step-by-step against the oracle's RingKV (slot->position map of RingKVCache.complete incl. SURVEY Q1)."""
    g = torch.Generator().manual_seed(H * D)
    ring = L.RingKV(B, H, D, cap, dtype=kv_dtype)
    kc = torch.zeros(B, H, cap, D, device=DEV, dtype=kv_dtype)
    vc = torch.zeros(B, H, cap, D, device=DEV, dtype=kv_dtype)
    pos = torch.zeros(1, dtype=torch.long, device=DEV)
    for s in range(steps):
        qkv = torch.randn(B, 3 * H * D, generator=g)
        q, k, v = qkv.view(B, 1, 3, H, D).permute(2, 0, 3, 1, 4)
        if rope:
            q, k = L.rope_interleaved_1(q, s, 10000.0), L.rope_interleaved_1(k, s, 10000.0)
        keys, vals, pos_k = ring.complete(k, v)
        delta = s - pos_k
        mask = (pos_k >= 0) & (delta >= 0)
        if context is not None:
            mask = mask & (delta < context)
        ref = F.scaled_dot_product_attention(q, keys, vals, mask.view(1, -1)).permute(0, 2, 1, 3).reshape(B, H * D)
        # alternate the two forms of the rotation: computed by the launch, or read from the once-per-step table (cap > 64 only)
        table = ops.lm_rope_table(pos, D) if rope and cap > 64 and s % 2 else None
        out = ops.lm_attn_decode(qkv.to(DEV), kc, vc, pos, rope=rope, context=context, rope_table=table)
        if rope and cap > 64 and s % 7 == 3:        # and bit for bit the same thing (a second ring copy takes the in-launch trig)
            k2, v2 = kc.clone(), vc.clone()
            assert torch.equal(ops.lm_attn_decode(qkv.to(DEV), k2, v2, pos, rope=rope, context=context,
                                                  rope_table=None if table is not None else ops.lm_rope_table(pos, D)), out)
            assert torch.equal(k2, kc) and torch.equal(v2, vc)
        pos.add_(1)
        assert rel_err(out, ref) < (1e-4 if kv_dtype == torch.float32 else 3e-3), f"step {s}"
    if kv_dtype == torch.float32:
        assert rel_err(kc, ring.k) < 1e-4   # rotated keys: fp32 sin/cos of angles up to ~300 rad
    else:
        # values are copied, so their bf16 images must be identical; rotated keys may differ by one bf16 ulp where the fp32
        # rotation (sin / cos of large angles) lands on the other side of a rounding boundary
        assert torch.equal(vc.float().cpu(), ring.v)
        assert rel_err(kc.float(), ring.k) < 1e-2


@pytest.mark.parametrize("V,k", [(2048, 250), (32, 7), (32000, 25), (50, 25), (4000, 250), (151936, 25), (40000, 300), (151936, 1000)])
def test_sampling_matches_oracle(V, k):
    g = torch.Generator().manual_seed(V)
    B = 3
    logits = torch.randn(B, V, generator=g) * 3
    logits[0, 5] = logits[0, 3]  # an exact tie: lowest index first
    noise = torch.empty(B, k).exponential_(1, generator=g)
    assert torch.equal(ops.lm_sample(logits.to(DEV), use_sampling=False, temp=0.8, top_k=k).cpu(), logits.argmax(-1))
    tok = ops.lm_sample(logits.to(DEV), use_sampling=True, temp=0.8, top_k=k, noise=noise.to(DEV))
    ref = L.sample_token(logits, True, 0.8, k, noise)
    assert torch.equal(tok.cpu(), ref)


@pytest.mark.parametrize("V,k", [(2048, 250), (32000, 25), (151936, 25)])
def test_sampling_plateaus_take_lowest_indices(V, k):
    """Rows made of a few distinct values (massive ties): top-k order is (value desc, index asc) like torch.topk's stable
    order on CPU; exercises the tie search and, for the large vocabulary, the candidate-overflow path."""
    g = torch.Generator().manual_seed(V + k)
    B = 2
    logits = torch.randint(0, 3, (B, V), generator=g).float()
    logits[1] = 1.5
    noise = torch.empty(B, k).exponential_(1, generator=g)
    tok = ops.lm_sample(logits.to(DEV), use_sampling=True, temp=0.8, top_k=k, noise=noise.to(DEV))
    probs = torch.softmax(logits / 0.8, -1)
    order = torch.argsort(probs, dim=-1, descending=True, stable=True)[:, :k]
    ref = order.gather(1, (probs.gather(1, order) / noise).argmax(-1, keepdim=True))[:, 0]
    assert torch.equal(tok.cpu(), ref)
    assert torch.equal(ops.lm_sample(logits.to(DEV), use_sampling=False, temp=0.8, top_k=k).cpu(), logits.argmax(-1))


@pytest.mark.parametrize("B,N,K,prologue", [(32, 4096, 4096, 0), (5, 1000, 1024, 1), (17, 96, 2816, 2), (64, 2048, 1024, 0), (1, 64, 32, 0)])
def test_gemm_skinny_fp8_matches_emulation(B, N, K, prologue):
    """fp8 e4m3 path: bit-level agreement of the quantisers with torch.float8_e4m3fn (per-row amax / 448 scales) and of the
    product with the fp32 product of the SAME quantised operands; and a sanity bound against the unquantised product."""
    g = torch.Generator().manual_seed(B + N + K)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    x = torch.randn(B, 2 * K if prologue == 2 else K, generator=g)
    alpha = 1 + 0.1 * torch.randn(K, generator=g)
    res = torch.randn(B, N, generator=g)
    if prologue == 0:
        px = x
    elif prologue == 1:
        px = L.rms_norm(x, alpha)
    else:
        px = F.silu(x[:, :K]) * x[:, K:]

    def q(t):
        sc = t.abs().amax(dim=1, keepdim=True).clamp_min(1e-30) / 448.0
        return (t / sc).to(torch.float8_e4m3fn).float() * sc
    ref_q = q(px).double() @ q(w.float()).double().t() + res.double()
    y = ops.gemm_skinny_fp8(x.to(DEV), w.to(DEV), prologue=prologue, alpha=alpha.to(DEV) if prologue == 1 else None, eps=1e-8,
                            res=res.to(DEV))
    assert rel_err(y, ref_q) < 2e-3
    exact = px.double() @ w.double().t() + res.double()
    assert rel_err(y, exact) < 0.1
    assert torch.equal(y, ops.gemm_skinny_fp8(x.to(DEV), w.to(DEV), prologue=prologue, alpha=alpha.to(DEV) if prologue == 1 else None,
                                              eps=1e-8, res=res.to(DEV)))


@pytest.mark.parametrize("B", [3, 4])
def test_lm_linear_wide_k_small_batch(B):
    """Batch 3-4 with K = 11264 (the 7B ffn-out): B*K floats exceed the GEMV's LDS stage, lm_linear must take the skinny path."""
    g = torch.Generator().manual_seed(B)
    K, N = 11264, 512
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    u = torch.randn(B, 2 * K, generator=g)
    res = torch.randn(B, N, generator=g)
    y = ops.lm_linear(u.to(DEV), w.to(DEV), prologue=ops.PROLOGUE_SILU_GATE, res=res.to(DEV))
    ref = res.double() + (F.silu(u[:, :K]) * u[:, K:]).double() @ w.double().t()
    assert rel_err(y, ref) < 5e-5


def _tiny():
    cfg = dict(synth.LM_TINY)
    sd = synth.lm_state_dict(cfg, cases.LM_SEED)
    model = LMModel.from_state_dict({k: v.to(DEV) for k, v in sd.items()}, cfg)
    return cfg, sd, model


@pytest.mark.parametrize("B,dep_q,n_user,delays", [(2, 3, 2, [0, 1, 0, 2, 0, 1]), (1, 8, 8, [0, 0, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1]),
                                                   (5, 2, 1, [0, 0, 0, 0])])
def test_token_ring_kernels_match_host_bookkeeping(B, dep_q, n_user, delays):
    """rst_lm_ring_begin / commit against the host-side restatement of LMGen.step's cache bookkeeping (models/model.py:506-562)."""
    K, max_delay = len(delays), max(delays)
    CT = max_delay + 2
    g = torch.Generator().manual_seed(B + K)
    ref = torch.full((B, K, CT), -2, dtype=torch.long)
    cache = ref.clone().to(DEV)
    initial = torch.arange(100, 100 + K)
    off_dev = torch.zeros(1, dtype=torch.long, device=DEV)
    d32 = torch.tensor(delays, dtype=torch.int32, device=DEV)
    for offset in range(3 * CT + 1):
        user = torch.randint(0, 50, (B, n_user), generator=g)
        gen = torch.randint(50, 99, (B, dep_q + 1), generator=g)
        for q in range(n_user):                                   # host restatement
            k = dep_q + 1 + q
            ref[:, k, (offset + delays[k]) % CT] = user[:, q]
        position = offset % CT
        for k, d in enumerate(delays):
            if offset <= d:
                ref[:, k, position] = initial[k]
        want_in = ref[:, :, position].clone()
        position = (offset + 1) % CT
        ref[:, :dep_q + 1, position] = gen
        idx = (offset + 1 - max_delay + torch.tensor(delays[:dep_q + 1])) % CT
        want_out = ref[:, :dep_q + 1].gather(2, idx.view(1, -1, 1).expand(B, -1, 1))[:, :, 0]
        got_in = ops.lm_ring_begin(cache, user.to(DEV), initial.to(DEV), d32, off_dev, dep_q + 1)
        got_out = ops.lm_ring_commit(cache, gen.to(DEV), d32, off_dev, max_delay)
        assert torch.equal(got_in.cpu(), want_in), offset
        assert int(off_dev) == offset + 1
        if offset + 1 > max_delay:
            assert torch.equal(got_out.cpu(), want_out), offset
        assert torch.equal(cache.cpu(), ref), offset


@pytest.mark.parametrize("name", list(cases.SAMPLING_CASES))
def test_sampler_matches_reference_fixture(name):
    """rst_lm_sample_f32 against the tokens the REFERENCE samplers returned (tests/golden/sampling.npz, F9): top-k sampling with
    the stored Exp(1) noise, id blanking (2049 / 2048), the 151 936-entry vocabulary; greedy."""
    g = np.load(os.path.join(G, "sampling.npz"))
    fn, B, V, k, temp, seed = cases.SAMPLING_CASES[name]
    limit = {"sample_token": 0, "sample_token_audio": 2049, "sample_token_audio_2048": 2048}[fn]
    lg = cases.sampling_logits(name).view(B, V).to(DEV)
    noise = torch.from_numpy(g[f"{name}.noise"]).to(DEV)
    tok = ops.lm_sample(lg, use_sampling=True, temp=temp, top_k=k, noise=noise, limit=limit)
    assert torch.equal(tok.cpu(), torch.from_numpy(g[f"{name}.tokens"]).long().view(B))
    tok = ops.lm_sample(lg, use_sampling=False, temp=temp, top_k=k)
    assert torch.equal(tok.cpu(), torch.from_numpy(g[f"{name}.greedy"]).long().view(B))


@pytest.mark.parametrize("name", list(cases.SAMPLING_TOP_P_CASES))
def test_top_p_sampler_matches_reference_fixture(name):
    """rst_lm_sample_f32(top_p > 0) against the tokens the REFERENCE's sample_token(top_p=...) returned (sampling.npz, nuclei of
    1 .. ~3000 entries, the 151 936-entry vocabulary) for the full-width noise its multinomial drew, and against the oracle on a row
    with exact ties (sorted lowest id first) and with id blanking."""
    g = np.load(os.path.join(G, "sampling.npz"))
    B, V, top_p, temp, seed, scale = cases.SAMPLING_TOP_P_CASES[name]
    lg = cases.sampling_top_p_logits(name).view(B, V)
    noise = cases.sampling_top_p_noise(name)
    tok = ops.lm_sample(lg.to(DEV), use_sampling=True, temp=temp, top_k=0, noise=noise.to(DEV), top_p=top_p)
    assert torch.equal(tok.cpu(), torch.from_numpy(g[f"top_p.{name}.tokens"]).long().view(B))
    # top_p wins over top_k when both are given (utils/sampling.py:96-99); greedy ignores both
    tok = ops.lm_sample(lg.to(DEV), use_sampling=True, temp=temp, top_k=25, noise=noise.to(DEV), top_p=top_p)
    assert torch.equal(tok.cpu(), torch.from_numpy(g[f"top_p.{name}.tokens"]).long().view(B))
    assert torch.equal(ops.lm_sample(lg.to(DEV), use_sampling=False, temp=temp, top_k=0, top_p=top_p).cpu(), lg.argmax(-1))
    # ties: quantised logits -> plateaus inside the nucleus
    lq = (lg * 2).round() / 2
    ref = L.sample_token(lq, True, temp, 0, noise, top_p=top_p)
    assert torch.equal(ops.lm_sample(lq.to(DEV), use_sampling=True, temp=temp, top_k=0, noise=noise.to(DEV), top_p=top_p).cpu(), ref)
    # id blanking: ids >= limit leave the nucleus, the softmax denominator stays that of the full row (as the top-k path blanks)
    limit = V // 2
    probs = torch.softmax(lq / temp, -1)
    probs[:, limit:] = 0.0
    ps, idx = torch.sort(probs, dim=-1, descending=True, stable=True)
    ps = ps * (~(torch.cumsum(ps, -1) - ps > top_p)).float()
    ref = idx.gather(-1, (ps / noise).argmax(-1, keepdim=True))[:, 0]
    got = ops.lm_sample(lq.to(DEV), use_sampling=True, temp=temp, top_k=0, noise=noise.to(DEV), top_p=top_p, limit=limit)
    assert torch.equal(got.cpu(), ref) and int(got.max()) < limit


@pytest.mark.parametrize("V,k", [(151936, 25), (65536, 50), (40000, 100), (151936, 1)])
def test_two_level_sampler_equals_one_level(V, k):
    """Vocabularies above 32768: the chunked two-level sampler (sample_split_kernel + sample_merge_kernel) against the one-workgroup-per-row
    kernel it replaces and the oracle -- random rows, an exact tie, massive plateaus, id blanking, greedy."""
    g = torch.Generator().manual_seed(V + k)
    B = 5
    logits = torch.randn(B, V, generator=g) * 3
    logits[0, 70000 % V] = logits[0, 3] = logits[0].max() + 1.0                # a tie for the first place across chunks
    logits[1] = torch.randint(0, 3, (V,), generator=g).float()                   # plateaus
    logits[2] = 1.5
    noise = torch.empty(B, k).exponential_(1, generator=g)
    for limit in (0, 30000, 11):
        if limit and limit < k:
            continue
        two = ops.lm_sample(logits.to(DEV), use_sampling=True, temp=0.8, top_k=k, noise=noise.to(DEV), limit=limit)
        one = ops.lm_sample(logits.to(DEV), use_sampling=True, temp=0.8, top_k=k, noise=noise.to(DEV), limit=limit, two_level=False)
        assert torch.equal(two, one), limit
        if not limit:
            probs = torch.softmax(logits / 0.8, -1)
            order = torch.argsort(probs, dim=-1, descending=True, stable=True)[:, :k]
            ref = order.gather(1, (probs.gather(1, order) / noise).argmax(-1, keepdim=True))[:, 0]
            assert torch.equal(two.cpu(), ref)
    assert torch.equal(ops.lm_sample(logits.to(DEV), use_sampling=False, temp=0.8, top_k=k).cpu(), logits.argmax(-1))


def test_lm_state_dict_keys():
    cfg, sd, model = _tiny()
    assert set(model.state_dict().keys()) == set(sd.keys())


@pytest.mark.parametrize("graphs,depth_frame", [(False, True), (True, True), (False, False), (True, False)])
def test_lmgen_greedy_matches_reference_fixture(graphs, depth_frame, monkeypatch):
    """LMGen.step token streams == the imported reference's (fixture); with and without HIP graphs, and with the depth phase as
    the persistent launch (rst_depth_decode_frame, the default at batch <= 2) or as the launch-per-op chain."""
    monkeypatch.setenv("NO_CUDA_GRAPH", "0" if graphs else "1")
    monkeypatch.setenv("RST_DEPTH_FRAME", "1" if depth_frame else "0")
    cfg, sd, model = _tiny()
    gold = torch.from_numpy(np.load(os.path.join(G, "lm_tiny.npz"))["tokens"]).long()
    user = cases.lm_user_tokens(cfg)
    gen = LMGen(model, use_sampling=False)
    outs = []
    with gen.streaming(cases.LM_BATCH):
        for s in range(cases.LM_STEPS):
            o = gen.step(user[s].to(DEV))
            outs.append(torch.full((cases.LM_BATCH, cfg["dep_q"] + 1, 1), -9, dtype=torch.long) if o is None else o.cpu())
    got = torch.cat(outs, -1)
    assert (got[..., 0] == -9).all() and (got[..., 1:] != -9).all()       # None exactly for the first max_delay steps (Q14)
    assert torch.equal(got, gold)
    if depth_frame:
        model.depth_frame_tables().check()


@pytest.mark.parametrize("depth_frame", [True, False])
def test_lmgen_sampling_matches_reference_fixture(depth_frame, monkeypatch):
    """LMGen.step with use_sampling=True against the token streams of the imported reference LMGen under a seeded RNG
    (tests/golden/lm_tiny_sampling.npz); the Exp(1) noise that run drew replaces the per-frame device draw (text draw, then the
    dep_q audio draws: the order of the one noise buffer of LMGen._frame).  Eager mode: a captured graph would freeze the noise."""
    monkeypatch.setenv("NO_CUDA_GRAPH", "1")
    monkeypatch.setenv("RST_DEPTH_FRAME", "1" if depth_frame else "0")
    cfg, sd, model = _tiny()
    g = np.load(os.path.join(G, "lm_tiny_sampling.npz"))
    sp = cases.LM_SAMPLING
    nt, na = torch.from_numpy(g["noise_text"]), torch.from_numpy(g["noise_audio"])      # [steps, B, k_text], [steps, dep_q, B, k]
    user = cases.lm_user_tokens(cfg)
    gen = LMGen(model, use_sampling=True, temp=sp["temp"], temp_text=sp["temp_text"], top_k=sp["top_k"], top_k_text=sp["top_k_text"])
    frame = {"s": 0}

    def noise(B, k):
        s = frame["s"]
        buf = torch.cat([nt[s]] + [na[s, c] for c in range(cfg["dep_q"])], dim=1)
        assert buf.shape == (B, k)
        return buf.to(DEV)
    monkeypatch.setattr(gen, "_noise", noise)
    outs = []
    with gen.streaming(cases.LM_BATCH):
        for s in range(cases.LM_STEPS):
            frame["s"] = s
            o = gen.step(user[s].to(DEV))
            outs.append(torch.full((cases.LM_BATCH, cfg["dep_q"] + 1, 1), -9, dtype=torch.long) if o is None else o.cpu())
    assert torch.equal(torch.cat(outs, -1), torch.from_numpy(g["tokens"]).long())


def test_forward_text_and_depformer_logits_match_oracle():
    cfg, sd, model = _tiny()
    g = np.load(os.path.join(G, "lm_tiny.npz"))
    ocfg = L.LMConfig(**cfg)
    sdf = {k: v.float() for k, v in sd.items()}
    B = cases.LM_BATCH
    gt = torch.Generator().manual_seed(5)
    st = L.new_transformer_state(B, ocfg.num_layers, ocfg.num_heads, ocfg.dim // ocfg.num_heads, ocfg.context)
    with model.streaming(B):
        for s in range(13):     # crosses the ring capacity (10)
            toks = torch.randint(0, cfg["card"], (B, cfg["n_q"] + 1, 1), generator=gt)
            toks[0, 2, 0] = -1  # a zero-embedding token
            ref_out, ref_logits = L.forward_text(sdf, ocfg, toks, st)
            out, logits = model.forward_text(toks.to(DEV))
            assert rel_err(out, ref_out) < 1e-3 and rel_err(logits, ref_logits) < 1e-3, f"step {s}"
            dst = L.new_transformer_state(B, ocfg.depformer_num_layers, ocfg.depformer_num_heads,
                                          ocfg.depformer_dim // ocfg.depformer_num_heads, ocfg.dep_q)
            model.depformer._streaming_state = model.depformer._init_streaming_state(B)
            prev = torch.randint(0, cfg["text_card"], (B, 1, 1), generator=gt)
            for cb in range(cfg["dep_q"]):
                rl = L.forward_depformer(sdf, ocfg, cb, prev, ref_out, dst)
                gl = model.forward_depformer(cb, prev.to(DEV), out)
                assert rel_err(gl, rl) < 1e-3, f"step {s} cb {cb}"
                prev = torch.randint(0, cfg["card"], (B, 1, 1), generator=gt)


def test_lmgen_requires_streaming():
    cfg, sd, model = _tiny()
    gen = LMGen(model, use_sampling=False)
    with pytest.raises(RuntimeError):
        gen.step(torch.zeros(1, cfg["n_q"] - cfg["dep_q"], 1, dtype=torch.long, device=DEV))


# ---- fused GEMV prologues of the depth transformer (rst_gemv_attn_bf16_f32, rst_gemv_embed_bf16_f32)

@pytest.mark.parametrize("B,H,D,cap,context,steps", [(1, 16, 64, 8, None, 8), (2, 16, 64, 8, None, 8), (2, 2, 64, 2, None, 7), (1, 2, 32, 3, None, 10),
                                                     (2, 4, 128, 8, 5, 20), (1, 1, 16, 8, None, 8), (2, 8, 4, 5, None, 12), (1, 32, 32, 8, 8, 17)])
def test_gemv_attn_out_proj_matches_oracle(B, H, D, cap, context, steps):
    """Out-projection with the short-ring attention as its prologue, step by step against the oracle's RingKV (no rope): covers the
    depth transformer's own shape (16 x 64, ring 8), rings that wrap (the `delta <= 0` slot), a context shorter than
    the ring, odd head counts / dims, both batch sizes; the ring contents written by the launch are checked at the end."""
    g = torch.Generator().manual_seed(H * D + cap)
    E, N = H * D, 3 * H * D // 2 + 5
    w = (torch.randn(N, E, generator=g) / E ** 0.5).bfloat16()
    ring = L.RingKV(B, H, D, cap)
    kc = torch.zeros(B, H, cap, D, device=DEV)
    vc = torch.zeros(B, H, cap, D, device=DEV)
    pos = torch.zeros(1, dtype=torch.long, device=DEV)
    for s in range(steps):
        qkv = torch.randn(B, 3 * E, generator=g)
        res = torch.randn(B, N, generator=g)
        q, k, v = qkv.view(B, 1, 3, H, D).permute(2, 0, 3, 1, 4)
        keys, vals, pos_k = ring.complete(k, v)
        delta = s - pos_k
        mask = (pos_k >= 0) & (delta >= 0)
        if context is not None:
            mask = mask & (delta < context)
        a = F.scaled_dot_product_attention(q, keys, vals, mask.view(1, -1)).permute(0, 2, 1, 3).reshape(B, E)
        ref = res + a @ w.float().t()
        out = ops.gemv_attn(qkv.to(DEV), kc, vc, pos, w.to(DEV), context=context, res=res.to(DEV))
        pos.add_(1)
        assert rel_err(out, ref) < 1e-4, f"step {s}"
    assert torch.equal(kc.cpu(), ring.k) and torch.equal(vc.cpu(), ring.v)


@pytest.mark.parametrize("B,K,N", [(1, 1024, 3072), (2, 1024, 3072), (2, 128, 384), (1, 64, 200), (2, 4096, 64)])
def test_gemv_embed_in_proj_matches_oracle(B, K, N):
    """First GEMV of a depth step: x = add + table[token] (id -1 -> zero row; `add` a column block of a wider buffer), RMSNorm,
    in-projection; the launch also returns x (the layer's residual)."""
    g = torch.Generator().manual_seed(K + N)
    rows = 37
    table = (0.5 * torch.randn(rows, K, generator=g)).bfloat16()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    alpha = 1 + 0.1 * torch.randn(K, generator=g)
    wide = torch.randn(B, 3 * K, generator=g)
    for toks in ([[5, 36, 0], [7, -1, 2]], [[0, 0, 0], [36, 1, 1]], [[3, -1, 9], [2, 2, 2]]):
        tokens = torch.tensor(toks[:B])
        for col in (0, 1):
            add = wide[:, K:2 * K]
            x_ref = add + L.scaled_embedding(table, tokens[:, col])
            y_ref = L.rms_norm(x_ref, alpha) @ w.float().t()
            y, x = ops.gemv_embed(wide.to(DEV)[:, K:2 * K], table.to(DEV), tokens.to(DEV), col, w.to(DEV), alpha=alpha.to(DEV), eps=1e-8)
            assert torch.equal(x.cpu(), x_ref), (toks, col)
            assert rel_err(y, y_ref) < 1e-5


# ---- the depth phase as one persistent launch (rst_depth_decode_frame, csrc/lm_depth.hip)

@pytest.mark.parametrize("B,sampling", [(1, False), (1, True), (2, True), (2, False)])
def test_depth_frame_equals_launch_per_op_path_at_the_real_shape(B, sampling, monkeypatch):
    """Moshi-7B's depth transformer (6 x 1024, 16 heads, 8 steps, 2048-way heads; the temporal stack cut to one layer): the
    persistent launch must give the tokens of the launch-per-op chain -- same noise, several frames, and the hand-offs must not
    have timed out.  (The launch-per-op chain is the path the oracle / fixture tests above pin.)"""
    cfg = dict(synth.LM_MOSHI_7B, num_layers=1)
    model = LMModel.from_state_dict(synth.lm_state_dict(cfg, seed=4, device=DEV), cfg)
    gen = LMGen(model, use_sampling=sampling)
    g = torch.Generator(device=DEV).manual_seed(8)
    for frame in range(4):
        h_t = torch.randn(B, cfg["dim"], device=DEV, generator=g)
        text = torch.randint(0, cfg["text_card"], (B,), device=DEV, generator=g)
        noise = torch.empty(B, cfg["dep_q"] * gen.top_k, device=DEV).exponential_(1, generator=g) if sampling else None
        got = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("RST_DEPTH_FRAME", mode)
            tokens = torch.full((B, cfg["dep_q"] + 1), -7, dtype=torch.long, device=DEV)
            tokens[:, 0] = text
            gen._depth(tokens, h_t, noise)
            got[mode] = tokens.cpu()
        assert (got["1"][:, 1:] >= 0).all() and (got["1"][:, 1:] < cfg["card"]).all()
        assert torch.equal(got["1"], got["0"]), f"frame {frame}: {got['1'].tolist()} vs {got['0'].tolist()}"
    model.depth_frame_tables().check()


def test_depth_frame_supported_shapes():
    assert ops.depth_frame_supported(1, 1024, 16, 2816, 2048, 8, 6, 250) and ops.depth_frame_supported(2, 128, 2, 352, 32, 2, 2, 8)
    assert not ops.depth_frame_supported(3, 1024, 16, 2816, 2048, 8, 6, 250)       # batch > 2: the skinny-GEMM chain
    assert not ops.depth_frame_supported(1, 1024, 16, 2816, 2048, 9, 6, 250)       # more steps than the tables hold
    assert not ops.depth_frame_supported(1, 1020, 15, 2816, 2048, 8, 6, 250)
