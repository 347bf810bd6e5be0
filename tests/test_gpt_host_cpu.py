"""Host-side logic of rstnet_amd.lm.gpt that needs no GPU: LoRA merge, legacy key remaps, config validation, the QKV row
permutation.  The oracle (pinned to the reference by gpt_tiny.npz) is the checker."""
import pytest
import torch

from oracle import gpt_oracle as Gp
from rstnet_amd import synth
from rstnet_amd.lm import gpt as G
from tests.golden import cases


def _cfgs(cfg_d):
    keep = set(Gp.GPTConfig.__dataclass_fields__)
    return G.Config.from_dict(cfg_d), Gp.GPTConfig(**{k: v for k, v in cfg_d.items() if k in keep})


@pytest.mark.parametrize("cfg_d", [synth.GPT_TINY_GQA, synth.GPT_TINY_MHA], ids=["gqa", "mha"])
def test_lora_merge_matches_oracle(cfg_d):
    cfg, ocfg = _cfgs(cfg_d)
    sd = {k: v.float() for k, v in synth.gpt_state_dict(cfg_d, cases.GPT_SEED).items()}
    mine = G.merge_lora_state_dict(sd, cfg)
    ref = Gp.merged_state(sd, ocfg)
    assert set(mine) == set(ref)
    for k in ref:
        assert torch.allclose(mine[k].float(), ref[k].float(), rtol=0, atol=1e-6), k


def test_legacy_keys_and_missing_adapter_config():
    cfg_d = dict(synth.GPT_TINY_GQA)
    sd = synth.gpt_state_dict(cfg_d, 1, lora=False)
    old = {}
    for k, v in sd.items():
        if ".linear." in k and (k.startswith("lm_head") or ".attn." in k or ".mlp." in k):
            k = k.replace(".linear.", ".")
        old[k] = v
    assert "lm_head.weight" in old and "transformer.h.0.attn.attn.weight" in old and "transformer.h.1.mlp.fc_2.weight" in old
    assert set(G._remap_legacy(old)) == set(sd)
    with pytest.raises(RuntimeError):
        G.GPT.from_state_dict(synth.gpt_state_dict(cfg_d, 1), G.Config.from_dict({**cfg_d, "lora_r": 0}))


def test_config_rejects_unimplemented_paths():
    ok = dict(synth.GPT_TINY_MHA)
    G.Config.from_dict(ok)
    for bad in ({"norm_class_name": "LayerNorm"}, {"mlp_class_name": "GptNeoxMLP", "intermediate_size": 8}, {"parallel_residual": True},
                {"rope_condense_ratio": 2}, {"attention_logit_softcapping": 30.0}, {"n_query_groups": 1}):
        with pytest.raises(NotImplementedError):
            G.Config.from_dict({**ok, **bad})


@pytest.mark.parametrize("cfg_d", [synth.GPT_TINY_GQA, synth.GPT_TINY_MHA], ids=["gqa", "mha"])
def test_qkv_row_order_is_the_rotate_half_to_interleaved_permutation(cfg_d):
    """Attention scores computed from the permuted rows with INTERLEAVED RoPE equal the reference formulation (interleaved
    GQA layout + rotate-half RoPE), and the value rows are only regrouped."""
    cfg, ocfg = _cfgs(cfg_d)
    order = G._qkv_row_order(cfg)
    H, Gq, hs, n = cfg.n_head, cfg.n_query_groups, cfg.head_size, cfg.rope_n_elem
    assert sorted(order.tolist()) == list(range((H + 2 * Gq) * hs))
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn((H + 2 * Gq) * hs, generator=g)
    pos = 7
    # reference side
    qpk = H // Gq
    ref = qkv.view(Gq, qpk + 2, hs)
    q_ref, k_ref, v_ref = ref[:, :qpk].reshape(H, hs), ref[:, qpk], ref[:, qpk + 1]
    cos, sin = Gp.build_rope_cache(16, n, cfg.rope_base)
    rot = lambda x: torch.cat((Gp.apply_rope_half(x[..., :n], cos[pos], sin[pos]), x[..., n:]), -1)
    s_ref = (rot(q_ref).view(Gq, qpk, hs) * rot(k_ref)[:, None]).sum(-1)
    # permuted side: interleaved rotation of the leading n dims
    p = qkv[order]
    q, k, v = p[:H * hs].view(H, hs), p[H * hs:(H + Gq) * hs].view(Gq, hs), p[(H + Gq) * hs:].view(Gq, hs)
    theta = torch.exp(torch.arange(n // 2).float() * (-torch.log(torch.tensor(float(cfg.rope_base))) * 2 / n)) * pos

    def rot_i(x):
        xr, xi = x[..., 0:n:2], x[..., 1:n:2]
        out = x.clone()
        out[..., 0:n:2] = xr * torch.cos(theta) - xi * torch.sin(theta)
        out[..., 1:n:2] = xr * torch.sin(theta) + xi * torch.cos(theta)
        return out
    s = (rot_i(q).view(Gq, qpk, hs) * rot_i(k)[:, None]).sum(-1)
    assert torch.allclose(s, s_ref, rtol=1e-5, atol=1e-5)
    assert torch.equal(v, v_ref)


@pytest.mark.parametrize("cfg_d", [synth.GPT_TINY_GQA, synth.GPT_TINY_MHA], ids=["gqa", "mha"])
def test_load_adapters_swaps_in_place(cfg_d):
    """GPT.load_adapters: a model built with keep_lora_base=True takes another adapter set (or none) and ends up with exactly the
    weights of a model built from [base ; that set] -- the reference keeps adapters unmerged for this (llama_streaming.py:113-143),
    here the swap re-merges in place (same tensors: the kernels' packed copies follow their `_version`)."""
    cfg = G.Config.from_dict(cfg_d)
    sd1 = synth.gpt_state_dict(cfg_d, 11)
    is_lora = lambda k: k.endswith((".lora_A", ".lora_B"))
    base = {k: v for k, v in sd1.items() if not is_lora(k)}
    ad2 = {k: v for k, v in synth.gpt_state_dict(cfg_d, 12).items() if is_lora(k)}
    assert ad2 and any(not torch.equal(ad2[k], sd1[k]) for k in ad2)

    model = G.GPT.from_state_dict({k: v.clone() for k, v in sd1.items()}, cfg, keep_lora_base=True)
    ptrs = {n: p.data_ptr() for n, p in model.named_parameters()}
    vers = {n: p._version for n, p in model.named_parameters()}

    def same_as(ref_sd):
        ref = G.GPT.from_state_dict({k: v.clone() for k, v in ref_sd.items()}, cfg)
        for (n, p), (n2, q) in zip(model.named_parameters(), ref.named_parameters()):
            assert n == n2 and torch.equal(p, q), n

    same_as(sd1)
    model.load_adapters(ad2)
    same_as({**base, **ad2})
    adapted = [n for n in ptrs if n in model._lora_base]
    assert adapted and all(dict(model.named_parameters())[n].data_ptr() == ptrs[n] for n in ptrs)          # in place
    assert all(dict(model.named_parameters())[n]._version > vers[n] for n in adapted)                       # packed copies will rebuild
    model.load_adapters(None)
    same_as(base if cfg.lora_r == 0 else {**base})           # back to the un-adapted weights
    model.load_adapters({k: v for k, v in sd1.items() if is_lora(k)})
    same_as(sd1)

    plain = G.GPT.from_state_dict({k: v.clone() for k, v in sd1.items()}, cfg)
    with pytest.raises(RuntimeError):
        plain.load_adapters(ad2)                              # built without keep_lora_base
    with pytest.raises(RuntimeError):
        model.load_adapters({"transformer.h.0.nope.lora_A": torch.zeros(1, 1), "transformer.h.0.nope.lora_B": torch.zeros(1, 1)})


@pytest.mark.parametrize("cfg_d", [synth.GPT_TINY_GQA, synth.GPT_TINY_MHA, {**synth.GPT_TINY_GQA, "lora_alpha": 6, "lora_key": True}],
                         ids=["gqa", "mha", "gqa_qkv_alpha6"])
def test_unmerged_adapter_forms_equal_the_merged_update(cfg_d):
    """``from_state_dict(..., merge_lora=False)``: the kernel-side adapter forms (rank padded to 16, fused-QKV / stacked-gate B laid out
    block-diagonally in the kernels' row order, alpha / r folded into B when a power of two) multiply out to exactly the update
    ``merge_lora_state_dict`` adds -- incl. the zero_pad scatter (q, v only) and its 'as is' return (q, k, v with grouped queries);
    the state dict keeps the reference's ``lora_A`` / ``lora_B`` keys; adapters for un-adapted linears / wrong shapes are refused."""
    cfg, _ = _cfgs(cfg_d)
    sd = synth.gpt_state_dict(cfg_d, cases.GPT_SEED)                       # bf16: adapters are exact in the kernel forms
    model = G.GPT.from_state_dict({k: v.clone() for k, v in sd.items()}, cfg, merge_lora=False)
    assert set(model.state_dict()) == set(sd)
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd[k]), k
    f32 = {k: v.float() for k, v in sd.items()}
    merged = G.merge_lora_state_dict(f32, cfg)
    order = G._qkv_row_order(cfg)

    def delta(ad):
        A, Bm, post = ad
        assert A.shape[0] % 16 == 0 and Bm.shape[1] == A.shape[0] and A.dtype == Bm.dtype == torch.bfloat16
        return (Bm.float() @ A.float()) * post

    for l, blk in enumerate(model.transformer.h):
        p = f"transformer.h.{l}"
        want = (merged[f"{p}.attn.attn.linear.weight"] - f32[f"{p}.attn.attn.linear.weight"])[order]
        assert torch.allclose(delta(blk.attn.packed_qkv_adapter()), want, rtol=0, atol=2e-6)
        if cfg.lora_mlp:
            want = torch.cat([merged[f"{p}.mlp.{n}.linear.weight"] - f32[f"{p}.mlp.{n}.linear.weight"] for n in ("fc_1", "fc_2")])
            assert torch.allclose(delta(blk.mlp.packed_fc_adapter()), want, rtol=0, atol=2e-6)
            assert torch.allclose(delta(blk.mlp.proj.adapter()), merged[f"{p}.mlp.proj.linear.weight"] - f32[f"{p}.mlp.proj.linear.weight"],
                                  rtol=0, atol=2e-6)
        else:
            assert blk.mlp.packed_fc_adapter() is None and blk.mlp.proj.adapter() is None
        if not cfg.lora_projection:
            assert blk.attn.proj.adapter() is None
    if cfg.lora_head:
        assert torch.allclose(delta(model.lm_head.adapter()), merged["lm_head.linear.weight"] - f32["lm_head.linear.weight"], rtol=0, atol=2e-6)
    post = model.transformer.h[0].attn.packed_qkv_adapter()[2]
    assert post == (1.0 if cfg.lora_alpha == 8 else cfg.lora_alpha / cfg.lora_r)
    # swapping: a set without the head adapter drops that branch; bad names / shapes are refused
    ad2 = {k: v for k, v in synth.gpt_state_dict(cfg_d, 5).items() if k.endswith((".lora_A", ".lora_B")) and not k.startswith("lm_head")}
    model.load_adapters(ad2)
    assert model.lm_head.adapter() is None
    assert torch.equal(model.transformer.h[1].attn.attn.lora_A, ad2["transformer.h.1.attn.attn.lora_A"])
    with pytest.raises(RuntimeError):
        model.load_adapters({"transformer.wte.lora_A": torch.zeros(4, 4), "transformer.wte.lora_B": torch.zeros(4, 4)})
    with pytest.raises(RuntimeError):
        model.load_adapters({"transformer.h.0.attn.attn.lora_A": torch.zeros(3, cfg.n_embd),
                             "transformer.h.0.attn.attn.lora_B": ad2["transformer.h.0.attn.attn.lora_B"]})
    model.load_adapters(None)
    assert all(blk.attn.packed_qkv_adapter() is None for blk in model.transformer.h)


def test_graphed_argument_contract_matches_cudagraphed():
    """`Graphed` keeps the argument contract of the reference's CUDAGraphed (utils/compile.py:216-256) -- checked here on the host-side
    paths that need no device: keyword arguments refused, a disabled / nested wrapper calls straight through, `reset(warmup_steps)`."""
    from rstnet_amd import graphs
    calls = []
    g = graphs.Graphed(lambda a, k: calls.append((a, k)) or a, disable=True)
    assert g(3, 4) == 3 and calls == [(3, 4)]
    with pytest.raises(RuntimeError):
        g(3, k=4)
    inner = graphs.Graphed(lambda a: a + 1)              # enabled, but called from inside another graphed call: runs plainly
    with graphs._set_in_graph():
        assert graphs.in_cuda_graph() and inner(1) == 2 and inner.graph is None and inner.calls == 0
    assert not graphs.in_cuda_graph()
    inner.reset(warmup_steps=3)
    assert inner.warmup == 3 and inner.graph is None
    # the value / type checks of a captured call (no capture needed to exercise them)
    g2 = graphs.Graphed(lambda a, n: a)
    g2.static_in = [torch.zeros(2, 3), 7]
    g2._refresh_inputs((torch.ones(2, 3), 7))
    assert float(g2.static_in[0].sum()) == 6.0
    for bad in ((torch.ones(2, 4), 7), (torch.ones(2, 3), 8), (5, 7), (torch.ones(2, 3), torch.ones(1)), (torch.ones(2, 3),)):
        with pytest.raises(ValueError):
            g2._refresh_inputs(bad)
        assert float(g2.static_in[0].sum()) == 6.0        # a refused call copies nothing
    assert graphs.CUDAGraphed is graphs.Graphed
