"""rst_codec_transformer_frame (csrc/codec_tr.hip) at head dims other than Mimi's 64: the ring attention's thread -> (4 dims, slot
class) layout, its lane-group sums and its 16-slot batches all depend on D (lane groups of D / 4 = 4 .. 64 lanes, 1024 / D slot
classes, 16384 / D slots per batch).  Streamed past the ring wrap, several positions and streams per step, against the oracle's
TransformerStream (modules/transformer.py:376-423,595-690)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mimi_oracle as O
from rstnet_amd import ops

DEV = "cuda:0"


def _state(E, H, F, L, seed, layer_scale):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for l in range(L):
        p = f"tr.transformer.layers.{l}"
        sd[f"{p}.norm1.weight"] = 1 + 0.1 * torch.randn(E, generator=g)
        sd[f"{p}.norm1.bias"] = 0.1 * torch.randn(E, generator=g)
        sd[f"{p}.norm2.weight"] = 1 + 0.1 * torch.randn(E, generator=g)
        sd[f"{p}.norm2.bias"] = 0.1 * torch.randn(E, generator=g)
        sd[f"{p}.self_attn.in_proj_weight"] = torch.randn(3 * E, E, generator=g) / E ** 0.5
        sd[f"{p}.self_attn.out_proj.weight"] = torch.randn(E, E, generator=g) / E ** 0.5
        sd[f"{p}.linear1.weight"] = torch.randn(F, E, generator=g) / E ** 0.5
        sd[f"{p}.linear2.weight"] = torch.randn(E, F, generator=g) / F ** 0.5
        sd[f"{p}.layer_scale_1.scale"] = torch.full((E,), layer_scale)
        sd[f"{p}.layer_scale_2.scale"] = torch.full((E,), layer_scale)
    return sd


@pytest.mark.parametrize("E,H,F,L,cap,B,chunks", [
    (128, 8, 512, 3, 20, 2, (2, 1, 2, 2, 1) * 6),        # D = 16: lane groups of 4, 64 slot classes
    (256, 8, 512, 2, 37, 1, (3, 4, 1, 2) * 5),           # D = 32; 3 and 4 positions per step
    (512, 4, 1024, 2, 250, 1, (4,) * 66),                # D = 128: two batches of slots once the ring fills (264 positions: past the wrap)
    (512, 2, 512, 2, 150, 2, (2,) * 80),                 # D = 256: 64 slots per batch -> three batches, a full wave per slot
    (512, 8, 2048, 8, 24, 4, (1,) * 30),                 # Mimi's layer shape, four streams of one position, a short ring
])
def test_frame_vs_oracle_stream(E, H, F, L, cap, B, chunks):
    D = E // H
    assert ops.codec_transformer_frame_supported(B, max(chunks), E, H, F, L, cap, device=DEV)
    sd = _state(E, H, F, L, seed=E + H + cap, layer_scale=0.3)
    cfg = O.MimiConfig(latent_dim=E, num_heads=H, num_layers=L, context=cap, dim_feedforward=F)
    ts = O.TransformerStream(sd, "tr", cfg, B)
    layers = []
    for l in range(L):
        p = f"tr.transformer.layers.{l}"
        layers.append({"in_proj": sd[f"{p}.self_attn.in_proj_weight"], "out_proj": sd[f"{p}.self_attn.out_proj.weight"],
                       "linear1": sd[f"{p}.linear1.weight"], "linear2": sd[f"{p}.linear2.weight"],
                       "norm1_w": sd[f"{p}.norm1.weight"], "norm1_b": sd[f"{p}.norm1.bias"], "norm2_w": sd[f"{p}.norm2.weight"],
                       "norm2_b": sd[f"{p}.norm2.bias"], "scale1": sd[f"{p}.layer_scale_1.scale"], "scale2": sd[f"{p}.layer_scale_2.scale"]})
    layers = [{k: v.to(DEV).contiguous() for k, v in ly.items()} for ly in layers]
    for ly in layers:
        # rings the caller did NOT zero: slots that were never written must not reach the output
        ly["k_cache"] = torch.full((B, H, cap, D), float("nan"), device=DEV)
        ly["v_cache"] = torch.full((B, H, cap, D), float("inf"), device=DEV)
    pos = torch.zeros(1, dtype=torch.long, device=DEV)
    g = torch.Generator().manual_seed(7)
    worst = 0.0
    for T in chunks:
        x = torch.randn(B, E, T, generator=g)
        with torch.no_grad():
            ref = ts.step(x).transpose(1, 2)
        y = ops.codec_transformer_frame(x.transpose(1, 2).contiguous().to(DEV), layers, pos, H=H, context=cap, rope=True,
                                        max_period=cfg.max_period, eps=1e-5)
        pos.add_(T)
        assert torch.isfinite(y).all(), "a never-written ring slot leaked into the output"
        worst = max(worst, float((y.cpu() - ref).abs().max() / ref.abs().max()))
    assert sum(chunks) > cap, "the test must pass the ring wrap"
    assert worst < 2e-4, worst
    assert ops.codec_transformer_status(torch.device(DEV)).tolist()[:3] == [0, 0, 0], "a hand-off of the persistent launch timed out"


@pytest.mark.parametrize("name", ["d16", "d32", "d128", "d256"])
def test_module_vs_reference_fixture(name):
    """tests/golden/transformer_dims.npz -- the REFERENCE's ProjectedTransformer streamed at head dims 16 / 32 / 128 / 256 -- through
    the package's module: steps of at most four rows take the persistent launch, the others the launch-per-op layer loop, on the
    same rings (a session may mix them)."""
    import os

    import numpy as np

    from rstnet_amd.codec.transformer import ProjectedTransformer
    from tests.golden import cases

    E, H, F, L, ctx, B, chunks = cases.TRANSFORMER_DIMS[name]
    ref = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), "golden", "transformer_dims.npz"))[name])
    m = ProjectedTransformer(input_dimension=E, output_dimensions=(E,), d_model=E, num_heads=H, num_layers=L, dim_feedforward=F,
                             causal=True, context=ctx, conv_layout=True, max_period=10000, gating="none", norm="layer_norm",
                             positional_embedding="rope", layer_scale=cases.TRANSFORMER_DIMS_LAYER_SCALE)
    sd = {k[len("tr."):]: v for k, v in cases.transformer_dims_state(name).items()}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    m = m.to(DEV).eval()
    x = cases.transformer_dims_input(name)
    # (the steps are the fixture's: past the wrap the result of a stream DOES depend on how it is cut -- a step's later positions
    # overwrite ring slots its earlier queries would still have seen, modules/transformer.py:254-278.  The launch-per-op attention
    # kernels serve head dims 32 / 64 / 128 only, so the 16- and 256-dim cases keep every step within the persistent launch's 4 rows.)
    if E // H not in (32, 64, 128):
        assert all(B * T <= 4 for T in chunks)
    ys, i, persistent = [], 0, 0
    with torch.no_grad(), m.streaming(B):
        for T in chunks:
            persistent += int(ops.codec_transformer_frame_supported(B, T, E, H, F, L, ctx, device=DEV) and B * T <= 4)
            ys.append(m(x[:, :, i:i + T].contiguous().to(DEV))[0].cpu())
            i += T
    y = torch.cat(ys, -1)
    assert persistent > 0, "no step of this case took the persistent launch"
    assert tuple(y.shape) == tuple(ref.shape)
    assert float((y - ref).abs().max() / ref.abs().max()) < 2e-4
    assert ops.codec_transformer_status(torch.device(DEV)).tolist()[:3] == [0, 0, 0]
