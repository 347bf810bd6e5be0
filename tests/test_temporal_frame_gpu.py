"""rst_temporal_decode_frame (csrc/lm_temporal.hip): all layers of the temporal transformer of a batch-1 LM step as ONE persistent launch
-- models/model.py:364-389 / modules/transformer.py:376-423, 551-592 -- against the launch-per-op chain (itself pinned to the reference by
lm_tiny*.npz) at the Moshi-7B layer width, plus its in-stream repair launch."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DIM, HEADS, CAP = 4096, 32, 3000


@pytest.fixture
def clean_health():
    """The repair counters these tests provoke must not retire the device's persistent launches for the tests that follow."""
    yield
    from rstnet_amd import ops
    torch.cuda.synchronize()
    ops._persist_off.clear()
    ops._persist_pending.clear()
    for refs in ops._persist_status.values():
        for r in refs:
            t = r()
            if t is not None:
                t.zero_()


def _transformer(layers, kvd):
    from rstnet_amd.lm.model import StreamingTransformer
    torch.manual_seed(0)
    return StreamingTransformer(DIM, HEADS, layers, int(4.125 * DIM), context=CAP, positional_embedding="rope", device="cuda:0",
                                dtype=torch.bfloat16, kv_dtype=kvd)


def _run(tr, xs, persistent, pos0=0, rings=None, plant=0):
    DIM_ = tr.d_model
    from rstnet_amd import ops
    old = ops.TEMPORAL_FRAME
    ops.TEMPORAL_FRAME = persistent
    try:
        outs = []
        with tr.streaming(1):
            st = tr._streaming_state
            if rings is not None:
                for l in range(len(st.k)):
                    st.k[l].copy_(rings[0][l]); st.v[l].copy_(rings[1][l])
            st.pos.fill_(pos0)
            for i, x in enumerate(xs):
                if plant and i == 1:
                    st.tables.status[0] = plant          # a time-out code as a timed-out hand-off leaves it
                outs.append(tr.step(x).clone())
            torch.cuda.synchronize()
            assert (st.tables is not None) == persistent, "the persistent launch was not taken / was taken unasked"
            status = st.tables.status.tolist() if st.tables is not None else None
            kv = [st.k[0].clone(), st.v[0].clone()]
        return outs, status, kv
    finally:
        ops.TEMPORAL_FRAME = old


def _rel(a, b):
    return max(((u - v).abs().max() / v.abs().max()).item() for u, v in zip(a, b))


@pytest.mark.parametrize("kv", ["f32", "bf16"])
def test_persistent_temporal_frame_equals_launch_per_op(kv):
    from rstnet_amd import ops
    kvd = torch.float32 if kv == "f32" else torch.bfloat16
    tr = _transformer(2, kvd)
    Hd = tr.layers[0].gating.linear_out.weight.shape[1]
    assert Hd == 11264
    old = ops.TEMPORAL_FRAME
    ops.TEMPORAL_FRAME = True
    try:
        assert ops.temporal_frame_supported(1, DIM, HEADS, Hd, 2, CAP, kvd == torch.bfloat16, "cuda:0")
    finally:
        ops.TEMPORAL_FRAME = old
    g = torch.Generator(device="cuda:0").manual_seed(1)
    # 140 positions from an empty ring: one split, then two (128 slots per workgroup), the new step's own slot in every block position
    xs = [torch.randn(1, DIM, device="cuda:0", generator=g) for _ in range(140)]
    ref, _, kv_ref = _run(tr, xs, False)
    got, status, kv_got = _run(tr, xs, True)
    assert status == [0, 0, 0, 0]
    # fp32 rings: the two paths differ in summation order only.  bf16 rings: a key / value that differs in its last fp32 bit between
    # the paths now and then rounds to the other bf16 neighbour (2^-8 relative on one element of one slot) -- the reference's own cache
    # precision; with few slots in the ring one such flip moves the output by ~1e-3 of its largest element
    tol = 1e-5 if kv == "f32" else 4e-3
    assert _rel(got, ref) < tol
    assert torch.allclose(kv_got[0].float(), kv_ref[0].float(), atol=2e-2 if kv == "bf16" else 1e-5) and \
        torch.allclose(kv_got[1].float(), kv_ref[1].float(), atol=2e-2 if kv == "bf16" else 1e-5)
    # a full ring across its wrap (the `delta <= 0` slot of RingKVCache.complete, eight splits per head, the combine by the head's owner)
    H, D = HEADS, DIM // HEADS
    rings = ([(0.5 * torch.randn(1, H, CAP, D, device="cuda:0", generator=g)).to(kvd) for _ in range(2)],
             [(0.5 * torch.randn(1, H, CAP, D, device="cuda:0", generator=g)).to(kvd) for _ in range(2)])
    ref2, _, _ = _run(tr, xs[:20], False, pos0=CAP - 8, rings=rings)
    got2, status2, _ = _run(tr, xs[:20], True, pos0=CAP - 8, rings=rings)
    assert status2 == [0, 0, 0, 0]
    assert _rel(got2, ref2) < (1e-5 if kv == "f32" else 1e-4)


def test_persistent_temporal_frame_repair_launch(clean_health):
    """A planted time-out code: the one-workgroup launch behind the persistent one recomputes the step from the untouched input and the
    rings, counts the repair and clears the code (csrc/persist.h).  Same arithmetic except that one workgroup walks a head's whole ring
    (one split, the persistent launch uses up to eight and merges them): equal to rounding, not bit for bit."""
    tr = _transformer(2, torch.bfloat16)
    g = torch.Generator(device="cuda:0").manual_seed(2)
    xs = [torch.randn(1, DIM, device="cuda:0", generator=g) for _ in range(3)]
    clean, st0, kv0 = _run(tr, xs, True, pos0=200)
    rep, st1, kv1 = _run(tr, xs, True, pos0=200, plant=8)
    assert st0 == [0, 0, 0, 0] and st1 == [0, 1, 8, 0]
    assert _rel(rep, clean) < 4e-3          # bf16 rings (see the parity test); fp32 rings below
    assert torch.allclose(kv0[0].float(), kv1[0].float(), atol=2e-2) and torch.allclose(kv0[1].float(), kv1[1].float(), atol=2e-2)
    tr = _transformer(1, torch.float32)
    clean, st0, kv0 = _run(tr, xs, True, pos0=200)
    rep, st1, kv1 = _run(tr, xs, True, pos0=200, plant=32)
    assert st0 == [0, 0, 0, 0] and st1 == [0, 1, 32, 0]
    assert _rel(rep, clean) < 1e-5
    assert torch.allclose(kv0[0], kv1[0], atol=1e-5) and torch.allclose(kv0[1], kv1[1], atol=1e-5)


def test_auto_mode_switches_the_captured_frame_at_the_threshold():
    """`ops.TEMPORAL_FRAME = "auto"` (the default): an `LMGen` session takes the launch-per-op chain while the temporal rings are short and
    re-captures its frame onto the persistent launch when the host-side position crosses `TEMPORAL_FRAME_AUTO_POS` -- token streams equal
    to the always-off session's (greedy; the two paths agree to rounding, far from the ties of a random model)."""
    from rstnet_amd import ops, synth
    from rstnet_amd.lm.model import LMGen, LMModel
    cfg = dict(synth.LM_MOSHI_7B, num_layers=2, depformer_num_layers=2)
    sd = synth.lm_state_dict(cfg, seed=5, device="cuda:0")
    model = LMModel.from_state_dict(sd, cfg, kv_dtype=torch.float32)
    g = torch.Generator(device="cuda:0").manual_seed(9)
    n = 12
    user = torch.randint(0, cfg["card"], (n, 1, cfg["n_q"] - cfg["dep_q"], 1), generator=g, device="cuda:0")
    old, old_pos = ops.TEMPORAL_FRAME, ops.TEMPORAL_FRAME_AUTO_POS

    def run(mode, threshold):
        ops.TEMPORAL_FRAME, ops.TEMPORAL_FRAME_AUTO_POS = mode, threshold
        gen = LMGen(model, use_sampling=False)
        outs, choices = [], []
        with gen.streaming(1):
            for s in range(n):
                o = gen.step(user[s])
                outs.append(None if o is None else o.clone())
                choices.append(gen._streaming_state.temporal_choice)
            tabs = model.transformer._streaming_state.tables
        return outs, choices, tabs
    try:
        ref, ch0, tabs0 = run(False, 1024)
        got, ch1, tabs1 = run("auto", 6)
    finally:
        ops.TEMPORAL_FRAME, ops.TEMPORAL_FRAME_AUTO_POS = old, old_pos
    assert tabs0 is None and not any(ch0)
    assert ch1[:6] == [False] * 6 and all(ch1[6:]) and tabs1 is not None and tabs1.status.tolist() == [0, 0, 0, 0]
    for a, b in zip(got, ref):
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b))


@pytest.mark.parametrize("dim,heads,cap,layers,kv", [(2048, 32, 300, 3, "bf16"), (1024, 16, 130, 3, "f32"), (1536, 12, 96, 2, "f32")])
def test_other_shapes(dim, heads, cap, layers, kv):
    """Head dim 64 and 128 away from the 7B width: fewer rows per wave than waves in places (1536 / 1024 wide: some waves own no row of
    the narrow ops), vectors that are not whole 4096-k blocks (zero-padded staging), a hidden width that is not a multiple of 512, short
    rings crossing their wrap."""
    from rstnet_amd import ops
    from rstnet_amd.lm.model import StreamingTransformer
    kvd = torch.float32 if kv == "f32" else torch.bfloat16
    torch.manual_seed(3)
    tr = StreamingTransformer(dim, heads, layers, int(4.125 * dim), context=cap, positional_embedding="rope", device="cuda:0", dtype=torch.bfloat16,
                              kv_dtype=kvd)
    Hd = tr.layers[0].gating.linear_out.weight.shape[1]
    old = ops.TEMPORAL_FRAME
    ops.TEMPORAL_FRAME = True
    try:
        if not ops.temporal_frame_supported(1, dim, heads, Hd, layers, cap, kvd == torch.bfloat16, "cuda:0"):
            pytest.skip("shape not served by the persistent launch")
    finally:
        ops.TEMPORAL_FRAME = old
    g = torch.Generator(device="cuda:0").manual_seed(4)
    xs = [torch.randn(1, dim, device="cuda:0", generator=g) for _ in range(cap + 40)]
    ref, _, _ = _run(tr, xs, False)
    got, status, _ = _run(tr, xs, True)
    assert status == [0, 0, 0, 0]
    assert _rel(got, ref) < (1e-5 if kv == "f32" else 4e-3)


def test_temporal_frame_without_full_residency_is_repaired(clean_health):
    """A REAL loss of residency: a helper kernel (tests/helpers/occupy.hip) holds 150 KB of LDS on all but 12 CUs while one step is
    launched, so most of the persistent launch's workgroups cannot start: the comm waves' sweeps time out (bounded spins, 0.1 s), the
    launch drains, and the one-workgroup launch behind it recomputes the step -- outputs and rings equal to an undisturbed run to
    rounding, status = [0, 1, codes, 0]; the later steps run on the rings the repair appended to."""
    import time
    from tests.test_persistent_safety_gpu import _hold_cus, _occ
    tr = _transformer(2, torch.float32)
    g = torch.Generator(device="cuda:0").manual_seed(6)
    xs = [torch.randn(1, DIM, device="cuda:0", generator=g) for _ in range(5)]
    lib, side = _occ(), torch.cuda.Stream()
    from rstnet_amd import ops
    old = ops.TEMPORAL_FRAME
    ops.TEMPORAL_FRAME = True

    def stream(disturb_at=None):
        outs = []
        with tr.streaming(1):
            st = tr._streaming_state
            st.pos.fill_(150)
            for i, x in enumerate(xs):
                keep = _hold_cus(lib, side, ms=2500) if disturb_at == i else None
                outs.append(tr.step(x).clone())
                torch.cuda.synchronize()
                if keep is not None:
                    side.synchronize()
            return outs, st.tables.status.tolist(), [st.k[1].clone(), st.v[1].clone()]
    try:
        want, st0, kv0 = stream()
        t0 = time.perf_counter()
        got, st1, kv1 = stream(disturb_at=2)
        took = time.perf_counter() - t0
    finally:
        ops.TEMPORAL_FRAME = old
    assert st0 == [0, 0, 0, 0]
    assert st1[0] == 0 and st1[1] == 1 and st1[2] != 0, f"status {st1}: the step was expected to time out and be repaired ({took:.2f} s)"
    assert _rel(got, want) < 1e-5
    assert torch.allclose(kv0[0], kv1[0], atol=1e-5) and torch.allclose(kv0[1], kv1[1], atol=1e-5)
