"""The persistent frame launches (csrc/lm_depth.hip, csrc/codec_tr.hip) on a device that does NOT grant them full residency.

A helper kernel (tests/helpers/occupy.hip) holds most CUs' LDS on a second stream, so that only a fraction of a persistent launch's
workgroups can be resident: its hand-offs time out (bounded spins), and the one-workgroup repair launch enqueued behind it must
recompute the frame -- the tokens / activations have to equal those of an undisturbed run, the repair counter must say 1, and the
host-side poll must then retire the persistent path for the device (models/model.py:564-597 -- the reference's depformer_step cannot
return wrong tokens silently; neither may this)."""
import ctypes as C
import os
import time

import pytest
import torch

from rstnet_amd import ops, synth
from rstnet_amd.codec.mimi import MimiCodec
from rstnet_amd.lm.model import LMGen, LMModel
from tests.golden import cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.abspath(__file__))


def _occ():
    path = os.path.join(ROOT, "helpers", "_build", "libocc.so")
    if not os.path.exists(path):
        from tests.helpers import build
        build.build()
    import torch  # noqa: F401  (HIP runtime first)
    lib = C.CDLL(path)
    lib.occ_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.occ_launch.restype = C.c_int
    lib.occ_cu_count.restype = C.c_int
    return lib


def _hold_cus(lib, side: torch.cuda.Stream, ms: int, leave: int = 12):
    """All but `leave` CUs lose 150 KB of their 160 KB LDS for `ms` milliseconds (one occupier workgroup each)."""
    sink = torch.zeros(1, dtype=torch.int32, device=DEV)
    n = lib.occ_cu_count() - leave
    assert n > 0
    rc = lib.occ_launch(n, 150 * 1024, ms, sink.data_ptr(), side.cuda_stream)
    assert rc == 0, rc
    time.sleep(0.1)            # the occupiers are resident before the launch under test is enqueued
    return sink


@pytest.fixture
def clean_health():
    yield
    torch.cuda.synchronize()
    ops._persist_off.clear()
    ops._persist_pending.clear()
    for refs in ops._persist_status.values():          # the repair counters of this test must not retire the path for later tests
        for r in refs:
            t = r()
            if t is not None:
                t.zero_()


@pytest.mark.parametrize("B", [1, 2])
def test_depth_frame_repair_launch_recomputes_the_frame(B, clean_health, monkeypatch):
    """The one-workgroup repair launch of rst_depth_decode_frame, driven deterministically: a time-out code is planted in status[0]
    before the call, so the launch enqueued behind the persistent one must recompute the whole frame alone (8 steps x 6 layers, all
    16 heads, their KV history in the global scratch) -- same tokens, status[1] counts the repair, status[0] is cleared, the planted
    code lands in status[2].  Then the host-side poll retires the persistent path of the device and the launch-per-op chain gives
    the same tokens again.  (How a REAL loss of residency is detected -- bounded spins on the hand-offs -- is exercised by the codec
    transformer test below with CUs taken away; both kernels share that code, csrc/persist.h.)"""
    monkeypatch.setenv("RST_DEPTH_FRAME", "1")
    cfg = dict(synth.LM_MOSHI_7B, num_layers=1)
    model = LMModel.from_state_dict(synth.lm_state_dict(cfg, seed=4, device=DEV), cfg)
    gen = LMGen(model, use_sampling=True)
    Q = cfg["dep_q"]
    g = torch.Generator(device=DEV).manual_seed(3 + B)
    h_t = torch.randn(B, cfg["dim"], device=DEV, generator=g)
    text = torch.randint(0, cfg["text_card"], (B,), device=DEV, generator=g)
    noise = torch.empty(B, Q * gen.top_k, device=DEV).exponential_(1, generator=g)

    def run():
        tokens = torch.full((B, Q + 1), -7, dtype=torch.long, device=DEV)
        tokens[:, 0] = text
        gen._depth(tokens, h_t, noise)
        torch.cuda.synchronize()
        return tokens.cpu()
    want = run()
    tables = model.depth_frame_tables()
    assert tables.status.tolist() == [0, 0, 0, 0]
    code = 1 << 20
    tables.status[0] = code
    t0 = time.perf_counter()
    got = run()
    took = time.perf_counter() - t0
    assert tables.status.tolist() == [0, 1, code, 0], tables.status.tolist()
    assert torch.equal(got, want), f"repaired frame {got.tolist()} vs undisturbed {want.tolist()} ({took * 1e3:.1f} ms)"
    # an undisturbed frame afterwards: the persistent launch alone, no further repair
    assert torch.equal(run(), want) and tables.status.tolist()[1] == 1
    # the host-side poll sees the repair and retires the persistent path of the device: the launch-per-op chain takes over
    with pytest.warns(RuntimeWarning, match="persistent frame launches retired"):
        ops.persistent_poll(torch.device(DEV), synchronize=True)
    assert not ops.depth_frame_enabled(torch.device(DEV)) and ops.persistent_epoch(DEV) >= 1
    assert torch.equal(run(), want) and tables.status.tolist()[1] == 1


def test_codec_transformer_frame_without_full_residency_is_repaired(clean_health, monkeypatch):
    monkeypatch.setenv("RST_DEPTH_FRAME", "1")
    sd = synth.mimi_state_dict(cases.MIMI_SEED, layer_scale=cases.TRANSFORMER_LAYER_SCALE)
    model = MimiCodec.from_state_dict(sd).to(DEV)
    tr = model.encoder_transformer
    x = cases.transformer_input(batch=2, frames=12).to(DEV)
    status = ops.codec_transformer_status(torch.device(DEV))
    base = status.tolist()

    def stream(disturb_at=None):
        outs = []
        with tr.streaming(2):
            for i in range(0, x.shape[-1], 2):
                keep = None
                if disturb_at == i:
                    keep = _hold_cus(lib, side, ms=1500)
                outs.append(tr(x[:, :, i:i + 2].contiguous())[0].clone())
                torch.cuda.synchronize()
                if keep is not None:
                    side.synchronize()
        return torch.cat(outs, -1).cpu()
    lib, side = _occ(), torch.cuda.Stream()
    want = stream()
    assert status.tolist() == base
    got = stream(disturb_at=4)                      # the third step loses its residency; later steps read the rings it appended to
    st = status.tolist()
    assert st[1] == base[1] + 1 and st[0] == 0, f"status {st}: the step was expected to time out and be repaired"
    assert torch.equal(got, want), float((got - want).abs().max())
