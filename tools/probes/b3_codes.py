"""RVQ decisions of the 8 x 10 s batch with the large GEMMs on the three-plane bf16 form vs on the f32 matrix instruction, both against
the CPU oracle (tests/test_mimi_gpu.py::test_encode_decode_8x10s_full_size_kernels_vs_oracle runs the product setting only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import mimi_oracle as O  # noqa: E402
from rstnet_amd import ops, synth  # noqa: E402
from rstnet_amd.codec.mimi import MimiCodec  # noqa: E402

DEV = "cuda:0"
sd = synth.mimi_state_dict(0)
model = MimiCodec.from_state_dict(sd).to(DEV)
cfg = O.MimiConfig()
audio = synth.synth_audio(8, 240000, seed=77)
with torch.no_grad():
    ref_codes = O.rvq_encode(sd, cfg, O.encode_latent(sd, cfg, audio))
    ref_wav = O.decode(sd, cfg, ref_codes)
for b3 in (True, False):
    ops.GEMM_B3 = b3
    codes = model.encode(audio.to(DEV)).cpu()
    wav = model.decode(ref_codes.to(DEV)).cpu()
    diff = codes != ref_codes
    print(f"GEMM_B3={b3}: code entries that differ from the oracle {int(diff.sum())} of {codes.numel()} (in {int(diff.any(1).sum())} of "
          f"{codes.shape[0] * codes.shape[2]} frames); decode of the oracle's codes: wav rel err {float((wav - ref_wav).abs().max() / ref_wav.abs().max()):.3e}",
          flush=True)
