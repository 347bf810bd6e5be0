"""Measure the per-level RVQ residual statistics used by rstnet_amd/synth.py (CPU oracle only).

    python tests/golden/calibrate_codebooks.py        # writes rstnet_amd/synth_calib.npz

Level l's codebook is  centre_l + scale_l * N_l  (N_l seeded noise); centre/scale are the mean
vector / std of the residual that reaches level l on a 4 x 10 s calibration clip.  Levels are
calibrated in order because the residual of level l depends on the codebooks of levels < l.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mimi_oracle as O  # noqa: E402
from rstnet_amd import synth  # noqa: E402

SEED = 0


def main():
    cfg = O.MimiConfig()
    K, D = cfg.rvq_layers, cfg.codebook_dim
    calib = {"seed": SEED, "center": torch.zeros(K, D), "scale": torch.full((K,), 0.08)}
    audio = synth.synth_audio(4, 240000, seed=7)
    with torch.no_grad():
        sd = synth.mimi_state_dict(SEED, calib=calib)
        z = O.encode_latent(sd, cfg, audio)
        for group, levels in (("quantizer.rvq_first", [0]), ("quantizer.rvq_rest", list(range(1, K)))):
            r = F.conv1d(z, sd[f"{group}.input_proj.weight"]).transpose(1, 2).reshape(-1, D)
            for j, lvl in enumerate(levels):
                calib["center"][lvl] = r.mean(0)
                calib["scale"][lvl] = (r - r.mean(0)).std()
                sd = synth.mimi_state_dict(SEED, calib=calib)
                emb = O.codebook(sd, f"{group}.vq.layers.{j}")
                idx = O.nearest_code(r, emb)
                print(f"level {lvl}: scale {calib['scale'][lvl]:.4f} |centre| {calib['center'][lvl].norm():.3f} "
                      f"distinct codes {idx.unique().numel()}/{r.shape[0]}")
                r = r - F.embedding(idx, emb)
    np.savez(os.path.join(ROOT, "rstnet_amd", "synth_calib.npz"), seed=np.int64(SEED),
             center=calib["center"].numpy(), scale=calib["scale"].numpy())


if __name__ == "__main__":
    main()
