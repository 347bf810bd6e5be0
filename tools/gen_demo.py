"""Streaming TTS-style generation at the BASELINE configs[4] shape (Qwen-0.5B-shaped GPT + LoRA, random weights): one
utterance through rstnet_amd.lm.generate.InferenceImp -- prompt prefill, then one graph-replayed global step + one depth graph
per frame -- and the resulting frames/s.  `python tools/gen_demo.py [frames] [prompt_len]`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rstnet_amd import synth  # noqa: E402
from rstnet_amd.lm.generate import GenIds, InferenceImp  # noqa: E402
from rstnet_amd.lm.gpt import GPT, Config  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n_text = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    cfg_d = dict(synth.GPT_QWEN_0_5B)
    dev = "cuda:0"
    model = GPT.from_state_dict(synth.gpt_state_dict(cfg_d, 0, device=dev), Config.from_dict(cfg_d))
    K, L = cfg_d["n_q"] + 1, n_text + frames
    g = torch.Generator().manual_seed(0)
    seq = torch.randint(0, 2048, (K, L), generator=g)
    seq[0, :n_text] = torch.randint(0, 100000, (n_text,), generator=g)
    seq[0, n_text:] = 128002                     # text_empty_token: the frames to generate
    imp = InferenceImp(None, model, "sample", temp_text=0.7, top_k_text=25, temp=0.8, top_k=250, task_name="TTS", ids=GenIds())
    torch.manual_seed(0)
    imp.generate(seq[:, : n_text + 8])           # warm-up (packing of weights, graph capture paths)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = imp.generate(seq)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = out["frames"].shape[0]
    print(f"generated {n} frames after a {n_text + 1}-position prefill in {dt * 1e3:.1f} ms: {n / dt:.1f} frames/s "
          f"({n / dt / 12.5:.1f}x real time); codes {tuple(out['codes'].shape)} max id {int(out['frames'].max())}")


if __name__ == "__main__":
    main()
