"""Soak run of the end-to-end streaming frame (Mimi encode -> LMGen.step -> Mimi decode as ONE captured graph, tools/../rstnet_amd/pipeline.py)
at the Moshi-7B shape: thousands of frames through the persistent launches (depth frame, codec transformer frames, the chained RVQ), past
the 250-slot wrap of the codec rings and the 3000-slot wrap of the temporal rings.  Reports the frame time per thousand frames, the repair
counters of the persistent launches (must stay 0 on a device this process owns) and that every waveform sample stays finite.

    python tools/soak_e2e.py [--streams 1] [--frames 3300]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--frames", type=int, default=3300)
    args = ap.parse_args()
    from rstnet_amd import ops, synth
    from rstnet_amd.codec.mimi import MimiCodec
    from rstnet_amd.lm.model import LMGen, LMModel
    from rstnet_amd.pipeline import StreamingPipeline
    dev = torch.device("cuda", 0)
    cfg = dict(synth.LM_MOSHI_7B)
    model = LMModel.from_state_dict(synth.lm_state_dict(cfg, seed=0, device=str(dev)), cfg)
    mimi = MimiCodec.from_state_dict(synth.mimi_state_dict(0)).to(dev)
    gen = LMGen(model, use_sampling=True)
    B = args.streams
    chunk = 100                                         # frames of audio generated at a time
    torch.manual_seed(1)
    bad = 0
    peak = 0.0
    with StreamingPipeline(mimi, gen, B) as pipe:
        done = 0
        t_block, f_block = time.perf_counter(), 0
        while done < args.frames:
            pcm = synth.synth_audio(B, 1920 * chunk, seed=300 + done).to(dev)
            outs = []
            for s in range(chunk):
                out = pipe.step(pcm[:, :, s * 1920:(s + 1) * 1920].contiguous())
                if out is not None:
                    outs.append(out)
            if outs:
                w = torch.cat(outs, -1)
                bad += int((~torch.isfinite(w)).sum())
                peak = max(peak, float(w.abs().max()))
            done += chunk
            f_block += chunk
            if done % 1000 == 0 or done >= args.frames:
                torch.cuda.synchronize()
                dt = time.perf_counter() - t_block
                print(f"frames {done - f_block:5d} .. {done:5d}: {dt / f_block * 1e3:7.3f} ms per frame incl. host loop and input generation "
                      f"({B} stream{'s' if B > 1 else ''}); fused graph {'on' if pipe._fused is not None and pipe._fused.graph is not None else 'off'}; "
                      f"repairs so far {ops.persistent_repairs(dev)}", flush=True)
                t_block, f_block = time.perf_counter(), 0
        pos = int(model.transformer._streaming_state.pos)
    print(f"{args.frames} frames, {B} stream(s): temporal ring position {pos} (capacity {cfg['context']}), non-finite samples {bad}, "
          f"peak |wav| {peak:.3f}, frames repaired by the persistent launches' repair paths {ops.persistent_repairs(dev)}, "
          f"persistent path retired: {dev in ops._persist_off}")
    assert bad == 0 and ops.persistent_repairs(dev) == 0 and dev not in ops._persist_off


if __name__ == "__main__":
    main()
