"""Offline dataset tokenisation on one GPU per process (the CLI of MLLM_v2/egs/pretraining/local/offline_codec_tokenization.py
for `--tokenizer mimi`, launched once per GPU by extract_token.sh:98-105 with --rank JOB):

    python tools/offline_codec_tokenization.py --input-file wav.JOB.scp --output-file codec.JOB.pt --tokenizer mimi --rank JOB \\
        --weights tokenizer-e351c8d8-checkpoint125.safetensors [--batch-seconds 600]
"""
import argparse
import logging
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rstnet_amd.codec import offline  # noqa: E402


def get_parser():
    p = argparse.ArgumentParser(description="convert a data list, do tokenization and save as a torch .pt file",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--input-file", type=str, default=None, help="text file in the format <example_id> <path>")
    p.add_argument("--wav-scp", type=str, default=None, help="kaldi wav.scp file (plain paths)")
    p.add_argument("--output-file", type=str, required=True, help="torch .pt dict {example_id: int16 codes [8, F]}")
    p.add_argument("--tokenizer", type=str, default="mimi", choices=["mimi"], help="what tokenizer to use")
    p.add_argument("--rank", type=int, default=1, help="1-based job id; GPU = (rank - 1) %% device_count")
    p.add_argument("--weights", type=str, required=True, help="Mimi checkpoint (.safetensors / torch.save'd {'model': ...})")
    p.add_argument("--batch-seconds", type=float, default=600.0, help="padded audio per encode batch")
    p.add_argument("--chunk-size", type=int, default=256, help="utterances read before sorting into batches")
    p.add_argument("--strict", action="store_true", help="exit with status 1 when any listed utterance produced no codes "
                   "(default: log and skip it, as the reference does)")
    return p


def main(argv=None):
    logging.basicConfig(stream=sys.stdout, level=logging.INFO, format="%(asctime)s %(levelname)s [%(filename)s:%(lineno)d] %(message)s")
    args = get_parser().parse_args(argv)
    assert (args.input_file is None) != (args.wav_scp is None), "give exactly one of --input-file / --wav-scp"
    from rstnet_amd.codec.loaders import get_mimi
    from rstnet_amd.codec.tokenizer import MimiTokenizer
    device = torch.device("cuda", offline.device_index(args.rank, torch.cuda.device_count()))
    torch.cuda.set_device(device)     # kernels, graphs and scratch of this process all live on the rank's GPU
    logging.info(f"Using device: {device}")
    tokenizer = MimiTokenizer(get_mimi(args.weights, device))
    items = offline.read_list(args.input_file or args.wav_scp)
    t0 = time.time()
    skipped = []
    data = offline.tokenize_list(tokenizer, items, chunk_size=args.chunk_size, max_batch_seconds=args.batch_seconds, skipped=skipped)
    torch.save(data, args.output_file)
    frames = sum(v.shape[1] for v in data.values())
    logging.info(f"processed {len(data)} / {len(items)} examples, {frames} frames in {time.time() - t0:.1f} s")
    if skipped:
        logging.warning(f"{len(skipped)} utterances produced no codes: {skipped[:10]}{' ...' if len(skipped) > 10 else ''}")
        if args.strict:
            return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
