"""CLI mirror of the reference's train.py for the training path (flags at /root/reference/train.py:15-46,65-71).

    python -m dvae_amd.train --train true --dataset_fp=<root> --batch-size=8 --latent-size=32 --speaker_size=4 \\
        --lr=1e-4 --epochs=10 --report-interval=5 --mse_cof=10 --kl_cof=10 --log_dir=./results

Kept flags: --batch-size --latent-size --speaker_size --lr --epochs --report-interval --mse_cof --kl_cof --seed
--dataset_fp --log_dir --train --samples_length (honoured here; the reference parses it but hard-codes 64,
train.py:53).  Parsed-and-ignored flags of the reference (--hidden-size --alpha --normalize --beta_cof --style_cof
--sample-size --no-cuda --do-not-resume --log-interval) are accepted for command-line compatibility.
--convert (voice conversion + vocoder) is out of scope.
"""
import argparse
import json
import os

import torch
from torch.utils.data import DataLoader


def get_parse():
    p = argparse.ArgumentParser()
    p.add_argument("--batch-size", type=int, default=2)
    p.add_argument("--hidden-size", type=str, default="400")
    p.add_argument("--speaker_size", type=int, default=4)
    p.add_argument("--latent-size", type=int, default=32)
    p.add_argument("--lr", default=1e-3, type=float)
    p.add_argument("--epochs", type=int, default=11)
    p.add_argument("--no-cuda", action="store_true", default=False)
    p.add_argument("--dataset", default="VCTK")
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--log-interval", type=int, default=500)
    p.add_argument("--report-interval", type=int, default=11)
    p.add_argument("--sample-size", type=int, default=64)
    p.add_argument("--do-not-resume", action="store_true", default=False)
    p.add_argument("--normalize", action="store_true", default=False)
    p.add_argument("--beta_cof", default=0.1, type=float)
    p.add_argument("--mse_cof", default=10, type=float)
    p.add_argument("--kl_cof", default=10, type=float)
    p.add_argument("--style_cof", default=0.1, type=float)
    p.add_argument("--samples_length", default=64, type=int)
    p.add_argument("--alpha", default=0.01, type=float)
    p.add_argument("--dataset_fp", default="/root/VCTK-Corpus/Autovc-known-speakers", type=str)
    p.add_argument("--log_dir", default="./results", type=str)
    p.add_argument("--train", type=bool, default=False)
    p.add_argument("--convert", type=bool, default=False)
    p.add_argument("--graph", type=int, default=1, help="replay the step from a hipGraph (fixed batch shape)")
    return p


def get_dataset(dataset_fp, batch_size, samples_length=64, seed=None):
    from .data import SpeechDatasetGVAE
    ds = SpeechDatasetGVAE(dataset_fp, samples_length=samples_length, seed=seed)
    # drop_last keeps the batch shape fixed (hipGraph replay); the reference uses drop_last=False (train.py:55-56)
    return DataLoader(ds, batch_size=batch_size, pin_memory=True, shuffle=True, drop_last=True), ds


def main(argv=None):
    args = get_parse().parse_args(argv)
    from .model.disentangled_vae import ConvolutionalMulVAE
    rank = int(os.environ.get("RANK", "0"))
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed(args.seed + 7919 * rank)      # every rank its own reparameterisation-noise stream
    loader, _ = get_dataset(args.dataset_fp, args.batch_size, args.samples_length, seed=args.seed)
    os.makedirs(args.log_dir, exist_ok=True)
    with open(os.path.join(args.log_dir, "config.json"), "w") as fp:
        json.dump(vars(args), fp, indent=4)
    vsc = ConvolutionalMulVAE(args.dataset, args.samples_length, 80, args.latent_size, args.lr, args.alpha,
                              args.log_interval, args.normalize, speaker_size=args.speaker_size,
                              latent_dim=args.latent_size, beta=args.beta_cof, batch_size=args.batch_size,
                              mse_cof=args.mse_cof, kl_cof=args.kl_cof, style_cof=args.style_cof)
    if args.graph:
        vsc.enable_graph(True)
    hist = None
    if args.train:
        hist = vsc.run_training(loader, loader, args.epochs, args.report_interval, args.sample_size,
                                reload_model=not args.do_not_resume,
                                checkpoints_path=os.path.join(args.log_dir, "checkpoints"),
                                images_path=os.path.join(args.log_dir, "images"),
                                logs_path=os.path.join(args.log_dir, "logs"),
                                estimation_dir=os.path.join(args.log_dir, "images", "estimation"))
    if args.convert:
        raise SystemExit("--convert (mel conversion + vocoder) is outside the training hot path (SURVEY.md §8f-3)")
    return hist


if __name__ == "__main__":
    main()
