#!/usr/bin/env python3
"""Class-conditioned trace sampling on the MI355X: ``sample()`` of the reference's src/generate.py:48-64 in bulk.

The reference's generate.py is a per-file defence simulator around ``sample(c_ind)`` and is stale (it builds the
Generator with a 3-argument constructor); the working loader idiom is gan_wt.py:51-65 (model
directory with ``scaler.gz`` + ``generator_seqlen*_cls*_latentdim*.ckpt``).  This module keeps generate.py's CLI
flags, loads the model the gan_wt way, and exposes

    TraceSampler(model_dir).sample(c_ind)                -> 1-D int array   (generate.py:48-64 semantics)
    TraceSampler(model_dir).sample_many(labels)          -> (ints (n, S) int32, lens (n,))   bulk, on device

    python -m wfd_gan_amd.generate -m <model_dir> --n 1000000 --out traces.npz      # bulk mode (config 5)

With torch.distributed (torchrun) every rank samples its own shard of the label list; no collective is needed.
"""
import argparse
import glob
import os
import re
from os.path import join

import joblib
import numpy as np
import torch

from . import common as cm
from . import parallel
from .sampler import BulkSampler


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Create traces from a trained GAN generator.")
    # the reference's flags (src/generate.py:23-36); --train/--test are only needed by its file-defence driver
    p.add_argument("--train", required=False, help="Train dataset path (unused: the scaler is loaded from the model dir).")
    p.add_argument("--test", required=False, help="Test dataset directory (raw traces); unused in bulk mode.")
    p.add_argument("--model", "-m", required=True, help="Path to the saved model directory (generator ckpt + scaler.gz).")
    p.add_argument("--latent_dim", type=int, default=16, help="dimensionality of the latent space (read from the "
                                                              "checkpoint name when it carries one)")
    p.add_argument("--format", metavar="<suffix of files>", help="The suffix of files", default=".cell")
    # additive: bulk sampling
    p.add_argument("--n", type=int, default=10000, help="number of traces; labels cycle over the classes")
    p.add_argument("--chunk", type=int, default=32768, help="rows per device pass (0.65 of the f32 MFMA peak from 16384 on)")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--out", default=None, help="npz to write (traces int32 zero-padded, lens, labels)")
    p.add_argument("--operands", choices=["f32", "bf16x3", "bf16"], default="f32",
                   help="f32 = exact-fp32 MFMA (parity path); bf16 = bf16-operand MFMA (about 3.7x faster, not bit-exact)")
    p.add_argument("--mode", choices=["generate", "wt", "glue", "synth"], default="generate",
                   help="integer epilogue: generate.py:59-63 (default), gan_wt.py:79-87, gan_glue.py:114-124 (signed), "
                        "synthesize.py:86-90")
    return p.parse_args(argv)


class TraceSampler:
    def __init__(self, mdir: str, device="cuda:0", chunk: int = 32768, seed: int = 0, operands: str = "f32",
                 mode: str = "generate"):
        # the fitted MinMaxScaler: model/scaler.npz (data_min_ / data_max_ as plain arrays, written by this package's
        # trainer next to scaler.gz) is preferred because loading it executes nothing; scaler.gz is the reference's
        # format (joblib pickle, src/mlp_df_wgan_train.py:312): unpickling runs code from the file, so only point -m at
        # model directories you trust
        if os.path.exists(join(mdir, "scaler.npz")):
            from types import SimpleNamespace
            with np.load(join(mdir, "scaler.npz"), allow_pickle=False) as f:
                scaler = SimpleNamespace(data_min_=f["data_min"], data_max_=f["data_max"])
        else:
            scaler = joblib.load(join(mdir, "scaler.gz"))
        paths = glob.glob(join(mdir, "generator*.ckpt"))
        assert len(paths) == 1, f"expected exactly one generator checkpoint in {mdir}"
        info = re.match(cm.GENERATOR_NAME_REG, os.path.basename(paths[0]))
        self.seq_len, self.class_dim, self.latent_dim = int(info["seqlen"]), int(info["cls"]), int(info["latentdim"])
        sd = torch.load(paths[0], map_location="cpu", weights_only=True)
        self.scaler = scaler
        self.bulk = BulkSampler(sd, self.seq_len, self.class_dim, self.latent_dim, scaler.data_min_, scaler.data_max_,
                                chunk=chunk, device=device, seed=seed, operands=operands, mode=mode)
        self._one = None
        self._sd, self._dev, self._seed = sd, device, seed

    def sample(self, c_ind: int, z: np.ndarray = None) -> np.ndarray:
        """One trace for class c_ind: int64 vector, burst count first (generate.py:48-64)."""
        if self._one is None:
            self._one = BulkSampler(self._sd, self.seq_len, self.class_dim, self.latent_dim, self.scaler.data_min_,
                                    self.scaler.data_max_, chunk=1, device=self._dev, seed=self._seed + 1)
        zt = None if z is None else torch.as_tensor(z, dtype=torch.float32).reshape(1, -1)
        out, lens = self._one.generate(torch.tensor([c_ind]), zt)
        n = int(lens[0])
        x = out[0, :n].cpu().numpy().astype(np.int64)
        assert (x > 0).all()
        return x

    def sample_many(self, labels):
        return self.bulk.generate(labels)


def main(argv=None):
    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("wfd_gan_amd needs an MI355X (HIP device); there is no CPU path")
    rank, world, local = parallel.init_from_env()
    torch.cuda.set_device(local)
    ts = TraceSampler(args.model, device=f"cuda:{local}", chunk=args.chunk, seed=args.seed + rank,
                      operands=args.operands, mode=args.mode)
    a, b = parallel.shard_rows(args.n, rank, world)
    labels = torch.arange(a, b) % ts.class_dim
    out, lens = ts.sample_many(labels)
    if args.out:
        path = args.out if world == 1 else args.out.replace(".npz", f".rank{rank}.npz")
        np.savez_compressed(path, traces=out.cpu().numpy(), lens=lens.cpu().numpy(), labels=labels.numpy())
    print(f"rank {rank}: sampled {b - a} traces, mean length {lens.float().mean().item():.1f}")
    # sampling issues no collective, but the chunk graphs were captured next to a live process group: same ordered end as the
    # trainer (sampler and graphs first, then barrier -> synchronize -> destroy_process_group)
    del ts, out, lens
    parallel.shutdown()


if __name__ == "__main__":
    main()
