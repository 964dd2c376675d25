#!/usr/bin/env python3
"""DF observer training on the MI355X kernels: same CLI, npz input and checkpoint as the reference's src/train_df.py
(positional feature path, --verbose, --cuda_id; 30 epochs, batch 128, Adamax, CrossEntropyLoss, MinMax-scaled
features without the burst-count column, stratified 90/10 split with random_state 42, checkpoint
``<dModelDir>/df_<name>.ckpt`` = ``DF.state_dict()``) -- the ``--f_model`` the GAN trainer consumes.

    python -m wfd_gan_amd.train_df <...>/feature/raw_feature_x.npz [--verbose] [--cuda_id 0]

Additive flags: --num_epochs, --batch_size, --seed, --out.  The training set is resident in HBM; every step is one
launch plan (wfd_gan_amd/df_engine.py); the only host syncs are the per-epoch accuracy read-outs.
"""
import argparse
import logging
import os
import random
from os.path import join

import numpy as np
import torch
from sklearn import preprocessing
from sklearn.model_selection import train_test_split

from . import common as cm
from . import utils
from .df_engine import DFTrainEngine
from .model import DF

num_epochs = 30          # src/train_df.py:19-20
batch_size = 128


def init_logger():
    logger = logging.getLogger("df")
    logger.setLevel(logging.DEBUG)
    if not logger.handlers:
        ch = logging.StreamHandler()
        ch.setFormatter(logging.Formatter(cm.LOG_FORMAT))
        logger.addHandler(ch)
    return logger


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="DF attack")
    p.add_argument("dir", metavar="<feature path>", help="Path to the directory of the extracted features")
    p.add_argument("--verbose", action="store_true", default=False, help="Enable verbose or not")
    p.add_argument("--cuda_id", type=int, default=7, help="The id number of cuda to be used.")
    # additive
    p.add_argument("--num_epochs", type=int, default=num_epochs)
    p.add_argument("--batch_size", type=int, default=batch_size)
    p.add_argument("--seed", type=int, default=None, help="seed for init / shuffling / dropout")
    p.add_argument("--out", default=None, help="checkpoint path (default <dModelDir>/df_<dataset name>.ckpt)")
    return p.parse_args(argv)


def main(argv=None):
    random.seed(0)
    logger = init_logger()
    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("wfd_gan_amd needs an MI355X (HIP device); there is no CPU path")
    dev_id = args.cuda_id if args.cuda_id < torch.cuda.device_count() else 0     # the reference's default is GPU 7
    device = torch.device(f"cuda:{dev_id}")
    torch.cuda.set_device(device)
    if args.seed is not None:
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)

    X, y = utils.load_dataset(args.dir)
    logger.info("Loaded dataset:{}, min burst:{} max burst:{}, min label:{}, max label:{}"
                .format(X.shape, X[:, 1:].min(), X[:, 1:].max(), y.min(), y.max()))
    y = y - y.min()
    if int(y.max()) >= len(np.unique(y)):
        raise IndexError("labels must be 0 .. num_classes - 1 after the shift by their minimum (got max {} for {} classes)"
                         .format(int(y.max()), len(np.unique(y))))
    X = preprocessing.MinMaxScaler().fit_transform(X)
    class_dim = len(np.unique(y))
    X = X[:, 1:]                       # the burst count is not a DF feature (src/train_df.py:77-80)
    seq_len = X.shape[1]
    assert seq_len > 1 and class_dim > 1
    logger.info("X shape {}, y shape {}, class num: {}".format((X.shape[0], 1, seq_len), y.shape, class_dim))
    X_train, X_test, y_train, y_test = train_test_split(X, y, test_size=0.1, random_state=42, stratify=y)
    Xtr = torch.from_numpy(X_train).float().to(device)
    ytr = torch.from_numpy(y_train).to(device)
    Xte = torch.from_numpy(X_test).float().to(device)
    yte = torch.from_numpy(y_test).to(device)

    model = DF(seq_len, class_dim)     # reference init (same RNG consumption as the reference's DF(length, num_classes))
    seed = args.seed if args.seed is not None else int(torch.initial_seed() & 0x7FFFFFFF)
    N = Xtr.shape[0]
    B = min(args.batch_size, N)
    eng = DFTrainEngine(seq_len, class_dim, B, model.state_dict(), device=device, seed=seed)
    n_full, tail_rows = N // B, N % B
    tail = DFTrainEngine(seq_len, class_dim, tail_rows, {}, device=device, seed=seed + 1, share=eng) \
        if tail_rows > 1 else None     # the DataLoader's ragged last batch; BatchNorm needs more than one row
    total_step = n_full + (1 if tail is not None else 0)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    loss_log = []
    for epoch in range(args.num_epochs):
        perm = torch.randperm(N, generator=gen, device=device)
        for step in range(total_step):
            e = eng if step < n_full else tail
            idx = perm[step * B: step * B + e.B]
            e.set_batch(Xtr[idx], ytr[idx])
            e.draw_masks()
            e.train_step()
            if args.verbose and (step + 1) % 50 == 0:
                logger.info("Epoch [{}/{}], Step [{}/{}], Loss: {:.4f}"
                            .format(epoch + 1, args.num_epochs, step + 1, total_step, e.loss()))
        loss_log.append(eng.loss())
        acc_train, acc_test = eng.accuracy(Xtr, ytr), eng.accuracy(Xte, yte)
        logger.info("Epoch [{}/{}] train acc: {:.4f} val accuracy: {:.4f}".format(
            epoch + 1, args.num_epochs, acc_train, acc_test))

    path = args.out
    if path is None:
        parts = os.path.abspath(args.dir).split("/")
        fname = parts[-3] if len(parts) >= 3 else "dataset"
        os.makedirs(cm.dModelDir, exist_ok=True)
        path = join(cm.dModelDir, "df_{}.ckpt".format(fname))
    torch.save({k: v.cpu() for k, v in eng.state_dict().items()}, path)
    logger.info("Model is saved to {}".format(path))
    return path, loss_log, (acc_train, acc_test)


if __name__ == "__main__":
    main()
