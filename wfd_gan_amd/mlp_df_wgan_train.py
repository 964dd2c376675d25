#!/usr/bin/env python3
"""cGAN + DF-observer trainer on the MI355X kernels: same CLI, npz input and output files as the reference's
src/mlp_df_wgan_train.py (flags :38-54, artefacts :104-112,286-313).

    python -m wfd_gan_amd.mlp_df_wgan_train -d <...>/feature/raw_feature_x.npz --f_model df.ckpt [--n_epochs 800 ...]

Additive flags (defaults reproduce the reference): --penalty {div,gp}, --seed, --no_graph, --dp {strong,weak}.

Multi-GPU: launch with torchrun, one process per GPU (``python -m torch.distributed.run --nproc-per-node N -m
wfd_gan_amd.mlp_df_wgan_train ...``).  ``--dp strong`` (default) splits every batch of --batch_size rows over the
ranks with synchronised BatchNorm statistics and a global selected-row count: the update equals the single-GPU one.
``--dp weak`` gives every rank its own --batch_size rows (DDP semantics: local BatchNorm, averaged gradients).  All
ranks hold the dataset and the same epoch permutation; rank 0 writes the artefacts.
Per iteration nothing leaves the GPU: the dataset is resident in HBM, batches / z / dropout masks are produced on
device, the D-step and G-step are HIP-graph replays, and the epoch's loss sums accumulate in device memory (the
reference's 2-5 ``.item()`` syncs and per-G-step D2H copies are gone; per epoch one sync reads the loss sums and one the
DF-accuracy count).
"""
import argparse
import json
import os
from os.path import join
from time import strftime

import joblib
import numpy as np
import torch

from . import common as cm
from . import parallel, utils
from ._lib import lib
from .engine import GroupedTrainer, TrainEngine
from .model import DF, Discriminator, Generator

w_dist_threshold = 0.05          # src/mlp_df_wgan_train.py:24

# --clip caps (walkie-talkie values the reference ships, :34-36)
MAX_BURST_LENGTH = 550
MAX_OUTGOING_SIZE = 13
MAX_INCOMING_SIZE = 122


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("-d", "--dir", required=True, help="Dataset directory.")
    p.add_argument("--clip", action="store_true", default=False,
                   help="Whether to clip the burst size of the dataset before training")
    p.add_argument("--f_model", type=str, required=True, help="The directory of the pre-trained DF.")
    p.add_argument("--n_epochs", type=int, default=800, help="number of epochs of training")
    p.add_argument("--batch_size", type=int, default=64, help="size of the batches")
    p.add_argument("--lr", type=float, default=0.0002, help="learning rate")
    p.add_argument("--n_cpu", type=int, default=2, help="kept for CLI compatibility; batches are built on the GPU")
    p.add_argument("--latent_dim", type=int, default=500, help="dimensionality of the latent space")
    p.add_argument("--n_critic", type=int, default=3, help="number of training steps for discriminator per iter")
    p.add_argument("--alpha", type=float, default=0.02, help="Max ratio of f loss")
    p.add_argument("--freq", type=int, default=20, help="Checkpoint every freq epochs")
    p.add_argument("--cuda_id", type=int, default=0, help="GPU ID")
    # additive
    p.add_argument("--penalty", choices=["div", "gp"], default="div",
                   help="div = W-div (what the reference runs); gp = interpolated WGAN-GP (utils.py:64-84)")
    p.add_argument("--operands", choices=["f32", "bf16x3", "bf16"], default="f32",
                   help="contraction operands: f32 = f32-input MFMA, exact f32 products (default, the parity path); "
                        "bf16x3 = split-bf16 on the bf16 matrix pipe (products ~1e-5 relative); bf16 = bf16 operands")
    p.add_argument("--seed", type=int, default=None, help="seed for init / shuffling / device RNG")
    p.add_argument("--no_graph", action="store_true", help="launch kernels eagerly instead of HIP-graph replay")
    p.add_argument("--no_group", action="store_true",
                   help="one iteration per launch group (default: the generator forwards of the n_critic iterations "
                        "that share generator parameters run as one stacked pass)")
    p.add_argument("--dp", choices=["strong", "weak"], default="strong",
                   help="torchrun launches only: strong = --batch_size is the global batch (SyncBN, result-equal to one "
                        "GPU); weak = --batch_size rows per GPU (local BatchNorm, averaged gradients)")
    return p.parse_args(argv)


def clip_dataset(X: np.ndarray, device) -> np.ndarray:
    """--clip on the device (float64, in place), then the reference's sanity asserts (:122-125)."""
    Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(device)
    lib().clip_bursts(Xd.data_ptr(), Xd.shape[0], Xd.shape[1], float(MAX_BURST_LENGTH), float(MAX_OUTGOING_SIZE),
                      float(MAX_INCOMING_SIZE), torch.cuda.current_stream().cuda_stream)
    X = Xd.cpu().numpy()
    assert X[:, 0].max() == MAX_BURST_LENGTH
    assert X[:, 1::2].max() == MAX_OUTGOING_SIZE
    assert X[:, 2::2].max() == MAX_INCOMING_SIZE
    return X


def df_accuracy(f_model: DF, fakes: torch.Tensor, labels: torch.Tensor, comm=None) -> float:
    """test_DF_acc_epoch (:57-79): DF argmax accuracy over this epoch's generated traces (summed over the ranks' shares in
    data-parallel runs).  The correct-prediction count accumulates in device memory
    (wfd_argmax_correct: first maximum, as torch.argmax) and is read once."""
    L = lib()
    dev = fakes.device
    correct = torch.zeros(1, dtype=torch.int32, device=dev)
    lab32 = labels.to(torch.int32).contiguous()
    total = fakes.shape[0]
    st = torch.cuda.current_stream(dev).cuda_stream
    C = f_model.num_classes
    # The reference walks the traces in batches of 128; in eval mode every row's logits are independent of the batch it is in,
    # so the batch size is a throughput knob only: at 128 rows the observer's ~25 launches are latency-bound (33 k traces =
    # 261 batches), at 2048 they are not -- 0.490 -> 0.467 s per config-2 epoch.  WFD_DF_ACC_BATCH=128 restores the walk.
    nb = max(1, int(os.environ.get("WFD_DF_ACC_BATCH", "2048")))
    for a in range(0, total, nb):
        x = fakes[a:a + nb, 1:]
        logits = f_model(x.reshape(x.shape[0], 1, x.shape[1]))
        L.argmax_correct(logits.data_ptr(), logits.stride(0), x.shape[0], C, lab32[a:a + nb].data_ptr(), correct.data_ptr(), st)
    n_ok = int(correct.item())
    if comm is not None and comm.world > 1:
        t = comm.all_reduce_sum_(torch.tensor([n_ok, total], dtype=torch.float64, device=dev))
        n_ok, total = t.tolist()
    return n_ok / max(total, 1)


def epoch_dump(scaler, X_rows: np.ndarray, fakes: torch.Tensor):
    """The two arrays of checkpoint/epoch_N.npz (:288-297) on the device: ``x`` = inverse_transform(transform(real rows)) in
    float64, ``recon_x`` = inverse_transform(generated rows) in float32 -- bit-identical to sklearn's in-place passes."""
    L = lib()
    dev = fakes.device
    st = torch.cuda.current_stream(dev).cuda_stream
    sd, md = torch.from_numpy(scaler.scale_.copy()).to(dev), torch.from_numpy(scaler.min_.copy()).to(dev)
    Xd = torch.from_numpy(np.ascontiguousarray(X_rows, dtype=np.float64)).to(dev)
    N, S = Xd.shape
    real = torch.empty_like(Xd)
    L.minmax_roundtrip_f64(Xd.data_ptr(), N, S, sd.data_ptr(), md.data_ptr(), real.data_ptr(), st)
    fk = fakes.contiguous()
    recon = torch.empty(fk.shape[0], S, dtype=torch.float32, device=dev)
    L.minmax_inverse_f32(fk.data_ptr(), fk.stride(0), fk.shape[0], S, sd.data_ptr(), md.data_ptr(), recon.data_ptr(), st)
    return real.cpu().numpy(), recon.cpu().numpy()


class _Quiet:
    """Stand-in logger of the ranks that do not write the log."""
    def info(self, *a, **k): pass
    debug = warning = info


def main(argv=None):
    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("wfd_gan_amd needs an MI355X (HIP device); there is no CPU path")
    rank, world, local = parallel.init_from_env()
    device = torch.device(f"cuda:{local if world > 1 else args.cuda_id}")
    torch.cuda.set_device(device)
    if args.seed is not None:
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)
    comm = parallel.Comm()

    pardir = modeldir = checkpointdir = None
    logger = _Quiet()
    if rank == 0:
        pardir, modeldir, checkpointdir = utils.init_directory(args.dir,
                                                               tag="training_{}".format(strftime("%m%d_%H%M%S")))
        logger = utils.init_logger("gan", join(pardir, "log.txt"))
        logger.info(args)
        logger.debug("Output to {}".format(pardir))
        with open(join(pardir, "args.json"), "wt") as f:
            json.dump(vars(args), f, indent=4)

    X, y = utils.load_dataset(args.dir)
    logger.info("Loaded dataset:{}, min burst:{} max burst:{}, min label:{}, max label:{}"
                .format(X.shape, X[:, 1:].min(), X[:, 1:].max(), y.min(), y.max()))
    if args.clip:
        X = clip_dataset(X, device)
        if rank == 0:
            np.savez_compressed(args.dir.split(".npz")[0] + "_clip.npz", features=X, labels=y)
        logger.info("Save back  the clipped dataset to {}".format(args.dir.split(".npz")[0] + "_clip.npz"))
    y = y - y.min()
    if int(y.max()) >= len(np.unique(y)):
        # np.eye(class_dim)[c] (:185) fails the same way on labels that are not 0 .. class_dim - 1 after the shift
        raise IndexError("index {} is out of bounds for axis 0 with size {}".format(int(y.max()), len(np.unique(y))))

    # MinMaxScaler fit + transform + the loop's .float() on the device, once (bit-identical to the host path)
    Xs_dev, scaler = utils.fit_transform_device(X, device)
    class_dim = len(np.unique(y))
    N, seq_len = Xs_dev.shape
    assert seq_len > 1 and class_dim > 1
    logger.info("X shape {}, y shape {}, class num: {}".format(tuple(Xs_dev.shape), y.shape, class_dim))

    # modules in the reference's construction order (:150-153) so that a seed gives the same initial weights
    generator = Generator(seq_len, class_dim, args.latent_dim, scaler_min=scaler.data_min_[0],
                          scaler_max=scaler.data_max_[0], is_gpu=True)
    discriminator = Discriminator(seq_len, class_dim)
    f_model = DF(seq_len - 1, class_dim)
    f_model.load_state_dict(torch.load(args.f_model, map_location="cpu", weights_only=True))
    f_model.eval().to(device)

    exact = world > 1 and args.dp == "strong"
    if exact and min(args.batch_size, N) % world:
        raise SystemExit(f"--dp strong needs --batch_size divisible by the number of ranks ({world})")
    B = min(args.batch_size, N) // (world if exact else 1)        # rows per rank
    Bg = B * world                                                # rows per iteration over all ranks
    seed = args.seed if args.seed is not None else int(torch.initial_seed() & 0x7FFFFFFF)
    if world > 1:       # one seed for the permutation on every rank (rank 0's); the device RNG streams differ per rank
        t = torch.tensor([seed], dtype=torch.int64, device=device)
        parallel.broadcast_state_([t])
        seed = int(t.item())
    # single GPU: RMSprop runs inside the weight-gradient stores (engine.TrainEngine fuse_opt); data parallel keeps the flat pass
    # WFD_FORCE_DP=1: the data-parallel code path (bucketed collectives, flat optimizer pass) on a 1-rank group (test rigs)
    force_dp = os.environ.get("WFD_FORCE_DP") == "1" and torch.distributed.is_initialized()
    dp = dict(world_size=world, rank=rank, sync_bn=exact, operands=args.operands, fuse_opt=(world == 1 and not force_dp),
              force_sync=force_dp)
    rng_seed = seed + 1000003 * rank
    eng = TrainEngine(seq_len, class_dim, args.latent_dim, B, scaler.data_min_[0], scaler.data_max_[0],
                      generator.state_dict(), discriminator.state_dict(), f_model.state_dict(), lr=args.lr,
                      alpha=args.alpha, n_critic=args.n_critic, penalty=args.penalty, lambda_gp=cm.lambda_gp,
                      device=device, seed=rng_seed, **dp)
    if world > 1:       # identical replicas whatever the host seeding was
        parallel.broadcast_state_([eng.Gp.flat, eng.Dp.flat] + list(eng.Gbuf.values()))
    eng.attach_dataset(Xs_dev, torch.from_numpy(y), shuffle_seed=seed)
    del Xs_dev
    eng.auto_reshuffle = False        # the epoch loop below owns the permutation (ragged last batch)
    if not args.no_graph:
        eng.capture_graphs()
    # iterations 1..3, 4..6, ... of an epoch (D D D G: same generator for the three D-steps) go out as groups
    grp = None
    if not exact and not args.no_group and args.n_critic >= 2 and B <= 128:
        grp = GroupedTrainer(eng, graphs=not args.no_graph)
        grp.auto_reshuffle = False
    n_full, tail_global = N // Bg, N % Bg
    tail_rows = tail_global // world
    tail = None
    # the DataLoader's ragged last batch (drop_last=False); BatchNorm needs > 1 row.  Data-parallel runs keep it only
    # when it splits evenly over the ranks.
    if tail_rows > 1 and tail_global % world == 0:
        tail = TrainEngine(seq_len, class_dim, args.latent_dim, tail_rows, scaler.data_min_[0], scaler.data_max_[0],
                           {}, {}, {}, lr=args.lr, alpha=args.alpha, n_critic=args.n_critic, penalty=args.penalty,
                           lambda_gp=cm.lambda_gp, device=device, seed=rng_seed + 1, share=eng, **dp)
    elif tail_global:
        logger.info("dropping the last {} rows of every epoch (ragged batch)".format(tail_global))
    n_batches = n_full + (1 if tail is not None else 0)
    n_gsteps = (n_batches + args.n_critic - 1) // args.n_critic
    cap = n_gsteps * B
    ep_fake = torch.zeros(cap, seq_len, dtype=torch.float32, device=device)
    ep_idx = torch.zeros(cap, dtype=torch.int64, device=device)
    y_dev = torch.from_numpy(y).to(device)

    losses = {"generator": [], "discriminator": [], "dist": []}
    for epoch in range(args.n_epochs):
        eng.buf["dacc"].zero_()
        eng.buf["gacc"].zero_()
        fill = 0
        i = 0
        nc = args.n_critic
        while i < n_batches:
            if grp is not None and i % nc == 1 and i + nc <= n_full:
                # iterations i .. i + nc - 1; the G-step belongs to the last one
                if i == 1:
                    eng.gf_state[1] = Bg                   # cursor of the stacked heads: batch 1 of this permutation
                grp.run_group()
                i += nc
                j = i - 1
                ep_fake[fill:fill + B].copy_(grp.slot(nc - 1).fake_rows[:, :seq_len])
                ep_idx[fill:fill + B].copy_(eng.perm[j * Bg + rank * B: j * Bg + (rank + 1) * B])
                fill += B
                if i >= n_full or i + nc > n_full:
                    eng.iter_state[1] = i * Bg             # the remaining full batches run one by one
                continue
            do_g = (i % nc == 0)
            if i < n_full:
                eng.iteration(i)
                src, rows, idx = eng, B, eng.perm[i * Bg + rank * B: i * Bg + (rank + 1) * B]
            else:
                idx = eng.perm[n_full * Bg + rank * tail_rows: n_full * Bg + (rank + 1) * tail_rows]
                tail.set_batch(eng.data_X[idx], y_dev[idx], torch.zeros(tail_rows, args.latent_dim))
                tail.draw_z()
                tail.draw_masks(do_g)
                tail.d_step()
                if do_g:
                    tail.g_step()
                src, rows = tail, tail_rows
            if do_g:        # total_real / total_fake / total_c of :241-243, kept on device
                ep_fake[fill:fill + rows].copy_(src.fake_rows[:, :seq_len])
                ep_idx[fill:fill + rows].copy_(idx)
                fill += rows
            i += 1
        eng.reshuffle()
        eng.check_length_index()
        dacc, gacc = eng.epoch_sums()                     # the epoch's only host sync (and loss collective)
        if eng._mesh is not None:
            eng._mesh.check()                             # peer-to-peer exchange: no wait for a peer may have timed out
        d_ep, w_ep = dacc[0] / n_batches, dacc[1] / n_batches
        div_g = max(n_batches // args.n_critic, 1)
        g_ep, f_ep, comb_ep = gacc[0] / div_g, gacc[1] / div_g, gacc[2] / div_g
        df_acc = df_accuracy(f_model, ep_fake[:fill], y_dev[ep_idx[:fill]], comm)
        logger.info(
            "[Epoch %2d/%2d] [D loss: %.4f] [G loss: %.4f + %.4f * %.4f = %.4f] [DF acc: %.4f] [w dist: %.4f]"
            % (epoch + 1, args.n_epochs, d_ep, g_ep, args.alpha, f_ep, comb_ep, df_acc, w_ep))
        stop = w_ep <= w_dist_threshold and df_acc >= 0.9
        if epoch == 0 or (epoch + 1) % args.freq == 0 or stop:
            idx_all, fake_all = ep_idx[:fill], ep_fake[:fill]
            if world > 1:                                 # every rank's share of the epoch's generated traces
                idx_all = comm.all_gather(idx_all).reshape(-1)
                fake_all = comm.all_gather(fake_all).reshape(-1, seq_len)
            if rank == 0:
                sel = idx_all.cpu().numpy()
                total_real, total_fake = epoch_dump(scaler, X[sel], fake_all)            # float64 / float32, as :291-292
                np.savez_compressed(join(checkpointdir, "epoch_{}".format(epoch + 1)), x=total_real,
                                    recon_x=total_fake, label=y[sel])
            if eng._mesh is not None:
                eng._mesh.wait_for_peers()                # nobody waits for rank 0's file INSIDE the next exchange kernel
        losses["generator"].append(g_ep)
        losses["discriminator"].append(d_ep)
        losses["dist"].append(w_ep)
        if stop:
            logger.info("Early stopping since the w-dist = {}, df acc = {}.".format(w_ep, df_acc))
            break

    if rank == 0:       # the write block of src/mlp_df_wgan_train.py:305-312
        np.savez_compressed(join(checkpointdir, "loss.npz"), generator=losses["generator"],
                            discriminator=losses["discriminator"], dist=losses["dist"])
        torch.save({k: v.cpu() for k, v in eng.generator_state().items()},
                   join(modeldir, "generator_seqlen{}_cls{}_latentdim{}.ckpt".format(seq_len, class_dim, args.latent_dim)))
        torch.save({k: v.cpu() for k, v in eng.discriminator_state().items()}, join(modeldir, "discriminator.ckpt"))
        logger.info("Model saved at {}".format(modeldir))
        joblib.dump(scaler, join(modeldir, "scaler.gz"))
        # the same fit as plain arrays: generate.py reads this one without unpickling anything
        np.savez(join(modeldir, "scaler.npz"), data_min=scaler.data_min_, data_max=scaler.data_max_)
        logger.info("Scaler saved at {}".format(modeldir))
    # Ordered end of a data-parallel run (parallel.shutdown): the other ranks wait at the barrier until rank 0 has written;
    # everything that holds a HIP graph with captured RCCL nodes goes BEFORE the communicator does
    del grp, tail, eng, comm
    parallel.shutdown()
    return pardir if rank == 0 else None


if __name__ == "__main__":
    main()
