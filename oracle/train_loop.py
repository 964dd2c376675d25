"""CPU oracle, part 2: the whole-run loop of src/mlp_df_wgan_train.py:130-313.  TEST INFRASTRUCTURE ONLY
(see oracle/wfdgan_oracle.py for the rules).  Uses the same torch / sklearn utilities in the same order as the
reference script so that, under the same torch/numpy seeds, it consumes the global RNG streams identically
(DataLoader shuffles, dropout draws, np.random.normal) and can be compared artefact-for-artefact with a run of
the unmodified script (tests/golden/make_golden.py does that).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.utils.data as Data
from sklearn import preprocessing

from . import wfdgan_oracle as O


def df_acc_epoch(df_sd, X: List[np.ndarray], y: List[np.ndarray]) -> float:
    """test_DF_acc_epoch, src/mlp_df_wgan_train.py:57-79 (DataLoader(num_workers=1) draws one base seed
    from the global torch generator; shuffle=False)."""
    Xt = torch.from_numpy(np.array(X)[:, 1:]).float()[:, np.newaxis, :]
    yt = torch.from_numpy(np.array(y))
    loader = Data.DataLoader(dataset=Data.TensorDataset(Xt, yt), num_workers=1, batch_size=128, shuffle=False)
    total, correct = 0.0, 0
    with torch.no_grad():
        for bx, by in loader:
            pred = torch.max(O.df_forward(df_sd, bx.float()), 1)[1]
            total += by.size(0)
            correct += int((pred == by.long()).sum())
    return correct / total


def run_training(X: np.ndarray, y: np.ndarray, df_state: Dict[str, torch.Tensor], n_epochs: int,
                 batch_size: int = 64, lr: float = 2e-4, n_cpu: int = 0, latent_dim: int = 500,
                 n_critic: int = 3, alpha: float = 0.02, freq: int = 20):
    """Everything after dataset load, src/mlp_df_wgan_train.py:130-313, returning what the script writes to
    disk: per-epoch log scalars, loss lists, final G/D state, epoch dumps, scaler."""
    y = y - y.min()                                               # :130
    scaler = preprocessing.MinMaxScaler()
    Xs = scaler.fit_transform(X)                                  # :132-133
    Xt, yt = torch.from_numpy(Xs), torch.from_numpy(y)
    class_dim = len(np.unique(yt))
    seq_len = Xt.size(1)
    loader = Data.DataLoader(dataset=Data.TensorDataset(Xt, yt), num_workers=n_cpu, batch_size=batch_size,
                             shuffle=True)                        # :141-147
    # construction draws from the global torch generator in the order G, D, DF (:150-153)
    G = O.make_generator_state(seq_len, class_dim, latent_dim)
    D = O.make_discriminator_state(seq_len, class_dim)
    _ = O.make_df_state(seq_len - 1, class_dim)                   # constructed then overwritten by the ckpt, :153-154
    st = O.OracleState(seq_len, class_dim, latent_dim, scaler.data_min_[0], scaler.data_max_[0], G, D,
                       {k: v.clone() for k, v in df_state.items()}, lr, alpha, n_critic, "div")
    for p in st.g_params() + st.d_params():
        p.requires_grad_(True)
    st.opt_G = torch.optim.RMSprop(st.g_params(), lr=lr)
    st.opt_D = torch.optim.RMSprop(st.d_params(), lr=lr)

    logs, dumps = [], {}
    losses = {"generator": [], "discriminator": [], "dist": []}
    for epoch in range(n_epochs):
        tot_real, tot_fake, tot_c = [], [], []
        g_ep = f_ep = comb_ep = d_ep = w_ep = 0.0
        nb = len(loader)
        for i, (traces, c) in enumerate(loader):
            z = torch.from_numpy(np.random.normal(0, 1, (traces.shape[0], latent_dim))).float()   # :194
            do_g = (i % n_critic == 0)
            r = O.iteration(st, traces, c, z, do_g, keep=do_g)
            w_ep += r["w_dist"] / nb                              # :226
            d_ep += r["d_loss"] / nb                              # :227
            if do_g:
                tot_real.extend(traces.cpu().numpy())             # :241
                tot_fake.extend(r["fake_g"].numpy())              # :242
                tot_c.extend(c.numpy())                           # :243
                g_ep += r["g_loss"] / (nb // n_critic)            # :269-271
                f_ep += r["f_loss"] / (nb // n_critic)
                comb_ep += r["g_combined"] / (nb // n_critic)
        acc = df_acc_epoch(st.DF, tot_fake, tot_c)                # :278
        logs.append(dict(epoch=epoch + 1, d_loss=d_ep, g_loss=g_ep, f_loss=f_ep, combined=comb_ep,
                         df_acc=acc, w_dist=w_ep))
        stop = (w_ep <= O.W_DIST_THRESHOLD and acc >= 0.9)
        if epoch == 0 or (epoch + 1) % freq == 0 or stop:         # :286-297
            dumps[epoch + 1] = dict(x=scaler.inverse_transform(np.array(tot_real)),
                                    recon_x=scaler.inverse_transform(np.array(tot_fake)),
                                    label=np.array(tot_c))
        losses["generator"].append(g_ep)
        losses["discriminator"].append(d_ep)
        losses["dist"].append(w_ep)
        if stop:
            break
    return dict(logs=logs, losses=losses, G={k: v.detach() for k, v in st.G.items()},
                D={k: v.detach() for k, v in st.D.items()}, dumps=dumps, scaler=scaler,
                seq_len=seq_len, class_dim=class_dim)
