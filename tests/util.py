"""Helpers shared by the parity tests: fixture loading and seed-addressed state re-derivation."""
import json
import os

import numpy as np
import torch

from oracle import wfdgan_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def checksum(sd):
    return float(sum(float(v.detach().double().abs().sum()) for v in sd.values() if v.is_floating_point()))


def load_iter_fixture(name):
    fx = np.load(os.path.join(GOLDEN, f"iter_{name}.npz"))
    meta = json.loads(str(fx["meta"]))
    return fx, meta


def rebuild_state(meta, fx, dtype=torch.float32, penalty="div"):
    """Re-derive the fixture's initial weights from its seeds (weights are not stored) and verify checksums."""
    S, C, Z = meta["seq_len"], meta["class_dim"], meta["latent_dim"]
    torch.manual_seed(meta["df_seed"])
    df_sd = O.make_df_state(S - 1, C)
    O.perturb_df_bn(df_sd, meta["df_bn_seed"])
    st = O.make_state(S, C, Z, meta["scaler_min"], meta["scaler_max"], meta["seed"], df_state=df_sd, lr=meta["lr"],
                      alpha=meta["alpha"], n_critic=meta["n_critic"], penalty=penalty, dtype=dtype)
    if dtype == torch.float32:
        assert abs(checksum(st.G) - float(fx["g_init_checksum"])) < 1e-6 * float(fx["g_init_checksum"]), \
            "generator init drifted from the fixture (different torch build?)"
        assert abs(checksum(st.D) - float(fx["d_init_checksum"])) < 1e-6 * float(fx["d_init_checksum"])
        assert abs(checksum(st.DF) - float(fx["df_checksum"])) < 1e-6 * float(fx["df_checksum"])
    return st


def unpack_masks(fx, prefix, batch):
    out = []
    for li, n in enumerate((2048, 1024, 512)):
        key = f"{prefix}{li}"
        if key not in fx:
            return None
        bits = np.unpackbits(fx[key])[: batch * n].reshape(batch, n)
        out.append(torch.from_numpy(bits.astype(np.float32) * 1.25))
    return out


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))
