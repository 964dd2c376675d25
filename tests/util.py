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


_SYNTH_CACHE = {}


def iter_inputs(fx, meta, it):
    """(traces float32 (B,S), labels int64 (B,), z float32 (B,Z)) of iteration ``it``: stored in the fixture, or -- for the
    large config-4 fixtures -- re-derived from the recipe make_golden.py recorded (dataset seed and arguments, row
    indices, z seed) and verified against the stored checksums."""
    p = f"it{it}_"
    if p + "traces" in fx:
        return fx[p + "traces"], fx[p + "labels"], fx[p + "z"]
    key = json.dumps(meta["synth_args"], sort_keys=True)
    if key not in _SYNTH_CACHE:
        X, y = O.synth_dataset(**meta["synth_args"])
        scale, min_ = O.scaler_params(X.min(0), X.max(0))
        _SYNTH_CACHE[key] = (X * scale + min_, y)
    Xs, y = _SYNTH_CACHE[key]
    rows = fx[p + "rows"]
    traces = Xs[rows].astype(np.float32)
    zr = np.random.RandomState(meta["z_seed"])
    for _ in range(it + 1):
        z64 = zr.normal(0, 1, (meta["batch"], meta["latent_dim"]))
    z = torch.Tensor(z64).numpy()
    assert abs(traces.astype(np.float64).sum() - float(fx[p + "traces_sum"])) <= 1e-9 * abs(float(fx[p + "traces_sum"]))
    assert abs(float(np.abs(z.astype(np.float64)).sum()) - float(fx[p + "z_sum"])) <= 1e-9 * float(fx[p + "z_sum"])
    return traces, y[rows], z


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


def state_from_engine(eng, meta, df_state, dtype=torch.float64):
    """An oracle state holding the ENGINE's current generator / critic parameters and BatchNorm buffers (cast to
    ``dtype``): the oracle then computes what exact arithmetic gives on the device's own parameters."""
    st = O.make_state(meta["seq_len"], meta["class_dim"], meta["latent_dim"], meta["scaler_min"], meta["scaler_max"], 0,
                      df_state=df_state, lr=meta["lr"], alpha=meta["alpha"], n_critic=meta["n_critic"], dtype=dtype)
    load_engine_params(st, eng, "GD")
    return st


def load_engine_params(st, eng, which="GD"):
    with torch.no_grad():
        if "G" in which:
            for k, v in eng.generator_state().items():
                st.G[k].copy_(v.to(st.G[k].dtype))
        if "D" in which:
            for k, v in eng.discriminator_state().items():
                st.D[k].copy_(v.to(st.D[k].dtype))
