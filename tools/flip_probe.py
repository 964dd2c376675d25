"""Diagnostic (GPU): where does the step differ from exact arithmetic on ITS OWN parameters?  For every iteration of a
fixture, the critic pass of the D-step is recomputed in fp64 from the engine's current parameters and its own fake rows;
reports (a) units whose LeakyReLU derivative differs (pre-activation within rounding of 0: "flips"), (b) the per-row
penalty error on rows with / without a flip.  Usage: python tools/flip_probe.py cfg4a"""
import sys
import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from tests.util import iter_inputs, load_iter_fixture, rebuild_state, unpack_masks  # noqa: E402
from wfd_gan_amd.engine import TrainEngine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4a"
fx, meta = load_iter_fixture(name)
st = rebuild_state(meta, fx)
B, S, C = meta["batch"], meta["seq_len"], meta["class_dim"]
eng = TrainEngine(S, C, meta["latent_dim"], B, meta["scaler_min"], meta["scaler_max"], {k: v.detach() for k, v in st.G.items()},
                  {k: v.detach() for k, v in st.D.items()}, st.DF, lr=meta["lr"], alpha=meta["alpha"], n_critic=meta["n_critic"])
for it in range(meta["n_iter"]):
    p = f"it{it}_"
    mr, mf, mg = (unpack_masks(fx, p + k, B) for k in ("mreal", "mfake", "mg"))
    tr, lab, z = iter_inputs(fx, meta, it)
    eng.set_batch(torch.from_numpy(tr), torch.from_numpy(lab), torch.from_numpy(z))
    eng.set_masks([torch.cat([a, b]) for a, b in zip(mr, mf)], mg)
    Dsd = {k: v.double().cpu() for k, v in eng.discriminator_state().items()}
    eng.d_step()
    torch.cuda.synchronize()
    X = eng.din[: 2 * B, : S + C].double().cpu().requires_grad_(True)       # [real ; fake] rows the critic saw
    masks = [torch.cat([a, b]).double() for a, b in zip(mr, mf)]
    h, flips = X, []
    for i, key in enumerate(("model.0", "model.3", "model.6")):
        zz = F.linear(h, Dsd[f"{key}.weight"], Dsd[f"{key}.bias"])
        d_gpu = eng.buf[f"D{i + 1}"].double().cpu()
        bad = (masks[i] > 0) & ((d_gpu > 0.5) != (zz > 0))
        flips.append(bad)
        print(f"it{it} layer{i + 1}: {int(bad.sum())} derivative flips of {bad.numel()}; min |z| at flips "
              f"{float(zz[bad].abs().max()) if bad.any() else 0:.2e}")
        h = F.leaky_relu(zz, 0.2) * masks[i]
    v = F.linear(h, Dsd["model.9.weight"], Dsd["model.9.bias"])
    g = torch.autograd.grad(v.sum(), X)[0][:, :S]
    n3 = g.pow(2).sum(1) ** 3
    pen = eng.buf["PEN"][: 2 * B].double().cpu()
    rel = ((pen - n3).abs() / n3.abs().max()).detach()
    rowflip = torch.zeros(2 * B, dtype=torch.bool)
    for b in flips:
        rowflip |= b.any(1)
    print(f"it{it}: per-row penalty error vs exact-on-own-params: rows with a flip ({int(rowflip.sum())}): "
          f"{float(rel[rowflip].max()) if rowflip.any() else 0:.2e}; rows without: {float(rel[~rowflip].max()) if (~rowflip).any() else 0:.2e}; "
          f"vs fixture n_r: {float(np.abs(pen[:B].numpy() - fx[p + 'n_r']).max() / np.abs(fx[p + 'n_r']).max()):.2e}")
    print(f"     critic scores vs exact: {float((eng.buf['V'][:2 * B].double().cpu() - v[:, 0]).abs().max()):.2e}")
    if it % meta["n_critic"] == 0:
        eng.g_step()
        torch.cuda.synchronize()
