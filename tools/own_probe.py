"""Diagnostic (GPU): error quantiles of the engine's gradients vs the fp64 oracle on the engine's own parameters, next to
the fp32 oracle's (the reference's arithmetic on the same parameters).  Usage: python tools/own_probe.py cfg4a"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import wfdgan_oracle as O  # noqa: E402
from tests.util import iter_inputs, load_iter_fixture, rebuild_state, state_from_engine, unpack_masks  # noqa: E402
from wfd_gan_amd.engine import TrainEngine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4a"
torch.set_num_threads(8)
fx, meta = load_iter_fixture(name)
st = rebuild_state(meta, fx)
B, S, C = meta["batch"], meta["seq_len"], meta["class_dim"]
eng = TrainEngine(S, C, meta["latent_dim"], B, meta["scaler_min"], meta["scaler_max"], {k: v.detach() for k, v in st.G.items()},
                  {k: v.detach() for k, v in st.D.items()}, st.DF, lr=meta["lr"], alpha=meta["alpha"], n_critic=meta["n_critic"])


def q(e, scale):
    e = e.reshape(-1) / scale
    return " ".join(f"{float(torch.quantile(e[:: max(1, e.numel() // 1000000)], p)):.1e}" for p in (0.5, 0.9, 0.99, 1.0))


for it in range(meta["n_iter"]):
    p = f"it{it}_"
    do_g = it % meta["n_critic"] == 0
    mr, mf, mg = (unpack_masks(fx, p + k, B) for k in ("mreal", "mfake", "mg"))
    tr, lab, z = iter_inputs(fx, meta, it)
    own64 = state_from_engine(eng, meta, st.DF)
    own32 = state_from_engine(eng, meta, st.DF, dtype=torch.float32)
    eng.set_batch(torch.from_numpy(tr), torch.from_numpy(lab), torch.from_numpy(z))
    eng.set_masks([torch.cat([a, b]) for a, b in zip(mr, mf)], mg)
    eng.d_step()
    torch.cuda.synchronize()
    d_after = {n: eng.Dp.p[n].double().cpu() for n, _ in eng.Dp.specs}
    dg = [eng.Dp.g[n].double().cpu() for n, _ in eng.Dp.specs]
    if do_g:
        eng.g_step()
        torch.cuda.synchronize()
    gg = [eng.Gp.g[n].double().cpu() for n, _ in eng.Gp.specs]
    m64 = lambda ms: None if ms is None else [t.double() for t in ms]
    hook = lambda o: [o.D[n].data.copy_(d_after[n].to(o.D[n].dtype)) for n in d_after]
    x64 = O.iteration(own64, torch.from_numpy(tr).double(), torch.from_numpy(lab), torch.from_numpy(z).double(), do_g,
                      m64(mr), m64(mf), m64(mg), keep=True, after_d_step=hook)
    x32 = O.iteration(own32, torch.from_numpy(tr).double(), torch.from_numpy(lab), torch.from_numpy(z), do_g, mr, mf, mg,
                      keep=True, after_d_step=hook)
    print(f"== {name} it{it}: error / max|exact| quantiles (50% 90% 99% max):  GPU | fp32 oracle")
    for k, (n, _) in enumerate(eng.Dp.specs):
        e = x64["d_grads"][k]
        sc = float(e.abs().max()) or 1.0
        print(f"  D {n:16s} {q((dg[k] - e).abs(), sc)} | {q((x32['d_grads'][k].double() - e).abs(), sc)}")
    if do_g:
        for k, (n, _) in enumerate(eng.Gp.specs):
            e = x64["g_grads"][k]
            sc = float(e.abs().max()) or 1.0
            print(f"  G {n:16s} {q((gg[k] - e).abs(), sc)} | {q((x32['g_grads'][k].double() - e).abs(), sc)}")
