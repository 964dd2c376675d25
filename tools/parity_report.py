#!/usr/bin/env python3
"""Print the relative error of every checked quantity (HIP engine vs reference fixtures). GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.util import load_iter_fixture, rebuild_state, rel_err, unpack_masks
from tests.test_gpu_step import make_engine

for name in ("tiny", "cfg1", "cfg2"):
    fx, meta = load_iter_fixture(name)
    st = rebuild_state(meta, fx)
    B = meta["batch"]
    eng = make_engine(st, meta, B)
    for it in range(meta["n_iter"]):
        p = f"it{it}_"
        do_g = it % meta["n_critic"] == 0
        mr, mf, mg = unpack_masks(fx, p + "mreal", B), unpack_masks(fx, p + "mfake", B), unpack_masks(fx, p + "mg", B)
        eng.set_batch(torch.from_numpy(fx[p + "traces"]), torch.from_numpy(fx[p + "labels"]), torch.from_numpy(fx[p + "z"]))
        eng.set_masks([torch.cat([a, b]) for a, b in zip(mr, mf)], mg)
        eng.d_step(); torch.cuda.synchronize()
        s = eng.d_stats()
        sc = fx[p + "scalars"]
        line = f"{name} it{it}: d_loss {s['d_loss']:.6f}/{sc[0]:.6f} pen {s['penalty']:.6e}/{sc[1]:.6e} w {s['w_dist']:.6f}/{sc[2]:.6f}"
        line += f" | n_r rel {rel_err(eng.buf['PEN'][:B].cpu().numpy(), fx[p+'n_r']):.2e} n_f rel {rel_err(eng.buf['PEN'][B:2*B].cpu().numpy(), fx[p+'n_f']):.2e}"
        fake = eng.fake_rows[:, :meta['seq_len']].cpu().numpy()
        line += f" fake rel {rel_err(fake.astype(np.float64).sum(1), fx[p+'fake_d_rowsum']):.2e}"
        if do_g:
            eng.g_step(); torch.cuda.synchronize()
            g = eng.g_stats(); gs = fx[p + "g_scalars"]
            line += f" | g {g['g_loss']:.6f}/{gs[0]:.6f} f {g['f_loss']:.6f}/{gs[1]:.6f} nsel {g['n_sel']}/{int(gs[3])}"
        dn = np.array([float(eng.Dp.p[n].double().norm()) for n, _ in eng.Dp.specs])
        gn = np.array([float(eng.Gp.p[n].double().norm()) for n, _ in eng.Gp.specs])
        line += f" | Dnorm rel {rel_err(dn, fx[p+'d_param_norms']):.2e} Gnorm rel {rel_err(gn, fx[p+'g_param_norms']):.2e}"
        # slices of params
        def sl(t, n=16):
            f = t.detach().flatten(); step = max(1, f.numel() // n); return f[::step][:n].cpu().numpy()
        dps = np.stack([sl(eng.Dp.p[n]) if eng.Dp.p[n].numel() >= 16 else np.resize(eng.Dp.p[n].cpu().numpy().flatten(), 16) for n, _ in eng.Dp.specs])
        gps = np.stack([sl(eng.Gp.p[n]) for n, _ in eng.Gp.specs])
        line += f" Dslice maxabs {np.abs(dps - fx[p+'d_param_slices']).max():.2e} Gslice maxabs {np.abs(gps - fx[p+'g_param_slices']).max():.2e}"
        print(line, flush=True)
