#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE itself and pin the oracle against it.

Run in the build container only (needs /root/reference; nothing here runs on the GPU box):

    python tests/golden/make_golden.py            # all pins + fixtures
    python tests/golden/make_golden.py --skip-script   # skip the end-to-end script run (slow part)

What it does
  1. imports /root/reference/src/model.py unmodified (a 2-line in-memory stub stands in for the absent
     ``torchsummaryX`` package, whose ``summary`` is only used under ``__main__``),
  2. PIN A: seeded construction of the reference Generator/Discriminator/DF == oracle.make_*_state (bit-exact),
  3. PIN B: replays the loop body src/mlp_df_wgan_train.py:184-275 around the reference's own nn.Modules
     (their nn.Dropout draws from the global generator), recovers the dropout masks by RNG replay, and asserts
     oracle.iteration(...) with those masks injected reproduces every scalar / tensor bit-for-bit,
  4. PIN C: runs the unmodified trainer script end-to-end (runpy, seeded) on a tiny dataset and compares its
     artefacts with oracle.train_loop.run_training under the same seeds,
  5. PIN D: reference-style sampling (generate.py:48-64 call sequence around the reference Generator and a real
     sklearn MinMaxScaler) == oracle.sample,
  6. PIN E: the reference's utils.compute_gradient_penalty (WGAN-GP, src/utils.py:64-84) == oracle.wgan_gp_penalty,
  7. PIN F: the other consumers' integer epilogues (gan_wt.py:79-87, gan_glue.py:114-124, synthesize.py:86-90), run
     by the reference's own functions == oracle.POSTPROCESS[...],
  8. PIN G: the reference DF module in train mode + torch.optim.Adamax (src/train_df.py:113-131) ==
     oracle.df_train.DFTrainer (loss, logits, every gradient, every parameter / BN buffer after the step),
  9. writes the fixtures (inputs, masks as packed bits, expected outputs).  Weights are NOT stored: tests
     re-derive them from the seed with oracle.make_state and check the stored checksums.
"""
import argparse
import json
import os
import runpy
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_SRC = "/root/reference/src"

from oracle import wfdgan_oracle as O          # noqa: E402
from oracle import train_loop as TL            # noqa: E402


def import_reference_model():
    stub = types.ModuleType("torchsummaryX")
    stub.summary = lambda *a, **k: None
    sys.modules.setdefault("torchsummaryX", stub)
    stub2 = types.ModuleType("torchsummary")
    stub2.summary = lambda *a, **k: None
    sys.modules.setdefault("torchsummary", stub2)
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    import model as ref_model                  # noqa
    return ref_model


def checksum(sd):
    return float(sum(float(v.detach().double().abs().sum()) for v in sd.values() if v.is_floating_point()))


def assert_same(a, b, what):
    a = torch.as_tensor(a)
    b = torch.as_tensor(b)
    if not torch.equal(a, b):
        diff = (a.double() - b.double()).abs().max()
        raise AssertionError(f"{what}: differs, max abs diff {diff}")


# ------------------------------------------------------------------------------------------------
def ref_build(ref_model, seq_len, class_dim, latent_dim, smin, smax, seed, df_sd):
    torch.manual_seed(seed)
    G = ref_model.Generator(seq_len, class_dim, latent_dim, scaler_min=smin, scaler_max=smax, is_gpu=False)
    D = ref_model.Discriminator(seq_len, class_dim)
    DFm = ref_model.DF(seq_len - 1, class_dim)
    if df_sd is not None:
        DFm.load_state_dict(df_sd)
    DFm.eval()
    return G, D, DFm


def ref_iteration(G, D, DFm, optG, optD, traces, c, z, do_g, alpha, class_dim):
    """Loop body of src/mlp_df_wgan_train.py:184-275 around the reference modules (W-div, as it runs)."""
    import torch.autograd as autograd
    out = {}
    real = traces.float().requires_grad_(True)
    c_onehot = torch.Tensor(np.eye(class_dim)[c.numpy()])
    c = c.long()
    optD.zero_grad()
    fake = G(z, c_onehot)
    rv = D(real, c_onehot)
    fv = D(fake, c_onehot)
    ones = torch.ones(real.size(0), 1)
    g_r = autograd.grad(rv, real, ones, create_graph=True, retain_graph=True, only_inputs=True)[0]
    n_r = g_r.view(g_r.size(0), -1).pow(2).sum(1) ** (O.P_DIV / 2)
    g_f = autograd.grad(fv, fake, ones, create_graph=True, retain_graph=True, only_inputs=True)[0]
    n_f = g_f.view(g_f.size(0), -1).pow(2).sum(1) ** (O.P_DIV / 2)
    div_gp = torch.mean(n_r + n_f) * O.K_DIV / 2
    d_loss = -torch.mean(rv) + torch.mean(fv) + div_gp
    out.update(n_r=n_r.detach().clone(), n_f=n_f.detach().clone(), penalty=float(div_gp),
               mean_rv=float(torch.mean(rv)), mean_fv=float(torch.mean(fv)),
               w_dist=float(torch.mean(rv) - torch.mean(fv)), d_loss=float(d_loss),
               fake_d=fake.detach().clone(), rv=rv.detach().clone(), fv=fv.detach().clone())
    d_loss.backward()
    out["d_grads"] = [p.grad.detach().clone() for p in D.parameters()]
    optD.step()
    optG.zero_grad()
    if do_g:
        fake = G(z, c_onehot)
        fv = D(fake, c_onehot)
        g_loss = -torch.mean(fv)
        sel = fv[:, 0] > 0
        if int(sel.sum()) > 0:
            conv = fake[sel][:, 1:]
            conv = conv.reshape(conv.size(0), 1, conv.size(1))
            f_loss = torch.nn.CrossEntropyLoss()(DFm(conv), c[sel])
            comb = g_loss + alpha * f_loss
            f_item = float(f_loss)
        else:
            comb = g_loss
            f_item = 0.0
        out.update(g_loss=float(g_loss), f_loss=f_item, g_combined=float(comb), n_sel=int(sel.sum()),
                   fake_g=fake.detach().clone(), fv_g=fv.detach().clone())
        comb.backward()
        out["g_grads"] = [p.grad.detach().clone() for p in G.parameters()]
        optG.step()
    return out


def pack_masks(masks):
    return [np.packbits((m.numpy() != 0).astype(np.uint8), axis=None) for m in masks]


def slices(t, n=16):
    f = t.detach().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].clone().numpy()


def make_case(ref_model, name, seq_len, class_dim, latent_dim, batch, n_iter, seed, data_seed, lr=2e-4,
              alpha=0.02, n_critic=3, full_rows=4, store_inputs=True):
    """One fixture: n_iter iterations from seeded init, on seeded synthetic data.
    store_inputs=False (the large config-4 shapes): the batch and z are NOT stored; the fixture keeps the recipe
    (dataset seed / arguments, row indices, z seed) and checksums, and tests/util.iter_inputs re-derives them."""
    print(f"[case {name}] S={seq_len} C={class_dim} Z={latent_dim} B={batch} iters={n_iter}")
    L = seq_len - 1
    synth_args = dict(n_cls=class_dim, n_inst=max(2, (batch * n_iter) // class_dim + 1), L=L, n_lo=max(2, L // 20),
                      max_out=75 if L > 600 else 13, max_in=179 if L > 600 else 122, seed=data_seed)
    X, y = O.synth_dataset(**synth_args)
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    Xs = X * scale + min_
    smin, smax = dmin[0], dmax[0]
    rng = np.random.default_rng(data_seed + 1)
    perm = rng.permutation(X.shape[0])

    # frozen observer: seeded init + perturbed BN statistics
    torch.manual_seed(seed + 100)
    df_sd = O.make_df_state(L, class_dim)
    O.perturb_df_bn(df_sd, seed + 101)

    G, D, DFm = ref_build(ref_model, seq_len, class_dim, latent_dim, smin, smax, seed, df_sd)
    optG = torch.optim.RMSprop(G.parameters(), lr=lr)
    optD = torch.optim.RMSprop(D.parameters(), lr=lr)
    st = O.make_state(seq_len, class_dim, latent_dim, smin, smax, seed, df_state=df_sd, lr=lr, alpha=alpha,
                      n_critic=n_critic)
    # PIN A
    for k, v in G.state_dict().items():
        assert_same(v, st.G[k], f"G init {k}")
    for k, v in D.state_dict().items():
        assert_same(v, st.D[k], f"D init {k}")
    for k, v in DFm.state_dict().items():
        assert_same(v, st.DF[k], f"DF init {k}")

    fx = dict(meta=json.dumps(dict(name=name, seq_len=seq_len, class_dim=class_dim, latent_dim=latent_dim,
                                   batch=batch, n_iter=n_iter, seed=seed, df_seed=seed + 100,
                                   df_bn_seed=seed + 101, lr=lr, alpha=alpha, n_critic=n_critic,
                                   scaler_min=float(smin), scaler_max=float(smax),
                                   torch=torch.__version__, synth_args=synth_args, z_seed=seed + 7)),
              g_init_checksum=checksum(st.G), d_init_checksum=checksum(st.D), df_checksum=checksum(st.DF))
    zrng = np.random.RandomState(seed + 7)
    torch.manual_seed(seed + 13)                     # dropout stream for the run
    for it in range(n_iter):
        idx = perm[it * batch:(it + 1) * batch]
        traces = torch.from_numpy(Xs[idx])           # float64, as the DataLoader hands it over
        c = torch.from_numpy(y[idx])
        z64 = zrng.normal(0, 1, (batch, latent_dim))
        z = torch.Tensor(z64)
        do_g = (it % n_critic == 0)
        rng_state = torch.get_rng_state()
        r_ref = ref_iteration(G, D, DFm, optG, optD, traces, c, z, do_g, alpha, class_dim)
        after = torch.get_rng_state()
        # recover the masks the reference's nn.Dropout layers drew
        torch.set_rng_state(rng_state)
        m_real = O.draw_dropout_masks(batch)
        m_fake = O.draw_dropout_masks(batch)
        m_g = O.draw_dropout_masks(batch) if do_g else None
        assert torch.equal(torch.get_rng_state(), after), "dropout replay consumed a different RNG stream"
        r = O.iteration(st, traces, c, z, do_g, m_real, m_fake, m_g, keep=True)
        # PIN B: oracle == reference modules, bit for bit
        for k in ("d_loss", "penalty", "w_dist", "mean_rv", "mean_fv"):
            assert r[k] == r_ref[k], (k, r[k], r_ref[k])
        for k in ("n_r", "n_f", "fake_d", "rv", "fv"):
            assert_same(r[k], r_ref[k], f"it{it} {k}")
        for a, b in zip(r["d_grads"], r_ref["d_grads"]):
            assert_same(a, b, f"it{it} d_grads")
        if do_g:
            for k in ("g_loss", "f_loss", "g_combined", "n_sel"):
                assert r[k] == r_ref[k], (k, r[k], r_ref[k])
            assert_same(r["fake_g"], r_ref["fake_g"], f"it{it} fake_g")
            for a, b in zip(r["g_grads"], r_ref["g_grads"]):
                assert_same(a, b, f"it{it} g_grads")
        for k, v in G.state_dict().items():
            assert_same(v, st.G[k].detach(), f"it{it} G param {k}")
        for k, v in D.state_dict().items():
            assert_same(v, st.D[k].detach(), f"it{it} D param {k}")
        print(f"   it{it}: d_loss={r['d_loss']:.6f} pen={r['penalty']:.6e} w={r['w_dist']:.6f}"
              + (f" g={r['g_loss']:.6f} f={r['f_loss']:.6f} nsel={r['n_sel']}" if do_g else ""))

        p = f"it{it}_"
        if store_inputs:
            fx[p + "traces"] = Xs[idx].astype(np.float32)
            fx[p + "labels"] = y[idx]
            fx[p + "z"] = z.numpy()
        else:
            fx[p + "rows"] = idx.astype(np.int64)
            fx[p + "traces_sum"] = float(Xs[idx].astype(np.float32).astype(np.float64).sum())
            fx[p + "z_sum"] = float(z.double().abs().sum())
        for nm, ms in (("mreal", m_real), ("mfake", m_fake), ("mg", m_g)):
            if ms is not None:
                for li, pk in enumerate(pack_masks(ms)):
                    fx[f"{p}{nm}{li}"] = pk
        fx[p + "scalars"] = np.array([r_ref["d_loss"], r_ref["penalty"], r_ref["w_dist"], r_ref["mean_rv"],
                                      r_ref["mean_fv"]], dtype=np.float64)
        fx[p + "n_r"], fx[p + "n_f"] = r_ref["n_r"].numpy(), r_ref["n_f"].numpy()
        fx[p + "idx_d"] = r["idx_d"].numpy()
        fx[p + "rv"], fx[p + "fv"] = r_ref["rv"].numpy(), r_ref["fv"].numpy()
        fx[p + "fake_d_rowsum"] = r_ref["fake_d"].double().sum(1).numpy()
        fx[p + "fake_d_rows"] = r_ref["fake_d"][:full_rows].numpy()
        fx[p + "d_grad_norms"] = np.array([float(g.double().norm()) for g in r_ref["d_grads"]])
        fx[p + "d_grad_slices"] = np.stack([slices(g) if g.numel() >= 16 else np.resize(g.numpy().flatten(), 16)
                                            for g in r_ref["d_grads"]])
        fx[p + "d_param_slices"] = np.stack([slices(q) if q.numel() >= 16 else np.resize(q.detach().numpy().flatten(), 16)
                                             for q in D.parameters()])
        fx[p + "d_param_norms"] = np.array([float(q.detach().double().norm()) for q in D.parameters()])
        if do_g:
            fx[p + "g_scalars"] = np.array([r_ref["g_loss"], r_ref["f_loss"], r_ref["g_combined"],
                                            r_ref["n_sel"]], dtype=np.float64)
            fx[p + "idx_g"] = r["idx_g"].numpy()
            fx[p + "fv_g"] = r_ref["fv_g"].numpy()
            fx[p + "fake_g_rowsum"] = r_ref["fake_g"].double().sum(1).numpy()
            fx[p + "fake_g_rows"] = r_ref["fake_g"][:full_rows].numpy()
            fx[p + "g_grad_norms"] = np.array([float(g.double().norm()) for g in r_ref["g_grads"]])
            fx[p + "g_grad_slices"] = np.stack([slices(g) for g in r_ref["g_grads"]])
        fx[p + "g_param_slices"] = np.stack([slices(q) for q in G.parameters()])
        fx[p + "g_param_norms"] = np.array([float(q.detach().double().norm()) for q in G.parameters()])
        gsd = G.state_dict()
        fx[p + "g_bn_stats"] = np.concatenate([gsd[f"model.{i}.{s}"].numpy() for i in (3, 6)
                                               for s in ("running_mean", "running_var")])
    np.savez_compressed(os.path.join(HERE, f"iter_{name}.npz"), **fx)
    print(f"   wrote iter_{name}.npz")


# ------------------------------------------------------------------------------------------------
def make_sampler_case(ref_model, name, seq_len, class_dim, latent_dim, n_cases, seed):
    """PIN D + fixture: reference-style sample() vs oracle.sample, including adversarial rows."""
    from sklearn import preprocessing
    print(f"[sampler {name}]")
    L = seq_len - 1
    X, y = O.synth_dataset(class_dim, 8, L, max(2, L // 20), 75, 179, seed=seed)
    X[:, 5] = 7.0                                   # a constant column: scale_ = 1 path
    scaler = preprocessing.MinMaxScaler()
    scaler.fit(X)
    scale, min_ = O.scaler_params(scaler.data_min_, scaler.data_max_)
    assert np.array_equal(scale, scaler.scale_) and np.array_equal(min_, scaler.min_)
    torch.manual_seed(seed)
    G = ref_model.Generator(seq_len, class_dim, latent_dim, scaler_min=scaler.data_min_[0],
                            scaler_max=scaler.data_max_[0], is_gpu=False)
    # non-trivial running statistics, as after training
    g = torch.Generator().manual_seed(seed + 1)
    sd = G.state_dict()
    for k in sd:
        if k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.05
        elif k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape, generator=g) * 0.2 + 0.05
    G.load_state_dict(sd)
    G.eval()
    osd = {k: v.clone() for k, v in G.state_dict().items()}
    zr = np.random.RandomState(seed + 2)
    Z = zr.randn(n_cases, latent_dim).astype("float32")
    C = zr.randint(0, class_dim, n_cases)
    outs, lens, ys = [], [], []
    for i in range(n_cases):
        with torch.no_grad():
            zt = torch.from_numpy(Z[i:i + 1])
            c = torch.zeros(1, class_dim)
            c[:, C[i]] = 1
            sx = G(zt, c).cpu().numpy()
            ys.append(sx[0].copy())
            sx = sx.mean(axis=0).reshape(1, -1)
            sx = scaler.inverse_transform(sx).flatten()
            sx = np.round(np.trim_zeros(sx)).astype(int)
            sx[sx == 0] = 1
        mine = O.sample(osd, int(C[i]), Z[i:i + 1], class_dim, scaler.data_min_[0], scaler.data_max_[0], scale, min_)
        assert np.array_equal(sx, mine), f"sampler case {i}"
        outs.append(sx)
        lens.append(len(sx))
    # adversarial post-processing rows: values whose inverse-transform lands on .5 exactly, zeros inside the
    # row, leading zeros (trim_zeros trims both ends), all-zero row
    adv = []
    r = np.random.RandomState(seed + 3)
    for i in range(32):
        row = np.zeros(seq_len, dtype=np.float32)
        n = int(r.randint(3, seq_len - 1))
        tgt = r.randint(0, 40, n).astype(np.float64) + 0.5 * (r.rand(n) < 0.5)
        row[:n] = ((tgt * scale[:n]) + min_[:n]).astype(np.float32)
        if i % 4 == 1:
            row[0] = 0.0
        if i % 4 == 2:
            row[n // 2] = 0.0
        if i == 31:
            row[:] = 0.0
        adv.append(row)
    adv = np.stack(adv)
    adv_out = []
    for row in adv:
        sx = scaler.inverse_transform(row.reshape(1, -1).copy()).flatten()
        sx = np.round(np.trim_zeros(sx)).astype(int)
        sx[sx == 0] = 1
        mine = O.sample_postprocess(row, scale, min_)
        assert np.array_equal(sx, mine)
        adv_out.append(sx)
    maxlen = seq_len
    pad = lambda rows: np.stack([np.pad(o, (0, maxlen - len(o))) for o in rows]).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, f"sampler_{name}.npz"),
                        meta=json.dumps(dict(seq_len=seq_len, class_dim=class_dim, latent_dim=latent_dim,
                                             seed=seed, n_cases=n_cases)),
                        data_min=scaler.data_min_, data_max=scaler.data_max_, z=Z, c=C,
                        y=np.stack(ys), out=pad(outs), out_len=np.array(lens),
                        adv_in=adv, adv_out=pad(adv_out), adv_len=np.array([len(o) for o in adv_out]),
                        g_checksum=checksum(osd),
                        bn_seed=seed + 1)
    print(f"   wrote sampler_{name}.npz ({n_cases} sampled + {len(adv)} adversarial rows)")


# ------------------------------------------------------------------------------------------------
class _FixedRowModel:
    """Stands where the consumers expect ``model``: eval() and __call__(z, c) -> the given generator output row."""
    def __init__(self, row):
        self.row = np.asarray(row, dtype=np.float32).reshape(1, -1)

    def eval(self):
        return self

    def __call__(self, z, c):
        return torch.from_numpy(self.row.copy())


def _reference_consumers():
    """The three other consumers of the generator in the reference, each returning its integer epilogue for the
    generator output row that ``model`` yields:
      wt    : gan_wt.WTGAN.get_ref_trace (imported module, object built without its file-loading constructor)
      synth : synthesize.sample_trace    (imported module, its globals set)
      glue  : gan_glue.query_from_gan    -- gan_glue imports the absent KDEpy at module top, so only this function's
              definition is taken from the file (ast) and executed with the globals it reads."""
    import ast
    import gan_wt
    import synthesize

    def wt(model, scaler, seq_len, class_dim, latent_dim):
        o = gan_wt.WTGAN.__new__(gan_wt.WTGAN)
        o.model, o.scaler, o.seq_len, o.class_dim, o.latent_dim = model, scaler, seq_len, class_dim, latent_dim
        return np.asarray(o.get_ref_trace(), dtype=np.int64)

    def synth(model, scaler, seq_len, class_dim, latent_dim):
        synthesize.model, synthesize.scaler = model, scaler
        synthesize.latent_dim, synthesize.class_dim = latent_dim, class_dim
        return np.asarray(synthesize.sample_trace(0), dtype=np.int64)

    path = os.path.join(REF_SRC, "gan_glue.py")
    tree = ast.parse(open(path).read(), filename=path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "query_from_gan"]
    assert len(fn) == 1
    code = compile(ast.Module(body=fn, type_ignores=[]), path, "exec")

    def glue(model, scaler, seq_len, class_dim, latent_dim):
        ns = dict(np=np, torch=torch, device=torch.device("cpu"), model=model, latent_dim=latent_dim, scaler=scaler,
                  cls_num=class_dim)
        exec(code, ns)
        return np.asarray(ns["query_from_gan"](0), dtype=np.int64)

    return dict(wt=wt, glue=glue, synth=synth)


def make_postprocess_modes(ref_model):
    """PIN F + fixture: the integer epilogues of gan_wt.py:79-87, gan_glue.py:114-124 and synthesize.py:86-90, run
    by the reference's own functions on (a) real generator outputs and (b) adversarial rows, vs the oracle."""
    from sklearn import preprocessing
    print("[sampler modes] gan_wt / gan_glue / synthesize epilogues")
    consumers = _reference_consumers()
    seq_len, class_dim, latent_dim, seed = 61, 5, 16, 41
    L = seq_len - 1
    X, y = O.synth_dataset(class_dim, 8, L, 4, 75, 179, seed=seed)
    X[:, 7] = 3.0                                   # constant column: scale_ = 1
    scaler = preprocessing.MinMaxScaler()
    scaler.fit(X)
    scale, min_ = O.scaler_params(scaler.data_min_, scaler.data_max_)
    torch.manual_seed(seed)
    G = ref_model.Generator(seq_len, class_dim, latent_dim, scaler_min=scaler.data_min_[0],
                            scaler_max=scaler.data_max_[0], is_gpu=False)
    G.eval()
    rows = []
    zr = np.random.RandomState(seed + 2)
    with torch.no_grad():
        for i in range(40):
            c = torch.zeros(1, class_dim)
            c[:, zr.randint(class_dim)] = 1
            rows.append(G(torch.from_numpy(zr.randn(1, latent_dim).astype("float32")), c).numpy()[0])
    r = np.random.RandomState(seed + 3)
    for i in range(48):         # adversarial: exact .5 values, zeros inside / at the ends, odd and even counts, short rows
        row = np.zeros(seq_len, dtype=np.float32)
        n = int(r.randint(0, seq_len))
        tgt = r.randint(0, 30, seq_len).astype(np.float64) + 0.5 * (r.rand(seq_len) < 0.5)
        tgt[0] = n + (0.75 if i % 3 == 0 else 0.0)          # fractional burst count: int() truncates
        if i % 8 == 7:
            tgt[0] = seq_len + 5                            # count beyond the row
        full = ((tgt * scale) + min_).astype(np.float32)
        m = min(n + 1, seq_len)
        row[:m] = full[:m]
        if i % 4 == 1 and m > 2:
            row[m - 1] = min_[m - 1]                        # trailing value that inverse-transforms to 0
        if i % 4 == 2 and m > 3:
            row[m // 2] = min_[m // 2]
        if i == 47:
            row[:] = 0.0
        rows.append(row)
    rows = np.stack(rows).astype(np.float32)
    fx = dict(rows=rows, data_min=scaler.data_min_, data_max=scaler.data_max_)
    for mode, ref_fn in consumers.items():
        outs = []
        for i, row in enumerate(rows):
            np.random.seed(i)
            want = ref_fn(_FixedRowModel(row), scaler, seq_len, class_dim, latent_dim)
            mine = O.POSTPROCESS[mode](row, scale, min_)
            assert np.array_equal(want, mine), (mode, i, want, mine)
            outs.append(want)
        fx[f"{mode}_out"] = np.stack([np.pad(o, (0, seq_len - len(o))) for o in outs]).astype(np.int64)
        fx[f"{mode}_len"] = np.array([len(o) for o in outs])
        print(f"   {mode}: {len(outs)} rows bit-exact (lengths {fx[f'{mode}_len'].min()}..{fx[f'{mode}_len'].max()})")
    np.savez_compressed(os.path.join(HERE, "sampler_modes.npz"), **fx)
    print("   wrote sampler_modes.npz")


# ------------------------------------------------------------------------------------------------
def make_df_train_case(ref_model, name, length, class_dim, batch, n_steps, seed):
    """PIN G + fixture: the reference DF module in train mode + torch.optim.Adamax (src/train_df.py:113-131) vs
    oracle.df_train.DFTrainer, bit-exact; dropout masks recovered by RNG replay and stored as packed bits."""
    from oracle import df_train as DT
    print(f"[df_train {name}] L={length} C={class_dim} B={batch}")
    torch.manual_seed(seed)
    model = ref_model.DF(length, class_dim)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    opt = torch.optim.Adamax(model.parameters())
    crit = torch.nn.CrossEntropyLoss()
    tr = DT.DFTrainer(sd0)
    tr2 = DT.DFTrainer(sd0)                                            # explicit-mask path
    g = torch.Generator().manual_seed(seed + 1)
    fx = dict(meta=json.dumps(dict(length=length, class_dim=class_dim, batch=batch, n_steps=n_steps, seed=seed)),
              init_checksum=checksum(sd0))
    keys = DT.df_param_keys(sd0)
    shapes = DT.dropout_mask_shapes(length, batch)
    for it in range(n_steps):
        x = torch.rand(batch, 1, length, generator=g)
        x = x * (torch.arange(length) < torch.randint(length // 4, length, (batch, 1, 1), generator=g)).float()
        y = torch.randint(0, class_dim, (batch,), generator=g)
        model.train()
        torch.manual_seed(1000 + it)
        out = model(x)
        loss = crit(out, y)
        opt.zero_grad()
        loss.backward()
        grads_ref = {k: p.grad.detach().clone() for k, p in zip(keys, model.parameters())}
        opt.step()
        # oracle, dropout drawn the same way
        torch.manual_seed(1000 + it)
        r = tr.step(x, y, keep=True)
        assert float(loss) == r["loss"], (it, float(loss), r["loss"])
        assert_same(out.detach(), r["logits"], f"df_train logits it{it}")
        for k in keys:
            assert_same(grads_ref[k], r["grads"][k], f"df_train grad {k} it{it}")
        for k, v in model.state_dict().items():
            assert_same(v, tr.sd[k].detach(), f"df_train state {k} it{it}")
        # masks by RNG replay, then the explicit-mask path
        torch.manual_seed(1000 + it)
        masks = [torch.nn.functional.dropout(torch.ones(s), p, True) for s, p in zip(shapes, DT.DROPOUT_P)]
        r2 = tr2.step(x, y, masks=masks)
        assert r2["loss"] == r["loss"]
        for k, v in model.state_dict().items():
            assert_same(v, tr2.sd[k].detach(), f"df_train(masks) state {k} it{it}")
        p_ = f"it{it}_"
        fx[p_ + "x"] = x.numpy()[:, 0, :]
        fx[p_ + "y"] = y.numpy()
        for i, pk in enumerate(pack_masks(masks)):
            fx[p_ + f"m{i}"] = pk
        fx[p_ + "loss"] = float(loss)
        fx[p_ + "logits"] = out.detach().numpy()
        fx[p_ + "grad_norms"] = np.array([float(grads_ref[k].double().norm()) for k in keys])
        fx[p_ + "param_norms"] = np.array([float(p.detach().double().norm()) for p in model.parameters()])
        fx[p_ + "bn_stats"] = np.concatenate([v.detach().numpy().ravel() for k, v in model.state_dict().items()
                                              if k.endswith("running_mean") or k.endswith("running_var")])
        if it == 0:
            for k in ("layer1.3.conv.weight", "layer2.0.conv.weight", "layer4.0.conv.weight", "layer5.0.weight",
                      "fc.weight", "layer1.0.conv.weight", "layer3.4.weight", "layer5.1.bias"):
                fx["it0_grad_" + k] = slices(grads_ref[k], 64)
    fx["param_keys"] = np.array(keys)
    np.savez_compressed(os.path.join(HERE, f"df_train_{name}.npz"), **fx)
    print(f"   {n_steps} train steps bit-exact (loss {fx['it0_loss']:.4f} -> {fx[f'it{n_steps-1}_loss']:.4f}); wrote df_train_{name}.npz")


# ------------------------------------------------------------------------------------------------
def make_cli_fixture():
    """The command-line contracts of the four scripts on the path, read from the reference's own argparse calls
    (ast of the source files; nothing is executed): flag names, dest-relevant options, types, defaults, required,
    action.  tests/test_cpu_abi.py compares the build's parsers against it."""
    import ast
    out = {}
    for script in ("mlp_df_wgan_train.py", "generate.py", "train_df.py", "extract.py"):
        tree = ast.parse(open(os.path.join(REF_SRC, script)).read())
        args = []
        for node in ast.walk(tree):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument":
                flags = [a.value for a in node.args if isinstance(a, ast.Constant)]
                kw = {}
                for k in node.keywords:
                    if k.arg in ("type", ):
                        kw["type"] = k.value.id if isinstance(k.value, ast.Name) else None
                    elif k.arg in ("default", "required", "action"):
                        try:
                            kw[k.arg] = ast.literal_eval(k.value)
                        except Exception:
                            kw[k.arg] = None
                args.append(dict(flags=flags, **kw))
        out[script] = args
        print(f"   {script}: {len(args)} arguments")
    with open(os.path.join(HERE, "cli_contract.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("   wrote cli_contract.json")


# ------------------------------------------------------------------------------------------------
def pin_script_end_to_end():
    """PIN C: unmodified reference trainer (runpy) vs oracle.train_loop.run_training, same seeds."""
    print("[script] running the unmodified reference trainer end-to-end (tiny config-1-like dataset)")
    n_cls, n_inst, L = 10, 20, 500
    X, y = O.synth_dataset(n_cls, n_inst, L, 20, 13, 122, seed=0)
    torch.manual_seed(0)
    df_sd = O.make_df_state(L, n_cls)
    tmp = tempfile.mkdtemp(prefix="wfdgan_ref_")
    ddir = os.path.join(tmp, "a", "b")
    os.makedirs(ddir)
    npz = os.path.join(ddir, "raw_feature_synth.npz")
    np.savez(npz, features=X, labels=y)
    ckpt = os.path.join(tmp, "df.ckpt")
    torch.save(df_sd, ckpt)
    import_reference_model()
    argv = sys.argv
    sys.argv = ["mlp_df_wgan_train.py", "-d", npz, "--f_model", ckpt, "--n_epochs", "2", "--n_cpu", "0",
                "--latent_dim", "64"]
    cwd = os.getcwd()
    os.chdir(REF_SRC)
    torch.manual_seed(1234)
    np.random.seed(4321)
    try:
        runpy.run_path(os.path.join(REF_SRC, "mlp_df_wgan_train.py"), run_name="__main__")
    finally:
        sys.argv = argv
        os.chdir(cwd)
    # utils.init_directory writes to the grandparent of --dir (src/utils.py:44-52)
    par = os.path.join(tmp, "a")
    outdir = [d for d in os.listdir(par) if d.startswith("training_")][0]
    base = os.path.join(par, outdir)
    ref_loss = np.load(os.path.join(base, "checkpoint", "loss.npz"))
    ref_ep1 = np.load(os.path.join(base, "checkpoint", "epoch_1.npz"))
    ref_G = torch.load(os.path.join(base, "model", "generator_seqlen501_cls10_latentdim64.ckpt"), weights_only=True)
    ref_D = torch.load(os.path.join(base, "model", "discriminator.ckpt"), weights_only=True)

    torch.manual_seed(1234)
    np.random.seed(4321)
    mine = TL.run_training(X.copy(), y.copy(), df_sd, n_epochs=2, n_cpu=0, latent_dim=64)
    for k in ("generator", "discriminator", "dist"):
        a, b = np.asarray(ref_loss[k]), np.asarray(mine["losses"][k])
        assert np.array_equal(a, b), (k, a, b)
    assert np.array_equal(ref_ep1["recon_x"], mine["dumps"][1]["recon_x"])
    assert np.array_equal(ref_ep1["x"], mine["dumps"][1]["x"])
    assert np.array_equal(ref_ep1["label"], mine["dumps"][1]["label"])
    for k, v in ref_G.items():
        assert_same(v, mine["G"][k], f"script G {k}")
    for k, v in ref_D.items():
        assert_same(v, mine["D"][k], f"script D {k}")
    print("   script artefacts == oracle run_training (loss.npz, epoch_1.npz, generator/discriminator ckpt): bit-exact")
    np.savez_compressed(os.path.join(HERE, "script_run.npz"),
                        meta=json.dumps(dict(n_cls=n_cls, n_inst=n_inst, L=L, torch_seed=1234, np_seed=4321,
                                             n_epochs=2, latent_dim=64, df_seed=0)),
                        **{f"loss_{k}": np.asarray(ref_loss[k]) for k in ("generator", "discriminator", "dist")},
                        recon_x_rowsum=ref_ep1["recon_x"].astype(np.float64).sum(1),
                        label=ref_ep1["label"],
                        g_checksum=checksum(ref_G), d_checksum=checksum(ref_D))
    print("   wrote script_run.npz")


def pin_gradient_penalty(ref_model):
    """PIN E: the reference's own utils.compute_gradient_penalty (src/utils.py:64-84; its call site in the trainer
    is commented out) vs oracle.wgan_gp_penalty with the same alpha / dropout masks, + a small fixture."""
    print("[gp] reference utils.compute_gradient_penalty vs oracle")
    import utils as ref_utils                      # /root/reference/src/utils.py
    S, C, B = 41, 5, 8
    torch.manual_seed(17)
    D = ref_model.Discriminator(S, C)
    dsd = {k: v.detach().clone() for k, v in D.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    real, fake = torch.rand(B, S, generator=g), torch.rand(B, S, generator=g)
    c = torch.eye(C)[torch.randint(0, C, (B,), generator=g)]
    np.random.seed(99)
    torch.manual_seed(5)
    rng_state = torch.get_rng_state()
    gp_ref = ref_utils.compute_gradient_penalty(D, real, fake, c, torch.FloatTensor)
    np.random.seed(99)
    alpha = torch.FloatTensor(np.random.random((B, 1)))
    torch.set_rng_state(rng_state)
    masks = O.draw_dropout_masks(B)
    for p_ in dsd.values():
        p_.requires_grad_(True)
    gp_mine = O.wgan_gp_penalty(dsd, real, fake, c, alpha, masks)
    assert float(gp_ref) == float(gp_mine), (float(gp_ref), float(gp_mine))
    grads_ref = torch.autograd.grad(gp_ref, list(D.parameters()), allow_unused=True)
    grads_mine = torch.autograd.grad(gp_mine, [dsd[k] for k in D.state_dict().keys()], allow_unused=True)
    for a, b in zip(grads_ref, grads_mine):
        if a is None:
            assert b is None or float(b.abs().max()) == 0.0
        else:
            assert_same(a, b, "gp grad")
    np.savez_compressed(os.path.join(HERE, "gp_tiny.npz"), real=real.numpy(), fake=fake.numpy(), c=c.numpy(),
                        alpha=alpha.numpy(), gp=float(gp_ref), d_seed=17,
                        **{f"m{i}": pk for i, pk in enumerate(pack_masks(masks))},
                        grad_norms=np.array([0.0 if a is None else float(a.double().norm()) for a in grads_ref]))
    print(f"   gp = {float(gp_ref):.6f}: bit-exact incl. parameter gradients; wrote gp_tiny.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-script", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.set_num_threads(8)
    ref_model = import_reference_model()
    if not args.only or args.only == "tiny":
        make_case(ref_model, "tiny", seq_len=41, class_dim=5, latent_dim=16, batch=8, n_iter=4, seed=3, data_seed=5)
    if not args.only or args.only == "cfg1":
        make_case(ref_model, "cfg1", seq_len=501, class_dim=10, latent_dim=500, batch=64, n_iter=4, seed=11,
                  data_seed=0)
    if not args.only or args.only == "cfg2":
        make_case(ref_model, "cfg2", seq_len=1401, class_dim=100, latent_dim=500, batch=64, n_iter=2, seed=21,
                  data_seed=0)
    # BASELINE config 4 shapes (large batch, L = 2500 / 5000; DF FC-in 256 / 512): inputs re-derived from seeds
    if not args.only or args.only == "cfg4":
        make_case(ref_model, "cfg4a", seq_len=2501, class_dim=100, latent_dim=500, batch=256, n_iter=2, seed=61,
                  data_seed=2, store_inputs=False)
        make_case(ref_model, "cfg4b", seq_len=5001, class_dim=20, latent_dim=500, batch=128, n_iter=1, seed=62,
                  data_seed=3, store_inputs=False)
    if not args.only or args.only == "sampler":
        make_sampler_case(ref_model, "cfg2", 1401, 100, 500, 100, seed=31)
        make_sampler_case(ref_model, "tiny", 41, 5, 16, 50, seed=32)
    if not args.only or args.only == "modes":
        make_postprocess_modes(ref_model)
    if not args.only or args.only == "df_train":
        make_df_train_case(ref_model, "tiny", length=64, class_dim=5, batch=16, n_steps=3, seed=51)
        make_df_train_case(ref_model, "cfg", length=1400, class_dim=100, batch=16, n_steps=2, seed=52)
    if not args.only or args.only == "cli":
        make_cli_fixture()
    if not args.only or args.only == "gp":
        pin_gradient_penalty(ref_model)
    if not args.skip_script and (not args.only or args.only == "script"):
        pin_script_end_to_end()
    print("ALL PINS PASSED")


if __name__ == "__main__":
    main()
