"""GPU parity of the training iteration: HIP engine vs the golden fixtures captured from the reference, and vs the
oracle run on the same seeded inputs.  Tolerance: north_star's 1e-3 relative in fp32 for losses / penalty / fake
tensors (the W-div penalty is a 6th power of a gradient norm, so it carries 6x the norm's relative error)."""
import numpy as np
import pytest
import torch

from tests.util import load_iter_fixture, rebuild_state, rel_err, unpack_masks

pytestmark = pytest.mark.gpu

RTOL = 1e-3


def make_engine(st, meta, batch):
    from wfd_gan_amd.engine import TrainEngine
    return TrainEngine(meta["seq_len"], meta["class_dim"], meta["latent_dim"], batch, meta["scaler_min"],
                       meta["scaler_max"], {k: v.detach() for k, v in st.G.items()},
                       {k: v.detach() for k, v in st.D.items()}, st.DF, lr=meta["lr"], alpha=meta["alpha"],
                       n_critic=meta["n_critic"])


def close(a, b, rtol=RTOL, atol=0.0, what=""):
    a, b = float(a), float(b)
    assert abs(a - b) <= rtol * abs(b) + atol, f"{what}: got {a}, want {b}"


@pytest.mark.parametrize("name", ["tiny", "cfg1", "cfg2"])
def test_iterations_match_reference_fixture(name):
    fx, meta = load_iter_fixture(name)
    st = rebuild_state(meta, fx)
    B = meta["batch"]
    eng = make_engine(st, meta, B)
    for it in range(meta["n_iter"]):
        p = f"it{it}_"
        do_g = it % meta["n_critic"] == 0
        mr, mf, mg = unpack_masks(fx, p + "mreal", B), unpack_masks(fx, p + "mfake", B), unpack_masks(fx, p + "mg", B)
        eng.set_batch(torch.from_numpy(fx[p + "traces"]), torch.from_numpy(fx[p + "labels"]),
                      torch.from_numpy(fx[p + "z"]))
        eng.set_masks([torch.cat([a, b]) for a, b in zip(mr, mf)], mg)
        eng.d_step()
        torch.cuda.synchronize()
        eng.check_length_index()
        s = eng.d_stats()
        d_loss, pen, w_dist, mrv, mfv = fx[p + "scalars"]
        # per-row gradient norms^3 (div_gp terms): 6x relative error amplification of the norm
        n_r = eng.buf["PEN"][:B].cpu().numpy()
        n_f = eng.buf["PEN"][B:2 * B].cpu().numpy()
        assert rel_err(n_r, fx[p + "n_r"]) < RTOL, f"it{it} n_r"
        assert rel_err(n_f, fx[p + "n_f"]) < RTOL, f"it{it} n_f"
        close(s["penalty"], pen, what=f"it{it} penalty")
        # the loss terms are O(1) sums of O(1) critic outputs: absolute floor 1e-5 on the means
        close(s["mean_rv"], mrv, atol=2e-5, what=f"it{it} mean_rv")
        close(s["mean_fv"], mfv, atol=2e-5, what=f"it{it} mean_fv")
        close(s["d_loss"], d_loss, atol=4e-5, what=f"it{it} d_loss")
        close(s["w_dist"], w_dist, atol=4e-5, what=f"it{it} w_dist")
        assert np.array_equal(eng.buf["idx"].cpu().numpy(), fx[p + "idx_d"]), f"it{it} idx"
        fake = eng.fake_rows[:, : meta["seq_len"]].cpu().numpy()
        assert rel_err(fake[: fx[p + "fake_d_rows"].shape[0]], fx[p + "fake_d_rows"]) < RTOL
        assert rel_err(fake.astype(np.float64).sum(1), fx[p + "fake_d_rowsum"]) < RTOL
        assert rel_err(eng.buf["V"][:B].cpu().numpy(), fx[p + "rv"][:, 0]) < RTOL or \
            np.abs(eng.buf["V"][:B].cpu().numpy() - fx[p + "rv"][:, 0]).max() < 2e-5
        if do_g:
            eng.g_step()
            torch.cuda.synchronize()
            g = eng.g_stats()
            g_loss, f_loss, comb, nsel = fx[p + "g_scalars"]
            close(g["g_loss"], g_loss, atol=2e-5, what=f"it{it} g_loss")
            assert g["n_sel"] == int(nsel), f"it{it} n_sel"
            close(g["f_loss"], f_loss, atol=2e-5, what=f"it{it} f_loss")
            close(g["g_combined"], comb, atol=4e-5, what=f"it{it} combined")
        # parameters after the optimizer steps: slices + norms
        dnorm = np.array([float(eng.Dp.p[n].double().norm()) for n, _ in eng.Dp.specs])
        assert rel_err(dnorm, fx[p + "d_param_norms"]) < RTOL, f"it{it} D param norms"
        gnorm = np.array([float(eng.Gp.p[n].double().norm()) for n, _ in eng.Gp.specs])
        assert rel_err(gnorm, fx[p + "g_param_norms"]) < RTOL, f"it{it} G param norms"
        bn = torch.cat([eng.Gbuf[f"model.{i}.{s}"] for i in (3, 6) for s in ("running_mean", "running_var")])
        assert rel_err(bn.cpu().numpy(), fx[p + "g_bn_stats"]) < RTOL, f"it{it} BN running stats"


@pytest.mark.parametrize("name", ["cfg1", "cfg2"])
def test_first_iteration_split_bf16_operands(name):
    """operands="bf16x3" (every contraction as hi*hi + hi*lo + lo*hi on the bf16 matrix pipe, gemm_mx.hip): products
    carry ~1e-5 relative error, so the first iteration (D-step + G-step) must still meet north_star's 1e-3 against the
    reference fixture.  Integer quantities derived from a float threshold (burst-length index, selected-row count) may
    flip where the float sits within 1e-4 of the threshold: at most one row is allowed to."""
    from wfd_gan_amd.engine import TrainEngine
    fx, meta = load_iter_fixture(name)
    st = rebuild_state(meta, fx)
    B = meta["batch"]
    eng = TrainEngine(meta["seq_len"], meta["class_dim"], meta["latent_dim"], B, meta["scaler_min"], meta["scaler_max"],
                      {k: v.detach() for k, v in st.G.items()}, {k: v.detach() for k, v in st.D.items()}, st.DF,
                      lr=meta["lr"], alpha=meta["alpha"], n_critic=meta["n_critic"], operands="bf16x3")
    p = "it0_"
    mr, mf, mg = unpack_masks(fx, p + "mreal", B), unpack_masks(fx, p + "mfake", B), unpack_masks(fx, p + "mg", B)
    eng.set_batch(torch.from_numpy(fx[p + "traces"]), torch.from_numpy(fx[p + "labels"]), torch.from_numpy(fx[p + "z"]))
    eng.set_masks([torch.cat([a, b]) for a, b in zip(mr, mf)], mg)
    eng.d_step()
    torch.cuda.synchronize()
    assert eng.L.get_gemm_mode() == 0            # the plan restores the process-wide mode
    s = eng.d_stats()
    d_loss, pen, w_dist, mrv, mfv = fx[p + "scalars"]
    idx_ok = eng.buf["idx"].cpu().numpy() == fx[p + "idx_d"]
    assert (~idx_ok).sum() <= 1
    assert rel_err(eng.buf["PEN"][:B].cpu().numpy(), fx[p + "n_r"]) < RTOL
    if idx_ok.all():
        assert rel_err(eng.buf["PEN"][B:2 * B].cpu().numpy(), fx[p + "n_f"]) < RTOL
        close(s["penalty"], pen, what="penalty")
        close(s["mean_fv"], mfv, atol=2e-5, what="mean_fv")
        close(s["d_loss"], d_loss, atol=4e-5, what="d_loss")
        fake = eng.fake_rows[:, : meta["seq_len"]].cpu().numpy()
        assert rel_err(fake[: fx[p + "fake_d_rows"].shape[0]], fx[p + "fake_d_rows"]) < RTOL
    close(s["mean_rv"], mrv, atol=2e-5, what="mean_rv")
    eng.g_step()
    torch.cuda.synchronize()
    g = eng.g_stats()
    g_loss, f_loss, comb, nsel = fx[p + "g_scalars"]
    assert abs(g["n_sel"] - int(nsel)) <= 1
    if idx_ok.all() and g["n_sel"] == int(nsel):
        close(g["g_loss"], g_loss, atol=2e-5, what="g_loss")
        close(g["f_loss"], f_loss, atol=2e-5, what="f_loss")
    dnorm = np.array([float(eng.Dp.p[n].double().norm()) for n, _ in eng.Dp.specs])
    assert rel_err(dnorm, fx[p + "d_param_norms"]) < RTOL


@pytest.mark.parametrize("name", ["cfg1", "cfg2"])
def test_first_dstep_bf16_operands(name):
    """operands="bf16" (bf16 operands / fp32 accumulate over the fp32 buffers; BASELINE config 4's numerics) is NOT a
    parity path.  Its stated tolerance against the reference fixture, first D-step (measured: means 8e-5, penalty 3e-4,
    d_loss 1e-2, per-row norm^3 1e-1, 3-4 of 64 length indices off by one):
        mean critic scores 1e-3 absolute; penalty 5e-3 relative; d_loss 5e-2 relative; per-row penalty terms 0.3 of the
        largest; burst-length index within 1 on every row and equal on at least 7 rows of 8."""
    from wfd_gan_amd.engine import TrainEngine
    fx, meta = load_iter_fixture(name)
    st = rebuild_state(meta, fx)
    B = meta["batch"]
    eng = TrainEngine(meta["seq_len"], meta["class_dim"], meta["latent_dim"], B, meta["scaler_min"], meta["scaler_max"],
                      {k: v.detach() for k, v in st.G.items()}, {k: v.detach() for k, v in st.D.items()}, st.DF,
                      lr=meta["lr"], alpha=meta["alpha"], n_critic=meta["n_critic"], operands="bf16")
    p = "it0_"
    mr, mf = unpack_masks(fx, p + "mreal", B), unpack_masks(fx, p + "mfake", B)
    eng.set_batch(torch.from_numpy(fx[p + "traces"]), torch.from_numpy(fx[p + "labels"]), torch.from_numpy(fx[p + "z"]))
    eng.set_masks([torch.cat([a, b]) for a, b in zip(mr, mf)], None)
    eng.d_step()
    torch.cuda.synchronize()
    s = eng.d_stats()
    d_loss, pen, w_dist, mrv, mfv = fx[p + "scalars"]
    close(s["mean_rv"], mrv, rtol=0.0, atol=1e-3, what="mean_rv")
    close(s["mean_fv"], mfv, rtol=0.0, atol=1e-3, what="mean_fv")
    close(s["penalty"], pen, rtol=5e-3, what="penalty")
    close(s["d_loss"], d_loss, rtol=5e-2, what="d_loss")
    assert rel_err(eng.buf["PEN"][:B].cpu().numpy(), fx[p + "n_r"]) < 0.3
    didx = np.abs(eng.buf["idx"].cpu().numpy().astype(np.int64) - fx[p + "idx_d"].astype(np.int64))
    assert didx.max() <= 1 and (didx != 0).sum() <= B // 8
