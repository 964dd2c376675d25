"""CPU: the oracle against the golden fixtures captured from the reference (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import wfdgan_oracle as O
from tests.util import GOLDEN, iter_inputs, load_iter_fixture, rebuild_state, unpack_masks


@pytest.mark.parametrize("name", ["tiny", "cfg1", "cfg4b"])
def test_oracle_iteration_reproduces_reference_fixture(name):
    torch.set_num_threads(8)
    fx, meta = load_iter_fixture(name)
    st = rebuild_state(meta, fx)
    B = meta["batch"]
    for it in range(meta["n_iter"]):
        p = f"it{it}_"
        do_g = it % meta["n_critic"] == 0
        traces, labels, z = iter_inputs(fx, meta, it)
        r = O.iteration(st, torch.from_numpy(traces).double(), torch.from_numpy(labels),
                        torch.from_numpy(z), do_g, unpack_masks(fx, p + "mreal", B),
                        unpack_masks(fx, p + "mfake", B), unpack_masks(fx, p + "mg", B), keep=True)
        sc = fx[p + "scalars"]
        # same machine class, same torch build: the fixture was asserted bit-exact at generation time; allow
        # last-ulp drift across CPU models
        for got, want in zip((r["d_loss"], r["penalty"], r["w_dist"], r["mean_rv"], r["mean_fv"]), sc):
            assert abs(got - want) <= 2e-5 * abs(want) + 1e-7, (it, got, want)
        assert np.allclose(r["n_r"].numpy(), fx[p + "n_r"], rtol=1e-4, atol=0)
        assert np.array_equal(r["idx_d"].numpy(), fx[p + "idx_d"])
        if do_g:
            gs = fx[p + "g_scalars"]
            assert abs(r["g_loss"] - gs[0]) <= 2e-5 * abs(gs[0]) + 1e-7
            assert abs(r["f_loss"] - gs[1]) <= 2e-5 * abs(gs[1]) + 1e-7
            assert r["n_sel"] == int(gs[3])


def test_oracle_training_run_reproduces_reference_script():
    """PIN C without a manual make_golden.py run: tests/golden/script_run.npz holds what the UNMODIFIED reference trainer
    (src/mlp_df_wgan_train.py run end-to-end by runpy, seeded) wrote -- loss.npz, the row sums of epoch_1.npz's recon_x,
    its labels, checksums of the two checkpoints.  oracle.train_loop.run_training under the same seeds must give them
    again (last-ulp drift across host CPUs allowed)."""
    from oracle import train_loop as TL
    from tests.util import checksum
    torch.set_num_threads(8)
    fx = np.load(os.path.join(GOLDEN, "script_run.npz"))
    meta = json.loads(str(fx["meta"]))
    X, y = O.synth_dataset(meta["n_cls"], meta["n_inst"], meta["L"], 20, 13, 122, seed=0)
    torch.manual_seed(meta["df_seed"])
    df_sd = O.make_df_state(meta["L"], meta["n_cls"])
    torch.manual_seed(meta["torch_seed"])
    np.random.seed(meta["np_seed"])
    mine = TL.run_training(X.copy(), y.copy(), df_sd, n_epochs=meta["n_epochs"], n_cpu=0, latent_dim=meta["latent_dim"])
    for k in ("generator", "discriminator", "dist"):
        got, want = np.asarray(mine["losses"][k], dtype=np.float64), np.asarray(fx[f"loss_{k}"], dtype=np.float64)
        assert got.shape == want.shape, k
        assert np.allclose(got, want, rtol=2e-5, atol=1e-6), (k, got, want)
    assert np.array_equal(mine["dumps"][1]["label"], fx["label"])
    rs = mine["dumps"][1]["recon_x"].astype(np.float64).sum(1)
    assert np.allclose(rs, fx["recon_x_rowsum"], rtol=1e-4, atol=1e-3)
    assert abs(checksum(mine["G"]) - float(fx["g_checksum"])) <= 1e-5 * float(fx["g_checksum"])
    assert abs(checksum(mine["D"]) - float(fx["d_checksum"])) <= 1e-5 * float(fx["d_checksum"])


def test_oracle_fp64_agrees_with_fp32_fixture():
    """fp64 run of the same iteration: the fp32 reference numbers sit within fp32 noise of it, which bounds how
    much of the 1e-3 GPU tolerance is the reference's own rounding."""
    fx, meta = load_iter_fixture("tiny")
    st = rebuild_state(meta, fx, dtype=torch.float64)
    B = meta["batch"]
    p = "it0_"
    m = lambda k: [t.double() for t in unpack_masks(fx, p + k, B)]
    r = O.iteration(st, torch.from_numpy(fx[p + "traces"]).double(), torch.from_numpy(fx[p + "labels"]),
                    torch.from_numpy(fx[p + "z"]).double(), True, m("mreal"), m("mfake"), m("mg"))
    sc = fx[p + "scalars"]
    assert abs(r["d_loss"] - sc[0]) < 1e-5
    assert abs(r["penalty"] - sc[1]) < 1e-4 * sc[1]


@pytest.mark.parametrize("name", ["tiny", "cfg2"])
def test_oracle_sampler_matches_reference_fixture(name):
    fx = np.load(os.path.join(GOLDEN, f"sampler_{name}.npz"))
    scale, min_ = O.scaler_params(fx["data_min"], fx["data_max"])
    for i in range(fx["y"].shape[0]):
        got = O.sample_postprocess(fx["y"][i], scale, min_)
        want = fx["out"][i][: fx["out_len"][i]]
        assert np.array_equal(got, want), i
    for i in range(fx["adv_in"].shape[0]):
        got = O.sample_postprocess(fx["adv_in"][i], scale, min_)
        assert np.array_equal(got, fx["adv_out"][i][: fx["adv_len"][i]]), i


@pytest.mark.parametrize("mode", ["wt", "glue", "synth"])
def test_oracle_epilogue_modes_match_reference_fixture(mode):
    """gan_wt / gan_glue / synthesize epilogues: outputs of the reference's own functions (PIN F)."""
    fx = np.load(os.path.join(GOLDEN, "sampler_modes.npz"))
    scale, min_ = O.scaler_params(fx["data_min"], fx["data_max"])
    for i, row in enumerate(fx["rows"]):
        got = O.POSTPROCESS[mode](row, scale, min_)
        assert np.array_equal(got, fx[f"{mode}_out"][i][: fx[f"{mode}_len"][i]]), (mode, i)


def test_oracle_df_training_matches_reference_fixture():
    """DF train step (reference DF module + Adamax, PIN G): the oracle with the stored masks reproduces loss, logits,
    gradient norms and parameter norms (1e-5: another host CPU may pick other conv kernels than the one that made
    the fixture)."""
    import json
    from oracle import df_train as DT
    fx = np.load(os.path.join(GOLDEN, "df_train_tiny.npz"))
    meta = json.loads(str(fx["meta"]))
    L, C, B = meta["length"], meta["class_dim"], meta["batch"]
    torch.manual_seed(meta["seed"])
    sd0 = O.make_df_state(L, C)
    tr = DT.DFTrainer(sd0)
    shapes = DT.dropout_mask_shapes(L, B)
    for it in range(meta["n_steps"]):
        p = f"it{it}_"
        masks = []
        for i, (shape, pr) in enumerate(zip(shapes, DT.DROPOUT_P)):
            bits = np.unpackbits(fx[f"{p}m{i}"])[: int(np.prod(shape))].reshape(shape)
            masks.append(torch.from_numpy(bits.astype(np.float32)) * (1.0 / (1.0 - pr)))
        r = tr.step(torch.from_numpy(fx[p + "x"])[:, None, :], torch.from_numpy(fx[p + "y"]), masks=masks, keep=True)
        assert abs(r["loss"] - float(fx[p + "loss"])) <= 1e-5 * abs(float(fx[p + "loss"]))
        assert np.abs(r["logits"].numpy() - fx[p + "logits"]).max() <= 1e-4
        gn = np.array([float(r["grads"][k].double().norm()) for k in tr.keys])
        big = fx[p + "grad_norms"] > 1e-6                 # biases in front of a BatchNorm carry rounding noise only
        assert np.allclose(gn[big], fx[p + "grad_norms"][big], rtol=1e-3)


def test_my_clip_matches_rowwise_definition():
    rng = np.random.default_rng(0)
    X = rng.integers(0, 700, size=(50, 41)).astype(np.float64)
    got = O.my_clip(X, 550, 13, 122)
    for r in range(X.shape[0]):
        row = X[r].copy()
        row[0] = min(row[0], 550)
        for i in range(1, len(row), 2):
            row[i] = min(row[i], 13)
        for i in range(2, len(row), 2):
            row[i] = min(row[i], 122)
        assert np.array_equal(got[r], row)


def test_generator_index_error_semantics():
    """A saturated length output must raise IndexError as the reference's advanced-index assignment does."""
    st = O.make_state(9, 2, 4, 0.0, 100.0, seed=0)
    st.G["model.8.bias"].data[0] = 1e4         # sigmoid -> 1.0 exactly -> idx = S
    with pytest.raises(IndexError):
        O.g_forward(st.G, torch.zeros(3, 4), torch.eye(2)[[0, 1, 0]], 0.0, 8.0, training=True)
