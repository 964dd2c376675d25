"""GPU: WGAN-GP penalty mode vs the oracle (pinned against the reference's utils.compute_gradient_penalty by
tests/golden/make_golden.py), the fast (device RNG + HIP graph) path, and the trainer / sampler CLIs end to end."""
import glob
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import wfdgan_oracle as O
from tests.util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(S, C, Z, B, penalty, seed=1):
    from wfd_gan_amd.engine import TrainEngine
    X, y = O.synth_dataset(C, max(8, 4 * B // C + 1), S - 1, max(2, S // 10), 13, 122, seed=3)
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    Xs = X * scale + min_
    st = O.make_state(S, C, Z, dmin[0], dmax[0], seed=seed, df_seed=seed + 1, penalty=penalty)
    eng = TrainEngine(S, C, Z, B, dmin[0], dmax[0], {k: v.detach() for k, v in st.G.items()},
                      {k: v.detach() for k, v in st.D.items()}, st.DF, penalty=penalty, device=DEV)
    return st, eng, Xs, y


@pytest.mark.parametrize("cfg", [(41, 5, 16, 8), (501, 10, 500, 64)])
def test_wgan_gp_mode_matches_oracle(cfg):
    S, C, Z, B = cfg
    st, eng, Xs, y = _setup(S, C, Z, B, "gp")
    g = torch.Generator().manual_seed(0)
    perm = torch.randperm(Xs.shape[0], generator=g)
    for it in range(4):
        idx = perm[it * B:(it + 1) * B].numpy()
        traces, labels = torch.from_numpy(Xs[idx]), torch.from_numpy(y[idx])
        z = torch.randn(B, Z, generator=g)
        alpha = torch.rand(B, 1, generator=g)
        mk = lambda: [(torch.rand(B, n, generator=g) > 0.2).float() * 1.25 for n in (2048, 1024, 512)]
        mr, mf, mgp, mg = mk(), mk(), mk(), mk()
        do_g = it % 3 == 0
        # every iteration starts from the oracle's parameter bits: otherwise last-bit differences accumulated over the
        # previous updates can put a LeakyReLU pre-activation on the other side of 0 (a 5x jump in one element of
        # d_i) and the per-tensor gradient comparison below turns into a test of that coin flip
        eng.Gp.load({k: v.detach() for k, v in st.G.items()})
        eng.Dp.load({k: v.detach() for k, v in st.D.items()})
        for k in eng.Gbuf:
            eng.Gbuf[k].copy_(st.G[k])
        ref = O.iteration(st, traces, labels, z, do_g, mr, mf, mg, gp_alpha=alpha, masks_gp=mgp, keep=True)
        eng.set_batch(traces, labels, z)
        eng.set_gp_alpha(alpha)
        eng.set_masks([torch.cat([a, b, c]) for a, b, c in zip(mr, mf, mgp)], mg)
        eng.d_step()
        if do_g:
            eng.g_step()
        torch.cuda.synchronize()
        d = eng.d_stats()
        assert abs(d["penalty"] - ref["penalty"]) <= 1e-3 * abs(ref["penalty"]), (it, d["penalty"], ref["penalty"])
        assert abs(d["d_loss"] - ref["d_loss"]) <= 1e-3 * abs(ref["d_loss"]) + 4e-5, (it, d["d_loss"], ref["d_loss"])
        assert abs(d["w_dist"] - ref["w_dist"]) <= 1e-3 * abs(ref["w_dist"]) + 4e-5
        # critic gradients of this iteration (per tensor) -- the grad buffer still holds them
        gtol = 1e-3
        for (name, _), gref in zip(eng.Dp.specs, ref["d_grads"]):
            got = eng.Dp.g[name].cpu()
            nref = float(gref.double().norm())
            assert float((got.double() - gref.double()).norm()) <= gtol * nref + 1e-7, (it, name)
        if do_g:
            gs = eng.g_stats()
            assert abs(gs["g_loss"] - ref["g_loss"]) <= 1e-3 * abs(ref["g_loss"]) + 2e-5
            assert gs["n_sel"] == ref["n_sel"]


def test_wdiv_gradients_match_oracle_per_tensor():
    """every critic / generator gradient tensor of one iteration vs autograd (oracle), relative L2 error."""
    S, C, Z, B = 501, 10, 64, 32
    st, eng, Xs, y = _setup(S, C, Z, B, "div", seed=4)
    g = torch.Generator().manual_seed(1)
    traces, labels = torch.from_numpy(Xs[:B]), torch.from_numpy(y[:B].copy())
    z = torch.randn(B, Z, generator=g)
    mk = lambda: [(torch.rand(B, n, generator=g) > 0.2).float() * 1.25 for n in (2048, 1024, 512)]
    mr, mf, mg = mk(), mk(), mk()
    # make the critic like some fakes so that the observer term is exercised
    with torch.no_grad():
        st.D["model.9.bias"] += 0.3
        eng.Dp.p["model.9.bias"] += 0.3
    ref = O.iteration(st, traces, labels, z, True, mr, mf, mg, keep=True)
    eng.set_batch(traces, labels, z)
    eng.set_masks([torch.cat([a, b]) for a, b in zip(mr, mf)], mg)
    eng.d_step()
    torch.cuda.synchronize()
    for (name, _), gref in zip(eng.Dp.specs, ref["d_grads"]):
        got = eng.Dp.g[name].cpu().double()
        assert float((got - gref.double()).norm()) <= 1e-3 * float(gref.double().norm()) + 1e-7, name
    eng.g_step()
    torch.cuda.synchronize()
    assert ref["n_sel"] > 0 and eng.g_stats()["n_sel"] == ref["n_sel"]
    assert abs(eng.g_stats()["f_loss"] - ref["f_loss"]) <= 1e-3 * ref["f_loss"]
    for (name, _), gref in zip(eng.Gp.specs, ref["g_grads"]):
        got = eng.Gp.g[name].cpu().double()
        nref = float(gref.double().norm())
        if name in ("model.2.bias", "model.5.bias"):
            # bias in front of BatchNorm: its true gradient is 0; both sides hold rounding noise only
            assert float(got.norm()) < 1e-5 and nref < 1e-5
            continue
        assert float((got - gref.double()).norm()) <= 1e-3 * nref + 1e-7, name


def test_fast_mode_graph_equals_eager():
    """device RNG + gather + HIP-graph replay reproduces the eager launches bit for bit (same Philox counters)."""
    from wfd_gan_amd.engine import TrainEngine
    S, C, Z, B = 41, 5, 16, 8
    st, e1, Xs, y = _setup(S, C, Z, B, "div")
    _, e2, _, _ = _setup(S, C, Z, B, "div")
    for e in (e1, e2):
        e.attach_dataset(torch.from_numpy(Xs), torch.from_numpy(y), shuffle_seed=5)
    e2.capture_graphs()
    for i in range(7):
        e1.iteration(i)
        e2.iteration(i)
    torch.cuda.synchronize()
    assert torch.equal(e1.Dp.flat, e2.Dp.flat) and torch.equal(e1.Gp.flat, e2.Gp.flat)
    assert torch.equal(e1.iter_state, e2.iter_state)
    assert e1.d_stats() == e2.d_stats()
    assert torch.isfinite(e1.Dp.flat).all() and torch.isfinite(e1.Gp.flat).all()


@pytest.fixture(scope="module")
def nccl_world1():
    """One 1-rank NCCL group for the whole module.  It is destroyed only after every engine that captured collectives into a
    HIP graph has been released: a graph holding RCCL nodes of a destroyed communicator aborted the process once (a later,
    unrelated allocation collected it)."""
    import gc
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    mine = not dist.is_initialized()
    if mine:
        dist.init_process_group("nccl", rank=0, world_size=1)
    yield
    gc.collect()
    torch.cuda.synchronize()
    if mine:
        dist.destroy_process_group()


# The sharded form (WFD_DP_SHARD=1, parallel.ShardedSync) is covered in this suite over gloo (tests/test_cpu_dp.py,
# tests/test_gpu_dp_exact.py); its RCCL world-1 runs live in tools/dp_shard_world1.py, outside the suite: one in-suite run of
# them aborted the interpreter without a message, and an abort here would take every later test down.
@pytest.mark.parametrize("shard", [False])
@pytest.mark.parametrize("captured", [True, False])
def test_data_parallel_world1_nccl(captured, shard, monkeypatch, nccl_world1):
    import gc
    monkeypatch.setenv("WFD_DP_SHARD", "1" if shard else "0")
    _data_parallel_world1_nccl(captured, monkeypatch, shard)
    gc.collect()                      # the engines and their graphs go before the next test (and before the group does)
    torch.cuda.synchronize()


def _data_parallel_world1_nccl(captured, monkeypatch, shard):
    """The data-parallel code path (bias gradients ahead of the weight gradients; bucketed asynchronous all-reduce over RCCL +
    flat optimizer pass, parallel.GradSync -- or, shard: per bucket reduce-scatter -> RMSprop on the rank's shard -> parameter
    all-gather on a side stream, parallel.ShardedSync) with a 1-rank NCCL group must reproduce the single-rank engine bit for
    bit -- with the collectives
    captured INSIDE the step graphs (one graph replay per D-step / G-step / group: the default on the "nccl" backend) and
    with them issued eagerly between separately captured stages (WFD_DP_GRAPH=0)."""
    import torch.distributed as dist
    from wfd_gan_amd.engine import GroupedTrainer, TrainEngine
    monkeypatch.setenv("WFD_DP_GRAPH", "1" if captured else "0")
    assert dist.is_initialized()
    if True:
        S, C, Z, B = 41, 5, 16, 8
        st, e1, Xs, y = _setup(S, C, Z, B, "div")
        mk = lambda: TrainEngine(S, C, Z, B, e1.scaler_min, e1.scaler_max, e1.generator_state(), e1.discriminator_state(),
                                 {k: v for k, v in st.DF.items()}, device=DEV, world_size=1, force_sync=True)
        e2 = mk()
        for e in (e1, e2):
            e.attach_dataset(torch.from_numpy(Xs), torch.from_numpy(y), shuffle_seed=5)
            e.capture_graphs()
        # eager collectives: stage, start, stage, start, stage, start + finish, stage (optimizer); the sharded optimizer runs
        # inside the buckets' chains, so no optimizer stage follows them
        assert e2.shard == shard
        assert len(e2.graph_d) == (1 if captured else (6 if shard else 7)) and len(e1.graph_d) == 1
        assert (e2.graph_group is not None) == captured
        for i in range(7):
            e1.iteration(i)
            e2.iteration(i)
        torch.cuda.synchronize()
        assert torch.equal(e1.Dp.flat, e2.Dp.flat) and torch.equal(e1.Gp.flat, e2.Gp.flat)
        assert e1.d_stats() == e2.d_stats()
        # the grouped driver (what bench.py runs under torchrun): one graph per group of n_critic iterations when captured
        e3, e4 = mk(), TrainEngine(S, C, Z, B, e1.scaler_min, e1.scaler_max, e1.generator_state(), e1.discriminator_state(),
                                   {k: v for k, v in st.DF.items()}, device=DEV)
        outs = []
        for e in (e3, e4):
            e.attach_dataset(torch.from_numpy(Xs), torch.from_numpy(y), shuffle_seed=5)
            grp = GroupedTrainer(e, graphs=True)
            if e is e3:
                assert len(grp.stages) == (1 if captured else (4 * 6 if shard else 1 + 3 * 6 + 6))
            for _ in range(3):
                grp.run_group()
            torch.cuda.synchronize()
            outs.append((e.Dp.flat.clone(), e.Gp.flat.clone(), e.buf["dacc"].clone()))
        for a, b in zip(*outs):
            assert torch.equal(a, b)


@pytest.mark.parametrize("shard", [False])
def test_capture_fallback_drops_stale_collective_handles(shard, monkeypatch, nccl_world1):
    """A capture WITH collectives that fails half-way must fall back to eager collectives between graph stages, and the
    host-side state of the abandoned capture must be gone, or the eager retry's first finish() waits on it.
    all-reduce form (GradSync): forced in the finish() of the first captured D-step, with the three buckets' Work handles
    pending.  sharded form (ShardedSync): the second bucket's reduce-scatter refuses, after the first bucket's chain went
    into the capture and the side stream was forked (the chain joins it before the error travels on).  The fallback engine
    then has to reproduce the single-rank engine bit for bit."""
    import gc
    import warnings
    import torch.distributed as dist
    from wfd_gan_amd import parallel
    from wfd_gan_amd.engine import TrainEngine
    monkeypatch.setenv("WFD_DP_GRAPH", "1")
    monkeypatch.setenv("WFD_DP_SHARD", "1" if shard else "0")
    S, C, Z, B = 41, 5, 16, 8
    st, e1, Xs, y = _setup(S, C, Z, B, "div")
    e2 = TrainEngine(S, C, Z, B, e1.scaler_min, e1.scaler_max, e1.generator_state(), e1.discriminator_state(),
                     {k: v for k, v in st.DF.items()}, device=DEV, world_size=1, force_sync=True)
    assert e2.shard == shard
    for e in (e1, e2):
        e.attach_dataset(torch.from_numpy(Xs), torch.from_numpy(y), shuffle_seed=5)
    e1.capture_graphs()
    seen = {"calls": 0, "forced": 0, "stale": None}
    if shard:
        real_rs = dist.reduce_scatter_tensor

        def refusing_reduce_scatter(*a, **k):
            if torch.cuda.is_current_stream_capturing() and not seen["forced"]:
                seen["calls"] += 1
                if seen["calls"] == 2:         # the second bucket's: the first bucket's chain is already in the capture
                    seen["forced"] = 1
                    raise RuntimeError("forced capture failure (test): collective refused")
            return real_rs(*a, **k)
        monkeypatch.setattr(dist, "reduce_scatter_tensor", refusing_reduce_scatter)
    else:
        real_finish = parallel.GradSync.finish

        def finish(self):
            if torch.cuda.is_current_stream_capturing() and not seen["forced"]:
                seen["forced"] = 1
                seen["stale"] = len(self.pending)
                raise RuntimeError("forced capture failure (test)")
            return real_finish(self)
        monkeypatch.setattr(parallel.GradSync, "finish", finish)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        e2.capture_graphs()
    assert seen["forced"] == 1 and (shard or seen["stale"] == 3)
    assert any("falling back to eager collectives" in str(x.message) for x in w)
    assert e2._gsync.pending == [] and not getattr(e2._gsync, "forked", False)
    assert len(e2.graph_d) == (6 if shard else 7)
    for i in range(7):
        e1.iteration(i)
        e2.iteration(i)
    torch.cuda.synchronize()
    assert torch.equal(e1.Dp.flat, e2.Dp.flat) and torch.equal(e1.Gp.flat, e2.Gp.flat)
    assert e1.d_stats() == e2.d_stats()
    del e1, e2
    gc.collect()
    torch.cuda.synchronize()


def test_trainer_world1_nccl_exits_cleanly(tmp_path):
    """The trainer CLI as a 1-rank torchrun-style launch on the real "nccl" (RCCL) backend, in a process of its own: the
    collectives are captured inside the step graphs, rank 0 writes the artefacts, and main() ends with the ordered
    shutdown (engines released before destroy_process_group) -- exit code 0, no abort at interpreter teardown."""
    import subprocess
    import sys
    n_cls, n_inst, L = 5, 21, 40
    X, y = O.synth_dataset(n_cls, n_inst, L, 4, 13, 122, seed=0)
    ddir = tmp_path / "ds" / "feature"
    ddir.mkdir(parents=True)
    npz = str(ddir / "raw_feature_synth.npz")
    np.savez(npz, features=X, labels=y)
    torch.manual_seed(0)
    ck = str(tmp_path / "df.ckpt")
    torch.save(O.make_df_state(L, n_cls), ck)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys, torch.distributed as dist\n"
            "from wfd_gan_amd import mlp_df_wgan_train as T, parallel\n"
            "dist.init_process_group('nccl', rank=0, world_size=1)\n"
            "parallel._OWN_PG = True\n"
            "out = T.main(sys.argv[1:])\n"
            "assert not dist.is_initialized()\n"
            "print('OUT', out)\n")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29537",
               HSA_ENABLE_IPC_MODE_LEGACY="0", WFD_FORCE_DP="1")
    p = subprocess.run([sys.executable, "-c", code, "-d", npz, "--f_model", ck, "--n_epochs", "2", "--batch_size", "16",
                        "--latent_dim", "24", "--seed", "7", "--freq", "2", "--dp", "weak"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    out = [ln for ln in p.stdout.splitlines() if ln.startswith("OUT ")][-1][4:]
    assert os.path.isfile(os.path.join(out, "model", "discriminator.ckpt"))
    assert os.path.isfile(os.path.join(out, "checkpoint", "loss.npz"))


def test_trainer_and_sampler_cli(tmp_path):
    from wfd_gan_amd import generate, mlp_df_wgan_train
    n_cls, n_inst, L = 5, 21, 40           # 105 rows, batch 16 -> 6 full batches + ragged tail of 9
    X, y = O.synth_dataset(n_cls, n_inst, L, 4, 13, 122, seed=0)
    ddir = tmp_path / "ds" / "feature"
    ddir.mkdir(parents=True)
    npz = str(ddir / "raw_feature_synth.npz")
    np.savez(npz, features=X, labels=y + 3)          # labels not starting at 0: the trainer re-indexes (:130)
    torch.manual_seed(0)
    ck = str(tmp_path / "df.ckpt")
    torch.save(O.make_df_state(L, n_cls), ck)
    out = mlp_df_wgan_train.main(["-d", npz, "--f_model", ck, "--n_epochs", "2", "--batch_size", "16",
                                  "--latent_dim", "24", "--seed", "7", "--freq", "2"])
    assert os.path.isfile(os.path.join(out, "args.json")) and os.path.isfile(os.path.join(out, "log.txt"))
    log = open(os.path.join(out, "log.txt")).read()
    assert len(re.findall(r"\[Epoch +\d+/ *2\] \[D loss: ", log)) == 2
    for ep in (1, 2):
        d = np.load(os.path.join(out, "checkpoint", f"epoch_{ep}.npz"))
        assert d["x"].dtype == np.float64 and d["recon_x"].dtype == np.float32 and d["label"].dtype == np.int64
        assert d["x"].shape == d["recon_x"].shape and d["x"].shape[1] == L + 1
        # 7 batches per epoch, G-steps at i = 0, 3, 6 -> 16 + 16 + 9 rows
        assert d["x"].shape[0] == 41
        assert np.allclose(d["x"], np.round(d["x"]))          # inverse-transformed real rows are the integer data
    ls = np.load(os.path.join(out, "checkpoint", "loss.npz"))
    assert all(len(ls[k]) == 2 for k in ("generator", "discriminator", "dist"))
    assert np.isfinite(ls["discriminator"]).all()
    gck = glob.glob(os.path.join(out, "model", "generator_seqlen41_cls5_latentdim24.ckpt"))
    assert len(gck) == 1
    gsd = torch.load(gck[0], weights_only=True)
    assert list(gsd.keys()) == list(O.make_generator_state(41, 5, 24).keys())
    assert int(gsd["model.3.num_batches_tracked"]) == 2 * (7 + 3)
    dsd = torch.load(os.path.join(out, "model", "discriminator.ckpt"), weights_only=True)
    assert list(dsd.keys()) == list(O.make_discriminator_state(41, 5).keys())
    assert os.path.isfile(os.path.join(out, "model", "scaler.gz"))
    # sampler on the artefacts: CLI + sample(c_ind) vs the oracle's sample() on the same z
    ts = generate.TraceSampler(os.path.join(out, "model"), device=DEV, chunk=64)
    outs, lens = ts.sample_many(torch.arange(100) % 5)
    assert outs.shape == (100, 41) and int(lens.min()) >= 1
    assert bool((outs[torch.arange(41)[None, :].to(DEV) < lens[:, None]] > 0).all())
    zr = np.random.RandomState(0).randn(1, 24).astype("float32")
    mine = ts.sample(2, zr)
    scale, min_ = O.scaler_params(ts.scaler.data_min_, ts.scaler.data_max_)
    want = O.sample({k: v for k, v in gsd.items()}, 2, zr, 5, ts.scaler.data_min_[0], ts.scaler.data_max_[0], scale, min_)
    assert len(mine) == len(want) and np.abs(mine - want).max() <= 1
    generate.main(["-m", os.path.join(out, "model"), "--n", "300", "--chunk", "128", "--out", str(tmp_path / "t.npz")])
    t = np.load(str(tmp_path / "t.npz"))
    assert t["traces"].shape == (300, 41) and t["lens"].shape == (300,)


@pytest.mark.parametrize("penalty", ["gp", "div"])
def test_trajectory_without_parameter_resets(penalty):
    """Multi-step drift: 6 iterations (G-steps at 0 and 3) with NO reset of the engine's parameters to the oracle's bits in
    between (the per-iteration gradient tests above reset them so that a LeakyReLU coin flip does not turn into a gradient
    mismatch).  Every loss scalar of the trajectory must stay within 1e-3 relative (+ the absolute floors of test_gpu_step)
    of the fp32 oracle's, plus twice the oracle's own fp32-vs-fp64 distance at that iteration -- the reference's own
    trajectory noise, which is what a different but equally valid fp32 summation order may pick up."""
    S, C, Z, B = 501, 10, 64, 32
    st, eng, Xs, y = _setup(S, C, Z, B, penalty, seed=6)
    st64 = O.make_state(S, C, Z, st.scaler_min, st.scaler_max, seed=6, df_seed=7, penalty=penalty, dtype=torch.float64)
    g = torch.Generator().manual_seed(2)
    perm = torch.randperm(Xs.shape[0], generator=g)
    worst = 0.0
    for it in range(6):
        idx = perm[torch.arange(it * B, (it + 1) * B) % perm.numel()].numpy()
        traces, labels = torch.from_numpy(Xs[idx]), torch.from_numpy(y[idx])
        z = torch.randn(B, Z, generator=g)
        alpha = torch.rand(B, 1, generator=g)
        mk = lambda: [(torch.rand(B, n, generator=g) > 0.2).float() * 1.25 for n in (2048, 1024, 512)]
        mr, mf, mgp, mg = mk(), mk(), mk(), mk()
        do_g = it % 3 == 0
        kw = dict(gp_alpha=alpha, masks_gp=mgp) if penalty == "gp" else {}
        d64 = lambda ms: [t.double() for t in ms]
        kw64 = dict(gp_alpha=alpha.double(), masks_gp=d64(mgp)) if penalty == "gp" else {}
        ref = O.iteration(st, traces, labels, z, do_g, mr, mf, mg, keep=True, **kw)
        ex = O.iteration(st64, traces.double(), labels, z.double(), do_g, d64(mr), d64(mf), d64(mg), keep=True, **kw64)
        eng.set_batch(traces, labels, z)
        if penalty == "gp":
            eng.set_gp_alpha(alpha)
            eng.set_masks([torch.cat([a, b, c]) for a, b, c in zip(mr, mf, mgp)], mg)
        else:
            eng.set_masks([torch.cat([a, b]) for a, b in zip(mr, mf)], mg)
        eng.d_step()
        if do_g:
            eng.g_step()
        torch.cuda.synchronize()
        d = eng.d_stats()
        got = dict(d)
        if do_g:
            got.update(eng.g_stats())
        for k, atol in (("penalty", 0.0), ("d_loss", 4e-5), ("w_dist", 4e-5), ("mean_rv", 2e-5), ("mean_fv", 2e-5)) + \
                ((("g_loss", 2e-5),) if do_g else ()):
            want, exact = float(ref[k]), float(ex[k])
            tol = 1e-3 * abs(want) + atol + 2 * abs(want - exact)
            worst = max(worst, abs(got[k] - want) / max(tol, 1e-30))
            assert abs(got[k] - want) <= tol, (penalty, it, k, got[k], want, exact)
    print(f"trajectory[{penalty}]: worst |got - want| / tolerance = {worst:.3f}")
