"""GPU: exact data-parallel mode (SURVEY.md 8e "strong": the global batch split over ranks, SyncBN, global selected-row
count, summed gradients) against the reference fixtures.  Two ranks share cuda:0 and exchange over gloo; each feeds
half of the fixture's batch, and the update must equal the reference's single-process update on the whole batch to
the fp32 tolerance (1e-3).  Also the one-rank degenerate case of the staged path in-process."""
import socket

import pytest
import torch
import torch.multiprocessing as mp

from tests.dp_exact_worker import generate_worker, trainer_worker, weak_worker, worker

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("name,world,mesh", [("tiny", 2, False), ("cfg2", 2, False), ("cfg2", 4, False), ("cfg2", 2, True),
                                             ("cfg2", 4, True)])
def test_exact_dp_equals_reference_update(name, world, mesh, monkeypatch):
    """mesh: the end of every step is the peer-to-peer exchange (WFD_DP_MESH=1: parameters and gradients in regions the
    ranks map from each other, one launch sums, applies RMSprop on the owner's shard and stores the parameters to all)."""
    monkeypatch.setenv("WFD_DP_MESH", "1" if mesh else "0")            # inherited by the spawned ranks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=280) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [m for _, msgs in out for m in msgs]
    assert not bad, "\n".join(bad)
    assert all(p.exitcode == 0 for p in procs)


@pytest.mark.parametrize("name,world,shard", [("cfg1", 2, False), ("cfg2", 3, False), ("cfg1", 2, True), ("cfg2", 3, True),
                                              ("cfg1", 2, "mesh"), ("cfg2", 3, "mesh"), ("cfg1", 8, "mesh")])
def test_weak_dp_buckets_sum_every_gradient(name, world, shard, monkeypatch):
    """Weak mode (per-rank batches): with the bucketed asynchronous all-reduce the flat gradient buffers hold the sum of the
    ranks' local gradients everywhere; with the sharded optimizer (WFD_DP_SHARD=1) every rank's shards hold it and the
    parameters take the step of the sum; with the peer-to-peer exchange (WFD_DP_MESH=1) the sum only exists in the owner's
    registers and the parameters take its step.  In every form the replicas stay bit-identical (ranks share cuda:0; gloo for
    the first two, for the third only the rendezvous)."""
    monkeypatch.setenv("WFD_DP_SHARD", "1" if shard is True else "0")          # inherited by the spawned ranks
    monkeypatch.setenv("WFD_DP_MESH", "1" if shard == "mesh" else "0")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=weak_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=280) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [m for _, msgs in out for m in msgs]
    assert not bad, "\n".join(bad)
    assert all(p.exitcode == 0 for p in procs)


def test_exact_mode_single_rank_is_plain_step():
    """sync_bn=True at world 1 runs the split BN kernels and the host stages but must reproduce the fused path."""
    from tests.test_gpu_step import make_engine
    from tests.util import load_iter_fixture, rebuild_state, unpack_masks
    from wfd_gan_amd.engine import TrainEngine
    fx, meta = load_iter_fixture("cfg1")
    B = meta["batch"]
    engs = []
    for exact in (False, True):
        st = rebuild_state(meta, fx)
        if not exact:
            eng = make_engine(st, meta, B)
        else:
            eng = TrainEngine(meta["seq_len"], meta["class_dim"], meta["latent_dim"], B, meta["scaler_min"],
                              meta["scaler_max"], {k: v.detach() for k, v in st.G.items()},
                              {k: v.detach() for k, v in st.D.items()}, st.DF, lr=meta["lr"], alpha=meta["alpha"],
                              n_critic=meta["n_critic"], sync_bn=True, force_sync=True)
        for it in range(2):
            p = f"it{it}_"
            mr, mf, mg = (unpack_masks(fx, p + k, B) for k in ("mreal", "mfake", "mg"))
            eng.set_batch(torch.from_numpy(fx[p + "traces"]), torch.from_numpy(fx[p + "labels"]),
                          torch.from_numpy(fx[p + "z"]))
            eng.set_masks([torch.cat([a, b]) for a, b in zip(mr, mf)], mg)
            eng.d_step()
            if it == 0:
                eng.g_step()
        torch.cuda.synchronize()
        engs.append(eng)
    a, b = engs
    for x, y in ((a.Dp.flat, b.Dp.flat), (a.Gp.flat, b.Gp.flat)):
        assert float((x - y).abs().max()) <= 1e-6 * float(x.abs().max())


@pytest.mark.parametrize("mode,mesh", [("strong", False), ("weak", False), ("weak", True), ("strong", True)])
def test_trainer_cli_two_ranks(tmp_path, mode, mesh, monkeypatch):
    """The trainer under a 2-rank launch: rank 0 writes the reference's artefacts, the per-epoch files hold every
    rank's generated rows, the ragged tail is split over the ranks.  mesh: with the peer-to-peer exchange (WFD_DP_MESH=1;
    the tail-batch engines share the first engine's mapped region)."""
    import glob
    monkeypatch.setenv("WFD_DP_MESH", "1" if mesh else "0")            # inherited by the spawned ranks
    import os
    import re

    import numpy as np
    from oracle import wfdgan_oracle as O
    n_cls, n_inst, L = 5, 24, 40           # 120 rows
    X, y = O.synth_dataset(n_cls, n_inst, L, 4, 13, 122, seed=0)
    ddir = tmp_path / "ds" / "feature"
    ddir.mkdir(parents=True)
    npz = str(ddir / "raw_feature_synth.npz")
    np.savez(npz, features=X, labels=y)
    torch.manual_seed(0)
    ck = str(tmp_path / "df.ckpt")
    torch.save(O.make_df_state(L, n_cls), ck)
    # strong: global batch 16 = 2 x 8 -> 7 full + tail 8 (4 per rank); weak: 2 x 16 -> 3 full + tail 24 (12 per rank)
    argv = ["-d", npz, "--f_model", ck, "--n_epochs", "2", "--batch_size", "16", "--latent_dim", "24", "--seed", "7",
            "--freq", "2", "--dp", mode]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=trainer_worker, args=(r, 2, port, argv, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    errs = [e for _, _, e in res if e]
    assert not errs, "\n".join(errs)
    outs = {r: o for r, o, _ in res}
    out = outs[0]
    assert outs[1] is None and os.path.isdir(out)
    log = open(os.path.join(out, "log.txt")).read()
    assert len(re.findall(r"\[Epoch +\d+/ *2\] \[D loss: ", log)) == 2
    d = np.load(os.path.join(out, "checkpoint", "epoch_2.npz"))
    # strong: 8 iterations, G-steps at i = 0, 3, 6 on 16 global rows; weak: 4 iterations, G-steps at i = 0 (32 rows)
    # and i = 3 (the tail, 24 rows)
    assert d["x"].shape == ((48, L + 1) if mode == "strong" else (56, L + 1))
    assert np.allclose(d["x"], np.round(d["x"])) and np.isfinite(d["recon_x"]).all()
    ls = np.load(os.path.join(out, "checkpoint", "loss.npz"))
    assert all(len(ls[k]) == 2 and np.isfinite(ls[k]).all() for k in ("generator", "discriminator", "dist"))
    assert len(glob.glob(os.path.join(out, "model", "generator_seqlen41_cls5_latentdim24.ckpt"))) == 1
    assert len(glob.glob(str(tmp_path / "ds" / "training_*"))) == 1          # only rank 0 made an output directory


@pytest.mark.parametrize("mode,mesh", [("weak", False), ("strong", False), ("weak", True)])
def test_bench_two_ranks_share_one_gpu(mode, mesh):
    """bench.py under a 2-rank launch (the driver's N > 1 form), here with both ranks on cuda:0 over gloo: the
    data-parallel plans, graph stages, bucketed gradient sync (or, mesh, the peer-to-peer launch: the weak-mode group then
    stays ONE graph even on this rig, there being no host-side collective left in it) and the JSON contract."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), WFD_DIST_BACKEND="gloo", WFD_DP_MESH="1" if mesh else "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6",
                                       "--warmup", "3", "--inst", "20", "--no-sampling", "--no-cpu-baseline", "--dp",
                                       mode], env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=280) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-2000:] for o in outs)
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["scaling"] == mode and res["value"] > 0
    assert res["config"]["global_batch"] == (128 if mode == "weak" else 64)
    assert res["config"]["dp_form"] == ("mesh" if mesh else "allreduce")
    steps = res["config"]["optimizer_steps_per_s"]
    assert abs(res["value"] - steps * (2 if mode == "weak" else 1)) < 1e-6 * res["value"]
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]          # only rank 0 prints


def test_bench_plain_launch_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO torchrun environment (the shape of the driver's 1-GPU command): bench.py starts
    the two ranks itself before any GPU call, relays rank 0's line and reports what the collective saw.  Both ranks share
    cuda:0 over gloo here (WFD_DIST_BACKEND=gloo; one GPU on the test box)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["WFD_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3",
                        "--inst", "20", "--no-sampling", "--no-cpu-baseline"], env=env, cwd=root, capture_output=True,
                       text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line for the whole job
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks_in_collective"] == 2 and res["dist_backend"] == "gloo"
    assert res["launcher"] == "self" and res["scaling"] == "weak" and res["config"]["global_batch"] == 128
    # a wrong --gpus under a torchrun-style environment is refused, never silently run on fewer ranks
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0"],
                        env=env2, cwd=root, capture_output=True, text=True, timeout=120)
    assert p2.returncode != 0 and "WORLD_SIZE=1" in p2.stderr
    # more ranks than devices on the real backend: refused with a clear message before anything runs
    env3 = {k: v for k, v in env.items() if k != "WFD_DIST_BACKEND"}
    n = torch.cuda.device_count() + 1
    p3 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "0"],
                        env=env3, cwd=root, capture_output=True, text=True, timeout=120)
    assert p3.returncode != 0 and "refusing to run fewer ranks" in p3.stderr and not p3.stdout.strip()


@pytest.mark.parametrize("world", [2, 3])
def test_generate_cli_shards_the_label_list(world, tmp_path):
    """generate.py under torchrun (src/generate.py's sample() in bulk, SURVEY.md 8e: embarrassingly parallel, no collective):
    the ranks' files together hold labels arange(n) % C in order, shard sizes differ by at most one, and every rank's traces
    are bit-identical to a single-process sampler with that rank's seed on that rank's labels."""
    import numpy as np
    from oracle import wfdgan_oracle as O
    from wfd_gan_amd import generate, parallel
    S, C, Z, n = 41, 5, 24, 301
    mdir = tmp_path / "model"
    mdir.mkdir()
    torch.manual_seed(5)
    torch.save(O.make_generator_state(S, C, Z), str(mdir / f"generator_seqlen{S}_cls{C}_latentdim{Z}.ckpt"))
    dmin, dmax = np.zeros(S), np.concatenate([[40.0], np.tile([13.0, 122.0], S // 2)])[:S]
    np.savez(str(mdir / "scaler.npz"), data_min=dmin, data_max=dmax)
    outp = str(tmp_path / "t.npz")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=generate_worker, args=(r, world, port, str(mdir), outp, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=280) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [m for _, msgs in out for m in msgs]
    assert not bad, "\n".join(bad)
    assert all(p.exitcode == 0 for p in procs)
    labels, sizes = [], []
    for r in range(world):
        f = np.load(outp.replace(".npz", f".rank{r}.npz"))
        a, b = parallel.shard_rows(n, r, world)
        assert f["traces"].shape == (b - a, S) and f["lens"].shape == (b - a,)
        labels.append(f["labels"])
        sizes.append(b - a)
        ts = generate.TraceSampler(str(mdir), device="cuda:0", chunk=64, seed=3 + r)
        want, wlens = ts.sample_many(torch.arange(a, b) % C)
        assert np.array_equal(f["traces"], want.cpu().numpy()) and np.array_equal(f["lens"], wlens.cpu().numpy())
        assert (f["lens"] >= 1).all() and (f["traces"][np.arange(S)[None, :] < f["lens"][:, None]] > 0).all()
    assert np.array_equal(np.concatenate(labels), np.arange(n) % C)
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n
