"""Worker of tests/test_gpu_dp_exact.py: one rank of the exact (global-batch, SyncBN) data-parallel mode.  The ranks
share cuda:0 and talk over gloo (device tensors staged through the host by wfd_gan_amd.parallel.Comm), so that the
strong-scaling arithmetic can be checked on a one-GPU box.  Each rank feeds its slice of the reference fixture's
batch; rank 0 compares the result with the fixture (= the reference's single-process update on the whole batch)."""
import os

import numpy as np
import torch

RTOL = 1e-3


def _close(a, b, rtol=RTOL, atol=0.0):
    return abs(float(a) - float(b)) <= rtol * abs(float(b)) + atol


def worker(rank, world, port, name, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    try:
        q.put((rank, _run(rank, world, name)))
    except Exception as e:  # noqa: BLE001 - report instead of hanging the parent on q.get
        import traceback
        q.put((rank, [f"rank {rank} raised: {e!r}\n{traceback.format_exc()}"]))


def _run(rank, world, name):
    import torch.distributed as dist
    from tests.util import load_iter_fixture, rebuild_state, rel_err, unpack_masks
    from wfd_gan_amd import parallel
    from wfd_gan_amd.engine import TrainEngine
    parallel.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    fx, meta = load_iter_fixture(name)
    st = rebuild_state(meta, fx)
    Bg = meta["batch"]
    B = Bg // world
    a, b = rank * B, (rank + 1) * B
    eng = TrainEngine(meta["seq_len"], meta["class_dim"], meta["latent_dim"], B, meta["scaler_min"], meta["scaler_max"],
                      {k: v.detach() for k, v in st.G.items()}, {k: v.detach() for k, v in st.D.items()}, st.DF,
                      lr=meta["lr"], alpha=meta["alpha"], n_critic=meta["n_critic"], world_size=world, rank=rank,
                      sync_bn=True)
    assert eng.Bg == Bg and eng.gscale == 1.0
    bad = []
    for it in range(meta["n_iter"]):
        p = f"it{it}_"
        do_g = it % meta["n_critic"] == 0
        mr, mf, mg = (unpack_masks(fx, p + k, Bg) for k in ("mreal", "mfake", "mg"))
        eng.set_batch(torch.from_numpy(fx[p + "traces"][a:b]), torch.from_numpy(fx[p + "labels"][a:b]),
                      torch.from_numpy(fx[p + "z"][a:b]))
        eng.set_masks([torch.cat([x[a:b], y[a:b]]) for x, y in zip(mr, mf)], None if mg is None else [m[a:b] for m in mg])
        eng.d_step()
        torch.cuda.synchronize()
        s = eng.d_stats()
        d_loss, pen, w_dist, mrv, mfv = fx[p + "scalars"]
        chk = [("penalty", _close(s["penalty"], pen)), ("mean_rv", _close(s["mean_rv"], mrv, atol=2e-5)),
               ("mean_fv", _close(s["mean_fv"], mfv, atol=2e-5)), ("d_loss", _close(s["d_loss"], d_loss, atol=4e-5)),
               ("w_dist", _close(s["w_dist"], w_dist, atol=4e-5)),
               ("idx", np.array_equal(eng.buf["idx"].cpu().numpy(), fx[p + "idx_d"][a:b])),
               ("n_r", rel_err(eng.buf["PEN"][:B].cpu().numpy(), fx[p + "n_r"][a:b]) < RTOL),
               ("n_f", rel_err(eng.buf["PEN"][B:2 * B].cpu().numpy(), fx[p + "n_f"][a:b]) < RTOL)]
        if do_g:
            eng.g_step()
            torch.cuda.synchronize()
            g = eng.g_stats()
            g_loss, f_loss, comb, nsel = fx[p + "g_scalars"]
            chk += [("g_loss", _close(g["g_loss"], g_loss, atol=2e-5)), ("n_sel", g["n_sel"] == int(nsel)),
                    ("f_loss", _close(g["f_loss"], f_loss, atol=2e-5)),
                    ("combined", _close(g["g_combined"], comb, atol=4e-5))]
        dnorm = np.array([float(eng.Dp.p[n].double().norm()) for n, _ in eng.Dp.specs])
        gnorm = np.array([float(eng.Gp.p[n].double().norm()) for n, _ in eng.Gp.specs])
        bn = torch.cat([eng.Gbuf[f"model.{i}.{s}"] for i in (3, 6) for s in ("running_mean", "running_var")])
        chk += [("D param norms", rel_err(dnorm, fx[p + "d_param_norms"]) < RTOL),
                ("G param norms", rel_err(gnorm, fx[p + "g_param_norms"]) < RTOL),
                ("BN running stats", rel_err(bn.cpu().numpy(), fx[p + "g_bn_stats"]) < RTOL)]
        bad += [f"rank {rank} it{it} {k}" for k, ok in chk if not ok]
    # replicas must hold identical parameters after the run
    flat = torch.cat([eng.Dp.flat, eng.Gp.flat]).cpu()
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    if not all(torch.equal(both[0], t) for t in both[1:]):
        bad.append(f"rank {rank}: replicas diverged (max |d| = {float((both[0] - both[-1]).abs().max()):.3e})")
    bad += _mesh_report(eng, rank)
    del eng
    parallel.shutdown(destroy=True)
    return bad


def _mesh_report(eng, rank):
    """WFD_DP_MESH=1 (the peer-to-peer exchange, parallel.MeshSync): it must be what ran, and no wait may have timed out."""
    want = os.environ.get("WFD_DP_MESH", "0") == "1"
    if bool(eng.mesh_dp) != want:
        return [f"rank {rank}: mesh_dp = {eng.mesh_dp}, WFD_DP_MESH asked for {want}"]
    if not want:
        return []
    done, timeouts = eng._mesh.status()
    bad = [] if done > 0 and timeouts == 0 else [f"rank {rank}: mesh exchanges {done}, timed-out waits {timeouts}"]
    # with every exchange a plain launch, no host-staged collective is left to split a step: one stage (= one HIP graph) each,
    # also on this rig (gloo between the ranks), in the weak AND in the exact mode
    for nme, pl in (("D-step", eng.plan_d), ("G-step", eng.plan_g)):
        if len(pl.stages()) != 1:
            bad.append(f"rank {rank}: the {nme} plan of the mesh form splits into {len(pl.stages())} stages")
    return bad


def trainer_worker(rank, world, port, argv, q):
    """One torchrun-style rank of the trainer CLI (ranks share cuda:0, gloo)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), WFD_DIST_BACKEND="gloo")
    try:
        import torch.distributed as dist
        from wfd_gan_amd import mlp_df_wgan_train
        out = mlp_df_wgan_train.main(argv)
        # main() ends with parallel.shutdown(): barrier (ranks != 0 outlive rank 0's writes) -> engines released ->
        # synchronize -> destroy_process_group, because it created the group itself
        assert not dist.is_initialized(), "the trainer left its process group behind"
        q.put((rank, out, None))
    except BaseException as e:  # noqa: BLE001
        import traceback
        q.put((rank, None, f"rank {rank} raised: {e!r}\n{traceback.format_exc()}"))


def weak_worker(rank, world, port, name, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    try:
        q.put((rank, _run_weak(rank, world, name)))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, [f"rank {rank} raised: {e!r}\n{traceback.format_exc()}"]))


def _run_weak(rank, world, name):
    """Weak (DDP-semantics) mode: every rank steps on its OWN batch (rank r feeds the fixture's iteration-r inputs); the
    bucketed exchange must apply the SUM of the ranks' local gradients to every element of both parameter buffers (an element
    outside every bucket / shard would keep its local update and differ between the ranks), and the replicas must stay
    bit-identical.  Runs the sharded optimizer (default) -- WFD_DP_SHARD=0 in the environment checks round 2's all-reduce form."""
    import torch.distributed as dist
    from tests.util import load_iter_fixture, rebuild_state, unpack_masks
    from wfd_gan_amd import parallel
    from wfd_gan_amd.engine import TrainEngine
    parallel.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    fx, meta = load_iter_fixture(name)
    B = meta["batch"]

    def mk(w):
        st = rebuild_state(meta, fx)
        return TrainEngine(meta["seq_len"], meta["class_dim"], meta["latent_dim"], B, meta["scaler_min"], meta["scaler_max"],
                           {k: v.detach() for k, v in st.G.items()}, {k: v.detach() for k, v in st.D.items()}, st.DF,
                           lr=meta["lr"], alpha=meta["alpha"], n_critic=meta["n_critic"], world_size=w, rank=rank if w > 1 else 0)
    eng, loc = mk(world), mk(1)
    assert eng.dp and not loc.dp and abs(eng.gscale - 1.0 / world) < 1e-12
    # the three buckets tile each flat gradient buffer exactly
    for which, fp in (("D", eng.Dp), ("G", eng.Gp)):
        cover = sorted(eng.grad_buckets(which))
        assert cover[0][0] == 0 and cover[-1][1] == fp.numel and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    p = f"it{rank % meta['n_iter']}_"
    mr, mf, mg = (unpack_masks(fx, p + k, B) for k in ("mreal", "mfake", "mg"))
    if mg is None:
        mg = unpack_masks(fx, "it0_mg", B)
    bad = []

    def gathered(t):
        h = t.detach().cpu()
        parts = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(parts, h)
        return parts
    for e in (eng, loc):
        e.set_batch(torch.from_numpy(fx[p + "traces"]), torch.from_numpy(fx[p + "labels"]), torch.from_numpy(fx[p + "z"]))
        e.set_masks([torch.cat([x, y]) for x, y in zip(mr, mf)], mg)
    before = eng.Dp.flat.clone()
    loc.d_step()
    eng.d_step()
    torch.cuda.synchronize()
    # the local engine's flat layout differs (bucket starts are padded to 4 * world elements in the data-parallel one): map
    # the summed local gradients into the data-parallel engine's layout tensor by tensor
    want = torch.zeros(eng.Dp.numel, dtype=torch.float64)
    for nme, shape in eng.Dp.specs:
        eng.Dp._view(want, nme, shape).copy_(torch.stack(gathered(loc.Dp.g[nme].contiguous())).double().sum(0))
    got = gathered(eng.Dp.grad)
    if eng.mesh_dp:
        # peer-to-peer form: the sum is formed in registers by the owner of each element and never stored; the buffers keep
        # the LOCAL gradients, and the parameters every rank ends up with are the RMSprop step of their sum
        summed = torch.stack(got).double().sum(0)
        err = float((summed - want).abs().max()) / float(want.abs().max())
        if err > 1e-6:
            bad.append(f"rank {rank}: the ranks' local critic gradients do not add up to the expected sum ({err:.2e})")
        g = want.float() / world
        sq = 0.01 * g * g
        step = -meta["lr"] * g / (sq.sqrt() + 1e-8)
        moved = (eng.Dp.flat.cpu() - before.cpu()).double()
        big = g.abs() > 1e-6 * float(g.abs().max())
        perr = float((moved - step.double())[big].abs().max())
        if perr > 1e-3 * 10 * meta["lr"]:
            bad.append(f"rank {rank}: critic parameters after the mesh step off by {perr:.2e}")
    elif eng.shard:
        # sharded optimizer: after the reduce-scatter every rank holds the SUM on its own shard of each bucket; the shards
        # of all ranks together tile the buffer exactly, and the parameters every rank ends up with are the RMSprop step of
        # that sum (owner's shard, gathered)
        covered = torch.zeros(eng.Dp.numel, dtype=torch.int32)
        summed = torch.zeros(eng.Dp.numel, dtype=torch.float64)
        for lo, hi in eng.grad_buckets("D"):
            sh = (hi - lo) // world
            if (hi - lo) % world:
                bad.append(f"bucket [{lo}, {hi}) does not split over {world} ranks")
            for r in range(world):
                a = lo + r * sh
                covered[a:a + sh] += 1
                summed[a:a + sh] = got[r][a:a + sh].double()
        if not bool((covered == 1).all()):
            bad.append("the ranks' shards do not tile the critic's gradient buffer exactly")
        err = float((summed - want).abs().max()) / float(want.abs().max())
        if err > 1e-6:
            bad.append(f"rank {rank}: reduce-scattered critic gradient off by {err:.2e}")
        g = want.float() / world
        sq = 0.01 * g * g
        step = -meta["lr"] * g / (sq.sqrt() + 1e-8)
        moved = (eng.Dp.flat.cpu() - before.cpu()).double()
        big = g.abs() > 1e-6 * float(g.abs().max())
        perr = float((moved - step.double())[big].abs().max())
        if perr > 1e-3 * 10 * meta["lr"]:
            bad.append(f"rank {rank}: critic parameters after the sharded step off by {perr:.2e}")
    else:
        if not all(torch.equal(got[0], t) for t in got[1:]):
            bad.append(f"rank {rank}: critic gradient buffers differ between ranks after the all-reduce")
        err = float((got[0].double() - want).abs().max()) / float(want.abs().max())
        if err > 1e-6:
            bad.append(f"rank {rank}: summed critic gradient off by {err:.2e}")
    eng.g_step()
    eng.d_step()
    torch.cuda.synchronize()
    # (sharded optimizer: gradients and square averages are only kept on the owner's shard; the PARAMETERS must be identical)
    for nme, t in ((("G params", eng.Gp.flat), ("D params", eng.Dp.flat)) if eng.shard or eng.mesh_dp else
                   (("G grad", eng.Gp.grad), ("D grad", eng.Dp.grad), ("G params", eng.Gp.flat), ("D params", eng.Dp.flat),
                    ("G sq", eng.Gp.sq), ("D sq", eng.Dp.sq))):
        parts = gathered(t)
        if not all(torch.equal(parts[0], x) for x in parts[1:]):
            bad.append(f"rank {rank}: {nme} differs between ranks (max |d| = {float((parts[0] - parts[-1]).abs().max()):.3e})")
    if float(eng.Gp.grad.abs().max()) == 0.0:
        bad.append("generator gradient is zero")
    bad += _mesh_report(eng, rank)
    del eng, loc
    parallel.shutdown(destroy=True)
    return bad


def generate_worker(rank, world, port, mdir, outp, n, q):
    """`python -m wfd_gan_amd.generate` under torchrun: every rank samples its own contiguous shard of the label list with
    its own seed and writes <out>.rank<r>.npz; no collective.  Ranks share cuda:0 (LOCAL_RANK 0), gloo."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      WFD_DIST_BACKEND="gloo")
    msgs = []
    try:
        from wfd_gan_amd import generate
        generate.main(["-m", mdir, "--n", str(n), "--chunk", "64", "--seed", "3", "--out", outp])
        assert not torch.distributed.is_initialized(), "generate.main() left its process group behind"
    except Exception as e:          # noqa: BLE001
        msgs.append(f"rank {rank}: {type(e).__name__}: {e}")
    q.put((rank, msgs))
