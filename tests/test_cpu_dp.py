"""CPU, world_size 2 over gloo: the data-parallel path's collective logic.  Each rank computes oracle gradients on
its own rows; allreduce_sum_ + the 1/world scale must reproduce the average of the per-rank gradients (DDP
semantics), and replicas must stay identical after the optimizer step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from oracle import wfdgan_oracle as O
    from wfd_gan_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    S, C, Z, B = 41, 5, 16, 8
    X, y = O.synth_dataset(C, 8, S - 1, 4, 13, 122, seed=3)
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    Xs = torch.from_numpy(X * scale + min_)
    a, b = parallel.shard_rows(2 * B, rank, world)            # this rank's rows of the global batch
    st = O.make_state(S, C, Z, dmin[0], dmax[0], seed=1, df_seed=2)
    g = torch.Generator().manual_seed(100 + rank)
    z = torch.randn(b - a, Z, generator=g)
    masks = [[(torch.rand(b - a, n, generator=g) > 0.2).float() * 1.25 for n in (2048, 1024, 512)] for _ in range(3)]
    res = O.iteration(st, Xs[a:b], torch.from_numpy(y[a:b]), z, True, *masks, keep=True)
    local = torch.cat([t.flatten() for t in res["d_grads"]])
    flat = local.clone()
    parallel.allreduce_sum_(flat)
    flat /= world
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered) / world
    ok = torch.allclose(flat, want, rtol=0, atol=1e-7)
    # bucketed asynchronous form used by the engine: two slices of one flat buffer
    flat2 = local.clone()
    gs = parallel.GradSync()
    cut = flat2.numel() // 3
    gs.start(flat2[:cut])
    gs.start(flat2[cut:])
    gs.finish()
    ok = ok and torch.allclose(flat2 / world, want, rtol=0, atol=1e-7) and not gs.pending
    # SyncBN combine: (mean, M2) of two row shards -> the pair over all rows
    xg = torch.Generator().manual_seed(5)
    xs = torch.randn(12, 7, generator=xg, dtype=torch.float64) * 3 + 1
    mine = xs[6 * rank: 6 * (rank + 1)]
    mean, m2 = mine.mean(0), ((mine - mine.mean(0)) ** 2).sum(0)
    comm = parallel.Comm()
    comm.combine_bn_stats_(mean, m2, 6)
    ok = ok and torch.allclose(mean, xs.mean(0), atol=1e-12) and torch.allclose(m2, ((xs - xs.mean(0)) ** 2).sum(0), atol=1e-10)
    ok = ok and comm.all_gather(torch.tensor([float(rank)])).flatten().tolist() == [0.0, 1.0]
    ok = ok and comm.all_reduce_sum_(torch.tensor([1.0 + rank])).item() == 3.0
    tmax = parallel.max_over_ranks(float(rank + 1), "cpu")
    tsum = parallel.sum_over_ranks(1.0, "cpu")
    t = torch.full((3,), float(rank))
    parallel.broadcast_state_([t])
    q.put((rank, bool(ok), tmax, tsum, t.tolist()))
    dist.destroy_process_group()


def test_gradient_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, tmax, tsum, t in out:
        assert ok, f"rank {rank}: all-reduced gradient != mean of per-rank gradients"
        assert tmax == 2.0 and tsum == 2.0
        assert t == [0.0, 0.0, 0.0]


class _HostRmsprop:
    """Stand-in for the library's wfd_rmsprop on HOST memory (same argument list), so that parallel.ShardedSync's bucket /
    shard / gather logic can run over gloo without a GPU.  Not a product path: the product passes the HIP library."""

    def rmsprop(self, p, g, sq, n, lr, alpha, eps, gscale, state, inc0, inc1, stream):
        import ctypes
        arr = lambda ptr: np.ctypeslib.as_array((ctypes.c_float * n).from_address(ptr))
        pv, gv, sv = arr(p), arr(g), arr(sq)
        gs = (gv * np.float32(gscale)).astype(np.float32)
        sv[:] = sv * np.float32(alpha) + np.float32(1.0 - alpha) * gs * gs
        pv[:] = pv - np.float32(lr) * gs / (np.sqrt(sv) + np.float32(eps))


def _shard_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from wfd_gan_amd import parallel
    parallel.init_from_env(backend="gloo")
    n, lr = 12 * world * 7, 2e-4                    # three buckets, each a multiple of the world size
    buckets = [(8 * world * 7, n), (3 * world * 7, 8 * world * 7), (0, 3 * world * 7)]      # backward order, as grad_buckets()
    g0 = torch.Generator().manual_seed(1)
    flat = torch.randn(n, generator=g0)             # identical replicas
    sq = torch.rand(n, generator=g0)
    g = torch.Generator().manual_seed(10 + rank)
    grad = torch.randn(n, generator=g)              # this rank's local gradient
    parts = [torch.empty_like(grad) for _ in range(world)]
    dist.all_gather(parts, grad)
    total = torch.stack(parts).sum(0)
    want_p, want_sq = flat.clone(), sq.clone()
    _HostRmsprop().rmsprop(want_p.data_ptr(), total.data_ptr(), want_sq.data_ptr(), n, lr, 0.99, 1e-8, 1.0 / world, 0, 0, 0, 0)
    ss = parallel.ShardedSync(_HostRmsprop())
    for lo, hi in buckets:
        ss.start(flat, grad, sq, lo, hi, lr, 1.0 / world)
    ss.finish()
    ok = torch.allclose(flat, want_p, rtol=0, atol=1e-7)          # every element took the step of the SUMMED gradient
    reps = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(reps, flat)
    ok = ok and all(torch.equal(reps[0], t) for t in reps[1:])    # replicas bit-identical
    # the square average is kept by the owner only: own shards updated, the others untouched
    for lo, hi in buckets:
        sh = (hi - lo) // world
        a = lo + rank * sh
        ok = ok and torch.allclose(sq[a:a + sh], want_sq[a:a + sh], rtol=0, atol=1e-7)
    try:
        ss.start(flat, grad, sq, 0, world + 1, lr, 1.0)
        ok = False                                                # a bucket that does not split evenly must be refused
    except ValueError:
        pass
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_optimizer_exchange_over_gloo(world):
    """parallel.ShardedSync (reduce-scatter -> RMSprop on the rank's shard -> parameter all-gather, per bucket) over gloo on
    host tensors: every parameter takes the step of the summed gradient, replicas stay bit-identical."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(ok for _, ok in res), res


def test_bench_plain_multi_gpu_launch_is_refused_without_devices():
    """bench.py --gpus 2 with no torchrun environment starts its own ranks -- or, with fewer HIP devices than ranks (none in
    the build container), exits non-zero with a clear message instead of silently running one."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("enough devices: the launch would run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "WFD_DIST_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "refusing to run fewer ranks" in p.stderr and not p.stdout.strip()
