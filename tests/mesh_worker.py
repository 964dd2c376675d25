"""One rank of the mesh-exchange tests (tests/test_gpu_mesh.py starts WORLD_SIZE of these on the one GPU of the test box:
the ranks' regions are then mapped between PROCESSES on one device instead of between devices -- same code path, no xGMI).
Rendezvous and the expected values travel over gloo; the exchange itself is the library's kernel."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfd_gan_amd import parallel  # noqa: E402
from wfd_gan_amd._lib import lib  # noqa: E402


def gather_host(t: torch.Tensor, world: int):
    h = t.detach().cpu().contiguous()
    if world == 1:
        return [h]
    parts = [torch.empty_like(h) for _ in range(world)]
    dist.all_gather(parts, h)
    return parts


def main():
    out_path = sys.argv[1]
    rank, world, _ = parallel.init_from_env("gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    L = lib()
    res = {"rank": rank, "world": world}
    sizes = [4, 1024, 4 * 7 * 13, 1 << 20, 3074048 + 2048]      # ragged shards (n / 4 not a multiple of world) included
    N = max(sizes)
    mesh = parallel.MeshSync(L, 2 * N, dev)
    grad, par = mesh.view(0, N), mesh.view(N, N)
    res["sharing"] = mesh.sharing
    ok_share = mesh.sharing == world              # the rig: every rank on the one GPU, and MeshSync must have noticed
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    ok = True
    # ---- all-reduce, eager, several sizes back to back (flags count up; a second call must not see the first one's state)
    for n in sizes:
        x = torch.randn(n, generator=g)
        grad[:n].copy_(x)
        torch.cuda.synchronize()
        parts = gather_host(grad[:n], world)
        want = parts[0].clone()
        for q in parts[1:]:
            want += q                                 # rank order, fp32: what the owner of every element computes
        L.mesh_allreduce_f32(mesh.mesh, mesh.offset_of(grad), n, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = grad[:n].cpu()
        same = bool(torch.equal(got, want))
        ok &= same
        res[f"allreduce_{n}"] = same
    # ---- the exact mode's small one-shot exchanges: vectors from the region's arena, summed / Chan-combined in rank order
    st = torch.cuda.current_stream().cuda_stream
    n_small = 0
    for n in (2, 1000, 4096):
        v = mesh.alloc_small(n)
        v.copy_(torch.randn(n, generator=g))
        torch.cuda.synchronize()
        parts = gather_host(v, world)
        want = parts[0].clone()
        for q in parts[1:]:
            want += q
        L.mesh_allreduce_small(mesh.mesh, mesh.offset_of(v), n, st)
        torch.cuda.synchronize()
        same = bool(torch.equal(v.cpu(), want))
        res[f"small_sum_{n}"] = same
        ok &= same
        n_small += 1
    n, rows = 2048, 8.0
    mean, m2 = mesh.alloc_small(n), mesh.alloc_small(n)
    mean.copy_(torch.randn(n, generator=g))
    m2.copy_(torch.rand(n, generator=g) * 5)
    torch.cuda.synchronize()
    pm, pq = gather_host(mean, world), gather_host(m2, world)
    tot = pm[0].clone()
    for q in pm[1:]:
        tot += q
    mu = tot / float(world)
    sm, sd = pq[0].clone(), (pm[0] - mu) * (pm[0] - mu)
    for a_, b_ in zip(pm[1:], pq[1:]):
        sm += b_
        sd += (a_ - mu) * (a_ - mu)
    want_m2 = sm + rows * sd
    L.mesh_bn_combine(mesh.mesh, mesh.offset_of(mean), mesh.offset_of(m2), n, rows, st)
    torch.cuda.synchronize()
    res["bn_combine_mean"] = bool(torch.equal(mean.cpu(), mu))
    res["bn_combine_m2"] = bool(torch.allclose(m2.cpu(), want_m2, rtol=1e-6, atol=0))
    res["bn_combine_m2_bits"] = bool(torch.equal(m2.cpu(), want_m2))
    ok &= res["bn_combine_mean"] and res["bn_combine_m2"]
    n_small += 1
    # ---- fused reduce + RMSprop + parameter push vs wfd_rmsprop on the summed gradients
    n = sizes[-1]
    p0 = torch.randn(n, generator=torch.Generator().manual_seed(7))            # identical replicas
    sq0 = torch.rand(n, generator=torch.Generator().manual_seed(8))
    sq = sq0.to(dev)
    par[:n].copy_(p0)
    grad[:n].copy_(torch.randn(n, generator=g))
    torch.cuda.synchronize()
    parts = gather_host(grad[:n], world)
    gsum = parts[0].clone()
    for q in parts[1:]:
        gsum += q
    rp, rg, rs = p0.to(dev), gsum.to(dev), sq0.to(dev)
    st = torch.cuda.current_stream().cuda_stream
    L.rmsprop(rp.data_ptr(), rg.data_ptr(), rs.data_ptr(), n, 5e-5, 0.99, 1e-8, 1.0 / world, 0, 0, 0, st)
    L.mesh_reduce_rmsprop(mesh.mesh, mesh.offset_of(grad), mesh.offset_of(par), sq.data_ptr(), n, 5e-5, 0.99, 1e-8,
                          1.0 / world, 0, 0, 0, st)
    torch.cuda.synchronize()
    res["fused_params_equal_flat_pass"] = bool(torch.equal(par[:n], rp))
    per = -(-(n // 4) // world) * 4
    a, b = rank * per, min((rank + 1) * per, n)
    res["fused_sq_own_shard"] = bool(torch.equal(sq[a:b], rs[a:b]))
    untouched = torch.cat([sq[:a], sq[b:]]).cpu()
    res["fused_sq_rest_untouched"] = bool(torch.equal(untouched, torch.cat([sq0[:a], sq0[b:]])))
    reps = gather_host(par[:n], world)
    res["replicas_identical"] = all(bool(torch.equal(reps[0], q)) for q in reps[1:])
    ok &= res["fused_params_equal_flat_pass"] and res["fused_sq_own_shard"] and res["fused_sq_rest_untouched"] \
        and res["replicas_identical"]
    # ---- captured in a HIP graph and replayed: the collective number lives on the device, so replays keep counting
    n = sizes[3]
    x = torch.randn(n, generator=g)
    stage = x.to(dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s, **parallel.graph_capture_kwargs()):
            grad[:n].copy_(stage)
            L.mesh_allreduce_f32(mesh.mesh, mesh.offset_of(grad), n, torch.cuda.current_stream().cuda_stream)
    parts = gather_host(stage, world)
    want = parts[0].clone()
    for q in parts[1:]:
        want += q
    good = True
    for _ in range(3):
        gr.replay()
        torch.cuda.synchronize()
        good &= bool(torch.equal(grad[:n].cpu(), want))
    res["graph_replays"] = good
    ok &= good
    done, timeouts = mesh.status()
    res["exchanges"], res["timeouts"] = done, timeouts
    ok &= timeouts == 0 and done == len(sizes) + n_small + 1 + 3 and ok_share
    # ---- a peer that stays away longer than the bound: counted, never a hang; the exchange numbering survives it
    if world == 2:
        import time
        L.mesh_set_timeout_ms(mesh.mesh, 300)
        dist.barrier()
        v = mesh.alloc_small(64)
        v.fill_(float(rank + 1))
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 1:
            time.sleep(3.0)                           # rank 0 is inside the kernel, waiting for START[1], for 0.3 s only
        t0 = time.time()
        L.mesh_allreduce_small(mesh.mesh, mesh.offset_of(v), 64, st)
        torch.cuda.synchronize()
        took = time.time() - t0
        _, t_after = mesh.status()
        res["late_peer"] = dict(took=round(took, 2), timeouts=t_after)
        ok &= took < 5.0 and ((t_after > 0) if rank == 0 else True)
        try:
            mesh.check()
            raised = False
        except RuntimeError:
            raised = True
        ok &= raised == (t_after > 0)
        dist.barrier()
        L.mesh_set_timeout_ms(mesh.mesh, 20000)
        v.fill_(float(rank + 1))
        torch.cuda.synchronize()
        dist.barrier()
        L.mesh_allreduce_small(mesh.mesh, mesh.offset_of(v), 64, st)
        torch.cuda.synchronize()
        _, t_again = mesh.status()
        res["after_late_peer"] = dict(value=float(v[0].item()), new_timeouts=t_again - t_after)
        ok &= t_again == t_after and float(v[0].item()) == 3.0 and float(v[63].item()) == 3.0
    res["ok"] = bool(ok)
    del gr, grad, par
    parallel.shutdown()
    with open(out_path + f".{rank}", "w") as f:
        json.dump(res, f)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
