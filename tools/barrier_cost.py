#!/usr/bin/env python3
"""What a grid-wide barrier with hand-over of data costs on this part, measured with the library's own exchange kernel at
world 1: wfd_mesh_allreduce_small on n floats = launch + start barrier (system-scope flag store / poll + one acquire) + n
loads + device-scope ticket + end barrier (flag store / poll) + n stores.  Compared with a trivial launch of the same graph
length (wfd_mul_mask on 4 floats).  Per-launch times from a captured graph of `reps` launches.  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wfd_gan_amd import parallel
from wfd_gan_amd._lib import lib

L = lib()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
mesh = parallel.MeshSync(L, 1 << 20, dev)
x = torch.ones(1 << 20, device=dev)
reps = 200


def timed(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn(s.cuda_stream)
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / (10 * reps))
    return best * 1e6


print(f"trivial launch (wfd_mul_mask, 4 floats): {timed(lambda st: L.mul_mask(x.data_ptr(), x.data_ptr(), 4, x.data_ptr(), st)):.2f} us per launch")
for n, blocks in ((2, 1), (1024, 1), (8192, 8), (65536, 64), (131072, 128)):
    v = mesh.alloc_small(n)
    L.mesh_set_max_blocks(mesh.mesh, 128)
    t = timed(lambda st: L.mesh_allreduce_small(mesh.mesh, mesh.offset_of(v), n, st))
    print(f"exchange kernel, {n:6d} floats, {blocks:3d} workgroups of 1024 threads: {t:.2f} us per launch")
# the fused optimizer form at world 1 on the critic's 5.7 M parameters, against the flat pass it replaces
n = 5_699_840
big = parallel.MeshSync(L, 2 * n, dev)
g_, p_ = big.view(0, n), big.view(n, n)
sq = torch.rand(n, device=dev)
for blocks in (64, 128, 256):
    L.mesh_set_max_blocks(big.mesh, blocks)
    t = timed(lambda st: L.mesh_reduce_rmsprop(big.mesh, 0, n, sq.data_ptr(), n, 1e-4, 0.99, 1e-8, 1.0, 0, 0, 0, st))
    print(f"mesh_reduce_rmsprop, {n} parameters, world 1, {blocks:3d} workgroups: {t:.1f} us per launch")
t = timed(lambda st: L.rmsprop(p_.data_ptr(), g_.data_ptr(), sq.data_ptr(), n, 1e-4, 0.99, 1e-8, 1.0, 0, 0, 0, st))
print(f"wfd_rmsprop (flat pass), {n} parameters: {t:.1f} us per launch")
done, timeouts = mesh.status()
print("exchanges", done, "timeouts", timeouts)
