"""Scheduling probes on one GPU: per-iteration time of (a) the sequential engine with one graph per step, (b) one graph
per group of n_critic iterations, (c) the pipelined driver; host enqueue time beside the total."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"]
import bench  # noqa: E402

args = types.SimpleNamespace(seq_len=1401, classes=100, latent_dim=500, batch=64, inst=1000, penalty="div", overlap=0,
                             dp="weak")
dev = torch.device("cuda:0")
eng, _ = bench.build_engine(args, 0, 1, dev)
eng.capture_graphs()
for i in range(30):
    eng.iteration(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(30, 330):
    eng.iteration(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("per-step graphs: enqueue ms/iter %.4f total ms/iter %.4f" % ((t1 - t0) / 300 * 1e3, (t2 - t0) / 300 * 1e3))

g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    eng.fast_d.run(); eng.fast_g.run(); eng.fast_d.run(); eng.fast_d.run()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("group graph:     enqueue ms/iter %.4f total ms/iter %.4f" % ((t1 - t0) / 300 * 1e3, (t2 - t0) / 300 * 1e3))

g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    for _ in range(4):
        eng.fast_d.run(); eng.fast_g.run(); eng.fast_d.run(); eng.fast_d.run()
for _ in range(5):
    g2.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(25):
    g2.replay()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("12-iteration graph: total ms/iter %.4f" % ((t2 - t0) / 300 * 1e3))
