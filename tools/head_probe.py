"""Time of the D-step head (gather, z / mask draws, generator forward) as one graph: batch 64 three times vs batch 192
once -- the premise of batching the heads of the iterations that share generator parameters."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"]
import bench  # noqa: E402

dev = torch.device("cuda:0")
for B in (64, 128, 192):
    args = types.SimpleNamespace(seq_len=1401, classes=100, latent_dim=500, batch=B, inst=100, penalty="div", overlap=0,
                                 dp="weak")
    eng, _ = bench.build_engine(args, 0, 1, dev)
    for part in ("gf", "rest"):
        plan = eng._build_d_plan(fast=True, part=part)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            plan.run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                plan.run()
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        print(f"B={B} {part}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us, {len(plan.calls)} launches")
