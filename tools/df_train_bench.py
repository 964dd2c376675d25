#!/usr/bin/env python3
"""DF observer training throughput at the reference's configuration (src/train_df.py:19-21: batch 128, L = 1400, 100
classes): steps/s of wfd_gan_amd.df_engine.DFTrainEngine on the GPU (eager and HIP-graph replay) and of the CPU oracle
on the host cores.  One JSON line."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--length", type=int, default=1400)
    ap.add_argument("--classes", type=int, default=100)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()
    from oracle import df_train as DT
    from oracle import wfdgan_oracle as O
    from wfd_gan_amd.df_engine import DFTrainEngine
    B, L, C = args.batch, args.length, args.classes
    torch.manual_seed(0)
    sd = O.make_df_state(L, C)
    eng = DFTrainEngine(L, C, B, sd)
    g = torch.Generator().manual_seed(1)
    x, y = torch.rand(B, L, generator=g), torch.randint(0, C, (B,), generator=g)
    eng.set_batch(x, y)

    def step():
        eng.draw_masks()
        eng.plan.run()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    eager = args.steps / (time.perf_counter() - t0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        graph.replay()
    torch.cuda.synchronize()
    graphed = args.steps / (time.perf_counter() - t0)
    loss = eng.loss()
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    tr = DT.DFTrainer(sd)
    tr.step(x[:, None, :], y)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < args.cpu_seconds:
        tr.step(x[:, None, :], y)
        n += 1
    cpu = n / (time.perf_counter() - t0)
    nkern = sum(1 for c in eng.plan.calls)
    print(json.dumps(dict(metric="DF observer training steps/s (batch 128, L=1400, 100 classes, fp32)", eager=eager,
                          graph=graphed, samples_per_s=graphed * B, launches_per_step=nkern, loss_after=loss,
                          cpu_oracle=dict(value=cpu, cores=threads, steps=n))))


if __name__ == "__main__":
    main()
