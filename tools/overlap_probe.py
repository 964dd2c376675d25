#!/usr/bin/env python3
"""Can the weight-gradient + RMSprop launch hide in the shadow of the chain's products?  Times, on a scratch engine of the
config-2 step (results discarded), (a) the critic's end-of-step launch alone, (b) one product of the t-chain alone, (c) both
back to back on one stream, (d) both at once on two streams -- each as a graph of `reps` pairs.  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
import bench

args = argparse.Namespace(seq_len=1401, classes=100, latent_dim=500, batch=64, inst=200, penalty="div", overlap=0, dp="weak",
                          operands="f32", fuse_opt="1", wk=1, force_dp=False)
dev = torch.device("cuda:0")
eng, _ = bench.build_engine(args, 0, 1, dev)
L = eng.L
calls = eng.fast_d.calls
rms = [c for c in calls if c[0] is L.gemm_f32_rmsprop_multi_colsum][0]
wks = [c for c in calls if c[0] is L.gemm_f32_wk]
reps = 50


def graph_time(build):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        build(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                build(torch.cuda.current_stream())
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        for _ in range(4):
            g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / (4 * reps))
    return best * 1e6


side = torch.cuda.Stream()
for wi, wk in enumerate(wks[-3:]):
    M, N, K = wk[1][5], wk[1][6], wk[1][7]
    a = graph_time(lambda s: rms[0](*rms[1], s.cuda_stream))
    b = graph_time(lambda s: wk[0](*wk[1], s.cuda_stream))
    c = graph_time(lambda s: (wk[0](*wk[1], s.cuda_stream), rms[0](*rms[1], s.cuda_stream)))

    def both(s):
        side.wait_stream(s)
        wk[0](*wk[1], s.cuda_stream)
        rms[0](*rms[1], side.cuda_stream)
        s.wait_stream(side)
    d = graph_time(both)
    print(f"chain product {M}x{N}x{K}: rms alone {a:.1f} us, product alone {b:.1f} us, back to back {c:.1f} us, "
          f"two branches {d:.1f} us", flush=True)
