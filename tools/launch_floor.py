#!/usr/bin/env python3
"""Per-launch cost inside a HIP graph on this box: a trivial kernel, and the flat split-K epilogue at the shapes of
the B=64 step (how much of its time is the launch floor and how much the slab traffic)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfd_gan_amd._lib import lib  # noqa: E402


def timed_graph(fn, n=200, reps=30):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * n) * 1e6


def main():
    L = lib()
    dev = "cuda:0"
    st = lambda: torch.cuda.current_stream().cuda_stream
    state = torch.zeros(2, dtype=torch.int64, device=dev)
    out = {"trivial_us": timed_graph(lambda: L.iter_advance(state.data_ptr(), 1, 1, st()))}
    for M, N in ((128, 2048), (128, 512), (64, 1024)):
        for ns in (1, 2, 4, 8, 16):
            P = torch.randn(ns, M, N, device=dev)
            D = torch.randn(M, N, device=dev)
            o = torch.zeros(M, N, device=dev)
            out[f"ep_mul_{M}x{N}_ns{ns}_us"] = timed_graph(
                lambda: L.ep_mul(P.data_ptr(), ns, M * N, N, M, N, D.data_ptr(), N, 1, 0, 0, o.data_ptr(), 0, N, st()))
    # a GEMM of the step with and without its epilogue
    for (M, N, K) in ((128, 2048, 1504), (128, 1024, 2048), (128, 512, 1024), (64, 1024, 512)):
        A = torch.randn(M, K, device=dev)
        B = torch.randn(N, K, device=dev)
        for ns in (1, 2, 4, 8, 16):
            if (K + 31) // 32 // ns < 2:
                continue
            P = torch.zeros(ns, M, N, device=dev)
            out[f"gemm_nt_{M}x{N}x{K}_ns{ns}_us"] = timed_graph(
                lambda: L.gemm_f32(0, A.data_ptr(), K, B.data_ptr(), K, P.data_ptr(), N, M * N, M, N, K, ns, st()))
        out[f"gemm_nt_{M}x{N}x{K}_picked"] = L.gemm_pick_splits(M, N, K)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
