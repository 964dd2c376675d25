#!/usr/bin/env python3
"""bf16-operand MFMA GEMM timing on the generator's large-batch shapes (GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wfd_gan_amd._lib import lib
L = lib(); dev = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
for M in (1024, 8192, 32768):
    for N, K in ((512, 600), (1024, 512), (2048, 1024), (1401, 2048), (4096, 4096)):
        A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
        O = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        sc = torch.ones(N, device=dev); sh = torch.zeros(N, device=dev)
        f = lambda: L.gemm_bf16_nt(A.data_ptr(), K, B.data_ptr(), K, M, N, K, 0, 0, O.data_ptr(), N, sc.data_ptr(), sh.data_ptr(), 1, st)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
