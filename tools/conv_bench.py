#!/usr/bin/env python3
"""Time the LDS-resident k=8 conv (Cin = 32) through the C-ABI at the DF observer's shapes.  GPU only.
    python tools/conv_bench.py [mode]          mode: 0 f32 (default), 1 bf16, 2 bf16x3"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wfd_gan_amd._lib import lib

L = lib()
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
L.set_gemm_mode(mode)
dev = "cuda:0"
peak = {0: 157.3, 1: 2516.6, 2: 2516.6 / 3}[mode]
for Bn, Ln, Cout in [(64, 1400, 32), (64, 175, 64), (256, 1400, 32), (1024, 1400, 32), (4096, 1400, 32), (4096, 175, 64)]:
    Cin = 32
    X = torch.randn(Bn * Ln, Cin, device=dev)
    W = torch.randn(Cout, 8 * Cin, device=dev)
    sc, sh = torch.randn(Cout, device=dev), torch.randn(Cout, device=dev)
    out = torch.empty(Bn * Ln, Cout, device=dev)
    call = lambda st: L.conv8_gemm_f32_fused(X.data_ptr(), Bn, Ln, Cin, 3, W.data_ptr(), Cout, out.data_ptr(), Cout, 1,
                                             sc.data_ptr(), sh.data_ptr(), 2, 0, 0, 0, st)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        call(s.cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    rep = 20
    with torch.cuda.graph(g):
        for _ in range(rep):
            call(torch.cuda.current_stream().cuda_stream)
    g.replay()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / rep * 1e3)
    fl = 2.0 * Bn * Ln * Cout * 8 * Cin
    print(f"mode {mode} B={Bn:5d} L={Ln:5d} 32->{Cout}: {best:8.1f} us  {fl / best / 1e6:7.1f} TF  ({fl / best / 1e6 / peak:.2f} of peak)",
          flush=True)
L.set_gemm_mode(0)
