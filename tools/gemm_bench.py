#!/usr/bin/env python3
"""Per-shape timing of the GEMM core on the shapes one training iteration launches (GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wfd_gan_amd._lib import lib

L = lib()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
S, C = 1401, 100
shapes = [  # (name, layout, M, N, K, lda, ldb)
    ("D fwd1 NT", 0, 128, 2048, 1501, 1504, 1504), ("D fwd2 NT", 0, 128, 1024, 2048, 2048, 2048),
    ("D fwd3 NT", 0, 128, 512, 1024, 1024, 1024), ("chain3 NN", 1, 128, 1024, 512, 512, 1024),
    ("chain2 NN", 1, 128, 2048, 1024, 1024, 2048), ("chain1 NN", 1, 128, 1401, 2048, 2048, 1504),
    ("t1 NT", 0, 128, 2048, 1401, 1504, 1504), ("dW1 TN", 2, 2048, 1501, 256, 2048, 1504),
    ("dW2 TN", 2, 1024, 2048, 256, 1024, 2048), ("dW3 TN", 2, 512, 1024, 256, 512, 1024),
    ("G fwd1 NT", 0, 64, 512, 600, 600, 600), ("G fwd2 NT", 0, 64, 1024, 512, 512, 512),
    ("G fwd3 NT", 0, 64, 2048, 1024, 1024, 1024), ("G fwd4 NT", 0, 64, 1401, 2048, 2048, 2048),
    ("dW8 TN", 2, 1401, 2048, 64, 1404, 2048), ("dh3 NN", 1, 64, 2048, 1401, 1404, 2048),
    ("sample L3 NT", 0, 8192, 2048, 1024, 1024, 1024), ("sample L4 NT", 0, 8192, 1401, 2048, 2048, 2048),
    ("big chain2 NN", 1, 8192, 2048, 1024, 1024, 2048), ("big dW2 TN", 2, 1024, 2048, 8192, 1024, 2048),
    ("big dW1 TN", 2, 2048, 1501, 8192, 2048, 1504),
]
ws = torch.empty(1 << 28, device=dev)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
L.set_gemm_mode(mode)
print("gemm mode", mode, flush=True)
for name, lay, M, N, K, lda, ldb in shapes:
    ra = M if lay != 2 else K
    rb = N if lay == 0 else K
    A = torch.randn(ra, lda, device=dev)
    B = torch.randn(rb, ldb, device=dev)
    ldp = (N + 3) // 4 * 4
    res = []
    for ns in sorted({1, 2, 4, 8, 16, L.gemm_pick_splits_mode(M, N, K, mode)}):
        if ns * M * ldp > ws.numel() or (K + 31) // 32 < ns:
            continue
        for _ in range(3):
            L.gemm_f32(lay, A.data_ptr(), lda, B.data_ptr(), ldb, ws.data_ptr(), ldp, M * ldp, M, N, K, ns, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.gemm_f32(lay, A.data_ptr(), lda, B.data_ptr(), ldb, ws.data_ptr(), ldp, M * ldp, M, N, K, ns, st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        res.append(f"ns={ns}: {us:6.1f}us {2.0*M*N*K/us/1e6:5.1f}TF")
    print(f"{name:14s} M={M:5d} N={N:5d} K={K:5d} pick={L.gemm_pick_splits_mode(M,N,K,mode):2d} | " + " | ".join(res), flush=True)
