#!/usr/bin/env python3
"""Run one wfd_gemm_f32_wk shape (for rocprofv3 --pmc passes).  usage: wk_one.py layout M N K lda ldb cfg reps [old]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wfd_gan_amd._lib import lib
L = lib()
lay, M, N, K, lda, ldb, cfg, reps = [int(x) for x in sys.argv[1:9]]
old = len(sys.argv) > 9
dev = "cuda:0"
rb = N if lay == 0 else K
A = torch.randn(M, lda, device=dev); B = torch.randn(rb, ldb, device=dev)
ldp = (N + 3) // 4 * 4
out = torch.zeros(M, ldp, device=dev)
ns = L.gemm_pick_splits(M, N, K)
ws = torch.zeros(ns * M * ldp, device=dev)
junk = torch.zeros(64 << 20, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    junk.add_(1.0)          # 512 MB of traffic between repetitions: caches cold for the operands
    if old:
        L.gemm_f32(lay, A.data_ptr(), lda, B.data_ptr(), ldb, ws.data_ptr(), ldp, M * ldp, M, N, K, ns, st)
    else:
        L.gemm_f32_wk(lay, A.data_ptr(), lda, B.data_ptr(), ldb, M, N, K, 0, out.data_ptr(), ldp, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0, cfg, st)
torch.cuda.synchronize()
