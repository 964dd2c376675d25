#!/usr/bin/env python3
"""Run one GEMM shape a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py layout M N K lda ldb ns reps [mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wfd_gan_amd._lib import lib
L = lib()
lay, M, N, K, lda, ldb, ns, reps = [int(x) for x in sys.argv[1:9]]
dev = "cuda:0"
ra = M if lay != 2 else K
rb = N if lay == 0 else K
A = torch.randn(ra, lda, device=dev); B = torch.randn(rb, ldb, device=dev)
ldp = (N + 3) // 4 * 4
ws = torch.empty(ns * M * ldp, device=dev)
st = torch.cuda.current_stream().cuda_stream
L.set_gemm_mode(int(sys.argv[9]) if len(sys.argv) > 9 else 0)
for _ in range(reps):
    L.gemm_f32(lay, A.data_ptr(), lda, B.data_ptr(), ldb, ws.data_ptr(), ldp, M * ldp, M, N, K, ns, st)
torch.cuda.synchronize()
