#!/usr/bin/env python3
"""wfd_gemm_f32_pair against the two separate launches on the critic's backward pairs (GPU only): results must be
bit-identical (same tile code), time per pair from graph replays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wfd_gan_amd._lib import lib
L = lib()
dev = "cuda:0"
ws = torch.empty(1 << 26, device=dev)
pairs = [  # TN (M1,N1,K1,lda1,ldb1) ; other (layout, M2,N2,K2,lda2,ldb2)
    ("dW1 | t1", (2048, 1501, 256, 2048, 1504), (0, 128, 2048, 1401, 1504, 1504)),
    ("dW2 | t2", (1024, 2048, 256, 1024, 2048), (0, 128, 1024, 2048, 2048, 2048)),
    ("dW3 | t3", (512, 1024, 256, 512, 1024), (0, 128, 512, 1024, 1024, 1024)),
    ("gW8 | dh3", (1401, 2048, 64, 1404, 2048), (1, 64, 2048, 1401, 1404, 2048)),
    ("gW5 | dh2", (2048, 1024, 64, 2048, 1024), (1, 64, 1024, 2048, 2048, 1024)),
]
def timeit(fn, rep=16):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn(st.cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(rep):
            fn(s)
    g.replay()
    tot = 0.0
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        tot += e0.elapsed_time(e1) / rep
    return tot / 3 * 1e3
for name, (M1, N1, K1, lda1, ldb1), (lay, M2, N2, K2, lda2, ldb2) in pairs:
    A1, B1 = torch.randn(K1, lda1, device=dev), torch.randn(K1, ldb1, device=dev)
    A2 = torch.randn(M2, lda2, device=dev)
    B2 = torch.randn(N2 if lay == 0 else K2, ldb2, device=dev)
    ldp1, ldp2 = (N1 + 3) // 4 * 4, (N2 + 3) // 4 * 4
    ns = L.gemm_pick_splits(M2, N2, K2)
    Pa, Pb = torch.zeros(M1, ldp1, device=dev), torch.zeros(M1, ldp1, device=dev)
    Sa, Sb = torch.zeros(ns, M2, ldp2, device=dev), torch.zeros(ns, M2, ldp2, device=dev)
    def sep(s):
        L.gemm_f32(2, A1.data_ptr(), lda1, B1.data_ptr(), ldb1, Pa.data_ptr(), ldp1, 0, M1, N1, K1, 1, s)
        L.gemm_f32(lay, A2.data_ptr(), lda2, B2.data_ptr(), ldb2, Sa.data_ptr(), ldp2, M2 * ldp2, M2, N2, K2, ns, s)
    def pair(s):
        L.gemm_f32_pair(A1.data_ptr(), lda1, B1.data_ptr(), ldb1, Pb.data_ptr(), ldp1, M1, N1, K1, lay, A2.data_ptr(), lda2,
                        B2.data_ptr(), ldb2, Sb.data_ptr(), ldp2, M2 * ldp2, M2, N2, K2, ns, s)
    ts, tp = timeit(sep), timeit(pair)
    same = torch.equal(Pa, Pb) and torch.equal(Sa, Sb)
    print(f"{name:10s} ns={ns:2d}  separate {ts:6.1f} us   pair {tp:6.1f} us   identical={same}", flush=True)
