#!/usr/bin/env python3
"""Isolated timing of the fused weight-gradient + RMSprop launch (wfd_gemm_f32_rmsprop_multi) at the step's shapes:
16 back-to-back launches in a HIP graph between events (hot operands) -- and the same with a 512 MB flush in between
single launches (cold).  usage: rms_multi_bench.py [d|g]   (switches: WFD_RMS_PERSIST, WFD_RMS_SLOTS)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wfd_gan_amd._lib import lib
L = lib()
which = sys.argv[1] if len(sys.argv) > 1 else "d"
dev = "cuda:0"
S, C, Z, B = 1401, 100, 500, 64
if which == "d":
    K = 4 * B
    dims = [(2048, S + C, 1504), (1024, 2048, 2048), (512, 1024, 1024)]
else:
    K = B
    dims = [(S, 2048, 2048), (2048, 1024, 1024), (1024, 512, 512), (512, Z + C, 600)]
desc, keep = [], []
for M, N, ldp in dims:
    lda, ldb = (M + 3) // 4 * 4, (N + 3) // 4 * 4
    A, Bm = torch.randn(K, lda, device=dev), torch.randn(K, ldb, device=dev)
    P, SQ = torch.randn(M, ldp, device=dev), torch.rand(M, ldp, device=dev)
    keep += [A, Bm, P, SQ]
    desc += [A.data_ptr(), lda, Bm.data_ptr(), ldb, M, N, K, P.data_ptr(), ldp, SQ.data_ptr()]
arr = (ctypes.c_longlong * len(desc))(*desc)
flops = sum(2.0 * M * N * K for M, N, _ in dims)
st = torch.cuda.current_stream().cuda_stream
L.gemm_f32_rmsprop_multi(arr, len(dims), 5e-5, 0.99, 1e-8, st)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    s2 = torch.cuda.current_stream().cuda_stream
    for _ in range(16):
        L.gemm_f32_rmsprop_multi(arr, len(dims), 5e-5, 0.99, 1e-8, s2)
g.replay()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) / 16 * 1e3)
hot = sorted(ts)[len(ts) // 2]
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
cs = []
for rep in range(7):
    flush.fill_(rep)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); L.gemm_f32_rmsprop_multi(arr, len(dims), 5e-5, 0.99, 1e-8, st); e1.record(); e1.synchronize()
    cs.append(e0.elapsed_time(e1) * 1e3)
cold = sorted(cs)[len(cs) // 2]
print(f"{which} persist={os.environ.get('WFD_RMS_PERSIST','1')} slots={os.environ.get('WFD_RMS_SLOTS','96')}: hot {hot:.2f} us "
      f"({flops / hot / 1e6:.1f} TF), cold single launch {cold:.2f} us")
