import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wfd_gan_amd._lib import lib
L = lib(); dev = "cuda:0"
REPS = 20
def timed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(REPS): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
    return best
for (lay, M, N, K, lda, ldb) in [(0, 128, 2048, 1501, 1504, 1504), (0, 128, 1024, 2048, 2048, 2048), (0, 64, 512, 512, 512, 512)]:
    A = torch.randn(M, lda, device=dev); B = torch.randn(N if lay == 0 else K, ldb, device=dev)
    ldp = (N + 3) // 4 * 4
    H = torch.zeros(M, ldp, device=dev)
    for cfg in (1, 2, 3, 4):
        row = []
        for dbg in (0, 1, 2, 4, 3, 7):
            def f():
                st = torch.cuda.current_stream().cuda_stream
                L.gemm_f32_wk(lay, A.data_ptr(), lda, B.data_ptr(), ldb, M, N, K, 0, H.data_ptr(), ldp, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0, cfg + 16 * dbg, st)
            row.append(f"dbg{dbg}: {timed(f):5.1f}")
        print(f"M={M} N={N} K={K} cfg{cfg} | " + "  ".join(row) + "   (1 = no loads, 2 = no mfma, 4 = no lds writes)", flush=True)
