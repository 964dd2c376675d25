#!/usr/bin/env python3
"""Split-inside-the-workgroup contraction (wfd_gemm_f32_wk) against the slab form + consumer launch it replaces, on the
batch-64 shapes of one training iteration.  Times are per dependent pair inside a captured HIP graph of 20 repetitions
(what the step pays), not back-to-back throughput of one kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wfd_gan_amd._lib import lib

L = lib()
dev = "cuda:0"
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 0          # operand mode: 0 f32, 1 bf16, 2 bf16x3 (wfd_set_gemm_mode)
L.set_gemm_mode(MODE)
shapes = [  # (name, layout, M, N, K, lda, ldb)
    ("D fwd1 NT", 0, 128, 2048, 1501, 1504, 1504), ("D fwd2 NT", 0, 128, 1024, 2048, 2048, 2048),
    ("D fwd3 NT", 0, 128, 512, 1024, 1024, 1024), ("chain3 NN", 1, 128, 1024, 512, 512, 1024),
    ("chain2 NN", 1, 128, 2048, 1024, 1024, 2048), ("chain1 NN", 1, 128, 1401, 2048, 2048, 1504),
    ("t1 NT", 0, 128, 2048, 1401, 1504, 1504), ("t2 NT", 0, 128, 1024, 2048, 2048, 2048), ("t3 NT", 0, 128, 512, 1024, 1024, 1024),
    ("Gc fwd1 NT", 0, 64, 2048, 1501, 1504, 1504), ("Gc fwd2 NT", 0, 64, 1024, 2048, 2048, 2048),
    ("Gc chain2 NN", 1, 64, 2048, 1024, 1024, 2048), ("Gc chain3 NN", 1, 64, 1024, 512, 512, 1024),
    ("head1 NT", 0, 192, 512, 600, 600, 600), ("head2 NT", 0, 192, 1024, 512, 512, 512),
    ("head3 NT", 0, 192, 2048, 1024, 1024, 1024), ("head4 NT", 0, 192, 1401, 2048, 2048, 2048),
    ("DF fc1 NT", 0, 64, 512, 256, 256, 256), ("DF fc2 NT", 0, 64, 512, 512, 512, 512), ("DF fc NT", 0, 64, 100, 512, 512, 512),
    ("DF dfc NN", 1, 64, 512, 100, 100, 512), ("DF dfc2 NN", 1, 64, 512, 512, 512, 512), ("DF dfc1 NN", 1, 64, 256, 512, 512, 256),
    ("G dh3 NN", 1, 64, 2048, 1401, 1404, 2048), ("G dh2 NN", 1, 64, 1024, 2048, 2048, 1024), ("G dh1 NN", 1, 64, 512, 1024, 1024, 512),
]
ws = torch.empty(1 << 26, device=dev)
REPS = 20


def timed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(REPS):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
    return best


for name, lay, M, N, K, lda, ldb in shapes:
    rb = N if lay == 0 else K
    A = torch.randn(M, lda, device=dev)
    B = torch.randn(rb, ldb, device=dev)
    ldp = (N + 3) // 4 * 4
    bias = torch.randn(N, device=dev)
    mask = torch.ones(M, N, device=dev)
    H, D = torch.zeros(M, ldp, device=dev), torch.zeros(M, ldp, device=dev)
    ns = L.gemm_pick_splits_mode(M, N, K, MODE)

    def old():
        st = torch.cuda.current_stream().cuda_stream
        L.gemm_f32(lay, A.data_ptr(), lda, B.data_ptr(), ldb, ws.data_ptr(), ldp, M * ldp, M, N, K, ns, st)
        L.ep_lrelu_drop(ws.data_ptr(), ns, M * ldp, ldp, M, N, bias.data_ptr(), mask.data_ptr(), N, 0.2, H.data_ptr(), ldp,
                        D.data_ptr(), ldp, st)

    def new(cfg):
        def f():
            st = torch.cuda.current_stream().cuda_stream
            L.gemm_f32_wk(lay, A.data_ptr(), lda, B.data_ptr(), ldb, M, N, K, 3, H.data_ptr(), ldp, D.data_ptr(), ldp, 0,
                          bias.data_ptr(), 0, 0, 0, 0, 0, mask.data_ptr(), N, 0.2, cfg, st)
        return f
    t_old = timed(old)
    res = [f"cfg{c}: {timed(new(c)):5.1f}" for c in (1, 2, 3, 4, 5)]
    pick = L.gemm_wk_pick(lay, M, N, K)
    print(f"{name:13s} M={M:4d} N={N:5d} K={K:5d} | slabs(ns={ns:2d})+consumer {t_old:5.1f} us | " + "  ".join(res) +
          f" | pick {pick} | mfma floor {2.0 * M * N * K / 137e6:4.1f} us", flush=True)
