#!/usr/bin/env python3
"""Where the fused weight-gradient + RMSprop launch (wfd_gemm_f32_rmsprop_multi) spends its time: per-workgroup
s_memrealtime stamps (100 MHz) from the diagnostic build (make -C wfd_gan_amd/csrc stamps).
usage: rms_multi_stamps.py [d|g]     d = the critic's three layers (K = 256), g = the generator's four (K = 64)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1] if len(sys.argv) > 1 else "d"
lib = ctypes.CDLL(os.path.join(here, "wfd_gan_amd", "libwfdgan_stamps.so"))
dev = "cuda:0"
S, C, Z, B = 1400, 100, 16, 64
if which == "d":
    K = 4 * B
    dims = [(2048, S + C, 1504), (1024, 2048, 2048), (512, 1024, 1024)]       # (M, N, ldp)
else:
    K = B
    dims = [(S, 2048, 2048), (2048, 1024, 1024), (1024, 512, 512), (512, Z + C, 116)]
NS = 65536
st = torch.zeros(len(dims) * NS, 4, dtype=torch.int64, device=dev)
desc, keep = [], []
for M, N, ldp in dims:
    lda, ldb = (M + 3) // 4 * 4, (N + 3) // 4 * 4
    A, Bm = torch.randn(K, lda, device=dev), torch.randn(K, ldb, device=dev)
    P, SQ = torch.randn(M, ldp, device=dev), torch.rand(M, ldp, device=dev)
    keep += [A, Bm, P, SQ]
    desc += [A.data_ptr(), lda, Bm.data_ptr(), ldb, M, N, K, P.data_ptr(), ldp, SQ.data_ptr()]
arr = (ctypes.c_longlong * len(desc))(*desc)
lib.wfd_rms_multi_set_stamps(ctypes.c_void_p(st.data_ptr()), NS)
fn = lib.wfd_gemm_f32_rmsprop_multi
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
s = torch.cuda.current_stream().cuda_stream
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for rep in range(5):
    flush.fill_(rep)                      # the step's weights come from HBM, not from the last launch's cache lines
    st.zero_()
    torch.cuda.synchronize()
    rc = fn(arr, len(dims), 5e-5, 0.99, 1e-8, s)
    assert rc == 0, rc
    torch.cuda.synchronize()
if os.environ.get("WFD_RMS_PERSIST", "1") != "0":
    # persistent form: 16 slots per workgroup: 0 entry, 1 first tile in LDS, 2 + 2t k-loop of tile t done, 3 + 2t its stores
    # issued, 15 all stores acknowledged
    t = st.cpu().numpy().reshape(-1, 16)[:8 * 256].astype(np.float64) * 0.01
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    q = lambda x: "n %4d  min %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % (len(x), x.min(), np.percentile(x, 10), np.percentile(x, 50), np.percentile(x, 90), x.max())
    print(f"{which}: K={K}, {len(t)} persistent workgroups")
    print("entry after first entry ", q(t[:, 0] - t0))
    print("prologue                ", q(t[:, 1] - t[:, 0]))
    prev = t[:, 1]
    for tn in range(6):
        has = t[:, 2 + 2 * tn] != 0
        if not has.any():
            break
        print(f"tile {tn} k-loop           ", q((t[:, 2 + 2 * tn] - (prev if tn == 0 else t[:, 1 + 2 * tn]))[has]))
        print(f"tile {tn} rmsprop + stores ", q((t[:, 3 + 2 * tn] - t[:, 2 + 2 * tn])[has]))
    print("store drain at the end  ", q(t[:, 15] - np.max(t[:, 2:15], axis=1)))
    print("exit after first entry  ", q(t[:, 15] - t0))
    sys.exit(0)
t = st.cpu().numpy()
t = t[t[:, 0] != 0].astype(np.float64) * 0.01          # us
t0 = t[:, 0].min()
q = lambda x: "min %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % (x.min(), np.percentile(x, 10), np.percentile(x, 50), np.percentile(x, 90), x.max())
print(f"{which}: K={K}, {len(t)} workgroups (weights flushed from the caches before the launch)")
print("entry after first entry  ", q(t[:, 0] - t0))
print("prologue (first tile)    ", q(t[:, 1] - t[:, 0]))
print("k-loop                   ", q(t[:, 2] - t[:, 1]))
print("stores                   ", q(t[:, 3] - t[:, 2]))
print("exit after first entry   ", q(t[:, 3] - t0))
end = (t[:, 3] - t0).max()
print("time   entering  prologue  k-loop  storing  done")
for x in np.arange(0, end + 1, 1.0):
    rel = t - t0
    print("%5.1f  %7d  %8d  %6d  %7d  %5d" % (x, (rel[:, 0] > x).sum(), ((rel[:, 0] <= x) & (rel[:, 1] > x)).sum(),
                                             ((rel[:, 1] <= x) & (rel[:, 2] > x)).sum(), ((rel[:, 2] <= x) & (rel[:, 3] > x)).sum(),
                                             (rel[:, 3] <= x).sum()))
