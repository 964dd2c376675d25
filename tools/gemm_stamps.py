#!/usr/bin/env python3
"""Where one GEMM launch spends its time: per-workgroup s_memrealtime stamps (100 MHz) from the diagnostic build
(make -C wfd_gan_amd/csrc stamps -> libwfdgan_stamps.so).  usage: gemm_stamps.py layout M N K lda ldb ns
Prints, relative to the first workgroup's entry: distribution of entry times, prologue (entry -> first tile in LDS), k-loop,
store phase, and the exit time of the last workgroup."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(here, "wfd_gan_amd", "libwfdgan_stamps.so"))
fn = lib.wfd_gemm_f32_stamped
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
               ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
lay, M, N, K, lda, ldb, ns = [int(x) for x in sys.argv[1:8]]
dev = "cuda:0"
ra, rb = (M if lay != 2 else K), (N if lay == 0 else K)
A, B = torch.randn(ra, lda, device=dev), torch.randn(rb, ldb, device=dev)
ldp = (N + 3) // 4 * 4
P = torch.empty(ns * M * ldp, device=dev)
gx, gy = (N + 63) // 64, (M + 63) // 64
assert gx <= 256 and gy <= 256
nslots = 65536 * ns
st = torch.zeros(nslots, 4, dtype=torch.int64, device=dev)
s = torch.cuda.current_stream().cuda_stream
for rep in range(5):
    st.zero_()
    rc = fn(lay, A.data_ptr(), lda, B.data_ptr(), ldb, P.data_ptr(), ldp, M * ldp, M, N, K, ns, st.data_ptr(), nslots, s)
    assert rc == 0
    torch.cuda.synchronize()
t = st.cpu().numpy()
t = t[t[:, 0] != 0].astype(np.float64) * 0.01          # us
t0 = t[:, 0].min()
q = lambda x: "min %.2f  p50 %.2f  p90 %.2f  max %.2f" % (x.min(), np.percentile(x, 50), np.percentile(x, 90), x.max())
print(f"layout {lay} M={M} N={N} K={K} ns={ns}: {len(t)} workgroups")
print("entry after first entry  ", q(t[:, 0] - t0))
print("prologue (first tile)    ", q(t[:, 1] - t[:, 0]))
print("k-loop                   ", q(t[:, 2] - t[:, 1]))
print("stores                   ", q(t[:, 3] - t[:, 2]))
print("exit after first entry   ", q(t[:, 3] - t0))
