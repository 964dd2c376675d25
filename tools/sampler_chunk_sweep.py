#!/usr/bin/env python3
"""Bulk sampler throughput against the chunk size (f32 operands, int32 bursts left on device).  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import wfdgan_oracle as O
from wfd_gan_amd.sampler import BulkSampler

S, C, Z = 1401, 100, 500
torch.manual_seed(0)
gsd = O.make_generator_state(S, C, Z)
dmin, dmax = np.zeros(S), np.full(S, 1400.0)
G_MAC = (Z + C) * 512 + 512 * 1024 + 1024 * 2048 + 2048 * S
for chunk in ([int(sys.argv[1])] if len(sys.argv) > 1 else (4096, 8192, 16384, 32768, 65536)):
    bs = BulkSampler(gsd, S, C, Z, dmin, dmax, chunk=chunk, device="cuda:0", seed=1)
    bs.capture()
    labels = (torch.arange(chunk, device="cuda:0") % C).to(torch.int32)
    n = max(4, (1 << 19) // chunk)
    for _ in range(3):
        bs.run_chunk(labels)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(n):
            bs.run_chunk(labels)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    tps = n * chunk / best
    print(f"chunk {chunk:6d}: {tps / 1e6:6.2f} M traces/s  {tps * 2 * G_MAC / 1e12:6.1f} TF  {tps * 2 * G_MAC / 1e12 / 157.3:5.3f} of peak", flush=True)
    del bs
