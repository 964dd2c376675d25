#!/usr/bin/env python3
"""Large-batch sweep (BASELINE.json configs[3]; --operands f32 | bf16 | bf16x3): whole-step throughput and GEMM-core TFLOP/s vs batch size
and burst length.  GPU only.   python tools/sweep_batch.py [--L 1400 2500 5000] [--B 64 256 1024 4096]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, nargs="+", default=[1400])
ap.add_argument("--B", type=int, nargs="+", default=[64, 256, 1024, 4096])
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--operands", default="f32", choices=["f32", "bf16", "bf16x3"])
a = ap.parse_args()
dev = torch.device("cuda:0")
for L in a.L:
    for B in a.B:
        args = argparse.Namespace(seq_len=L + 1, classes=100, latent_dim=500, batch=B, inst=max(200, 2 * B // 100 + 1),
                                  penalty="div", overlap=0, dp="weak", operands=a.operands, fuse_opt="1", wk=1, force_dp=False)
        eng, _ = bench.build_engine(args, 0, 1, dev)
        eng.capture_graphs()
        # pilot, then a warm-up of >= 0.1 s and a timed region of >= 0.3 s (whole groups of 3): a short region reads high,
        # the clocks settle only after ~25 ms of load
        bench.run_steps(eng, 0, 6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_steps(eng, 6, 6)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / 6
        warm = 3 * max(1, int(0.1 / per / 3))
        steps = 3 * max(a.steps // 3, int(0.3 / per / 3))
        bench.run_steps(eng, 12, warm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_steps(eng, 12 + warm, steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        a_steps = steps
        roof = bench.gemm_roofline(eng, rep=4, rounds=2)
        print(json.dumps(dict(operands=a.operands, L=L, B=B, it_per_s=a_steps / dt, samples_per_s=a_steps * B / dt, steps=a_steps,
                              gemm_tflops=roof["tflops"], gemm_frac_of_f32_mfma_peak=roof["tflops"] / 157.3,
                              avg_us_per_gemm=roof["avg_us"])), flush=True)
        del eng
        torch.cuda.empty_cache()
