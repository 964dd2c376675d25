#!/usr/bin/env python3
"""Per-launch timing of every contraction of one training step (D-step + G-step plans) at a given batch / operand mode.
GPU only.   python tools/step_gemm_profile.py --B 4096 --operands bf16"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=1400)
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--operands", default="f32", choices=["f32", "bf16", "bf16x3"])
ap.add_argument("--fuse-opt", dest="fuse_opt", default="1")
ap.add_argument("--wk", type=int, default=1)
a = ap.parse_args()
args = argparse.Namespace(seq_len=a.L + 1, classes=100, latent_dim=500, batch=a.B, inst=max(200, 2 * a.B // 100 + 1),
                          penalty="div", overlap=0, dp="weak", operands=a.operands, fuse_opt=a.fuse_opt, wk=a.wk)
eng, _ = bench.build_engine(args, 0, 1, torch.device("cuda:0"))
roof = bench.gemm_roofline(eng, rep=4, rounds=2)
tot = sum(us * w for _, _, us, w, _ in roof["per_call"])
for kind, fl, us, w, arg in roof["per_call"]:
    if kind == "wk_multi":
        w = arg[0]
        shape = "wk x%d: " % arg[1] + " | ".join(f"lay{w[24*k]} {w[24*k+5]}x{w[24*k+6]}x{w[24*k+7]}" for k in range(arg[1]))
    elif kind == "rms_multi":
        w = arg[0]
        shape = "TN+rmsprop x%d: " % arg[1] + " | ".join(f"{w[10*k+4]}x{w[10*k+5]}x{w[10*k+6]}" for k in range(arg[1]))
    elif kind == "pair":
        shape = f"TN {arg[6]}x{arg[7]}x{arg[8]} | lay{arg[9]} {arg[17]}x{arg[18]}x{arg[19]} ns={arg[20]}"
    elif kind == "gemm":
        shape = f"lay{arg[0]} M={arg[8]} N={arg[9]} K={arg[10]} ns={arg[11]}"
    elif kind == "gemm_fused":
        shape = f"lay{arg[0]} M={arg[7]} N={arg[8]} K={arg[9]} fused"
    elif kind in ("gemm_wk", "gemm_rms"):
        shape = f"lay{arg[0]} M={arg[5]} N={arg[6]} K={arg[7]}" + (f" epi{arg[8]} cfg{arg[-1]}" if kind == "gemm_wk" else " +rmsprop")
    else:
        shape = f"conv B={arg[1]} L={arg[2]} Cin={arg[3]} Cout={arg[6]} pad={arg[4]}"
    print(f"{kind:10s} {shape:48s} x{w}  {us:9.1f} us  {fl / us / 1e6:7.1f} TF  {100 * us * w / tot:5.1f} %", flush=True)
print(f"total {tot:.1f} us per group (D x{eng.n_critic} + G), {roof['tflops']:.1f} TF average")
