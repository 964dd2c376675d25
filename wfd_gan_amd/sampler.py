"""Bulk class-conditioned sampling: ``sample()`` of src/generate.py:48-64 for many traces at once.

Per chunk of rows: one-hot + z on device, eval-mode generator (BatchNorm folded to the running-statistics affine),
sigmoid + learned-length mask, then the fused integer epilogue (MinMaxScaler.inverse_transform in float64 ->
np.trim_zeros -> round-half-even -> int -> 0 becomes 1).  Embarrassingly parallel: every rank samples its own
(label, seed) shard, no collective.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from ._lib import lib
from .plan import ACT_RELU, NT, Plan, ptr, rup


def minmax_params(data_min: np.ndarray, data_max: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """scale_, min_ of sklearn's MinMaxScaler(feature_range=(0, 1)) from data_min_/data_max_ (float64)."""
    rng = np.asarray(data_max, dtype=np.float64) - np.asarray(data_min, dtype=np.float64)
    rng[rng == 0.0] = 1.0
    scale = 1.0 / rng
    return scale, 0.0 - np.asarray(data_min, dtype=np.float64) * scale


POST_MODES = {"generate": 0, "wt": 1, "glue": 2, "synth": 3}


class BulkSampler:
    def __init__(self, G_state: Dict[str, torch.Tensor], seq_len: int, class_dim: int, latent_dim: int,
                 data_min: np.ndarray, data_max: np.ndarray, chunk: int = 8192, device="cuda:0", seed: int = 0,
                 operands: str = "f32", mode: str = "generate"):
        """mode selects the integer epilogue: "generate" (src/generate.py:59-63), "wt" (gan_wt.py:79-87), "glue"
        (gan_glue.py:114-124, signed bursts) or "synth" (synthesize.py:86-90).
        operands = "f32": exact-fp32 MFMA (the parity path: integers match the reference except at rounding
        boundaries); "bf16": bf16-operand MFMA for the four Linear layers (weights and hidden activations rounded to
        bf16, fp32 accumulate and epilogues) -- several times faster, values within ~1e-2 of the fp32 path."""
        if operands not in ("f32", "bf16", "bf16x3"):
            raise ValueError("operands must be 'f32', 'bf16x3' or 'bf16'")
        if mode not in POST_MODES:
            raise ValueError(f"mode must be one of {sorted(POST_MODES)}")
        self.operands, self.mode = operands, mode
        if not torch.cuda.is_available():
            raise RuntimeError("wfd_gan_amd.BulkSampler needs a HIP device: there is no CPU path")
        self.L = lib()
        self.dev = torch.device(device)
        S, C, Z, M = seq_len, class_dim, latent_dim, chunk
        self.S, self.C, self.Z, self.M = S, C, Z, M
        self.seed = seed
        dev = self.dev
        f = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        g = {k: v.detach().to(dev, torch.float32) if v.is_floating_point() else v for k, v in G_state.items()}
        self.w = [g["model.0.weight"].contiguous(), g["model.2.weight"].contiguous(),
                  g["model.5.weight"].contiguous(), g["model.8.weight"].contiguous()]
        self.b0, self.b8 = g["model.0.bias"].contiguous(), g["model.8.bias"].contiguous()
        self.aff = []
        for lin, bn in (("model.2", "model.3"), ("model.5", "model.6")):
            sc = g[f"{bn}.weight"] / torch.sqrt(g[f"{bn}.running_var"] + 0.8)
            sh = (g[f"{lin}.bias"] - g[f"{bn}.running_mean"]) * sc + g[f"{bn}.bias"]
            self.aff.append((sc.contiguous(), sh.contiguous()))
        scale, min_ = minmax_params(data_min, data_max)
        self.scale_d = torch.from_numpy(scale).to(dev)
        self.min_d = torch.from_numpy(min_).to(dev)
        self.smin, self.srange = float(data_min[0]), float(data_max[0] - data_min[0])
        self.ldz, self.ldy = rup(Z + C, 4), rup(S, 4)
        self.zin, self.h1, self.h2, self.h3 = f(M, self.ldz), f(M, 512), f(M, 1024), f(M, 2048)
        self.y = f(M, self.ldy)
        self.labels = torch.zeros(M, dtype=torch.int32, device=dev)
        self.out = torch.zeros(M, S, dtype=torch.int32, device=dev)
        self.lens = torch.zeros(M, dtype=torch.int32, device=dev)
        self.idx = torch.zeros(M, dtype=torch.int32, device=dev)
        self.err = torch.zeros(1, dtype=torch.int32, device=dev)
        self.state = torch.zeros(2, dtype=torch.int64, device=dev)
        self.ws = torch.empty(max(M, 64) * 2048 * 2, dtype=torch.float32, device=dev)
        if operands == "bf16":
            bf = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=dev)
            self.k0 = rup(Z + C, 8)
            w0 = torch.zeros(512, self.k0, dtype=torch.float32, device=dev)
            w0[:, : Z + C] = self.w[0]
            self.w16 = [w0.to(torch.bfloat16).contiguous()] + [w.to(torch.bfloat16).contiguous() for w in self.w[1:]]
            self.zin16, self.h1_16, self.h2_16, self.h3_16 = bf(M, self.k0), bf(M, 512), bf(M, 1024), bf(M, 2048)
        self.plan_rng, self.plan = self._build()
        self.graph = None

    def _build(self):
        S, C, Z, M = self.S, self.C, self.Z, self.M
        # "bf16x3": the f32 plan with its contractions on the bf16 pipe as split-bf16 (gemm_mx.hip; products ~1e-5 relative)
        mode = 2 if self.operands == "bf16x3" else 0
        pr = Plan(self.ws)
        pr.add(self.L.rng_normal_rows, ptr(self.zin), self.ldz, M, Z, self.seed & (2**64 - 1), self.state.data_ptr(), 0)
        pr.add(self.L.iter_advance, self.state.data_ptr(), (M * Z + 3) // 4, 0)
        p = Plan(self.ws, gemm_mode=mode)
        p.add(self.L.onehot_rows, ptr(self.labels), M, C, ptr(self.zin), self.ldz, Z)
        if self.operands == "bf16":
            L = self.L
            p.add(L.f32_to_bf16, ptr(self.zin), self.ldz, ptr(self.zin16), self.k0, M, Z + C, self.k0)
            p.add(L.gemm_bf16_nt, ptr(self.zin16), self.k0, ptr(self.w16[0]), self.k0, M, 512, self.k0, 0, 0,
                  ptr(self.h1_16), 512, 0, ptr(self.b0), ACT_RELU)
            p.add(L.gemm_bf16_nt, ptr(self.h1_16), 512, ptr(self.w16[1]), 512, M, 1024, 512, 0, 0, ptr(self.h2_16), 1024,
                  ptr(self.aff[0][0]), ptr(self.aff[0][1]), ACT_RELU)
            p.add(L.gemm_bf16_nt, ptr(self.h2_16), 1024, ptr(self.w16[2]), 1024, M, 2048, 1024, 0, 0, ptr(self.h3_16),
                  2048, ptr(self.aff[1][0]), ptr(self.aff[1][1]), ACT_RELU)
            ldp = rup(S, 4)
            p.add(L.gemm_bf16_nt, ptr(self.h3_16), 2048, ptr(self.w16[3]), 2048, M, S, 2048, ptr(self.ws), ldp, 0, 0, 0, 0,
                  0)
            p.add(L.ep_sigmoid_lenmask, ptr(self.ws), 1, 0, ldp, M, S, ptr(self.b8), self.srange, self.smin, ptr(self.y),
                  self.ldy, ptr(self.idx), ptr(self.err))
            self._post(p)
            return pr, p
        p.gemm_affine(NT, self.zin, self.ldz, self.w[0], Z + C, M, 512, Z + C, None, self.b0, ACT_RELU, self.h1, 512)
        p.gemm_affine(NT, self.h1, 512, self.w[1], 512, M, 1024, 512, self.aff[0][0], self.aff[0][1], ACT_RELU,
                      self.h2, 1024)
        p.gemm_affine(NT, self.h2, 1024, self.w[2], 1024, M, 2048, 1024, self.aff[1][0], self.aff[1][1], ACT_RELU,
                      self.h3, 2048)
        s = p.gemm(NT, self.h3, 2048, self.w[3], 2048, M, S, 2048)
        p.add(self.L.ep_sigmoid_lenmask, s.ptr, s.ns, s.ss, s.ldp, M, S, ptr(self.b8), self.srange, self.smin,
              ptr(self.y), self.ldy, ptr(self.idx), ptr(self.err))
        self._post(p)
        return pr, p

    def _post(self, p: Plan):
        S, M = self.S, self.M
        if self.mode == "generate":
            p.add(self.L.sample_post, ptr(self.y), self.ldy, M, S, ptr(self.min_d), ptr(self.scale_d), ptr(self.out), S,
                  ptr(self.lens))
        else:
            p.add(self.L.sample_post_mode, ptr(self.y), self.ldy, M, S, ptr(self.min_d), ptr(self.scale_d),
                  POST_MODES[self.mode], ptr(self.out), S, ptr(self.lens))

    def capture(self):
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.plan_rng.run()
            self.plan.run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        from .parallel import graph_capture_kwargs
        with torch.cuda.graph(self.graph, **graph_capture_kwargs()):
            self.plan_rng.run()
            self.plan.run()
        torch.cuda.synchronize()

    def run_chunk(self, labels: torch.Tensor, z: Optional[torch.Tensor] = None):
        """labels: (chunk,) ints already on device or host; z: optional explicit (chunk, Z) noise (parity mode).
        Results stay on device in ``self.out`` (int32, zero padded) and ``self.lens``."""
        self.labels.copy_(labels.to(torch.int32))
        if z is None:
            if self.graph is not None:
                self.graph.replay()
                return
            self.plan_rng.run()
        else:
            self.zin[:, : self.Z].copy_(z.to(self.dev, torch.float32))
        self.plan.run()

    def check_length_index(self):
        """Host sync: raise what the reference's Generator.forward raises when a row's length output saturates
        (trunc(y0 * range + min) + 1 >= seq_size: IndexError at src/model.py:44); the kernel clamps the index and sets a flag."""
        if int(self.err.item()) != 0:
            self.err.zero_()
            raise IndexError("index out of bounds in the generator's burst-length mask (model.py:44 semantics)")

    def generate(self, labels, z: Optional[torch.Tensor] = None):
        """Any number of rows: returns (ints (n,S) int32 on device, lens (n,) int32).  Pads the last chunk.  Raises
        IndexError (once, after the last chunk) if any row's length index left the trace, as the reference would."""
        labels = torch.as_tensor(labels)
        n = labels.numel()
        outs, lens = [], []
        for a in range(0, n, self.M):
            m = min(self.M, n - a)
            lab = torch.zeros(self.M, dtype=torch.int32)
            lab[:m] = labels[a:a + m].to(torch.int32).cpu()
            zz = None
            if z is not None:
                zz = torch.zeros(self.M, self.Z)
                zz[:m] = z[a:a + m].cpu()
            self.run_chunk(lab.to(self.dev), zz)
            outs.append(self.out[:m].clone())
            lens.append(self.lens[:m].clone())
        self.check_length_index()
        return torch.cat(outs), torch.cat(lens)
