"""Training-step engine: the loop body of the reference trainer (src/mlp_df_wgan_train.py:182-275) as two launch
plans (D-step, G-step) over preallocated HBM buffers, in closed form (no autograd graph).

Closed form used (verified against autograd by the oracle tests):
  forward      z_i = h_{i-1} W_i^T + b_i ; d_i = mask_i * (z_i > 0 ? 1 : .2) ; h_i = z_i o d_i ; v = h_3 w_4 + b_4
  chain        delta_3 = w_4 o d_3 ; delta_2 = (delta_3 W_3) o d_2 ; delta_1 = (delta_2 W_2) o d_1 ; g = (delta_1 W_1)[:, :S]
  penalty      n = ||g||^2 ; W-div: mean(n_r^3 + n_f^3) ; seed ghat = 6 n^2 g / B
  t-chain      t_1 = (ghat W_1[:, :S]^T) o d_1 ; t_2 = (t_1 W_2^T) o d_2 ; t_3 = (t_2 W_3^T) o d_3
  grads        dW_i = [delta_i ; dv o delta_i]^T [t_{i-1} ; h_{i-1}]  (t_0 = ghat, h_0 = [x|c]) ; db_i = sum dv o delta_i
               dw_4 = sum t_3 + sum dv o h_3 ; dv = -1/B (real rows), +1/B (fake rows)
Real and fake rows are stacked (2B rows) so every weight matrix is streamed once per pass.
"""
from __future__ import annotations

import ctypes
import os
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from ._lib import lib
from .plan import (ACT_ELU, ACT_NONE, ACT_RELU, DM_ELU, DM_MUL, DM_NONE, DM_RELU, NN, NT, TN, Plan, Slabs, ptr, rup)

G_PARAM_SPECS = lambda S, C, Z: [  # nn.Module.parameters() order of Generator, src/model.py:27-34
    ("model.0.weight", (512, Z + C)), ("model.0.bias", (512,)),
    ("model.2.weight", (1024, 512)), ("model.2.bias", (1024,)),
    ("model.3.weight", (1024,)), ("model.3.bias", (1024,)),
    ("model.5.weight", (2048, 1024)), ("model.5.bias", (2048,)),
    ("model.6.weight", (2048,)), ("model.6.bias", (2048,)),
    ("model.8.weight", (S, 2048)), ("model.8.bias", (S,)),
]
D_PARAM_SPECS = lambda S, C: [  # Discriminator, src/model.py:56-67
    ("model.0.weight", (2048, S + C)), ("model.0.bias", (2048,)),
    ("model.3.weight", (1024, 2048)), ("model.3.bias", (1024,)),
    ("model.6.weight", (512, 1024)), ("model.6.bias", (512,)),
    ("model.9.weight", (1, 512)), ("model.9.bias", (1,)),
]
G_BUFFER_KEYS = ["model.3.running_mean", "model.3.running_var", "model.6.running_mean", "model.6.running_var"]

GEMM_MODES = {"f32": 0, "bf16": 1, "bf16x3": 2}

DF_CONVS = [  # (conv prefix, bn prefix, Cin, Cout, act, pool_after) -- src/model.py:140-185
    ("layer1.0", "layer1.1", 1, 32, ACT_ELU, False), ("layer1.3", "layer1.4", 32, 32, ACT_ELU, True),
    ("layer2.0", "layer2.1", 32, 64, ACT_RELU, False), ("layer2.3", "layer2.4", 64, 64, ACT_RELU, True),
    ("layer3.0", "layer3.1", 64, 128, ACT_RELU, False), ("layer3.3", "layer3.4", 128, 128, ACT_RELU, True),
    ("layer4.0", "layer4.1", 128, 256, ACT_RELU, True),
]


class FlatParams:
    """Parameters of one network in ONE flat fp32 buffer (tensor offsets 16-byte aligned) with matching flat
    gradient and RMSprop square-average buffers: the optimizer and the gradient all-reduce are one pass each.
    Weight matrices are stored with their row stride ``ld[name]`` rounded up to 4 floats, so that every row starts
    16-byte aligned and the GEMM kernels can stage them with 128-bit loads (the critic's first layer has 1501
    columns).  The pad columns stay zero: their gradient is never written, and RMSprop maps (p=0, g=0, sq=0) to 0.
    ``p[name]`` / ``g[name]`` are views of the logical shape."""

    @staticmethod
    def layout(specs, bucket_starts: Sequence[str] = (), align: int = 4):
        """(offsets, row strides, number of elements) of the flat buffer."""
        offsets: Dict[str, int] = {}
        ld: Dict[str, int] = {}
        off = 0
        for name, shape in specs:
            if name in bucket_starts:
                off = rup(off, align)
            offsets[name] = off
            ld[name] = rup(shape[-1], 4) if len(shape) == 2 else shape[-1]
            off += rup(math.prod(shape[:-1]) * ld[name], 4)
        return offsets, ld, rup(off, align)

    def __init__(self, specs, device, bucket_starts: Sequence[str] = (), align: int = 4, storage=None):
        """bucket_starts / align (data parallel with the sharded optimizer, parallel.ShardedSync): the named tensors -- the
        first tensor of each gradient bucket -- start at multiples of ``align`` = 4 * world elements and the buffer's length is
        one too, so that every bucket splits evenly over the ranks into 16-byte aligned shards.  The pad elements stay 0.
        storage = (flat, grad): zeroed tensors of ``layout(...)[2]`` elements to use for the parameters and the gradients
        (the peer-mapped exchange region of parallel.MeshSync) instead of fresh allocations."""
        self.specs = specs
        self.offsets, self.ld, off = FlatParams.layout(specs, bucket_starts, align)
        self.numel = off
        if storage is not None:
            self.flat, self.grad = storage
            if self.flat.numel() != off or self.grad.numel() != off:
                raise ValueError("storage does not match the layout")
        else:
            self.flat = torch.zeros(off, dtype=torch.float32, device=device)
            self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.sq = torch.zeros(off, dtype=torch.float32, device=device)
        self.p = {n: self._view(self.flat, n, s) for n, s in specs}
        self.g = {n: self._view(self.grad, n, s) for n, s in specs}
        self.s = {n: self._view(self.sq, n, s) for n, s in specs}

    def vector_ranges(self):
        """Element ranges [(begin, end)] of the flat buffer that hold no weight matrix (biases, BatchNorm affine, the
        critic's head), maximal runs in storage order."""
        out, start = [], None
        for name, shape in self.specs:
            if len(shape) == 2 and shape[0] > 1:
                if start is not None:
                    out.append((start, self.offsets[name]))
                    start = None
            elif start is None:
                start = self.offsets[name]
        if start is not None:
            out.append((start, self.numel))
        return out

    def _view(self, flat: torch.Tensor, name: str, shape):
        o, ld = self.offsets[name], self.ld[name]
        if len(shape) == 2:
            return flat[o: o + shape[0] * ld].view(shape[0], ld)[:, : shape[1]]
        return flat[o: o + shape[0]]

    def load(self, sd: Dict[str, torch.Tensor]):
        for n, s in self.specs:
            t = sd[n]
            if tuple(t.shape) != tuple(s):
                raise RuntimeError(f"size mismatch for {n}: expected {tuple(s)}, got {tuple(t.shape)}")
            self.p[n].copy_(t.detach().to(torch.float32))

    def state(self) -> Dict[str, torch.Tensor]:
        return {n: self.p[n].detach().clone(memory_format=torch.contiguous_format) for n, _ in self.specs}


def subtract_interval(ranges, lo: int, hi: int):
    """[(begin, end)] minus [lo, hi): runs that overlap it are cut (a run containing it is split in two)."""
    out = []
    for a, b in ranges:
        if b <= lo or a >= hi:
            out.append((a, b))
            continue
        if a < lo:
            out.append((a, lo))
        if b > hi:
            out.append((hi, b))
    return out


def fold_bn(sd, bn: str, lin_bias: torch.Tensor, eps: float = 1e-5):
    """eval-mode BatchNorm after a conv / linear with bias -> per-channel (scale, shift)."""
    scale = sd[f"{bn}.weight"].double() / torch.sqrt(sd[f"{bn}.running_var"].double() + eps)
    shift = (lin_bias.double() - sd[f"{bn}.running_mean"].double()) * scale + sd[f"{bn}.bias"].double()
    return scale.float().contiguous(), shift.float().contiguous()


class ObserverDF:
    """Frozen DF observer (src/model.py:135-212, eval mode) packed for the channels-last kernels: conv weights as
    [Cout][8*Cin] (tap-major), their tap-flipped transposes for the input gradient, BN folded to scale/shift."""

    def __init__(self, sd: Dict[str, torch.Tensor], length: int, num_classes: int, device):
        sd = {k: v.detach().to(device=device) for k, v in sd.items()}
        self.L = length
        self.C = num_classes
        self.Ls = [length]
        for _ in range(4):
            self.Ls.append((self.Ls[-1] + 7) // 8)
        self.convs = []
        for conv, bn, cin, cout, act, pool in DF_CONVS:
            w = sd[f"{conv}.conv.weight"].float()
            assert tuple(w.shape) == (cout, cin, 8), (conv, tuple(w.shape))
            scale, shift = fold_bn(sd, bn, sd[f"{conv}.conv.bias"])
            fwd = w.permute(0, 2, 1).reshape(cout, 8 * cin).contiguous()
            bwd = w.flip(2).permute(1, 2, 0).reshape(cin, 8 * cout).contiguous()
            self.convs.append(dict(cin=cin, cout=cout, act=act, pool=pool, w=fwd, wT=bwd, scale=scale, shift=shift,
                                   w_raw=w.reshape(cout, cin * 8).contiguous()))
        L4 = self.Ls[4]
        w1 = sd["layer5.0.weight"].float()
        assert tuple(w1.shape) == (512, 256 * L4), tuple(w1.shape)
        self.fc1_w = w1.view(512, 256, L4).permute(0, 2, 1).reshape(512, L4 * 256).contiguous()
        self.fc1_scale, self.fc1_shift = fold_bn(sd, "layer5.1", sd["layer5.0.bias"])
        self.fc2_w = sd["layer5.4.weight"].float().contiguous()
        self.fc2_scale, self.fc2_shift = fold_bn(sd, "layer5.5", sd["layer5.4.bias"])
        self.fc_w = sd["fc.weight"].float().contiguous()
        self.fc_b = sd["fc.bias"].float().contiguous()
        assert tuple(self.fc_w.shape) == (num_classes, 512)


def observer_buffers(df: "ObserverDF", B: int, device) -> Dict[str, object]:
    """Activation buffers of one DF forward on B rows (channels-last)."""
    z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
    Ls = df.Ls
    ldc = rup(df.C, 4)
    return dict(
        Y=[z(B * Ls[0], 32), z(B * Ls[0], 32), z(B * Ls[1], 64), z(B * Ls[1], 64), z(B * Ls[2], 128),
           z(B * Ls[2], 128), z(B * Ls[3], 256)],
        PO=[z(B * Ls[1], 32), z(B * Ls[2], 64), z(B * Ls[3], 128), z(B * Ls[4], 256)],
        F1=z(B, 512), F2=z(B, 512), LOG=z(B, ldc), ldc=ldc)


def observer_forward(p: Plan, df: "ObserverDF", X, ldx: int, xoff: int, B: int, Y, PO, F1, F2, LOG, ldc: int,
                     convs_only: bool = False):
    """DF.forward in eval mode (src/model.py:198-206) on x[b][l] = X[b*ldx + xoff + l] -> logits LOG[B][ldc]."""
    L = p.L
    Ls, cv = df.Ls, df.convs
    p.add(L.df_conv1_fwd, ptr(X), ldx, xoff, B, Ls[0], ptr(cv[0]["w_raw"]), ptr(cv[0]["scale"]), ptr(cv[0]["shift"]),
          cv[0]["act"], ptr(Y[0]), 32)
    cur, lvl, pi = Y[0], 0, 0
    for k in range(1, 7):
        c = cv[k]
        if c["pool"]:
            p.conv8_affine_pool(cur, B, Ls[lvl], c["cin"], 3, c["w"], c["cout"], c["scale"], c["shift"], c["act"], Y[k], PO[pi])
            cur, lvl, pi = PO[pi], lvl + 1, pi + 1
        else:
            p.conv8_affine(cur, B, Ls[lvl], c["cin"], 3, c["w"], c["cout"], c["scale"], c["shift"], c["act"], Y[k])
            cur = Y[k]
    if convs_only:
        return
    for step in observer_fc_steps(p, df, B, PO, F1, F2, LOG, ldc):
        step()


def observer_fc_steps(p: Plan, df: "ObserverDF", B: int, PO, F1, F2, LOG, ldc: int):
    """The three FC layers of DF.forward (src/model.py:186-196, 204-205) as separately issuable steps."""
    kin = 256 * df.Ls[4]
    return [
        lambda: p.gemm_affine(NT, PO[3], kin, df.fc1_w, kin, B, 512, kin, df.fc1_scale, df.fc1_shift, ACT_RELU, F1, 512),
        lambda: p.gemm_affine(NT, F1, 512, df.fc2_w, 512, B, 512, 512, df.fc2_scale, df.fc2_shift, ACT_RELU, F2, 512),
        lambda: p.gemm_affine(NT, F2, 512, df.fc_w, 512, B, df.C, 512, None, df.fc_b, ACT_NONE, LOG, ldc)]


class TrainEngine:
    """One rank's state for the training hot path.  ``B`` is the per-rank batch (rows per side)."""

    def __init__(self, seq_len: int, class_dim: int, latent_dim: int, batch: int, scaler_min: float,
                 scaler_max: float, G_state: Dict[str, torch.Tensor], D_state: Dict[str, torch.Tensor],
                 DF_state: Dict[str, torch.Tensor], lr: float = 2e-4, alpha: float = 0.02, n_critic: int = 3,
                 penalty: str = "div", lambda_gp: float = 10.0, device="cuda:0", global_batch: Optional[int] = None,
                 seed: int = 0, workspace_mb: int = 256, process_group=None, world_size: int = 1,
                 share: Optional["TrainEngine"] = None, overlap: bool = False, force_sync: bool = False,
                 sync_bn: bool = False, rank: int = 0, stacked: Optional[Dict[str, torch.Tensor]] = None,
                 slot: int = 0, operands: str = "f32", pair: bool = True, wk: bool = True, fuse_opt: bool = False):
        """Data parallel, two modes:
          * sync_bn=False (weak scaling, DDP semantics): every rank trains on its own ``batch`` rows with local
            BatchNorm statistics and a local selected-row count; gradients are averaged over ranks.
          * sync_bn=True (strong scaling, exact): ``batch`` is this rank's slice of a global batch of
            ``batch * world_size`` rows; BatchNorm statistics, BatchNorm-backward sums and the selected-row count are
            combined over ranks, every mean is over the global batch and gradients are summed -- the update equals the
            single-GPU update on the global batch (up to fp32 summation order)."""
        if not torch.cuda.is_available():
            raise RuntimeError("wfd_gan_amd.TrainEngine needs a HIP device: there is no CPU path")
        if penalty not in ("div", "gp"):
            raise ValueError("penalty must be 'div' (what the reference runs) or 'gp' (src/utils.py:64-84)")
        if operands not in GEMM_MODES:
            raise ValueError(f"operands must be one of {sorted(GEMM_MODES)}")
        # contraction operands: "f32" = f32-input MFMA (exact products; the parity path), "bf16" = bf16 operands /
        # fp32 accumulate (BASELINE config 4), "bf16x3" = split-bf16 (hi*hi + hi*lo + lo*hi).  HBM stays fp32.
        self.operands, self.gemm_mode = operands, GEMM_MODES[operands]
        # pair: a weight gradient and the next product of the backward chain (independent of each other) go out as one
        # launch (Plan.gemm_pair) unless the two-stream schedule (overlap) is on
        self.pair = bool(pair) and not overlap
        # wk: small-batch products with the K split inside the workgroup and the consumer fused into the store (gemm_wk.hip)
        self.wk = bool(wk)
        # fuse_opt (single-GPU path): every weight matrix is updated inside its gradient's store (wfd_gemm_f32_rmsprop: the
        # gradient is never written or read back), the vectors by one small wfd_rmsprop_ranges launch.  Same bits as the
        # flat wfd_rmsprop pass; the weight-gradient views Dp.g / Gp.g of the matrices then stay unwritten.  Data parallel
        # must all-reduce the gradients first and keeps the flat pass; so do K-split weight gradients (large batches).
        self.fuse_opt = fuse_opt if fuse_opt in ("d", "g") else bool(fuse_opt)      # "d" / "g": one network only
        self.L = lib()
        self.dev = torch.device(device)
        S, C, Z, B = seq_len, class_dim, latent_dim, batch
        self.S, self.C, self.Z, self.B = S, C, Z, B
        self.exact = bool(sync_bn)
        self.rank = int(rank)
        # rows the batch means run over: the global batch in exact mode, this rank's batch otherwise
        self.Bg = global_batch or (B * int(world_size) if self.exact else B)
        self.lr, self.alpha, self.n_critic = lr, alpha, n_critic
        self.penalty, self.lambda_gp = penalty, lambda_gp
        # python floats -> the fp32 constants torch would use for `tensor * scalar`
        self.range_f = float(scaler_max - scaler_min)
        self.smin_f = float(scaler_min)
        self.scaler_min, self.scaler_max = scaler_min, scaler_max
        self.seed = seed
        self.rng_offset = 0
        # data parallel (DDP semantics): every rank computes its local-batch losses and gradients, the flat
        # gradient buffers are all-reduced (sum) and RMSprop scales them by 1/world_size.
        self.pg, self.world = process_group, int(world_size)
        self._gsync = None
        self.force_sync = bool(force_sync)
        self.dp = self.world > 1 or force_sync      # force_sync: run the staged data-parallel path at world 1 (tests)
        self.gscale = 1.0 if self.exact else 1.0 / self.world     # factor RMSprop applies to the summed gradients
        # data parallel, two forms of the exchange: all-reduce of the gradient buckets + the flat RMSprop pass on every rank
        # (parallel.GradSync; the default), or WFD_DP_SHARD=1: per bucket reduce-scatter -> RMSprop on this rank's shard ->
        # all-gather of the parameters (parallel.ShardedSync: optimizer pass world times smaller, twice as many collective
        # launches plus a forked side stream).  Measured on ONE GPU only (1-rank RCCL group, where the sharded form's two
        # collectives are 22.8 MB device copies the all-reduce does not make): 3130 vs 2685 it/s -- the default stays the
        # all-reduce form until a multi-GPU run says otherwise.
        self.shard = self.dp and os.environ.get("WFD_DP_SHARD", "0") == "1"
        # ... or WFD_DP_MESH=1: the hand-written peer-to-peer exchange (parallel.MeshSync, csrc/mesh.hip): parameters and
        # gradients live in a region every rank maps, and the data-parallel end of a step is ONE launch (start barrier ->
        # reduce the rank's shard from all ranks -> RMSprop on the shard -> store the parameters to all ranks -> end barrier).
        # Validated with ranks sharing one GPU; unmeasured on real links, hence opt-in too.
        self.mesh_dp = self.dp and os.environ.get("WFD_DP_MESH", "0") == "1"
        if self.mesh_dp and self.shard:
            raise ValueError("WFD_DP_MESH and WFD_DP_SHARD are two forms of the same step: choose one")
        self._mesh = None
        # "late gradients": the data-parallel step keeps the single-GPU shape (whole chain first, then all weight gradients
        # and column sums as ONE launch, then one exchange of the whole buffer) instead of interleaving weight gradients with
        # the chain so that buckets can leave early.  Default with the mesh form (its exchange is in-stream, nothing can leave
        # early: +5 % at world 1); WFD_DP_LATE_GRADS=1 selects it for the all-reduce form too (one collective instead of
        # three, no early buckets), =0 switches it off everywhere.
        lg = os.environ.get("WFD_DP_LATE_GRADS", "")
        self.late_grads = self.dp and not self.shard and not overlap and (lg == "1" or (self.mesh_dp and lg != "0"))
        self._comm = None
        self.nbt = {"model.3": 0, "model.6": 0}
        dev = self.dev
        f32 = dict(dtype=torch.float32, device=dev)

        if share is not None:
            # a second engine for another batch size (the ragged last batch of an epoch) on the SAME parameters,
            # optimizer state, BN buffers and packed observer
            self.Gp, self.Dp, self.Gbuf, self.nbt, self.df = share.Gp, share.Dp, share.Gbuf, share.nbt, share.df
            self._mesh = share._mesh
        else:
            al = 4 * max(1, self.world)
            g_lay = (G_PARAM_SPECS(S, C, Z), dev, ("model.5.weight", "model.8.weight"), al)
            d_lay = (D_PARAM_SPECS(S, C), dev, ("model.3.weight", "model.6.weight"), al)
            g_store = d_store = None
            if self.mesh_dp:
                from .parallel import MeshSync
                nG = rup(FlatParams.layout(g_lay[0], g_lay[2], al)[2], 64)
                nD = rup(FlatParams.layout(d_lay[0], d_lay[2], al)[2], 64)
                self._mesh = MeshSync(self.L, 2 * (nG + nD), dev, self.pg)
                mv = self._mesh.view
                g_store = (mv(0, nG)[:FlatParams.layout(g_lay[0], g_lay[2], al)[2]],
                           mv(nG, nG)[:FlatParams.layout(g_lay[0], g_lay[2], al)[2]])
                d_store = (mv(2 * nG, nD)[:FlatParams.layout(d_lay[0], d_lay[2], al)[2]],
                           mv(2 * nG + nD, nD)[:FlatParams.layout(d_lay[0], d_lay[2], al)[2]])
            self.Gp = FlatParams(*g_lay, storage=g_store)
            self.Dp = FlatParams(*d_lay, storage=d_store)
            self.Gp.load(G_state)
            self.Dp.load(D_state)
            self.Gbuf = {k: G_state[k].detach().to(**f32).clone() for k in G_BUFFER_KEYS}
            for k in self.nbt:
                if f"{k}.num_batches_tracked" in G_state:
                    self.nbt[k] = int(G_state[f"{k}.num_batches_tracked"])
            self.df = ObserverDF(DF_state, S - 1, C, dev)

        # ---- activations -------------------------------------------------------------------------------
        self.nside = 2 if penalty == "div" else 3     # row groups per D-step: real, fake (, interpolates)
        M = self.nside * B
        self.M = M
        ldx, ldz = rup(S + C, 4), rup(Z + C, 4)
        self.ldx, self.ldz = ldx, ldz
        z = lambda *s: torch.zeros(*s, **f32)
        b = self.buf = {}
        b["XC"] = z(2 * M, ldx)          # rows [0,M): penalty seeds ghat ; rows [M,2M): D inputs [x|c]
        b["ZIN"] = z(B, ldz)             # generator input [z|c]
        b["labels"] = torch.zeros(B, dtype=torch.int32, device=dev)
        b["MKflat"] = z(M * 3584)        # the three dropout keep-masks {0,1.25}, one RNG launch
        for i, n, o in ((1, 2048, 0), (2, 1024, 2048), (3, 512, 3072)):
            b[f"MK{i}"] = b["MKflat"][M * o: M * (o + n)].view(M, n)
            b[f"D{i}"] = z(M, n)         # local derivatives d_i
            b[f"DL{i}"] = z(2 * M, n)    # [delta_i ; dv o delta_i]
        b["T1H1"] = z(2 * M, 2048)       # [t_1 ; h_1]
        b["T2H2"] = z(2 * M, 1024)       # [t_2 ; h_2]
        b["H3"], b["T3"] = z(M, 512), z(M, 512)
        b["V"], b["NSQ"], b["PEN"], b["DV"], b["PENFLAG"] = z(M), z(M), z(M), z(M), z(M)
        b["GPA"] = z(B)                  # WGAN-GP interpolation weights alpha (B,1)
        for nme in ("NDV", "PW", "INVB", "RSELR", "RSELF"):
            b[nme] = z(M)                # constant row weights that turn the loss scalars into column sums
        # exact mode over the peer-to-peer exchange: the vectors the ranks combine (BatchNorm statistics, BatchNorm backward
        # sums, [-sum fv / Bg, #selected]) live in the mapped region, where the peers' launches read them
        zx = self._mesh.alloc_small if (self.mesh_dp and self.exact) else z
        b["dstats"], b["gstats"], b["ghead"] = z(8), z(8), zx(2)
        b["dacc"] = share.buf["dacc"] if share is not None else z(4)      # epoch sums live on device
        b["gacc"] = share.buf["gacc"] if share is not None else z(4)
        # generator
        b["GH1"] = z(B, 512)
        for i, n in ((2, 1024), (3, 2048)):
            b[f"GZ{i}"], b[f"GXH{i}"], b[f"GH{i}"] = z(B, n), z(B, n), z(B, n)
            b[f"mean{i}"], b[f"m2{i}"], b[f"istd{i}"], b[f"bns{i}"] = zx(n), zx(n), z(n), zx(2 * n)
            b[f"GDZ{i}"] = z(B, n)
        b["GDZ1"] = z(B, 512)
        self.lds4 = rup(S, 4)
        b["GDZ4"] = z(B, self.lds4)
        b["idx"] = torch.zeros(B, dtype=torch.int32, device=dev)
        b["err"] = torch.zeros(1, dtype=torch.int32, device=dev)
        # G-step critic pass (fake rows only)
        b["gMKflat"] = z(B * 3584)
        for i, n, o in ((1, 2048, 0), (2, 1024, 2048), (3, 512, 3072)):
            b[f"gMK{i}"] = b["gMKflat"][B * o: B * (o + n)].view(B, n)
            b[f"gD{i}"], b[f"gDL{i}"] = z(B, n), z(B, n)
        b["gH1"], b["gH2"], b["gH3"], b["gV"] = z(B, 2048), z(B, 1024), z(B, 512), z(B)
        # observer
        df = self.df
        Ls = df.Ls
        b["Y"] = [z(B * Ls[0], 32), z(B * Ls[0], 32), z(B * Ls[1], 64), z(B * Ls[1], 64), z(B * Ls[2], 128),
                  z(B * Ls[2], 128), z(B * Ls[3], 256)]
        b["PO"] = [z(B * Ls[1], 32), z(B * Ls[2], 64), z(B * Ls[3], 128), z(B * Ls[4], 256)]
        b["DC"] = [torch.zeros_like(t) for t in b["Y"]]
        b["DPO"] = [torch.zeros_like(t) for t in b["PO"]]
        b["F1"], b["F2"], b["DF1"], b["DF2"] = z(B, 512), z(B, 512), z(B, 512), z(B, 512)
        self.ldc = rup(C, 4)
        b["LOG"], b["DLOG"], b["CE"] = z(B, self.ldc), z(B, self.ldc), z(B)
        self.ldl = rup(S - 1, 4)
        b["DDF"] = z(B, self.ldl)
        # slab workspace: at least the largest raw product of a step (first DF conv block: B*L x 32, split-K x2)
        b["ws"] = torch.empty(max(workspace_mb * (1 << 18), B * S * 64), **f32)
        b["ws2"] = torch.empty(max(64 * (1 << 18), 2 * B * rup(S, 4)), **f32)        # slab workspace of the side lane
        b["ws_gf"] = torch.empty(max(16 * B * 2048, 1 << 20), **f32)   # slabs of a generator forward running ahead
        if stacked is not None:
            # GroupedTrainer: this engine is one slot of a stack; the buffers the batched generator forward writes are
            # views of tensors shaped [slots, ...]
            for nme, t in stacked.items():
                assert tuple(t[slot].shape) == tuple(b[nme].shape), nme
                b[nme] = t[slot]
        self.overlap = overlap

        # [philox counter, perm cursor] of the step plans; gf_state: the same pair for the stand-alone generator-forward
        # plans of the pipelined driver (they run ahead on a second stream and advance their own state)
        self.iter_state = share.iter_state if share is not None else torch.zeros(2, dtype=torch.int64, device=dev)
        self.gf_state = share.gf_state if share is not None else torch.zeros(2, dtype=torch.int64, device=dev)
        # fast mode, data parallel: all ranks hold the same permutation; rank r takes rows [r*B, (r+1)*B) of each
        # run of B*world indices and the cursor advances by the rows all ranks consumed
        self.g_tail_fused = not self.exact and B <= 256 and C <= 128
        # the G-step's independent products of equal depth (critic layer k | observer FC layer k, and their two backward
        # chains) share launches; WFD_G_MERGE=0: one launch each (same-session A/B)
        self.g_merge = self.g_tail_fused and not overlap and self.wk and os.environ.get("WFD_G_MERGE", "1") != "0"
        # (every operand mode: the weight gradients -- contractions over the 64 .. 256 batch rows, where the bf16 pipe buys
        # nothing -- always go through the f32-operand launch wfd_gemm_f32_rmsprop_multi, so the bf16 / bf16x3 modes differ from
        # the f32 mode in the arithmetic of the forward / chain products only and share its launch structure.  Round 2's bf16
        # modes kept gradient buffers and the flat pass because the bf16-pipe core's own RMSprop store lost: 2743 vs 3076 it/s)
        fuse = bool(self.fuse_opt) and not self.dp and not overlap
        # the weight gradients contract over 2 M (critic) / B (generator) rows: unsplit below 1024 (Plan.gemm)
        self.fuse_d = fuse and self.fuse_opt != "g" and 2 * M < 1024
        self.fuse_g = fuse and self.fuse_opt != "d" and B < 1024
        self.row_offset = self.rank * B if self.dp else 0
        self.cursor_step = B * self.world
        self._fill_row_constants()
        self.plan_d = self._build_d_plan()
        self.plan_g = self._build_g_plan()
        self.graph_d = self.graph_g = self.graph_group = None

    # ------------------------------------------------------------------------------------------------------
    @property
    def din(self):
        return self.buf["XC"][self.M:]

    @property
    def real_rows(self):
        return self.din[: self.B]

    @property
    def fake_rows(self):
        return self.din[self.B: 2 * self.B]

    def _fill_row_constants(self):
        B, b = self.B, self.buf
        dv = torch.zeros(self.M)
        dv[:B] = -1.0 / self.Bg
        dv[B:2 * B] = 1.0 / self.Bg
        flag = torch.zeros(self.M)
        if self.penalty == "div":
            flag[:] = 1.0
        else:
            flag[2 * B:] = 1.0
        b["DV"].copy_(dv)
        b["PENFLAG"].copy_(flag)
        b["NDV"].copy_(-dv)
        b["PW"].fill_((1.0 if self.penalty == "div" else self.lambda_gp) / self.Bg)
        b["INVB"].fill_(1.0 / self.Bg)
        sel = torch.zeros(self.M)
        sel[:B] = 1.0 / self.Bg
        b["RSELR"].copy_(sel)
        sel = torch.zeros(self.M)
        sel[B:2 * B] = 1.0 / self.Bg
        b["RSELF"].copy_(sel)

    # ------------------------------------------------------------------------------------------------------
    def _g_forward(self, p: Plan, nupd: int):
        """Generator.forward in training mode (src/model.py:36-48) -> fake rows of the critic input."""
        B, S, C, Z, b, G = self.B, self.S, self.C, self.Z, self.buf, self.Gp.p
        p.gemm_affine(NT, b["ZIN"], self.ldz, G["model.0.weight"], self.Gp.ld["model.0.weight"], B, 512, Z + C, None,
                      G["model.0.bias"], ACT_RELU, b["GH1"], 512)
        prev, kprev = b["GH1"], 512
        for i, lin, bn, n in ((2, "model.2", "model.3", 1024), (3, "model.5", "model.6", 2048)):
            s = p.gemm(NT, prev, kprev, G[f"{lin}.weight"], kprev, B, n, kprev)
            if not self.dp:
                p.add(self.L.bn_train_relu, s.ptr, s.ns, s.ss, s.ldp, B, n, ptr(G[f"{lin}.bias"]), 0.8,
                      ptr(G[f"{bn}.weight"]), ptr(G[f"{bn}.bias"]), ptr(b[f"GXH{i}"]), ptr(b[f"GH{i}"]), n,
                      ptr(b[f"mean{i}"]), ptr(b[f"m2{i}"]), ptr(b[f"istd{i}"]), ptr(self.Gbuf[f"{bn}.running_mean"]),
                      ptr(self.Gbuf[f"{bn}.running_var"]), 0.1, nupd)
                prev, kprev = b[f"GH{i}"], n
                continue
            p.add(self.L.bn_stats, s.ptr, s.ns, s.ss, s.ldp, B, n, ptr(G[f"{lin}.bias"]), ptr(b[f"GZ{i}"]), n,
                  ptr(b[f"mean{i}"]), ptr(b[f"m2{i}"]))
            self._bn_sync_hook(p, i)
            p.add(self.L.bn_apply_relu, ptr(b[f"GZ{i}"]), n, B, n, ptr(b[f"mean{i}"]), ptr(b[f"m2{i}"]),
                  float(self.Bg), 0.8, ptr(G[f"{bn}.weight"]), ptr(G[f"{bn}.bias"]), ptr(b[f"GXH{i}"]),
                  ptr(b[f"GH{i}"]), n, ptr(b[f"istd{i}"]), ptr(self.Gbuf[f"{bn}.running_mean"]),
                  ptr(self.Gbuf[f"{bn}.running_var"]), 0.1, nupd)
            prev, kprev = b[f"GH{i}"], n
        s = p.gemm(NT, prev, 2048, G["model.8.weight"], 2048, B, S, 2048)
        p.add(self.L.ep_sigmoid_lenmask, s.ptr, s.ns, s.ss, s.ldp, B, S, ptr(G["model.8.bias"]), self.range_f,
              self.smin_f, ptr(self.fake_rows), self.ldx, ptr(b["idx"]), ptr(b["err"]))

    @property
    def comm(self):
        if self._comm is None:
            from .parallel import Comm
            self._comm = Comm(self.pg)
        return self._comm

    def _bn_sync_hook(self, p: Plan, layer: int):
        """Exact data-parallel mode: the ranks' per-feature (mean, M2) are combined into the global-batch pair between
        the statistics kernel and the normalisation (parallel.Comm.combine_bn_stats_); otherwise nothing."""
        if not self.exact:
            return
        mean, m2, B = self.buf[f"mean{layer}"], self.buf[f"m2{layer}"], self.B
        if self.mesh_dp:        # one launch: every rank reads all ranks' pairs and forms the global pair itself
            ms = self._mesh
            p.add(self.L.mesh_bn_combine, ms.mesh, ms.offset_of(mean), ms.offset_of(m2), mean.numel(), float(B))
            return
        p.host(lambda: self.comm.combine_bn_stats_(mean, m2, B), self.comm.capturable)

    def _critic_forward(self, p: Plan, X, M: int, mk, H1, H2, H3, D1, D2, D3, V, DL3, DL3s, rowscale):
        """Discriminator.forward (src/model.py:69-72) on M rows of X, keeping d_i; ends with the head + delta_3."""
        S, C, D = self.S, self.C, self.Dp.p
        p.gemm_lrelu_drop(NT, X, self.ldx, D["model.0.weight"], self.Dp.ld["model.0.weight"], M, 2048, S + C,
                          D["model.0.bias"], mk[0], 2048, H1, 2048, D1, 2048)
        p.gemm_lrelu_drop(NT, H1, 2048, D["model.3.weight"], 2048, M, 1024, 2048, D["model.3.bias"], mk[1], 1024, H2, 1024,
                          D2, 1024)
        s = p.gemm(NT, H2, 1024, D["model.6.weight"], 1024, M, 512, 1024)
        # layer-3 epilogue + head (nn.Linear(512, 1)) + first step of the input-gradient chain in one launch
        p.add(self.L.d_head_lrelu, s.ptr, s.ns, s.ss, s.ldp, ptr(D["model.6.bias"]), ptr(mk[2]), 512, 0.2, ptr(H3), 512,
              ptr(D3), 512, ptr(D["model.9.weight"]), ptr(D["model.9.bias"]), M, 512, ptr(rowscale), ptr(V), ptr(DL3),
              ptr(DL3s), 512)

    def _critic_chain(self, p: Plan, M: int, DL3, DL2, DL2s, DL1, DL1s, D2, D1, rowscale) -> Slabs:
        """d validity / d traces in closed form; returns the slabs of g = (delta_1 W_1)[:, :S]."""
        S, C, D = self.S, self.C, self.Dp.p
        p.gemm_mul(NN, DL3, 512, D["model.6.weight"], 1024, M, 1024, 512, D2, 1024, DM_MUL, None, DL2, 1024,
                   rowscale=rowscale, out2=DL2s)
        p.gemm_mul(NN, DL2, 1024, D["model.3.weight"], 2048, M, 2048, 1024, D1, 2048, DM_MUL, None, DL1, 2048,
                   rowscale=rowscale, out2=DL1s)
        return p.gemm(NN, DL1, 2048, D["model.0.weight"], self.Dp.ld["model.0.weight"], M, S, 2048)

    GOLD = 0x9E3779B97F4A7C15

    def _rng_inc(self) -> int:
        return max((self.B * self.Z + 3) // 4, (self.buf["MKflat"].numel() + 3) // 4)

    def _build_d_plan(self, fast: bool = False, part: str = "all", draw_masks: bool = True) -> Plan:
        """fast: the batch, z and the dropout masks are produced on device from the resident dataset / Philox and
        the step ends by advancing the device-resident iteration state (graph-replayable); otherwise the caller
        supplies them (parity mode).
        part: "all" = the whole D-step; "gf" = only its head (batch gather, z / mask draws, generator forward), which
        depends on nothing the critic update writes; "rest" = everything after it (PipelinedTrainer)."""
        B, M, S, C, Z, b, D = self.B, self.M, self.S, self.C, self.Z, self.buf, self.Dp
        p = Plan(b["ws"] if part != "gf" else b["ws_gf"], b["ws2"], self.overlap, self.gemm_mode, self.wk)
        st = self.iter_state.data_ptr() if part != "gf" else self.gf_state.data_ptr()
        if part in ("rest", "rest_draw"):
            if part == "rest_draw":
                # the head ran stacked with other iterations' heads (GroupedTrainer) and drew only z: the dropout masks
                # (and the WGAN-GP interpolates) of this iteration are produced here
                # (draw_masks = False: the group's head already drew them, GroupedTrainer._build_head)
                t = b["MKflat"]
                if draw_masks:
                    p.add(self.L.rng_dropout_mask, ptr(t), t.numel(), 0.2, (self.seed + self.GOLD) & (2**64 - 1), st, 0)
                if self.penalty == "gp":
                    p.add(self.L.rng_uniform, ptr(b["GPA"]), B, (self.seed + 11 * self.GOLD) & (2**64 - 1), st, 0)
                    p.add(self.L.interpolate_rows, ptr(self.real_rows), ptr(self.fake_rows), ptr(b["GPA"]),
                          ptr(self.din[2 * B:]), B, S, S + C, self.ldx)
            return self._build_d_rest(p, fast, own_cursor=False)
        if fast:
            p.add(self.L.gather_batch, ptr(self.data_X), S, ptr(self.data_y), ptr(self.perm), st, self.N, self.N,
                  self.row_offset, B, S, C, ptr(self.real_rows), ptr(self.fake_rows), self.ldx, ptr(b["ZIN"]), self.ldz, Z, ptr(b["labels"]))
            p.add(self.L.rng_step_inputs, ptr(b["ZIN"]), self.ldz, B, Z, (self.seed + 7 * self.GOLD) & (2**64 - 1),
                  ptr(b["MKflat"]), b["MKflat"].numel(), 0.2, (self.seed + self.GOLD) & (2**64 - 1), st)
            if self.penalty == "gp":
                p.add(self.L.rng_uniform, ptr(b["GPA"]), B, (self.seed + 11 * self.GOLD) & (2**64 - 1), st, 0)
        self._g_forward(p, nupd=1)
        if self.penalty == "gp":
            # interpolates = third row group of the critic input (src/utils.py:67-69)
            p.add(self.L.interpolate_rows, ptr(self.real_rows), ptr(self.fake_rows), ptr(b["GPA"]),
                  ptr(self.din[2 * B:]), B, S, S + C, self.ldx)
        if part == "gf":
            p.add(self.L.iter_advance, st, self._rng_inc(), self.cursor_step)
            return p
        return self._build_d_rest(p, fast, own_cursor=True)

    def _build_d_rest(self, p: Plan, fast: bool, own_cursor: bool) -> Plan:
        """Critic forward on the stacked rows, closed-form penalty, backward, RMSprop (everything of the D-step
        after the generator forward)."""
        B, M, S, C, Z, b, D = self.B, self.M, self.S, self.C, self.Z, self.buf, self.Dp
        st = self.iter_state.data_ptr()
        din = self.din
        H1, H2 = b["T1H1"][M:], b["T2H2"][M:]
        self._critic_forward(p, din, M, (b["MK1"], b["MK2"], b["MK3"]), H1, H2, b["H3"], b["D1"], b["D2"], b["D3"],
                             b["V"], b["DL3"][:M], b["DL3"][M:], b["DV"])
        g = self._critic_chain(p, M, b["DL3"][:M], b["DL2"][:M], b["DL2"][M:], b["DL1"][:M], b["DL1"][M:], b["D2"],
                               b["D1"], b["DV"])
        bias_segs = [
            (b["DL1"][M:], 2048, None, None, 0, None, M, 2048, D.g["model.0.bias"]),
            (b["DL2"][M:], 1024, None, None, 0, None, M, 1024, D.g["model.3.bias"]),
            (b["DL3"][M:], 512, None, None, 0, None, M, 512, D.g["model.6.bias"])]
        # data parallel over the peer-to-peer exchange: nothing leaves before the end of the step, so the step keeps the
        # single-GPU shape -- the whole t-chain first, then ALL weight gradients and column sums as one launch (stored, not
        # consumed), then the one launch that sums the ranks' gradients, applies RMSprop and redistributes the parameters
        late = self.late_grads
        if self.dp and not late:
            # data parallel: the bias gradients (column sums of dv o delta_i, final once the chain is) are produced here,
            # ahead of the weight gradients, so that [W1, b1] and [W2, b2] can go out as whole buckets early
            self._colsum_multi(p, "d_bias", bias_segs)
            bias_segs = []
        mode = 0 if self.penalty == "div" else 1
        p.add(self.L.ep_penalty_row, g.ptr, g.ns, g.ss, g.ldp, M, S, mode, 1.0 / self.Bg, self.lambda_gp,
              ptr(b["PENFLAG"]), ptr(b["NSQ"]), ptr(b["PEN"]), ptr(b["XC"]), self.ldx)
        # t-chain (second-order part of d_loss.backward()) on the main lane; each layer's weight gradient (one TN
        # product over the stacked [penalty ; standard] rows) runs beside it on the side lane as soon as its
        # operands exist
        Dp = D.p
        d_problems = None
        if self.fuse_d or late:
            # single GPU: the whole t-chain first (it reads W_1..W_3), then the three weight gradients, each consumed by RMSprop
            # in its own store, as ONE launch
            ld1 = D.ld["model.0.weight"]
            p.gemm_mul(NT, b["XC"], self.ldx, Dp["model.0.weight"], ld1, M, 2048, S, b["D1"], 2048, DM_MUL, None, b["T1H1"][:M], 2048)
            p.gemm_mul(NT, b["T1H1"][:M], 2048, Dp["model.3.weight"], 2048, M, 1024, 2048, b["D2"], 1024, DM_MUL, None, b["T2H2"][:M], 1024)
            p.gemm_mul(NT, b["T2H2"][:M], 1024, Dp["model.6.weight"], 1024, M, 512, 1024, b["D3"], 512, DM_MUL, None, b["T3"], 512)
            d_problems = [
                (b["DL1"], 2048, b["XC"], self.ldx, 2048, S + C, 2 * M, Dp["model.0.weight"], ld1, D.s["model.0.weight"]),
                (b["DL2"], 1024, b["T1H1"], 2048, 1024, 2048, 2 * M, Dp["model.3.weight"], 2048, D.s["model.3.weight"]),
                (b["DL3"], 512, b["T2H2"], 1024, 512, 1024, 2 * M, Dp["model.6.weight"], 1024, D.s["model.6.weight"])]
            if late:
                # (stored to the gradient matrices instead: same operands, the gradient matrix in place of the parameter matrix)
                d_problems = [pr[:7] + (D.g[n], D.ld[n]) for pr, n in zip(d_problems, ("model.0.weight", "model.3.weight",
                                                                                       "model.6.weight"))]
            elif os.environ.get("WFD_D_END_MERGE", "1") == "0":             # (the switch: same-session A/B)
                p.gemm_rms_multi(d_problems, self.lr)
                d_problems = None
        elif self.pair:
            # each layer's weight gradient (one TN product over the stacked [penalty ; standard] rows) shares a launch
            # with the t-chain product that follows it: neither reads what the other writes
            p.gemm_pair_mul((b["DL1"], 2048, b["XC"], self.ldx, 2048, S + C, 2 * M, D.g["model.0.weight"], D.ld["model.0.weight"]),
                            (NT, b["XC"], self.ldx, Dp["model.0.weight"], D.ld["model.0.weight"], M, 2048, S),
                            b["D1"], 2048, DM_MUL, b["T1H1"][:M], 2048)
            self._grad_bucket_hook(p, "D", 0)
            p.gemm_pair_mul((b["DL2"], 1024, b["T1H1"], 2048, 1024, 2048, 2 * M, D.g["model.3.weight"], 2048),
                            (NT, b["T1H1"][:M], 2048, Dp["model.3.weight"], 2048, M, 1024, 2048),
                            b["D2"], 1024, DM_MUL, b["T2H2"][:M], 1024)
            self._grad_bucket_hook(p, "D", 1)
            p.gemm_pair_mul((b["DL3"], 512, b["T2H2"], 1024, 512, 1024, 2 * M, D.g["model.6.weight"], 1024),
                            (NT, b["T2H2"][:M], 1024, Dp["model.6.weight"], 1024, M, 512, 1024),
                            b["D3"], 512, DM_MUL, b["T3"], 512)
        else:
            self._d_backward_two_lane(p)
        st_arg = (st if fast else 0, self._rng_inc() if fast else 0, (self.cursor_step if own_cursor else 0) if fast else 0)
        # single GPU with the matrices updated in their gradients' stores: every remaining critic parameter (biases, head) has
        # its gradient produced by THIS launch, which then applies their RMSprop step too and ends the D-step
        d_segs = bias_segs + [
            (b["T3"], 512, None, b["H3"], 512, b["DV"], M, 512, D.g["model.9.weight"]),
            (b["DV"], 1, None, None, 0, None, M, 1, D.g["model.9.bias"]),
            # loss scalars of :215-227 as weighted column sums of the per-row critic scores / penalties:
            # dstats = [d_loss, penalty, w_dist, mean_rv, mean_fv]; dacc += [d_loss, w_dist] (epoch sums)
            (b["V"], 1, b["DV"], b["PEN"], 1, b["PW"], M, 1, b["dstats"][0:1], b["dacc"][0:1]),
            (b["PEN"], 1, b["INVB"], None, 0, None, M, 1, b["dstats"][1:2]),
            (b["V"], 1, b["NDV"], None, 0, None, M, 1, b["dstats"][2:3], b["dacc"][1:2]),
            (b["V"], 1, b["RSELR"], None, 0, None, M, 1, b["dstats"][3:4]),
            (b["V"], 1, b["RSELF"], None, 0, None, M, 1, b["dstats"][4:5])]
        if late:
            desc, blk = self._colsum_desc(p, d_segs)
            if not p.gemm_multi_grad(d_problems, colsum=(ptr(desc), len(d_segs), blk)):
                self._colsum_multi(p, "d_small", d_segs, opt=None)
            self._grad_sync_hook(p, "D", st_arg)
            self._optimizer(p, D, False, *st_arg)
            return p
        if self.fuse_d and d_problems is not None:
            # ... and those column sums ride in the first workgroups of the weight-gradient launch: the end of the D-step
            # is ONE launch (they read the chain's activations and row scalars, never the parameter matrices)
            desc, blk = self._colsum_desc(p, d_segs)
            if not p.gemm_rms_multi(d_problems, self.lr, colsum=(ptr(desc), len(d_segs), blk, ptr(D.grad), ptr(D.flat), ptr(D.sq),
                                                                  D.numel) + st_arg):
                self._colsum_multi(p, "d_small", d_segs, opt=(D, st_arg))
            return p
        self._colsum_multi(p, "d_small", d_segs, opt=(D, st_arg) if self.fuse_d else None)
        if self.fuse_d:
            return p
        if self.overlap:
            p.join()
        self._grad_sync_hook(p, "D", st_arg)
        self._optimizer(p, D, False, *st_arg)
        return p

    def _optimizer(self, p: Plan, fp: FlatParams, fused: bool, state: int, inc0: int, inc1: int):
        """optimizer.step() (:230, :275): the flat RMSprop pass, or -- the matrices having been updated in their gradients'
        stores -- the vectors only.  Either also advances the device-resident Philox counter / batch cursor."""
        if not fused:
            if self.shard or self.mesh_dp:   # done shard by shard inside the exchange (_grad_bucket_hook)
                return
            p.add(self.L.rmsprop, ptr(fp.flat), ptr(fp.grad), ptr(fp.sq), fp.numel, self.lr, 0.99, 1e-8, self.gscale,
                  state, inc0, inc1)
            return
        flat = [v for r in fp.vector_ranges() for v in r]
        arr = (ctypes.c_long * len(flat))(*flat)
        p.keep.append(arr)
        p.add(self.L.rmsprop_ranges, ptr(fp.flat), ptr(fp.grad), ptr(fp.sq), arr, len(flat) // 2, self.lr, 0.99, 1e-8,
              state, inc0, inc1)

    def _d_backward_two_lane(self, p: Plan):
        """The same backward with every product its own launch: weight gradients on the side lane (overlap=True runs it
        on a second stream), the t-chain on the main lane."""
        B, M, S, C, b, D = self.B, self.M, self.S, self.C, self.buf, self.Dp
        Dp = D.p
        p.fork()
        with p.side():
            p.gemm(TN, b["DL1"], 2048, b["XC"], self.ldx, 2048, S + C, 2 * M, out=D.g["model.0.weight"],
                   ldp=D.ld["model.0.weight"])
        self._grad_bucket_hook(p, "D", 0)
        s = p.gemm(NT, b["XC"], self.ldx, Dp["model.0.weight"], D.ld["model.0.weight"], M, 2048, S)
        p.mul(s, b["D1"], 2048, DM_MUL, None, None, b["T1H1"][:M], None, 2048)
        p.fork()
        with p.side():
            p.gemm(TN, b["DL2"], 1024, b["T1H1"], 2048, 1024, 2048, 2 * M, out=D.g["model.3.weight"], ldp=2048)
        self._grad_bucket_hook(p, "D", 1)
        s = p.gemm(NT, b["T1H1"][:M], 2048, Dp["model.3.weight"], 2048, M, 1024, 2048)
        p.mul(s, b["D2"], 1024, DM_MUL, None, None, b["T2H2"][:M], None, 1024)
        p.fork()
        with p.side():
            p.gemm(TN, b["DL3"], 512, b["T2H2"], 1024, 512, 1024, 2 * M, out=D.g["model.6.weight"], ldp=1024)
        s = p.gemm(NT, b["T2H2"][:M], 1024, Dp["model.6.weight"], 1024, M, 512, 1024)
        p.mul(s, b["D3"], 512, DM_MUL, None, None, b["T3"], None, 512)

    def _colsum_desc(self, p: Plan, segs):
        """Device descriptor of wfd_colsum_multi's segments (11 words each) and the launch's number of 32-column blocks."""
        rows, blk = [], 0
        for sg in segs:
            X, ldx, rs, X2, ldx2, rs2, M, N, out = sg[:9]
            acc = sg[9] if len(sg) > 9 else None
            rows.append([ptr(X), ldx, ptr(rs), ptr(X2), ldx2, ptr(rs2), M, N, ptr(out), blk, ptr(acc)])
            blk += (N + 31) // 32
        desc = torch.tensor(rows, dtype=torch.int64, device=self.dev)
        p.keep.append(desc)          # owned by the plan: a later rebuild of a D plan must not free what this one points at
        return desc, blk

    def _colsum_multi(self, p: Plan, name: str, segs, opt=None):
        """segs: (X, ldx, rowscale, X2, ldx2, rowscale2, M, N, out) -> one launch of wfd_colsum_multi.
        opt = (FlatParams, (state, inc0, inc1)): outputs inside that network's gradient buffer also take their RMSprop
        step in the launch (wfd_colsum_multi_rmsprop), which then does the step's bookkeeping."""
        desc, blk = self._colsum_desc(p, segs)
        rows = segs
        if opt is None:
            p.add(self.L.colsum_multi, ptr(desc), len(rows), blk)
        else:
            fp, (state, inc0, inc1) = opt
            p.add(self.L.colsum_multi_rmsprop, ptr(desc), len(rows), blk, ptr(fp.grad), ptr(fp.flat), ptr(fp.sq), fp.numel,
                  self.lr, 0.99, 1e-8, state, inc0, inc1)

    def _grad_bucket_hook(self, p: Plan, which: str, stage: int, st_arg=None):
        """Data parallel: all-reduce the flat gradient buffer in three buckets, each launched (asynchronously, on the
        collective's stream) as soon as its slice is final, so that only the last, small one is exposed:
          critic     0: [W1, b1] (12.3 MB, right after the first weight-gradient GEMM)   1: [W2, b2] (8.4 MB)
                     2: [W3, b3, w4, b4] (2.1 MB, end of the step)
          generator  0: [W8, b8] (11.5 MB, produced first by the backward)   1: [W5, b5, BN6] (8.4 MB)
                     2: [W0, b0, W2, b2, BN3] (3.3 MB, end of the step)
        The optimizer waits for all three (stage 2 also finishes)."""
        if not self.dp:
            return
        if self.mesh_dp:
            # peer-to-peer form: nothing per bucket (an in-stream exchange in the middle of the backward would only make the
            # ranks wait for each other three times); at the end of the step ONE launch sums, updates and redistributes
            if stage == 2:
                fp, ms = (self.Dp if which == "D" else self.Gp), self._mesh
                st = tuple(st_arg) if st_arg is not None else (0, 0, 0)
                p.add(self.L.mesh_reduce_rmsprop, ms.mesh, ms.offset_of(fp.grad), ms.offset_of(fp.flat), ptr(fp.sq), fp.numel,
                      self.lr, 0.99, 1e-8, self.gscale, *st)
                p.keep.append(ms.base)
            return
        if self._gsync is None:
            from .parallel import GradSync, ShardedSync
            self._gsync = ShardedSync(self.L, self.pg) if self.shard else GradSync(self.pg)
        if self._gsync.reset not in p.abort_hooks:
            p.abort_hooks.append(self._gsync.reset)
        lo, hi = self.grad_buckets(which)[stage]
        fp = self.Dp if which == "D" else self.Gp
        gs = self._gsync
        if self.shard:
            # sharded optimizer: this bucket's reduce-scatter -> RMSprop shard -> parameter all-gather chain starts now (side
            # stream); the last bucket also does the step's bookkeeping and joins every chain
            lr, gscale = self.lr, self.gscale
            if stage < 2:
                p.host(lambda: gs.start(fp.flat, fp.grad, fp.sq, lo, hi, lr, gscale), gs.capturable)
            else:
                st = tuple(st_arg) if st_arg is not None else (0, 0, 0)
                p.host(lambda: (gs.start(fp.flat, fp.grad, fp.sq, lo, hi, lr, gscale, st), gs.finish()), gs.capturable)
            return
        bucket = fp.grad if self.late_grads else fp.grad[lo:hi]       # late gradients: ONE collective over the whole buffer
        if stage < 2:
            p.host(lambda: gs.start(bucket), gs.capturable)
        else:
            p.host(lambda: (gs.start(bucket), gs.finish()), gs.capturable)

    def grad_buckets(self, which: str):
        """[(begin, end)] x 3 element ranges of the flat gradient buffer, in the order the backward completes them."""
        if which == "D":
            o, n = self.Dp.offsets, self.Dp.numel
            return [(0, o["model.3.weight"]), (o["model.3.weight"], o["model.6.weight"]), (o["model.6.weight"], n)]
        o, n = self.Gp.offsets, self.Gp.numel
        return [(o["model.8.weight"], n), (o["model.5.weight"], o["model.8.weight"]), (0, o["model.5.weight"])]

    def _grad_sync_hook(self, p: Plan, which: str, st_arg=None):
        self._grad_bucket_hook(p, which, 2, st_arg)

    def _observer_forward(self, p: Plan):
        b = self.buf
        observer_forward(p, self.df, self.fake_rows, self.ldx, 1, self.B, b["Y"], b["PO"], b["F1"], b["F2"], b["LOG"], self.ldc)

    def _observer(self, p: Plan, forward: bool = True):
        """DF forward on fake[:, 1:], masked CE, and the input gradient dDF (src/mlp_df_wgan_train.py:251-263,273).
        forward = False: the forward already ran (GroupedTrainer issues it beside the group's critic updates)."""
        B, b, df = self.B, self.buf, self.df
        Ls, Y, PO, DC, DPO = df.Ls, b["Y"], b["PO"], b["DC"], b["DPO"]
        cv = df.convs
        kin = 256 * Ls[4]
        if forward:
            self._observer_forward(p)
        p.join()                       # critic scores of the fake rows (side lane) are needed from here on
        if self.g_tail_fused:
            # g_head + masked CE + G-step scalars as one single-workgroup launch (small batch, no all-reduce in between)
            p.add(self.L.g_tail, ptr(b["LOG"]), self.ldc, B, self.C, ptr(b["labels"]), ptr(b["gV"]), 1.0 / self.Bg, self.alpha,
                  ptr(b["CE"]), ptr(b["DLOG"]), self.ldc, ptr(b["ghead"]), ptr(b["gstats"]), ptr(b["gacc"]))
        else:
            self._nsel_sync_hook(p)
            p.add(self.L.masked_ce, ptr(b["LOG"]), self.ldc, B, self.C, ptr(b["labels"]), ptr(b["gV"]),
                  b["ghead"].data_ptr() + 4, self.alpha, ptr(b["CE"]), ptr(b["DLOG"]), self.ldc)
            p.add(self.L.g_loss, ptr(b["CE"]), B, ptr(b["ghead"]), self.alpha, ptr(b["gstats"]), ptr(b["gacc"]))
        self._observer_backward(p)

    def _observer_backward_fc_steps(self, p: Plan):
        """Input gradient through the observer's three FC layers (src/model.py:186-196), as separately issuable steps."""
        B, b, df = self.B, self.buf, self.df
        kin = 256 * df.Ls[4]
        return [
            lambda: p.gemm_mul(NN, b["DLOG"], self.ldc, df.fc_w, 512, B, 512, self.C, b["F2"], 512, DM_RELU, df.fc2_scale, b["DF2"], 512),
            lambda: p.gemm_mul(NN, b["DF2"], 512, df.fc2_w, 512, B, 512, 512, b["F1"], 512, DM_RELU, df.fc1_scale, b["DF1"], 512),
            lambda: p.gemm_mul(NN, b["DF1"], 512, df.fc1_w, kin, B, kin, 512, None, 0, DM_NONE, None, b["DPO"][3], kin)]

    def _g_front_merged(self, p: Plan) -> Slabs:
        """The G-step up to the generator's backward with the INDEPENDENT products of equal depth sharing launches
        (Plan.batch -> wfd_gemm_f32_wk_multi): the critic's three layers on the generated rows beside the observer's three FC
        layers, and the critic's input-gradient chain beside the observer's FC backward chain.  Same kernels on the same
        operands as the separate launches (bit-identical); 6 launches fewer.  Returns the slabs of g = (delta_1 W_1)[:, :S]."""
        B, S, C, b, df = self.B, self.S, self.C, self.buf, self.df
        D, ld1 = self.Dp.p, self.Dp.ld["model.0.weight"]
        observer_forward(p, df, self.fake_rows, self.ldx, 1, B, b["Y"], b["PO"], b["F1"], b["F2"], b["LOG"], self.ldc,
                         convs_only=True)
        fc = observer_fc_steps(p, df, B, b["PO"], b["F1"], b["F2"], b["LOG"], self.ldc)
        with p.batch():
            fc[0]()
            p.gemm_lrelu_drop(NT, self.fake_rows, self.ldx, D["model.0.weight"], ld1, B, 2048, S + C, D["model.0.bias"],
                              b["gMK1"], 2048, b["gH1"], 2048, b["gD1"], 2048)
        with p.batch():
            fc[1]()
            p.gemm_lrelu_drop(NT, b["gH1"], 2048, D["model.3.weight"], 2048, B, 1024, 2048, D["model.3.bias"], b["gMK2"], 1024,
                              b["gH2"], 1024, b["gD2"], 1024)
        with p.batch():
            fc[2]()
            with p.side(stream=False):           # slabs that must outlive the observer's main-lane products: side workspace
                s = p.gemm(NT, b["gH2"], 1024, D["model.6.weight"], 1024, B, 512, 1024)
        p.add(self.L.d_head_lrelu, s.ptr, s.ns, s.ss, s.ldp, ptr(D["model.6.bias"]), ptr(b["gMK3"]), 512, 0.2, ptr(b["gH3"]), 512,
              ptr(b["gD3"]), 512, ptr(D["model.9.weight"]), ptr(D["model.9.bias"]), B, 512, 0, ptr(b["gV"]), ptr(b["gDL3"]), 0, 512)
        p.add(self.L.g_tail, ptr(b["LOG"]), self.ldc, B, self.C, ptr(b["labels"]), ptr(b["gV"]), 1.0 / self.Bg, self.alpha,
              ptr(b["CE"]), ptr(b["DLOG"]), self.ldc, ptr(b["ghead"]), ptr(b["gstats"]), ptr(b["gacc"]))
        bw = self._observer_backward_fc_steps(p)
        with p.batch():
            bw[0]()
            p.gemm_mul(NN, b["gDL3"], 512, D["model.6.weight"], 1024, B, 1024, 512, b["gD2"], 1024, DM_MUL, None, b["gDL2"], 1024)
        with p.batch():
            bw[1]()
            p.gemm_mul(NN, b["gDL2"], 1024, D["model.3.weight"], 2048, B, 2048, 1024, b["gD1"], 2048, DM_MUL, None, b["gDL1"], 2048)
        with p.batch():
            bw[2]()
            with p.side(stream=False):
                g = p.gemm(NN, b["gDL1"], 2048, D["model.0.weight"], ld1, B, S, 2048)
        self._observer_backward(p, fc=False)
        return g

    def _observer_backward(self, p: Plan, fc: bool = True):
        B, b, df = self.B, self.buf, self.df
        Ls, Y, PO, DC, DPO = df.Ls, b["Y"], b["PO"], b["DC"], b["DPO"]
        cv = df.convs
        if fc:
            for step in self._observer_backward_fc_steps(p):
                step()
        # walk the conv stack backwards; `lvl` = index of the length the conv runs at
        k, lvl, pi = 6, 3, 3
        while k >= 1:
            c = cv[k]
            if c["pool"]:
                p.add(self.L.df_pool8_bwd_act, ptr(DPO[pi]), ptr(Y[k]), B, Ls[lvl], c["cout"], ptr(c["scale"]),
                      c["act"], ptr(DC[k]))
                pi -= 1
            # DC[k] = gradient at conv k's raw output.  Push it through conv k to the activation feeding it.
            prev = cv[k - 1]
            if prev["pool"]:
                p.conv8_mul(DC[k], B, Ls[lvl], c["cout"], 4, c["wT"], c["cin"], None, 0, DM_NONE, None, DPO[pi])
                lvl -= 1
            else:
                dm = DM_ELU if prev["act"] == ACT_ELU else DM_RELU
                p.conv8_mul(DC[k], B, Ls[lvl], c["cout"], 4, c["wT"], c["cin"], Y[k - 1], prev["cout"], dm,
                            prev["scale"], DC[k - 1])
            k -= 1
        p.add(self.L.df_conv1_bwd, ptr(DC[0]), B, Ls[0], 32, ptr(cv[0]["w_raw"]), ptr(b["DDF"]), self.ldl)

    def _nsel_sync_hook(self, p: Plan):
        """Exact mode: ghead = [-sum fv / Bg, #selected] are per-rank partial sums; their all-reduce gives the global
        g_loss and the global normaliser of the masked cross-entropy."""
        if not self.exact:
            return
        ghead = self.buf["ghead"]
        if self.mesh_dp:
            p.add(self.L.mesh_allreduce_small, self._mesh.mesh, self._mesh.offset_of(ghead), 2)
            return
        p.host(lambda: self.comm.all_reduce_sum_(ghead), self.comm.capturable)

    def _build_g_plan(self, fast: bool = False, draw_masks: bool = True, bn_stacked: Optional[Dict[str, torch.Tensor]] = None,
                      observer_forward: bool = True, plan: Optional[Plan] = None) -> Plan:
        """bn_stacked (GroupedTrainer): the [slots, N] statistics of the group's stacked generator forwards, whose
        running-statistics updates were left to this plan: all slots' batches in order, the last one twice."""
        B, S, C, Z, b, G = self.B, self.S, self.C, self.Z, self.buf, self.Gp
        p = plan if plan is not None else Plan(b["ws"], b["ws2"], self.overlap, self.gemm_mode, self.wk)
        st = self.iter_state.data_ptr()
        if fast and draw_masks:
            t = b["gMKflat"]
            p.add(self.L.rng_dropout_mask, ptr(t), t.numel(), 0.2, (self.seed + 4 * self.GOLD) & (2**64 - 1), st, 0)
        # The generator parameters and z are unchanged since the D-step's forward, so the second
        # generator(z, c) of :240 reproduces the same activations; only its BN running-stat update is new.
        src = bn_stacked if bn_stacked is not None else b
        groups = bn_stacked["mean2"].shape[0] if bn_stacked is not None else 1
        bn_run = (ptr(src["mean2"]), ptr(src["m22"]), 1024,
                  ptr(self.Gbuf["model.3.running_mean"]), ptr(self.Gbuf["model.3.running_var"]), ptr(src["mean3"]), ptr(src["m23"]),
                  2048, ptr(self.Gbuf["model.6.running_mean"]), ptr(self.Gbuf["model.6.running_var"]), float(self.Bg), 0.1, groups,
                  2 if bn_stacked is not None else 1)
        # fused single-GPU G-step: the running statistics ride on the step's last launch (nothing reads them in between)
        end_merge = self.fuse_g and os.environ.get("WFD_G_END_MERGE", "1") != "0"          # (the switch: same-session A/B)
        if not end_merge:
            p.add(self.L.bn_running_update_layers, *bn_run)
        # side lane: critic pass on the fake rows + its input-gradient chain (slabs stay in the side workspace until
        # g_out_bwd reads them); main lane: the DF observer forward.  They meet at the masked CE.
        merged = self.g_merge and observer_forward
        if merged:
            g = self._g_front_merged(p)
        else:
            if observer_forward:
                p.fork()
            with p.side(stream=observer_forward):       # the slabs of g stay in the side workspace either way
                self._critic_forward(p, self.fake_rows, B, (b["gMK1"], b["gMK2"], b["gMK3"]), b["gH1"], b["gH2"],
                                     b["gH3"], b["gD1"], b["gD2"], b["gD3"], b["gV"], b["gDL3"], None, None)
                if not self.g_tail_fused:
                    p.add(self.L.g_head, ptr(b["gV"]), B, 1.0 / self.Bg, ptr(b["ghead"]))
                g = self._critic_chain(p, B, b["gDL3"], b["gDL2"], None, b["gDL1"], None, b["gD2"], b["gD1"], None)
            self._observer(p, forward=observer_forward)
        if B > 512:      # tall batch: dz in one fully parallel pass, its column sums (the bias gradient) by row chunks
            part = b.setdefault("gout_part", torch.zeros(((B + 127) // 128) * S, dtype=torch.float32, device=self.dev))
            p.add(self.L.g_out_bwd, g.ptr, g.ns, g.ss, g.ldp, B, S, -1.0 / self.Bg, ptr(b["DDF"]), self.ldl,
                  ptr(self.fake_rows), self.ldx, ptr(b["idx"]), ptr(b["GDZ4"]), self.lds4, 0)
            p.add(self.L.colsum_tall, ptr(b["GDZ4"]), self.lds4, B, S, 128, ptr(part), ptr(G.g["model.8.bias"]))
        else:
            p.add(self.L.g_out_bwd, g.ptr, g.ns, g.ss, g.ldp, B, S, -1.0 / self.Bg, ptr(b["DDF"]), self.ldl,
                  ptr(self.fake_rows), self.ldx, ptr(b["idx"]), ptr(b["GDZ4"]), self.lds4, ptr(G.g["model.8.bias"]))
        # generator backward: the activation-gradient chain on the main lane, each weight gradient on the side lane
        Gp = G.p
        # (weight gradient of layer k) and (activation gradient into layer k - 1) both read dz_k only: one launch each pair
        pr = self.pair
        if self.fuse_g:
            # single GPU: the activation gradient into layer k - 1 first (it reads W_k), then W_k's gradient with RMSprop in
            # its store
            s = p.gemm(NN, b["GDZ4"], self.lds4, Gp["model.8.weight"], 2048, B, 2048, S)
            self._bn_bwd(p, s, 3, "model.5", "model.6", 2048)
            s = p.gemm(NN, b["GDZ3"], 2048, Gp["model.5.weight"], 1024, B, 1024, 2048)
            self._bn_bwd(p, s, 2, "model.2", "model.3", 1024)
            p.gemm_mul(NN, b["GDZ2"], 1024, Gp["model.2.weight"], 512, B, 512, 1024, b["GH1"], 512, DM_RELU, None, b["GDZ1"], 512)
            if not end_merge:
                p.add(self.L.colsum, ptr(b["GDZ1"]), 512, B, 512, 0, ptr(G.g["model.0.bias"]), 0)
            # the activation-gradient chain is done (it read W_8, W_5, W_2): all four weight gradients, one launch
            g_problems = [
                (b["GDZ4"], self.lds4, b["GH3"], 2048, S, 2048, B, Gp["model.8.weight"], 2048, G.s["model.8.weight"]),
                (b["GDZ3"], 2048, b["GH2"], 1024, 2048, 1024, B, Gp["model.5.weight"], 1024, G.s["model.5.weight"]),
                (b["GDZ2"], 1024, b["GH1"], 512, 1024, 512, B, Gp["model.2.weight"], 512, G.s["model.2.weight"]),
                (b["GDZ1"], 512, b["ZIN"], self.ldz, 512, Z + C, B, Gp["model.0.weight"], G.ld["model.0.weight"],
                 G.s["model.0.weight"])]
            one = end_merge and os.environ.get("WFD_G_END_MERGE", "1") != "2"       # ("2": wfd_g_end as its own launch)
            if not one:
                p.gemm_rms_multi(g_problems, self.lr)
            if end_merge:
                # first layer's bias gradient + its step, the other vectors' steps, the BN running statistics: one launch
                desc = torch.tensor([[ptr(b["GDZ1"]), 512, 0, 0, 0, 0, B, 512, ptr(G.g["model.0.bias"]), 0, 0]],
                                    dtype=torch.int64, device=self.dev)
                # the column-sum block already applies model.0.bias's RMSprop step: take exactly its interval out of the
                # ranges (splitting a run where it sits inside one), so that no element is updated twice or not at all
                b0 = G.offsets["model.0.bias"]
                flat = [v for r in subtract_interval(G.vector_ranges(), b0, b0 + 512) for v in r]
                assert all(hi <= b0 or lo >= b0 + 512 for lo, hi in zip(flat[0::2], flat[1::2]))
                assert sum(hi - lo for lo, hi in zip(flat[0::2], flat[1::2])) + 512 == sum(hi - lo for lo, hi in G.vector_ranges())
                arr = (ctypes.c_long * len(flat))(*flat)
                p.keep += [desc, arr]
                st_arg = (st if fast else 0, self._rng_inc() if fast else 0, 0)
                # ... riding in the first workgroups of the weight-gradient launch: the end of the G-step is ONE launch
                if not (one and p.gemm_rms_multi(g_problems, self.lr, colsum=(ptr(desc), 1, 512 // 32, ptr(G.grad), ptr(G.flat),
                                                                               ptr(G.sq), G.numel) + st_arg
                                                 + (arr, len(flat) // 2) + bn_run)):
                    p.add(self.L.g_end, ptr(desc), 1, 512 // 32, ptr(G.grad), ptr(G.flat), ptr(G.sq), G.numel, arr, len(flat) // 2,
                          self.lr, 0.99, 1e-8, *st_arg, *bn_run)
            else:
                self._optimizer(p, G, True, st if fast else 0, self._rng_inc() if fast else 0, 0)
            return p
        if self.late_grads:
            # data parallel over the peer-to-peer exchange: the single-GPU shape -- the activation-gradient chain, then all four
            # weight gradients as one launch (stored), then the exchange launch
            s = p.gemm(NN, b["GDZ4"], self.lds4, Gp["model.8.weight"], 2048, B, 2048, S)
            self._bn_bwd(p, s, 3, "model.5", "model.6", 2048)
            s = p.gemm(NN, b["GDZ3"], 2048, Gp["model.5.weight"], 1024, B, 1024, 2048)
            self._bn_bwd(p, s, 2, "model.2", "model.3", 1024)
            p.gemm_mul(NN, b["GDZ2"], 1024, Gp["model.2.weight"], 512, B, 512, 1024, b["GH1"], 512, DM_RELU, None, b["GDZ1"], 512)
            desc = torch.tensor([[ptr(b["GDZ1"]), 512, 0, 0, 0, 0, B, 512, ptr(G.g["model.0.bias"]), 0, 0]],
                                dtype=torch.int64, device=self.dev)
            p.keep.append(desc)
            if not p.gemm_multi_grad([
                    (b["GDZ4"], self.lds4, b["GH3"], 2048, S, 2048, B, G.g["model.8.weight"], 2048),
                    (b["GDZ3"], 2048, b["GH2"], 1024, 2048, 1024, B, G.g["model.5.weight"], 1024),
                    (b["GDZ2"], 1024, b["GH1"], 512, 1024, 512, B, G.g["model.2.weight"], 512),
                    (b["GDZ1"], 512, b["ZIN"], self.ldz, 512, Z + C, B, G.g["model.0.weight"], G.ld["model.0.weight"])],
                    colsum=(ptr(desc), 1, 512 // 32)):
                p.add(self.L.colsum, ptr(b["GDZ1"]), 512, B, 512, 0, ptr(G.g["model.0.bias"]), 0)
            g_st = (st if fast else 0, self._rng_inc() if fast else 0, 0)
            self._grad_sync_hook(p, "G", g_st)
            self._optimizer(p, G, False, *g_st)
            return p
        s = p.gemm_pair((b["GDZ4"], self.lds4, b["GH3"], 2048, S, 2048, B, G.g["model.8.weight"], 2048),
                        (NN, b["GDZ4"], self.lds4, Gp["model.8.weight"], 2048, B, 2048, S), pair=pr)
        self._grad_bucket_hook(p, "G", 0)
        self._bn_bwd(p, s, 3, "model.5", "model.6", 2048)
        s = p.gemm_pair((b["GDZ3"], 2048, b["GH2"], 1024, 2048, 1024, B, G.g["model.5.weight"], 1024),
                        (NN, b["GDZ3"], 2048, Gp["model.5.weight"], 1024, B, 1024, 2048), pair=pr)
        self._grad_bucket_hook(p, "G", 1)
        self._bn_bwd(p, s, 2, "model.2", "model.3", 1024)
        p.gemm_pair_mul((b["GDZ2"], 1024, b["GH1"], 512, 1024, 512, B, G.g["model.2.weight"], 512),
                        (NN, b["GDZ2"], 1024, Gp["model.2.weight"], 512, B, 512, 1024), b["GH1"], 512, DM_RELU, b["GDZ1"], 512,
                        pair=pr)
        p.add(self.L.colsum, ptr(b["GDZ1"]), 512, B, 512, 0, ptr(G.g["model.0.bias"]), 0)
        p.gemm(TN, b["GDZ1"], 512, b["ZIN"], self.ldz, 512, Z + C, B, out=G.g["model.0.weight"],
               ldp=G.ld["model.0.weight"])
        if self.overlap:
            p.join()
        g_st = (st if fast else 0, self._rng_inc() if fast else 0, 0)
        self._grad_sync_hook(p, "G", g_st)
        self._optimizer(p, G, False, *g_st)
        return p

    def _bn_bwd(self, p: Plan, s: Slabs, i: int, lin: str, bn: str, n: int):
        b, G = self.buf, self.Gp
        args = (s.ptr, s.ns, s.ss, s.ldp, self.B, n, ptr(b[f"GH{i}"]), ptr(b[f"GXH{i}"]), n,
                ptr(G.p[f"{bn}.weight"]), ptr(b[f"istd{i}"]), float(self.Bg), ptr(b[f"bns{i}"]), ptr(b[f"GDZ{i}"]), n,
                ptr(G.g[f"{bn}.weight"]), ptr(G.g[f"{bn}.bias"]), ptr(G.g[f"{lin}.bias"]))
        if not self.exact:
            if self.B > 512:     # tall batch: row chunks in parallel, partial sums combined in chunk order
                part = b.setdefault(f"bnpart{i}", torch.zeros(2 * ((self.B + 127) // 128) * n, dtype=torch.float32, device=self.dev))
                p.add(self.L.bn_bwd_tall, *args, ptr(part))
            else:
                p.add(self.L.bn_bwd, *args, 3)
            return
        # exact mode: local [sum dy, sum dy*xhat] -> all-reduce -> dz with the global sums.  dgamma / dbeta come out
        # as the GLOBAL sums on every rank, so they are pre-divided by the world size before the gradient buffer
        # is summed over ranks.
        sums, dgam, dbet, w = b[f"bns{i}"], G.g[f"{bn}.weight"], G.g[f"{bn}.bias"], float(self.world)
        p.add(self.L.bn_bwd, *args, 1)
        if self.mesh_dp:
            p.add(self.L.mesh_allreduce_small, self._mesh.mesh, self._mesh.offset_of(sums), sums.numel())
        else:
            p.host(lambda: self.comm.all_reduce_sum_(sums), self.comm.capturable)
        p.add(self.L.bn_bwd, *args, 2)
        if self.world > 1:
            # (plain device work on the current stream: inside the graph whenever no host-staged collective splits the plan)
            p.host(lambda: (dgam.div_(w), dbet.div_(w)), self.comm.capturable or self.mesh_dp)

    # ------------------------------------------------------------------------------------------------------
    # inputs
    def set_batch(self, traces: torch.Tensor, labels: torch.Tensor, z: torch.Tensor):
        """Explicit batch (parity mode): traces (B,S) scaled features, labels (B,) ints, z (B,Z)."""
        B, S, C, Z, b = self.B, self.S, self.C, self.Z, self.buf
        labels = labels.to(self.dev)
        onehot = torch.zeros(B, C, dtype=torch.float32, device=self.dev)
        onehot[torch.arange(B, device=self.dev), labels.long()] = 1.0
        self.real_rows[:, :S].copy_(traces.to(self.dev, torch.float32))
        self.real_rows[:, S:S + C].copy_(onehot)
        self.fake_rows[:, S:S + C].copy_(onehot)
        b["ZIN"][:, :Z].copy_(z.to(self.dev, torch.float32))
        b["ZIN"][:, Z:Z + C].copy_(onehot)
        b["labels"].copy_(labels.to(torch.int32))

    def set_masks(self, d_masks: Optional[Sequence[torch.Tensor]] = None,
                  g_masks: Optional[Sequence[torch.Tensor]] = None):
        """Explicit dropout keep-masks (values {0, 1.25}).  d_masks: 3 tensors of M rows (real rows first, then
        fake); g_masks: 3 tensors of B rows for the G-step's critic pass."""
        if d_masks is not None:
            for i, m in enumerate(d_masks):
                self.buf[f"MK{i + 1}"].copy_(m.to(self.dev, torch.float32))
        if g_masks is not None:
            for i, m in enumerate(g_masks):
                self.buf[f"gMK{i + 1}"].copy_(m.to(self.dev, torch.float32))

    def set_gp_alpha(self, alpha: torch.Tensor):
        """Explicit WGAN-GP interpolation weights (B,1) or (B,) (parity mode)."""
        self.buf["GPA"].copy_(alpha.reshape(-1).to(self.dev, torch.float32))

    def draw_masks(self, g_step: bool):
        """Fast mode: Philox keep-masks on device (one launch per mask tensor)."""
        st = torch.cuda.current_stream().cuda_stream
        names = ["MKflat"] + (["gMKflat"] if g_step else [])
        for k, nme in enumerate(names):
            t = self.buf[nme]
            self.L.rng_dropout_mask(t.data_ptr(), t.numel(), 0.2, (self.seed + 31 * (k + 1)) & (2**64 - 1), 0,
                                    self.rng_offset, st)
        self.rng_offset += (self.buf["MKflat"].numel() + 3) // 4

    def draw_z(self):
        st = torch.cuda.current_stream().cuda_stream
        self.L.rng_normal_rows(self.buf["ZIN"].data_ptr(), self.ldz, self.B, self.Z, self.seed ^ 0x5A5A5A5A, 0,
                               self.rng_offset, st)
        self.rng_offset += (self.B * self.Z + 3) // 4

    # ------------------------------------------------------------------------------------------------------
    def d_step(self):
        self.plan_d.run()
        self.nbt["model.3"] += 1
        self.nbt["model.6"] += 1

    def g_step(self):
        self.plan_g.run()
        self.nbt["model.3"] += 1
        self.nbt["model.6"] += 1

    def check_length_index(self):
        """Host sync: raise what the reference raises when the sigmoid saturates (src/model.py:44)."""
        if int(self.buf["err"].item()) != 0:
            raise IndexError("index out of bounds in the generator's burst-length mask (model.py:44 semantics)")

    def d_stats(self) -> Dict[str, float]:
        """Exact mode: the device scalars are per-rank partial sums of global-batch means -> summed over ranks here
        (a collective: every rank must call it)."""
        t = self.buf["dstats"]
        if self.exact:
            t = self.comm.all_reduce_sum_(t.clone())
        v = t.tolist()
        return dict(d_loss=v[0], penalty=v[1], w_dist=v[2], mean_rv=v[3], mean_fv=v[4])

    def g_stats(self) -> Dict[str, float]:
        v = self._g_scalars(self.buf["gstats"])
        return dict(g_loss=v[0], f_loss=v[1], g_combined=v[2], n_sel=int(v[3]))

    def _g_scalars(self, t: torch.Tensor):
        """[g_loss, f_loss, combined, count].  Exact mode: g_loss and the count are already global on every rank,
        f_loss is a per-rank partial sum (collective readout)."""
        if not self.exact:
            return t.tolist()
        f = self.comm.all_reduce_sum_(t[1:2].clone())
        v = t.tolist()
        v[1] = float(f.item())
        v[2] = v[0] + self.alpha * v[1]
        return v

    def epoch_sums(self):
        """(dacc, gacc): the epoch's running sums of [d_loss, w_dist] and [g_loss, f_loss, combined, #G-steps] (one
        host sync per epoch; a collective in exact mode)."""
        d, g = self.buf["dacc"], self.buf["gacc"]
        if self.exact:
            return self.comm.all_reduce_sum_(d.clone()).tolist(), self._g_scalars(g)
        if self.world > 1:          # local-batch losses: report the average over ranks
            d = self.comm.all_reduce_sum_(d.clone()) / self.world
            g = self.comm.all_reduce_sum_(g.clone()) / self.world
        return d.tolist(), g.tolist()

    def generator_state(self) -> Dict[str, torch.Tensor]:
        sd = {}
        for key in ("model.0", "model.2", "model.3", "model.5", "model.6", "model.8"):
            sd[f"{key}.weight"] = self.Gp.p[f"{key}.weight"].detach().clone(memory_format=torch.contiguous_format)
            sd[f"{key}.bias"] = self.Gp.p[f"{key}.bias"].detach().clone()
            if key in ("model.3", "model.6"):
                sd[f"{key}.running_mean"] = self.Gbuf[f"{key}.running_mean"].clone()
                sd[f"{key}.running_var"] = self.Gbuf[f"{key}.running_var"].clone()
                sd[f"{key}.num_batches_tracked"] = torch.tensor(self.nbt[key], dtype=torch.int64)
        return sd

    def discriminator_state(self) -> Dict[str, torch.Tensor]:
        return self.Dp.state()

    # ------------------------------------------------------------------------------------------------------
    # fast mode: dataset resident in HBM, batch assembly + z + dropout masks drawn on device, HIP-graph replay
    def attach_dataset(self, X: torch.Tensor, y: torch.Tensor, shuffle_seed: int = 0):
        """X (N,S) scaled fp32 features and y (N,) int labels, both moved to HBM once (100000 x 1401 fp32 = 560 MB)."""
        self.data_X = X.to(self.dev, torch.float32).contiguous()
        self.data_y = y.to(self.dev, torch.int32).contiguous()
        self.N = self.data_X.shape[0]
        self.iters_per_epoch = self.N // self.cursor_step    # full (global) batches; the trainer handles a ragged tail
        self.auto_reshuffle = True                       # iteration() starts a new permutation when one is used up
        self._pos = 0
        self.iter_state.zero_()
        self.perm_gen = torch.Generator(device=self.dev)
        self.perm_gen.manual_seed(shuffle_seed)
        self.perm = torch.randperm(self.N, generator=self.perm_gen, device=self.dev)
        self.fast_d, self.fast_g = self._build_fast_plans()

    def reshuffle(self):
        """New epoch permutation (DataLoader(shuffle=True), src/mlp_df_wgan_train.py:142-147); cursor back to 0."""
        self.perm.copy_(torch.randperm(self.N, generator=self.perm_gen, device=self.dev))
        self.iter_state[1] = 0
        self._pos = 0

    def _build_fast_plans(self):
        return self._build_d_plan(fast=True), self._build_g_plan(fast=True)

    def _snapshot(self):
        snap = dict(G=self.Gp.flat.clone(), Gsq=self.Gp.sq.clone(), D=self.Dp.flat.clone(), Dsq=self.Dp.sq.clone(),
                    bn={k: v.clone() for k, v in self.Gbuf.items()}, st=self.iter_state.clone(),
                    dacc=self.buf["dacc"].clone(), gacc=self.buf["gacc"].clone(), nbt=dict(self.nbt))
        return snap

    def _restore(self, snap):
        self.Gp.flat.copy_(snap["G"]); self.Gp.sq.copy_(snap["Gsq"])
        self.Dp.flat.copy_(snap["D"]); self.Dp.sq.copy_(snap["Dsq"])
        for k, v in snap["bn"].items():
            self.Gbuf[k].copy_(v)
        self.iter_state.copy_(snap["st"])
        self.buf["dacc"].copy_(snap["dacc"]); self.buf["gacc"].copy_(snap["gacc"])
        self.nbt = dict(snap["nbt"])

    def capture_graphs(self):
        """Capture the D-step and the G-step as HIP graphs (one launch each per replay).  A warm-up run loads the
        kernels first; training state is restored afterwards so the capture does not consume an iteration."""
        snap = self._snapshot()
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.fast_d.run()
            self.fast_g.run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph_d = self._capture_stages(self.fast_d)
        self.graph_g = self._capture_stages(self.fast_g)
        # one graph for a whole group of n_critic iterations (D G D ... D), when no host action splits the plans:
        # saves the graph-launch boundaries inside the group (iteration_group)
        self.graph_group = None
        if len(self.graph_d) == 1 and len(self.graph_g) == 1:
            g = torch.cuda.CUDAGraph()
            from .parallel import graph_capture_kwargs
            kw = graph_capture_kwargs()
            try:
                with torch.cuda.graph(g, **kw):
                    self.fast_d.run()
                    self.fast_g.run()
                    for _ in range(self.n_critic - 1):
                        self.fast_d.run()
                self.graph_group = g
            except RuntimeError:
                if not kw:
                    raise
                for hook in self.fast_d.abort_hooks + self.fast_g.abort_hooks:
                    hook()
                torch.cuda.synchronize()       # collectives refused this capture: single-step graphs only
        torch.cuda.synchronize()
        self._restore(snap)

    @staticmethod
    def _capture_stages(plan: Plan, split_all: bool = False):
        """One HIP graph per run of launches; host actions that cannot be captured stay eager between the replays.  If
        capturing a run WITH its collectives fails (a collective backend that refuses stream capture), the plan is split
        at every host action instead and the collectives run eagerly between the stages."""
        out = []
        for st in plan.stages(split_all):
            if isinstance(st, Plan):
                g = torch.cuda.CUDAGraph()
                # collectives captured with the launches (RCCL): the backend's watchdog thread makes HIP calls of its own
                # during the capture, which only the thread-local capture mode tolerates
                from .parallel import graph_capture_kwargs
                coll = st.has_collectives
                kw = graph_capture_kwargs()             # (thread-local whenever a process group exists, collectives or not)
                try:
                    with torch.cuda.graph(g, **kw):
                        st.run()
                except RuntimeError:
                    if not coll:
                        raise
                    import warnings
                    warnings.warn("capturing the data-parallel collectives inside the step graph failed; "
                                  "falling back to eager collectives between graph stages")
                    # Work handles created under the abandoned capture (a start() that ran before the failure) must not
                    # be waited on by the eager retry's first finish(): drop them, then drain the device
                    for hook in plan.abort_hooks:
                        hook()
                    torch.cuda.synchronize()
                    return TrainEngine._capture_stages(plan, split_all=True)
                g.keep = plan.keep          # the captured launches point into the plan's descriptor tables
                out.append(g.replay)
            else:
                out.append(st)
        return out

    def iteration(self, i: int):
        """One pass of the reference loop body in fast mode: D-step, plus a G-step when i % n_critic == 0."""
        do_g = (i % self.n_critic == 0)
        if self.graph_d is not None:
            for st in self.graph_d:
                st()
            if do_g:
                for st in self.graph_g:
                    st()
        else:
            self.fast_d.run()
            if do_g:
                self.fast_g.run()
        k = 2 if do_g else 1
        self.nbt["model.3"] += k
        self.nbt["model.6"] += k
        self._pos += 1
        if self.auto_reshuffle and self._pos >= self.iters_per_epoch:
            self.reshuffle()
        return do_g

    def iteration_group(self, i: int) -> int:
        """Iterations i .. i + n_critic - 1 (i a multiple of n_critic: G-step after the first D-step) as ONE graph
        replay when available and the group stays inside the epoch permutation; otherwise n_critic single iterations.
        For throughput runs: the generated rows of the G-step are overwritten by the following D-steps, so a caller
        that collects them (the trainer) uses iteration().  Returns the number of iterations done."""
        n = self.n_critic
        if self.graph_group is None or i % n != 0 or (self.auto_reshuffle and self._pos + n > self.iters_per_epoch):
            for k in range(n):
                self.iteration(i + k)
            return n
        self.graph_group.replay()
        self.nbt["model.3"] += n + 1
        self.nbt["model.6"] += n + 1
        self._pos += n
        if self.auto_reshuffle and self._pos >= self.iters_per_epoch:
            self.reshuffle()
        return n


class PipelinedTrainer:
    """Fast-mode driver that runs the generator forward of iteration i+1 on a second HIP stream while the critic
    update of iteration i executes (software pipelining across iterations).

    The head of a D-step -- batch gather, z / dropout-mask draws, generator forward -- reads only the generator's
    parameters, which change in G-steps alone.  So whenever iteration i has no G-step (two iterations out of three
    with n_critic = 3), the head of iteration i+1 is independent of everything iteration i computes.  Two buffer sets
    (two TrainEngines on shared parameters: "slots", used by even / odd iterations) make that overlap legal; each
    part is its own HIP graph (no fork/join inside a graph), ordered by events:

        stream main :  rest(i) [G-step(i)]            rest(i+1) ...
        stream side :         head(i+1)  --event-->

    Arithmetic and the order of every state update (BatchNorm running statistics, RMSprop) are those of the
    sequential engine (tests/test_gpu_pipeline.py replays every pipelined iteration sequentially: bit-identical);
    only the Philox streams are partitioned differently.

    MEASURED (round 1, one MI355X, config 2): 2231 it/s against 2309 it/s for the sequential engine -- the extra graph
    launches and cross-stream event waits cost more than the overlap returns (host enqueue 0.20 ms/iteration, not
    the limit).  Kept as an option (bench.py --pipeline 1), off by default."""

    def __init__(self, eng: "TrainEngine", graphs: bool = True):
        if eng.exact:
            raise ValueError("the pipelined driver does not cover the SyncBN mode (collectives inside the generator forward)")
        if not hasattr(eng, "data_X"):
            raise RuntimeError("attach_dataset() first")
        self.e = [eng, TrainEngine(eng.S, eng.C, eng.Z, eng.B, eng.scaler_min, eng.scaler_max, {}, {}, {}, lr=eng.lr,
                                   alpha=eng.alpha, n_critic=eng.n_critic, penalty=eng.penalty, lambda_gp=eng.lambda_gp,
                                   device=eng.dev, seed=eng.seed + 0x51ED, process_group=eng.pg, world_size=eng.world,
                                   share=eng, rank=eng.rank, operands=eng.operands, wk=eng.wk, fuse_opt=eng.fuse_opt, pair=eng.pair, force_sync=eng.force_sync)]
        other = self.e[1]
        for name in ("data_X", "data_y", "N", "perm", "perm_gen", "iters_per_epoch"):
            setattr(other, name, getattr(eng, name))
        other._gsync = eng._gsync
        self.main = torch.cuda.current_stream(eng.dev)
        self.side = torch.cuda.Stream(device=eng.dev)
        self.gf = [e._build_d_plan(fast=True, part="gf") for e in self.e]
        self.rest = [e._build_d_plan(fast=True, part="rest") for e in self.e]
        self.gs = [e._build_g_plan(fast=True) for e in self.e]
        self.ready = [None, None]          # event: head of the slot's next iteration is done
        self.prefetched = [False, False]
        self._pos = 0
        self.auto_reshuffle = True
        eng.gf_state.zero_()
        self.run_gf, self.run_rest, self.run_gs = ([pl.run for pl in plans] for plans in (self.gf, self.rest, self.gs))
        if graphs:
            self._capture()

    def _capture(self):
        eng = self.e[0]
        snap = eng._snapshot()
        gfs = eng.gf_state.clone()
        warm = torch.cuda.Stream(device=eng.dev)
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            for k in range(2):
                self.gf[k].run(); self.rest[k].run(); self.gs[k].run()
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()

        def runner(plan):
            stages = TrainEngine._capture_stages(plan)
            return lambda: [st() for st in stages]
        self.run_gf = [runner(pl) for pl in self.gf]
        self.run_rest = [runner(pl) for pl in self.rest]
        self.run_gs = [runner(pl) for pl in self.gs]
        torch.cuda.synchronize()
        eng._restore(snap)
        eng.gf_state.copy_(gfs)

    def slot(self, i: int) -> "TrainEngine":
        """The engine whose buffers hold iteration i's batch, generated rows and statistics."""
        return self.e[i & 1]

    def iteration(self, i: int, prefetch_next: bool = True) -> bool:
        """One pass of the reference loop body (D-step, plus a G-step when i % n_critic == 0).  ``prefetch_next``:
        iteration i+1 follows in the same epoch permutation (its head may start now)."""
        k = i & 1
        eng = self.e[0]
        do_g = (i % eng.n_critic == 0)
        main = torch.cuda.current_stream(eng.dev)
        if self.prefetched[k]:
            main.wait_event(self.ready[k])
            self.prefetched[k] = False
        else:
            self.run_gf[k]()                                   # pipeline start-up (and after every G-step / epoch end)
        last = self.auto_reshuffle and self._pos + 1 >= eng.iters_per_epoch
        ahead = prefetch_next and not do_g and not last
        if ahead:
            # slot k^1 is free: everything that read it (iteration i-1, the caller's copies) is already on `main`
            free = torch.cuda.Event()
            free.record(main)
            self.side.wait_event(free)
            with torch.cuda.stream(self.side):
                self.run_gf[k ^ 1]()
                ev = torch.cuda.Event()
                ev.record(self.side)
            self.ready[k ^ 1], self.prefetched[k ^ 1] = ev, True
        self.run_rest[k]()
        if do_g:
            self.run_gs[k]()
        n = 2 if do_g else 1
        eng.nbt["model.3"] += n
        eng.nbt["model.6"] += n
        self._pos += 1
        if last:
            self.reshuffle()
        return do_g

    def reshuffle(self):
        """New epoch permutation; the cursor of the head plans goes back to 0 (no head is in flight here)."""
        eng = self.e[0]
        assert not any(self.prefetched)
        eng.perm.copy_(torch.randperm(eng.N, generator=eng.perm_gen, device=eng.dev))
        eng.gf_state[1] = 0
        self._pos = 0

    def check_length_index(self):
        for e in self.e:
            e.check_length_index()


STACKED_NAMES = ("XC", "ZIN", "labels", "idx", "GH1", "GXH2", "GH2", "GXH3", "GH3", "mean2", "m22", "istd2", "mean3",
                 "m23", "istd3")


class GroupedTrainer:
    """Fast-mode driver that runs the generator forwards of the n_critic iterations that share generator parameters
    as ONE stacked pass.

    The generator changes only in G-steps (every n_critic-th iteration, after that iteration's D-step).  With
    n_critic = 3 the reference's schedule is  D0 G0 | D1 D2 D3 G3 | D4 D5 D6 G6 | ...: inside a group the three
    D-steps see the same generator, and their heads (batch gather, z draw, generator forward: nothing a critic update
    writes) depend on nothing but it.  So a group runs as

        head(3 batches stacked: one gather, one z draw, four GEMMs over 3*B rows, BatchNorm per 64-row batch,
             running statistics updated once per batch in order)
        rest(i) rest(i+1) rest(i+2)  G-step(i+2)

    i.e. 12 launches for the three heads instead of 33, on GEMMs three times as tall (the B = 64
    step is launch/latency-bound).  Each iteration keeps its own buffers ("slot" engines on shared parameters whose
    head buffers are views of the stacked tensors), so the critic update, the penalty and the G-step are the
    sequential engine's plans unchanged; arithmetic and the order of every state update are the reference's."""

    def __init__(self, eng: "TrainEngine", graphs: bool = True, observer_ahead: bool = False):
        """observer_ahead: the frozen DF observer's forward on the last slot's generated rows depends on nothing the group's
        critic updates write, so it is issued on a side lane (second stream / graph branch) right after the stacked heads
        and joins the main lane at the G-step's masked cross-entropy.  Bit-identical to the single-lane group
        (tests/test_gpu_fused_opt.py).  MEASURED (round 2, one MI355X, config 2, same session): 3010 vs 3075 it/s -- and moving
        the observer's backward to the branch as well (unit row weights, applied afterwards) gave the same 3015 vs 3080: a
        second branch in a HIP graph does not overlap with the first on this stack, it only adds the fork / join.  Off by
        default; kept as the measured record (bench.py --observer-ahead 1)."""
        self.observer_ahead = bool(observer_ahead)
        if eng.exact:
            raise ValueError("the grouped driver does not cover the SyncBN mode (collectives inside the generator forward)")
        if not hasattr(eng, "data_X"):
            raise RuntimeError("attach_dataset() first")
        if eng.B > 128:
            raise ValueError("stacked BatchNorm kernel covers batches up to 128 rows")
        n = self.n = eng.n_critic
        dev = eng.dev
        self.stacked = {k: torch.zeros((n,) + tuple(eng.buf[k].shape), dtype=eng.buf[k].dtype, device=dev)
                        for k in STACKED_NAMES}
        self.e = [TrainEngine(eng.S, eng.C, eng.Z, eng.B, eng.scaler_min, eng.scaler_max, {}, {}, {}, lr=eng.lr,
                              alpha=eng.alpha, n_critic=n, penalty=eng.penalty, lambda_gp=eng.lambda_gp, device=dev,
                              seed=eng.seed + 0x9E37 * (k + 1), process_group=eng.pg, world_size=eng.world, share=eng,
                              rank=eng.rank, stacked=self.stacked, slot=k, operands=eng.operands, wk=eng.wk, fuse_opt=eng.fuse_opt, pair=eng.pair,
                              force_sync=eng.force_sync) for k in range(n)]
        self.base = eng
        for e in self.e:
            for name in ("data_X", "data_y", "N", "perm", "perm_gen", "iters_per_epoch"):
                setattr(e, name, getattr(eng, name))
        eng.gf_state.zero_()
        self.ws = torch.empty(max(16 * n * eng.B * 2048, 1 << 20), dtype=torch.float32, device=dev)
        self.merged_masks = n + 1 <= 8          # one launch draws the dropout masks of the whole group (8 buffers at most)
        self.head = self._build_head()
        self.rest = [e._build_d_plan(fast=True, part="rest_draw", draw_masks=not self.merged_masks) for e in self.e]
        # the running-statistics updates of the three stacked forwards ride on the G-step's own update launch
        last = self.e[n - 1]
        if self.observer_ahead:
            # lanes: main = critic updates + G-step, side = observer forward (own slab workspace)
            # (own slab workspace for the observer lane: the G-step's critic pass keeps its input-gradient slabs in the engine's
            # side workspace, and may run while the observer still does)
            self.ws_obs = torch.empty(max(eng.B * eng.S * 64, 1 << 20), dtype=torch.float32, device=dev)
            mk = lambda: Plan(last.buf["ws"], last.buf["ws2"], True, eng.gemm_mode, eng.wk)
            self.obs = Plan(last.buf["ws"], self.ws_obs, True, eng.gemm_mode, eng.wk)
            self.obs.fork()
            with self.obs.side():
                last._observer_forward(self.obs)
            self.gstep = last._build_g_plan(fast=True, draw_masks=not self.merged_masks, bn_stacked=self.stacked,
                                            observer_forward=False, plan=mk())
            self.group = Plan(self.ws, last.buf["ws2"], True, eng.gemm_mode, eng.wk)
        else:
            self.obs = None
            self.gstep = last._build_g_plan(fast=True, draw_masks=not self.merged_masks, bn_stacked=self.stacked)
            self.group = Plan(self.ws, gemm_mode=eng.gemm_mode, wk=eng.wk)
        self.group.extend(self.head)
        if self.obs is not None:
            self.group.extend(self.obs)
        for r in self.rest:
            self.group.extend(r)
        self.group.extend(self.gstep)
        self._pos = 0
        self._phase = 0                 # iterations of the current group already run by run_iterations()
        self.part_plans = self.part_stages = None
        self.auto_reshuffle = True
        self.stages = None
        if graphs:
            self._capture()

    def _build_head(self) -> Plan:
        eng, n, L = self.base, self.n, self.base.L
        B, S, C, Z, G, st = eng.B, eng.S, eng.C, eng.Z, eng.Gp.p, self.stacked
        M = n * B
        p = Plan(self.ws, gemm_mode=eng.gemm_mode, wk=eng.wk)
        # the dropout keep-masks of the group's n critic passes and of its G-step in one launch: the same numbers the
        # iterations would draw themselves (iteration k starts k optimizer steps, i.e. k counter increments, after this
        # point; the G-step after all n)
        if self.merged_masks:
            inc, M64 = eng._rng_inc(), 2**64 - 1
            slots = []
            for k, e in enumerate(self.e):
                t = e.buf["MKflat"]
                slots += [t.data_ptr(), t.numel(), (e.seed + eng.GOLD) & M64, k * inc]
            ge = self.e[n - 1]
            slots += [ge.buf["gMKflat"].data_ptr(), ge.buf["gMKflat"].numel(), (ge.seed + 4 * eng.GOLD) & M64, n * inc]
            self._mask_slots = (ctypes.c_ulonglong * len(slots))(*slots)
        state = eng.gf_state.data_ptr()
        xc = st["XC"]                                   # [n][2*Mslot][ldx]
        slot_stride = xc.stride(0)
        real0, fake0 = self.e[0].real_rows, self.e[0].fake_rows
        # z has its own Philox key: the single-iteration plans (fast_d, key seed + 7 GOLD) count from 0 as well, and the trainer
        # runs iteration 0 of an epoch on them -- with a shared key iteration 1 would repeat iteration 0's noise
        zseed = (eng.seed + 13 * eng.GOLD) & (2**64 - 1)
        fold = self.merged_masks and os.environ.get("WFD_HEAD_MERGE", "1") != "0"       # (the switch: same-session A/B)
        if fold:
            # masks + gathered batches + z: three independent draws, one launch
            p.add(L.group_inputs, self._mask_slots, len(slots) // 4, 0.2, eng.iter_state.data_ptr(),
                  ptr(eng.data_X), S, ptr(eng.data_y), ptr(eng.perm), state, eng.N, eng.N, eng.row_offset,
                  B, n, eng.cursor_step, S, C, ptr(real0), ptr(fake0), eng.ldx, slot_stride, ptr(st["ZIN"]), eng.ldz, Z,
                  ptr(st["labels"]), zseed)
        else:
            if self.merged_masks:
                p.add(L.rng_dropout_mask_slots, self._mask_slots, len(slots) // 4, 0.2, eng.iter_state.data_ptr())
            p.add(L.gather_batch_slots, ptr(eng.data_X), S, ptr(eng.data_y), ptr(eng.perm), state, eng.N, eng.N, eng.row_offset,
                  B, n, eng.cursor_step, S, C, ptr(real0), ptr(fake0), eng.ldx, slot_stride, ptr(st["ZIN"]), eng.ldz, Z,
                  ptr(st["labels"]))
            p.add(L.rng_normal_rows, ptr(st["ZIN"]), eng.ldz, M, Z, zseed, state, 0)
        p.gemm_affine(NT, st["ZIN"], eng.ldz, G["model.0.weight"], eng.Gp.ld["model.0.weight"], M, 512, Z + C, None,
                      G["model.0.bias"], ACT_RELU, st["GH1"], 512)
        prev, kprev = st["GH1"], 512
        for i, lin, bn, width in ((2, "model.2", "model.3", 1024), (3, "model.5", "model.6", 2048)):
            s = p.gemm(NT, prev, kprev, G[f"{lin}.weight"], kprev, M, width, kprev)
            p.add(L.bn_train_relu_grouped, s.ptr, s.ns, s.ss, s.ldp, B, n, width, ptr(G[f"{lin}.bias"]), 0.8,
                  ptr(G[f"{bn}.weight"]), ptr(G[f"{bn}.bias"]), ptr(st[f"GXH{i}"]), ptr(st[f"GH{i}"]), width,
                  ptr(st[f"mean{i}"]), ptr(st[f"m2{i}"]), ptr(st[f"istd{i}"]), 0, 0, 0.1)      # running statistics: see gstep
            prev, kprev = st[f"GH{i}"], width
        s = p.gemm(NT, prev, 2048, G["model.8.weight"], 2048, M, S, 2048)
        if os.environ.get("WFD_HEAD_MERGE", "1") != "0":
            # the last launch of the head also advances the z counter and the permutation cursor
            p.add(L.ep_sigmoid_lenmask_slots_adv, s.ptr, s.ns, s.ss, s.ldp, M, S, ptr(G["model.8.bias"]), eng.range_f, eng.smin_f,
                  ptr(fake0), eng.ldx, B, slot_stride, ptr(st["idx"]), ptr(eng.buf["err"]), state, (M * Z + 3) // 4,
                  n * eng.cursor_step)
        else:
            p.add(L.ep_sigmoid_lenmask_slots, s.ptr, s.ns, s.ss, s.ldp, M, S, ptr(G["model.8.bias"]), eng.range_f, eng.smin_f,
                  ptr(fake0), eng.ldx, B, slot_stride, ptr(st["idx"]), ptr(eng.buf["err"]))
            p.add(L.iter_advance, state, (M * Z + 3) // 4, n * eng.cursor_step)
        return p

    def _capture(self):
        eng = self.base
        snap, gfs = eng._snapshot(), eng.gf_state.clone()
        warm = torch.cuda.Stream(device=eng.dev)
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            self.group.run()
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        self.stages = TrainEngine._capture_stages(self.group)
        torch.cuda.synchronize()
        eng._restore(snap)
        eng.gf_state.copy_(gfs)

    def slot(self, k: int) -> "TrainEngine":
        """Buffers of the k-th iteration of the last group (k = n_critic - 1: the one with the G-step)."""
        return self.e[k]

    def run_group(self) -> int:
        """n_critic iterations: D D ... D G (the G-step belongs to the last one).  Returns n_critic."""
        eng, n = self.base, self.n
        if self._phase:
            raise RuntimeError("run_group() inside a group that run_iterations() has started; finish it with run_iterations()")
        if self.auto_reshuffle and self._pos + n > eng.iters_per_epoch:
            self.reshuffle()                 # the remaining < n_critic batches of the permutation are skipped
        if self.stages is not None:
            for st in self.stages:
                st()
        else:
            self.group.run()
        eng.nbt["model.3"] += n + 1
        eng.nbt["model.6"] += n + 1
        self._pos += n
        return n

    def _build_parts(self):
        """The group cut into one piece per iteration -- [head + D-step 0], [D-step 1], ..., [D-step n-1 + G-step] -- each
        captured as a graph of its own: what run_iterations() replays when a caller stops or starts inside a group."""
        eng = self.base
        if self.obs is not None:
            raise RuntimeError("run_iterations() is built for the single-lane group (observer_ahead=0)")
        plans = []
        for k in range(self.n):
            p = Plan(self.ws, gemm_mode=eng.gemm_mode, wk=eng.wk)
            if k == 0:
                p.extend(self.head)
            p.extend(self.rest[k])
            if k == self.n - 1:
                p.extend(self.gstep)
            plans.append(p)
        self.part_plans = plans
        self.part_stages = [TrainEngine._capture_stages(p) for p in plans] if self.stages is not None else None
        torch.cuda.synchronize()

    def run_iterations(self, k: int) -> int:
        """k consecutive iterations of the schedule D D ... D G, continuing where the previous call stopped: whole groups
        go out as the group graph, a call that starts or stops inside a group replays that group's per-iteration pieces
        (same launches in the same order, so the state after any number of iterations does not depend on how the calls
        cut them)."""
        eng, n, done = self.base, self.n, 0
        while done < k:
            if self._phase == 0 and k - done >= n:
                self.run_group()
                done += n
                continue
            if self.part_plans is None:
                self._build_parts()
            if self._phase == 0 and self.auto_reshuffle and self._pos + n > eng.iters_per_epoch:
                self.reshuffle()
            if self.part_stages is not None:
                for st in self.part_stages[self._phase]:
                    st()
            else:
                self.part_plans[self._phase].run()
            self._phase += 1
            done += 1
            if self._phase == n:            # the group is complete: the bookkeeping run_group() does
                self._phase = 0
                eng.nbt["model.3"] += n + 1
                eng.nbt["model.6"] += n + 1
                self._pos += n
        return k

    def reshuffle(self):
        eng = self.base
        eng.perm.copy_(torch.randperm(eng.N, generator=eng.perm_gen, device=eng.dev))
        eng.gf_state[1] = 0
        self._pos = 0

    def check_length_index(self):
        self.base.check_length_index()
