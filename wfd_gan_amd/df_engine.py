"""Training step of the DF observer on the MI355X kernels: the loop body of src/train_df.py:121-131 (DF in train mode,
CrossEntropyLoss, Adamax) as one launch plan over preallocated HBM buffers, in closed form (no autograd graph).

Layout: activations are channels-last [B][L][C]; conv weights live in the packed forward layout [Cout][t*Cin+ci]
(tap-major) that the implicit-GEMM conv reads, layer5.0.weight with its input index permuted to (l, c); the
gradients come out in the same layouts, so Adamax (element-wise) runs on one flat buffer.  ``state_dict()`` /
``load`` convert to and from the reference's layouts.

Per conv block: conv (MFMA implicit GEMM) + bias -> Z; BatchNorm over all B*L rows of a channel (two-stage
deterministic statistics, running stats updated) + ELU/ReLU -> Y; [zero-pad max-pool 8 + dropout(0.1)].  Backward walks
the same stack: pool routing * act' -> BatchNorm backward -> bias / weight gradients (weight gradient = TN GEMM over the
im2col view, split over B*L) -> input gradient (conv with tap-flipped transposed weights, dropout mask or act' fused
into its epilogue).  The Linear + BatchNorm1d + ReLU + Dropout head reuses the generator's BatchNorm kernels.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from ._lib import lib
from .engine import DF_CONVS, FlatParams, ObserverDF, observer_buffers, observer_forward
from .plan import ACT_ELU, ACT_NONE, DM_ELU, DM_MUL, DM_RELU, NN, NT, TN, Plan, ptr, rup

DROPOUT_P = (0.1, 0.1, 0.1, 0.1, 0.7, 0.5)       # src/model.py:152,164,176,183,190,194
BN_EPS, BN_MOM = 1e-5, 0.1


def _df_specs(length: int, C: int):
    """(state_dict key, packed shape) in nn.Module.parameters() order."""
    Ls = [length]
    for _ in range(4):
        Ls.append((Ls[-1] + 7) // 8)
    specs = []
    for conv, bn, cin, cout, _a, _p in DF_CONVS:
        specs += [(f"{conv}.conv.weight", (cout, 8 * cin)), (f"{conv}.conv.bias", (cout,)),
                  (f"{bn}.weight", (cout,)), (f"{bn}.bias", (cout,))]
    specs += [("layer5.0.weight", (512, 256 * Ls[4])), ("layer5.0.bias", (512,)),
              ("layer5.1.weight", (512,)), ("layer5.1.bias", (512,)),
              ("layer5.4.weight", (512, 512)), ("layer5.4.bias", (512,)),
              ("layer5.5.weight", (512,)), ("layer5.5.bias", (512,)),
              ("fc.weight", (C, 512)), ("fc.bias", (C,))]
    return specs, Ls


class DFTrainEngine:
    def __init__(self, length: int, num_classes: int, batch: int, state: Dict[str, torch.Tensor], lr: float = 2e-3,
                 betas=(0.9, 0.999), eps: float = 1e-8, device="cuda:0", seed: int = 0, workspace_mb: int = 256,
                 share: Optional["DFTrainEngine"] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("wfd_gan_amd.DFTrainEngine needs a HIP device: there is no CPU path")
        if batch < 2:
            raise ValueError("BatchNorm in training mode needs more than one row")
        self.L = lib()
        self.dev = dev = torch.device(device)
        self.length, self.C, self.B = length, num_classes, batch
        self.lr, self.betas, self.eps, self.seed = lr, betas, eps, seed
        specs, self.Ls = _df_specs(length, num_classes)
        Ls, B, C = self.Ls, batch, num_classes
        f32 = dict(dtype=torch.float32, device=dev)
        z = lambda *s: torch.zeros(*s, **f32)
        if share is not None:
            self.P, self.m, self.bufs, self.step, self.nbt, self.rng_state = (share.P, share.m, share.bufs, share.step,
                                                                             share.nbt, share.rng_state)
        else:
            self.P = FlatParams(specs, dev)
            self.m = torch.zeros_like(self.P.flat)                      # Adamax exp_avg; P.sq holds exp_inf
            self.bufs: Dict[str, torch.Tensor] = {}
            self.nbt: Dict[str, int] = {}
            self.step = torch.zeros(1, dtype=torch.int64, device=dev)
            self.rng_state = torch.zeros(2, dtype=torch.int64, device=dev)
            self.load(state)
        b = self.buf = {}
        self.kin = 256 * Ls[4]
        b["X"] = z(B, length)
        b["labels"] = torch.zeros(B, dtype=torch.int32, device=dev)
        lv, chans = [0, 0, 1, 1, 2, 2, 3], [c[3] for c in DF_CONVS]
        self.lv = lv
        b["Z"] = [z(B * Ls[lv[k]], chans[k]) for k in range(7)]
        b["Y"] = [z(B * Ls[lv[k]], chans[k]) for k in range(7)]
        b["dA"] = [z(B * Ls[lv[k]], chans[k]) for k in range(7)]
        pc = [32, 64, 128, 256]
        n_pool = [B * Ls[i + 1] * pc[i] for i in range(4)]
        # dropout keep-masks: the four pooled activations (p = .1) contiguous, then the two FC ones
        self.mask_sizes = n_pool + [B * 512, B * 512]
        b["MKflat"] = z(sum(self.mask_sizes))
        off, b["MK"] = 0, []
        for n in self.mask_sizes:
            b["MK"].append(b["MKflat"][off: off + n])
            off += n
        b["PO"] = [z(B * Ls[i + 1], pc[i]) for i in range(4)]            # pooled, pre-dropout
        b["POd"] = [z(B * Ls[i + 1], pc[i]) for i in range(4)]           # after dropout (input of the next block)
        b["DPO"] = [z(B * Ls[i + 1], pc[i]) for i in range(4)]
        for k in range(7):
            b[f"mean{k}"], b[f"istd{k}"], b[f"sums{k}"] = z(chans[k]), z(chans[k]), z(2 * chans[k])
        for i in (1, 2):
            b[f"XH{i}"], b[f"H{i}"], b[f"Hd{i}"], b[f"DH{i}"], b[f"DZ{i}"] = (z(B, 512) for _ in range(5))
            b[f"fmean{i}"], b[f"fm2{i}"], b[f"fistd{i}"], b[f"fsums{i}"] = z(512), z(512), z(512), z(1024)
        self.ldc = rup(C, 4)
        b["LOG"], b["DLOG"], b["CE"] = z(B, self.ldc), z(B, self.ldc), z(B)
        b["ones"], b["nsel"], b["invb"] = torch.ones(max(B, 256), **f32), torch.full((1,), float(B), **f32), \
            torch.full((B,), 1.0 / B, **f32)
        b["loss"] = z(1)
        b["WT"] = [None] + [z(DF_CONVS[k][2], 8 * DF_CONVS[k][3]) for k in range(1, 7)]
        self.rpc = max(64, (B * Ls[0] + 1023) // 1024)                   # rows per chunk of the tall reductions
        nchunk = (B * Ls[0] + self.rpc - 1) // self.rpc
        b["part"] = z(max(nchunk * 2 * 256, ((Ls[0] + 127) // 128) * B * 256))
        b["ws"] = torch.empty(max(workspace_mb * (1 << 18), B * Ls[0] * 64), **f32)
        self.fwd_plan = Plan(b["ws"])              # DF.forward in train mode alone (module shell, model.DF.forward)
        self._build_forward(self.fwd_plan)
        self.plan = self._build()
        self.graph = None

    # ------------------------------------------------------------------------------------------------------
    # reference layout <-> packed layout
    def load(self, sd: Dict[str, torch.Tensor]):
        L4 = self.Ls[4]
        packed = {}
        for key, _shape in self.P.specs:
            t = sd[key].detach().to(self.dev, torch.float32)
            if key.endswith("conv.weight"):
                t = t.permute(0, 2, 1).reshape(t.shape[0], -1)
            elif key == "layer5.0.weight":
                t = t.view(512, 256, L4).permute(0, 2, 1).reshape(512, L4 * 256)
            packed[key] = t
        self.P.load(packed)
        for key, v in sd.items():
            if key.endswith("running_mean") or key.endswith("running_var"):
                if key in self.bufs:            # the launch plan holds this buffer's address: update in place
                    self.bufs[key].copy_(v.detach())
                else:
                    self.bufs[key] = v.detach().to(self.dev, torch.float32).clone()
            elif key.endswith("num_batches_tracked"):
                self.nbt[key] = int(v)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """The reference's DF.state_dict(): same keys, order, shapes and dtypes (src/model.py:136-196)."""
        L4 = self.Ls[4]
        out = {}
        cf = torch.contiguous_format

        def put_param(key):
            t = self.P.p[key].detach().clone(memory_format=cf)
            if key.endswith("conv.weight"):
                cout = t.shape[0]
                t = t.view(cout, 8, -1).permute(0, 2, 1).contiguous()
            elif key == "layer5.0.weight":
                t = t.view(512, L4, 256).permute(0, 2, 1).reshape(512, 256 * L4).contiguous()
            out[key] = t

        def put_bn(bn):
            put_param(f"{bn}.weight"); put_param(f"{bn}.bias")
            out[f"{bn}.running_mean"] = self.bufs[f"{bn}.running_mean"].clone()
            out[f"{bn}.running_var"] = self.bufs[f"{bn}.running_var"].clone()
            out[f"{bn}.num_batches_tracked"] = torch.tensor(self.nbt.get(f"{bn}.num_batches_tracked", 0),
                                                            dtype=torch.int64)
        for conv, bn, *_ in DF_CONVS:
            put_param(f"{conv}.conv.weight"); put_param(f"{conv}.conv.bias")
            put_bn(bn)
        for lin, bn in (("layer5.0", "layer5.1"), ("layer5.4", "layer5.5")):
            put_param(f"{lin}.weight"); put_param(f"{lin}.bias")
            put_bn(bn)
        put_param("fc.weight"); put_param("fc.bias")
        return out

    # ------------------------------------------------------------------------------------------------------
    def _build_forward(self, p: Plan):
        """DF.forward in train mode (src/model.py:198-206): batch statistics, running-stat updates, dropout masks."""
        L, b, P, B, C, Ls, lv = self.L, self.buf, self.P, self.B, self.C, self.Ls, self.lv
        w = P.p
        rpc, part = self.rpc, ptr(b["part"])
        Z, Y, PO, POd, MK = b["Z"], b["Y"], b["PO"], b["POd"], b["MK"]
        cur, pi = None, 0
        for k, (conv, bn, cin, cout, act, pool) in enumerate(DF_CONVS):
            M = B * Ls[lv[k]]
            if k == 0:
                p.add(L.df_conv1_fwd, ptr(b["X"]), self.length, 0, B, Ls[0], ptr(w[f"{conv}.conv.weight"]),
                      ptr(b["ones"]), ptr(w[f"{conv}.conv.bias"]), ACT_NONE, ptr(Z[0]), cout)
            else:
                p.conv8_affine(cur, B, Ls[lv[k]], cin, 3, w[f"{conv}.conv.weight"], cout, None, w[f"{conv}.conv.bias"],
                               ACT_NONE, Z[k])
            p.add(L.bn_tall_stats, ptr(Z[k]), cout, M, cout, rpc, part, BN_EPS, BN_MOM, ptr(b[f"mean{k}"]),
                  ptr(b[f"istd{k}"]), ptr(self.bufs[f"{bn}.running_mean"]), ptr(self.bufs[f"{bn}.running_var"]))
            p.add(L.bn_tall_apply, ptr(Z[k]), M, cout, ptr(b[f"mean{k}"]), ptr(b[f"istd{k}"]), ptr(w[f"{bn}.weight"]),
                  ptr(w[f"{bn}.bias"]), act, ptr(Y[k]))
            cur = Y[k]
            if pool:
                p.add(L.df_pool8_fwd, ptr(Y[k]), B, Ls[lv[k]], cout, ptr(PO[pi]))
                p.add(L.mul_mask, ptr(PO[pi]), ptr(MK[pi]), PO[pi].numel(), ptr(POd[pi]))
                cur = POd[pi]
                pi += 1
        kin = self.kin
        prev, kp = POd[3], kin
        for i, lin, bn in ((1, "layer5.0", "layer5.1"), (2, "layer5.4", "layer5.5")):
            s = p.gemm(NT, prev, kp, w[f"{lin}.weight"], P.ld[f"{lin}.weight"], B, 512, kp)
            p.add(L.bn_train_relu, s.ptr, s.ns, s.ss, s.ldp, B, 512, ptr(w[f"{lin}.bias"]), BN_EPS, ptr(w[f"{bn}.weight"]),
                  ptr(w[f"{bn}.bias"]), ptr(b[f"XH{i}"]), ptr(b[f"H{i}"]), 512, ptr(b[f"fmean{i}"]), ptr(b[f"fm2{i}"]),
                  ptr(b[f"fistd{i}"]), ptr(self.bufs[f"{bn}.running_mean"]), ptr(self.bufs[f"{bn}.running_var"]), BN_MOM, 1)
            p.add(L.mul_mask, ptr(b[f"H{i}"]), ptr(MK[3 + i]), B * 512, ptr(b[f"Hd{i}"]))
            prev, kp = b[f"Hd{i}"], 512
        p.gemm_affine(NT, prev, 512, w["fc.weight"], 512, B, C, 512, None, w["fc.bias"], ACT_NONE, b["LOG"], self.ldc)

    def _build(self) -> Plan:
        L, b, P, B, C, Ls, lv = self.L, self.buf, self.P, self.B, self.C, self.Ls, self.lv
        p = Plan(b["ws"])
        w, g = P.p, P.g
        rpc, part = self.rpc, ptr(b["part"])
        Z, Y, dA, PO, POd, DPO, MK, WT = b["Z"], b["Y"], b["dA"], b["PO"], b["POd"], b["DPO"], b["MK"], b["WT"]
        kin = self.kin
        for k in range(1, 7):
            conv, _bn, cin, cout, _a, _p = DF_CONVS[k]
            p.add(L.conv_wflip, ptr(w[f"{conv}.conv.weight"]), cout, cin, ptr(WT[k]))
        self._build_forward(p)
        # ---- CrossEntropyLoss (mean) and its gradient --------------------------------------------------------
        p.add(L.masked_ce, ptr(b["LOG"]), self.ldc, B, C, ptr(b["labels"]), ptr(b["ones"]), ptr(b["nsel"]), 1.0,
              ptr(b["CE"]), ptr(b["DLOG"]), self.ldc)
        p.add(L.colsum, ptr(b["CE"]), 1, B, 1, ptr(b["invb"]), ptr(b["loss"]), 0)
        # ---- backward: head --------------------------------------------------------------------------------
        p.gemm(TN, b["DLOG"], self.ldc, b["Hd2"], 512, C, 512, B, out=g["fc.weight"], ldp=512)
        p.add(L.colsum, ptr(b["DLOG"]), self.ldc, B, C, 0, ptr(g["fc.bias"]), 0)
        dz_next, w_next, ld_next, k_next = b["DLOG"], w["fc.weight"], 512, C
        lda_next = self.ldc
        for i, lin, bn, inp, kp in ((2, "layer5.4", "layer5.5", b["Hd1"], 512), (1, "layer5.0", "layer5.1", POd[3], kin)):
            s = p.gemm(NN, dz_next, lda_next, w_next, ld_next, B, 512, k_next)
            p.mul(s, MK[3 + i].view(B, 512), 512, DM_MUL, None, None, b[f"DH{i}"], None, 512)
            p.add(L.bn_bwd, ptr(b[f"DH{i}"]), 1, 0, 512, B, 512, ptr(b[f"H{i}"]), ptr(b[f"XH{i}"]), 512,
                  ptr(w[f"{bn}.weight"]), ptr(b[f"fistd{i}"]), float(B), ptr(b[f"fsums{i}"]), ptr(b[f"DZ{i}"]), 512,
                  ptr(g[f"{bn}.weight"]), ptr(g[f"{bn}.bias"]), ptr(g[f"{lin}.bias"]), 3)
            p.gemm(TN, b[f"DZ{i}"], 512, inp, kp, 512, kp, B, out=g[f"{lin}.weight"], ldp=P.ld[f"{lin}.weight"])
            dz_next, w_next, ld_next, k_next, lda_next = b[f"DZ{i}"], w[f"{lin}.weight"], P.ld[f"{lin}.weight"], 512, 512
        s = p.gemm(NN, b["DZ1"], 512, w["layer5.0.weight"], P.ld["layer5.0.weight"], B, kin, 512)
        p.mul(s, MK[3].view(B, kin), kin, DM_MUL, None, None, DPO[3], None, kin)
        # ---- backward: conv stack ------------------------------------------------------------------------------
        pi = 3
        for k in range(6, -1, -1):
            conv, bn, cin, cout, act, pool = DF_CONVS[k]
            Lk = Ls[lv[k]]
            M = B * Lk
            if pool:
                p.add(L.df_pool8_bwd_act, ptr(DPO[pi]), ptr(Y[k]), B, Lk, cout, ptr(b["ones"]), act, ptr(dA[k]))
                pi -= 1
            p.add(L.bn_tall_bwd, ptr(dA[k]), ptr(Z[k]), M, cout, rpc, part, ptr(b[f"mean{k}"]), ptr(b[f"istd{k}"]),
                  ptr(w[f"{bn}.weight"]), ptr(b[f"sums{k}"]), ptr(g[f"{bn}.weight"]), ptr(g[f"{bn}.bias"]), ptr(dA[k]))
            p.add(L.colsum_tall, ptr(dA[k]), cout, M, cout, rpc, part, ptr(g[f"{conv}.conv.bias"]))
            if k == 0:
                p.add(L.conv1_wgrad, ptr(dA[0]), ptr(b["X"]), self.length, 0, B, Lk, cout, part,
                      ptr(g[f"{conv}.conv.weight"]))
                break
            prev = DF_CONVS[k - 1]
            inp = POd[pi] if prev[5] else Y[k - 1]
            ldp = 8 * cin
            tiles = ((cout + 63) // 64) * ((ldp + 63) // 64)
            ns = max(1, min(256, 1024 // tiles, (M + 63) // 64))
            while ns > 1 and ns * cout * ldp > b["ws"].numel():
                ns -= 1
            p.add(L.conv8_wgrad_f32, ptr(dA[k]), ptr(inp), B, Lk, cin, cout, ptr(b["ws"]), ldp, cout * ldp, ns)
            p.add(L.ep_affine_act, ptr(b["ws"]), ns, cout * ldp, ldp, cout, ldp, 0, 0, ACT_NONE,
                  ptr(g[f"{conv}.conv.weight"]), ldp)
            if prev[5]:
                p.conv8_mul(dA[k], B, Lk, cout, 4, WT[k], cin, MK[pi].view(M, cin), cin, DM_MUL, None, DPO[pi])
            else:
                dm = DM_ELU if prev[4] == ACT_ELU else DM_RELU
                p.conv8_mul(dA[k], B, Lk, cout, 4, WT[k], cin, Y[k - 1], cin, dm, None, dA[k - 1])
        # ---- torch.optim.Adamax (src/train_df.py:115,128-131) ------------------------------------------------
        p.add(L.adamax, ptr(P.flat), ptr(P.grad), ptr(self.m), ptr(P.sq), P.numel, self.lr, self.betas[0], self.betas[1],
              self.eps, self.step.data_ptr(), 1)
        return p

    # ------------------------------------------------------------------------------------------------------
    def set_batch(self, x: torch.Tensor, y: torch.Tensor):
        """x (B, L) or (B, 1, L) scaled features without the burst-count column; y (B,) labels."""
        self.buf["X"].copy_(x.reshape(self.B, self.length).to(self.dev, torch.float32))
        self.buf["labels"].copy_(y.to(self.dev).to(torch.int32))

    def set_masks(self, masks: Sequence[torch.Tensor]):
        """Explicit dropout keep-masks in the reference's layouts ((B, C, L) x4, (B, 512) x2), values {0, 1/(1-p)}."""
        for i, m in enumerate(masks):
            m = m.to(self.dev, torch.float32)
            if m.dim() == 3:
                m = m.permute(0, 2, 1)
            self.buf["MK"][i].copy_(m.reshape(-1))

    def draw_masks(self):
        """Fast mode: Philox keep-masks on device (one launch per dropout rate)."""
        st = torch.cuda.current_stream().cuda_stream
        n4 = sum(self.mask_sizes[:4])
        flat = self.buf["MKflat"]
        GOLD = 0x9E3779B97F4A7C15
        for k, (a, n, pr) in enumerate(((0, n4, 0.1), (n4, self.B * 512, 0.7), (n4 + self.B * 512, self.B * 512, 0.5))):
            self.L.rng_dropout_mask(flat.data_ptr() + 4 * a, n, pr, (self.seed + (k + 1) * GOLD) & (2**64 - 1),
                                    self.rng_state.data_ptr(), 0, st)
        self.L.iter_advance(self.rng_state.data_ptr(), (n4 + 3) // 4, 0, st)

    def forward_train(self) -> torch.Tensor:
        """Forward pass alone in train mode on the batch / masks set before; returns the logits (B, C)."""
        self.fwd_plan.run()
        self._count_batch()
        return self.buf["LOG"][:, : self.C]

    def train_step(self):
        self.plan.run()
        self._count_batch()

    def _count_batch(self):
        for _conv, bn, *_ in DF_CONVS:
            self.nbt[f"{bn}.num_batches_tracked"] = self.nbt.get(f"{bn}.num_batches_tracked", 0) + 1
        for bn in ("layer5.1", "layer5.5"):
            self.nbt[f"{bn}.num_batches_tracked"] = self.nbt.get(f"{bn}.num_batches_tracked", 0) + 1

    def loss(self) -> float:
        return float(self.buf["loss"].item())

    # ------------------------------------------------------------------------------------------------------
    def accuracy(self, X: torch.Tensor, y: torch.Tensor, bs: int = 128) -> float:
        """Eval-mode argmax accuracy (src/train_df.py:134-157) of the current parameters on device-resident
        X (N, L) / y (N,), in batches of ``bs``."""
        df = ObserverDF(self.state_dict(), self.length, self.C, self.dev)
        correct = torch.zeros(1, dtype=torch.int32, device=self.dev)
        st = torch.cuda.current_stream().cuda_stream
        N = X.shape[0]
        plans = {}
        y32 = y.to(self.dev).to(torch.int32).contiguous()
        for a in range(0, N, bs):
            n = min(bs, N - a)
            if n not in plans:
                ob = observer_buffers(df, n, self.dev)
                xin = torch.zeros(n, self.length, dtype=torch.float32, device=self.dev)
                pl = Plan(self.buf["ws"])
                observer_forward(pl, df, xin, self.length, 0, n, ob["Y"], ob["PO"], ob["F1"], ob["F2"], ob["LOG"], ob["ldc"])
                plans[n] = (pl, xin, ob)
            pl, xin, ob = plans[n]
            xin.copy_(X[a:a + n].reshape(n, self.length))
            pl.run()
            self.L.argmax_correct(ob["LOG"].data_ptr(), ob["ldc"], n, self.C, y32[a:a + n].data_ptr(), correct.data_ptr(), st)
        return int(correct.item()) / max(1, N)
