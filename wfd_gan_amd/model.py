"""nn.Module shells with the reference's constructor / forward signatures and state_dict layout
(src/model.py:10-212), whose forward runs the HIP kernels of libwfdgan_hip.so.

* ``Generator(seq_size, class_dim, latent_dim, scaler_min, scaler_max, is_gpu=False).forward(z, c)``
* ``Discriminator(seq_size, class_dim).forward(trace, c)``
* ``DF(length, num_classes=100).forward(x)``, ``.linear_input()``

The submodule tree is the same (``model.0`` ... ``model.9``, ``layer1.0.conv`` ...), so checkpoints written by the
reference load with ``strict=True`` and vice versa, and constructing the modules in the reference's order under
``torch.manual_seed(s)`` yields the same initial weights.  ``forward`` is inference-style (no autograd graph): the
training step lives in ``wfd_gan_amd.engine.TrainEngine`` in closed form.  There is no CPU path: a forward on a
CPU tensor raises.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn as nn

from ._lib import lib
from .engine import ObserverDF, observer_buffers, observer_forward
from .plan import ACT_NONE, ACT_RELU, NT, Plan, ptr, rup


def _need_gpu(t: torch.Tensor, who: str):
    if not t.is_cuda:
        raise RuntimeError(f"{who}: the wfd-gan hot path runs on the MI355X only (got a {t.device} tensor); "
                           "there is no CPU fallback")


class _Workspace:
    _ws: Dict[torch.device, torch.Tensor] = {}

    @classmethod
    def get(cls, device, floats: int) -> torch.Tensor:
        w = cls._ws.get(device)
        if w is None or w.numel() < floats:
            w = torch.empty(max(floats, 1 << 24), dtype=torch.float32, device=device)
            cls._ws[device] = w
        return w


class Generator(nn.Module):
    """MLP generator: Linear+ReLU, 2x (Linear+BatchNorm1d(eps=0.8)+ReLU), Linear+Sigmoid, then the learned-length
    tail mask (src/model.py:10-48)."""

    def __init__(self, seq_size, class_dim, latent_dim, scaler_min, scaler_max, is_gpu=False):
        super().__init__()
        self.seq_size, self.class_dim, self.latent_dim = seq_size, class_dim, latent_dim
        self.scaler_min, self.scaler_max = scaler_min, scaler_max
        self.is_gpu = is_gpu
        widths = [latent_dim + class_dim, 512, 1024, 2048]
        layers = []
        for i in range(3):
            layers.append(nn.Linear(widths[i], widths[i + 1]))
            if i > 0:
                layers.append(nn.BatchNorm1d(widths[i + 1], 0.8))     # eps = 0.8, as the reference passes it
            layers.append(nn.ReLU(inplace=True))
        layers += [nn.Linear(2048, seq_size), nn.Sigmoid()]
        self.model = nn.Sequential(*layers)
        self._cache = {}

    def forward(self, z, c):
        _need_gpu(z, "Generator.forward")
        L = lib()
        B, S, Z, C = z.shape[0], self.seq_size, self.latent_dim, self.class_dim
        dev = z.device
        key = (B, self.training, self.model[0].weight.data_ptr(), dev)
        ent = self._cache.get(key)
        if ent is None:
            self._cache.clear()
            ldz = rup(Z + C, 4)
            f = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
            bufs = dict(zin=f(B, ldz), h1=f(B, 512), z2=f(B, 1024), xh2=f(B, 1024), h2=f(B, 1024), z3=f(B, 2048),
                        xh3=f(B, 2048), h3=f(B, 2048), out=f(B, S), idx=torch.zeros(B, dtype=torch.int32, device=dev),
                        err=torch.zeros(1, dtype=torch.int32, device=dev), st=[f(1024), f(1024), f(1024), f(2048),
                                                                              f(2048), f(2048)],
                        sc=[f(1024), f(1024), f(2048), f(2048)])
            ws = _Workspace.get(dev, 16 * max(B, 64) * 2048)
            p = Plan(ws)
            m = self.model
            s = p.gemm(NT, bufs["zin"], ldz, m[0].weight, Z + C, B, 512, Z + C)
            p.affine_act(s, None, m[0].bias, ACT_RELU, bufs["h1"], 512)
            prev, kp = bufs["h1"], 512
            for j, (li, bi, n) in enumerate(((2, 3, 1024), (5, 6, 2048))):
                s = p.gemm(NT, prev, kp, m[li].weight, kp, B, n, kp)
                if self.training:
                    mean, m2, istd = bufs["st"][3 * j: 3 * j + 3]
                    zb, xh, h = (bufs["z2"], bufs["xh2"], bufs["h2"]) if j == 0 else (bufs["z3"], bufs["xh3"], bufs["h3"])
                    p.add(L.bn_stats, s.ptr, s.ns, s.ss, s.ldp, B, n, ptr(m[li].bias), ptr(zb), n, ptr(mean), ptr(m2))
                    p.add(L.bn_apply_relu, ptr(zb), n, B, n, ptr(mean), ptr(m2), float(B), 0.8, ptr(m[bi].weight),
                          ptr(m[bi].bias), ptr(xh), ptr(h), n, ptr(istd), ptr(m[bi].running_mean),
                          ptr(m[bi].running_var), 0.1, 1)
                else:
                    h = bufs["h2"] if j == 0 else bufs["h3"]
                    sc, sh = bufs["sc"][2 * j], bufs["sc"][2 * j + 1]
                    p.affine_act(s, sc, sh, ACT_RELU, h, n)
                prev, kp = h, n
            s = p.gemm(NT, prev, 2048, m[8].weight, 2048, B, S, 2048)
            p.add(L.ep_sigmoid_lenmask, s.ptr, s.ns, s.ss, s.ldp, B, S, ptr(m[8].bias),
                  float(self.scaler_max - self.scaler_min), float(self.scaler_min), ptr(bufs["out"]), S,
                  ptr(bufs["idx"]), ptr(bufs["err"]))
            ent = (p, bufs)
            self._cache[key] = ent
        p, bufs = ent
        bufs["zin"][:, :Z].copy_(z)
        bufs["zin"][:, Z:Z + C].copy_(c)
        if not self.training:
            # BatchNorm in eval mode = per-feature affine of the running statistics (eps 0.8)
            for j, (li, bi) in enumerate(((2, 3), (5, 6))):
                bn = self.model[bi]
                sc = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
                bufs["sc"][2 * j].copy_(sc)
                bufs["sc"][2 * j + 1].copy_((self.model[li].bias.detach() - bn.running_mean) * sc + bn.bias.detach())
        elif B < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        p.run()
        if self.training:
            for bi in (3, 6):
                self.model[bi].num_batches_tracked += 1
        if int(bufs["err"].item()) != 0:
            bufs["err"].zero_()
            raise IndexError("index out of bounds: generated burst length reaches seq_size (model.py:44 semantics)")
        return bufs["out"].clone()


class Discriminator(nn.Module):
    """MLP critic, LeakyReLU(0.2) + Dropout(0.2) after each hidden Linear (src/model.py:51-72).  Dropout follows
    ``self.training`` (the reference trainer never leaves training mode); masks come from Philox on device."""

    def __init__(self, seq_size, class_dim):
        super().__init__()
        self.seq_size, self.class_dim = seq_size, class_dim
        widths = [seq_size + class_dim, 2048, 1024, 512]
        layers = []
        for i in range(3):
            layers += [nn.Linear(widths[i], widths[i + 1]), nn.LeakyReLU(0.2, inplace=True), nn.Dropout(0.2)]
        layers.append(nn.Linear(512, 1))
        self.model = nn.Sequential(*layers)
        self._cache = {}
        self.seed = 0x5eed
        self._calls = 0

    def forward(self, trace, c, masks=None):
        _need_gpu(trace, "Discriminator.forward")
        L = lib()
        B, S, C = trace.shape[0], self.seq_size, self.class_dim
        dev = trace.device
        key = (B, self.model[0].weight.data_ptr(), dev)
        ent = self._cache.get(key)
        if ent is None:
            self._cache.clear()
            ldx = rup(S + C, 4)
            f = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
            bufs = dict(x=f(B, ldx), mk=[f(B, 2048), f(B, 1024), f(B, 512)], h=[f(B, 2048), f(B, 1024), f(B, 512)],
                        d=[f(B, 2048), f(B, 1024), f(B, 512)], v=f(B), dl=f(B, 512))
            p = Plan(_Workspace.get(dev, 16 * max(B, 64) * 2048))
            m = self.model
            prev, kp, ld = bufs["x"], S + C, ldx
            for j, (li, n) in enumerate(((0, 2048), (3, 1024), (6, 512))):
                s = p.gemm(NT, prev, ld, m[li].weight, kp, B, n, kp)
                p.lrelu_drop(s, m[li].bias, bufs["mk"][j], n, bufs["h"][j], n, bufs["d"][j], n)
                prev, kp, ld = bufs["h"][j], n, n
            p.add(L.d_head, ptr(bufs["h"][2]), 512, ptr(bufs["d"][2]), 512, ptr(m[9].weight), ptr(m[9].bias), B, 512, 0,
                  ptr(bufs["v"]), ptr(bufs["dl"]), 0, 512)
            ent = (p, bufs)
            self._cache[key] = ent
        p, bufs = ent
        bufs["x"][:, :S].copy_(trace)
        bufs["x"][:, S:S + C].copy_(c)
        st = torch.cuda.current_stream().cuda_stream
        for j, mk in enumerate(bufs["mk"]):
            if masks is not None:
                mk.copy_(masks[j])
            elif self.training:
                L.rng_dropout_mask(mk.data_ptr(), mk.numel(), 0.2, self.seed + j, 0, self._calls * (1 << 20), st)
            else:
                mk.fill_(1.0)
        self._calls += 1
        p.run()
        return bufs["v"].clone().unsqueeze(1)


def _channels_last(x: torch.Tensor) -> torch.Tensor:
    """(B, C, L) -> contiguous [B][L][C] rows, the layout of the conv / pool kernels."""
    return x.detach().to(torch.float32).permute(0, 2, 1).contiguous()


class MyConv1dPadSame(nn.Module):
    """k=8 stride-1 Conv1d with "SAME" zero padding, left 3 / right 4 (src/model.py:75-105; state_dict key
    ``<name>.conv.*``).  forward runs the HIP conv kernels: (B, Cin, L) -> (B, Cout, L)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        if kernel_size != 8 or stride != 1:
            raise NotImplementedError("the DF observer uses kernel_size=8, stride=1 everywhere")
        self.in_channels, self.out_channels, self.kernel_size, self.stride = in_channels, out_channels, kernel_size, stride
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, stride)

    def forward(self, x):
        _need_gpu(x, "MyConv1dPadSame.forward")
        L = lib()
        B, cin, Ln = x.shape
        cout, dev = self.out_channels, x.device
        if cin != self.in_channels:
            raise RuntimeError(f"expected {self.in_channels} input channels, got {cin}")
        w, bias = self.conv.weight.detach().float(), self.conv.bias.detach().float().contiguous()
        out = torch.empty(B * Ln, cout, dtype=torch.float32, device=dev)
        if cin == 1:
            xin = x.detach().to(torch.float32).reshape(B, Ln).contiguous()
            ones = torch.ones(cout, dtype=torch.float32, device=dev)
            wr = w.reshape(cout, 8).contiguous()
            L.df_conv1_fwd(xin.data_ptr(), Ln, 0, B, Ln, wr.data_ptr(), ones.data_ptr(), bias.data_ptr(), ACT_NONE,
                           out.data_ptr(), cout, torch.cuda.current_stream().cuda_stream)
        elif cin % 4 == 0:
            xin = _channels_last(x)
            wp = w.permute(0, 2, 1).reshape(cout, 8 * cin).contiguous()        # [Cout][tap * Cin + ci]
            p = Plan(_Workspace.get(dev, max(1 << 24, 16 * B * Ln * rup(cout, 4))))
            p.conv8_affine(xin, B, Ln, cin, 3, wp, cout, None, bias, ACT_NONE, out)
            p.run()
        else:
            raise NotImplementedError("in_channels must be 1 or a multiple of 4 (the DF observer's are 1, 32, 64, 128)")
        return out.view(B, Ln, cout).permute(0, 2, 1).contiguous()


class MyMaxPool1dPadSame(nn.Module):
    """Zero-pad (3,4) then MaxPool1d(8) with stride 8 (src/model.py:108-132; the ``stride_size`` argument only enters the
    reference's padding arithmetic); no parameters.  forward: (B, C, L) -> (B, C, ceil(L / 8))."""

    def __init__(self, kernel_size, stride_size):
        super().__init__()
        if kernel_size != 8 or stride_size != 1:
            raise NotImplementedError("the DF observer pools with kernel_size=8, stride_size=1 everywhere")
        self.kernel_size, self.stride = kernel_size, stride_size

    def forward(self, x):
        _need_gpu(x, "MyMaxPool1dPadSame.forward")
        B, C, Ln = x.shape
        xin = _channels_last(x)
        Lo = (Ln + 7) // 8
        out = torch.empty(B * Lo, C, dtype=torch.float32, device=x.device)
        lib().df_pool8_fwd(xin.data_ptr(), B, Ln, C, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return out.view(B, Lo, C).permute(0, 2, 1).contiguous()


class DF(nn.Module):
    """Deep-Fingerprinting observer, 7 convs + 3 FC (src/model.py:135-212).  In eval mode forward is inference on the
    frozen observer (BN from running statistics, dropout off), which is how the GAN trainer uses it
    (src/mlp_df_wgan_train.py:155).  In train mode forward is the reference's train-mode pass (BatchNorm on batch
    statistics with running-stat updates, dropout masks from Philox on device -- or the masks in ``self.masks`` when a
    test sets them) on the kernels of df_engine.DFTrainEngine; like the other shells it builds no autograd graph: the
    observer's training step (forward + backward + Adamax, src/train_df.py:121-131) is DFTrainEngine.train_step."""

    def __init__(self, length, num_classes=100):
        super().__init__()
        self.length, self.num_classes = length, num_classes
        chans = [1, 32, 64, 128, 256]
        drops = [0.1, 0.1, 0.1, 0.1]
        for b in range(4):
            cin, cout = chans[b], chans[b + 1]
            act = nn.ELU if b == 0 else nn.ReLU
            mods = [MyConv1dPadSame(cin, cout, 8, 1), nn.BatchNorm1d(cout), act()]
            if b < 3:
                mods += [MyConv1dPadSame(cout, cout, 8, 1), nn.BatchNorm1d(cout), act()]
            mods += [MyMaxPool1dPadSame(8, 1), nn.Dropout(drops[b])]
            setattr(self, f"layer{b + 1}", nn.Sequential(*mods))
        self.layer5 = nn.Sequential(nn.Linear(256 * self.linear_input(), 512), nn.BatchNorm1d(512), nn.ReLU(),
                                    nn.Dropout(0.7), nn.Linear(512, 512), nn.BatchNorm1d(512), nn.ReLU(),
                                    nn.Dropout(0.5))
        self.fc = nn.Linear(512, num_classes)
        self._cache = {}

    def linear_input(self):
        res = self.length
        for _ in range(4):
            res = int(math.ceil(res / 8))
        return res

    def _forward_train(self, x):
        from .df_engine import DFTrainEngine
        B, dev = x.shape[0], x.device
        key = ("train", B, dev)
        eng = self._cache.get(key)
        if eng is None:
            eng = self._cache[key] = DFTrainEngine(self.length, self.num_classes, B, self.state_dict(), device=dev,
                                                   seed=0xD0F, workspace_mb=64)
        else:
            eng.load(self.state_dict())              # parameters / running statistics may have changed since
        eng.set_batch(x.reshape(B, self.length), torch.zeros(B, dtype=torch.int64))
        masks = getattr(self, "masks", None)
        if masks is not None:
            eng.set_masks(masks)
        else:
            eng.draw_masks()
        logits = eng.forward_train().clone()
        with torch.no_grad():                        # what nn.BatchNorm1d leaves behind after a train-mode forward
            for k, v in self.state_dict().items():
                if k.endswith("running_mean") or k.endswith("running_var"):
                    v.copy_(eng.bufs[k])
                elif k.endswith("num_batches_tracked"):
                    v += 1
        # the packed eval-mode observers (keyed by batch size / fc.weight storage) folded the OLD running statistics into
        # their scale / shift: drop them, so that the next eval forward re-packs from the buffers just updated
        for k in [k for k in self._cache if k[0] != "train"]:
            del self._cache[k]
        return logits

    def forward(self, x):
        _need_gpu(x, "DF.forward")
        if self.training:
            return self._forward_train(x)
        B = x.shape[0]
        dev = x.device
        key = (B, self.fc.weight.data_ptr(), dev)
        ent = self._cache.get(key)
        if ent is None:
            # one packed observer + buffer set per batch size (the accuracy pass alternates 128-row and remainder batches)
            for k in [k for k in self._cache if k[1] != key[1] or k[0] == "train"]:
                del self._cache[k]
            df = ObserverDF(self.state_dict(), self.length, self.num_classes, dev)
            ob = observer_buffers(df, B, dev)
            xin = torch.zeros(B, rup(self.length, 4), dtype=torch.float32, device=dev)
            p = Plan(_Workspace.get(dev, 1 << 26))
            observer_forward(p, df, xin, xin.shape[1], 0, B, ob["Y"], ob["PO"], ob["F1"], ob["F2"], ob["LOG"], ob["ldc"])
            ent = (p, ob, xin, df)
            self._cache[key] = ent
        p, ob, xin, _ = ent
        xin[:, : self.length].copy_(x.reshape(B, self.length))
        p.run()
        return ob["LOG"][:, : self.num_classes].clone()

    def refresh(self):
        """Drop the packed weights (call after load_state_dict when the storage was re-used in place)."""
        self._cache.clear()

    def load_state_dict(self, *a, **k):
        self._cache.clear()
        return super().load_state_dict(*a, **k)
