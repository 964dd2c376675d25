"""CPU oracle for the wfd-gan training / sampling hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (torch-CPU ops, functional style) of the reference's per-iteration
algorithm.  It exists to CHECK the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg may import it.  Nothing in the product package (wfd_gan_amd/) imports it, and the
product has no CPU fallback.

Pinning: the reference ships no tests / golden vectors, so this oracle is pinned against
outputs of the reference itself run in the build container: tests/golden/make_golden.py imports
/root/reference/src/model.py, replays the loop body of src/mlp_df_wgan_train.py:184-275 around the
reference's own nn.Modules, and asserts this restatement matches it (bit-exact on CPU, same torch build)
before writing the fixtures under tests/golden/.  It also runs the unmodified reference trainer script
end-to-end on a tiny dataset and compares artefacts with ``run_training`` below.

All citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# W-div constants, src/mlp_df_wgan_train.py:22-24
K_DIV = 2
P_DIV = 6
W_DIST_THRESHOLD = 0.05
# WGAN-GP weight, src/common.py:10
LAMBDA_GP = 10

G_LINEAR_KEYS = ("model.0", "model.2", "model.5", "model.8")
G_BN_KEYS = ("model.3", "model.6")
D_LINEAR_KEYS = ("model.0", "model.3", "model.6", "model.9")
BN_EPS_G = 0.8          # BatchNorm1d(out_feat, 0.8): 2nd positional arg is eps, src/model.py:23
BN_MOMENTUM = 0.1
LEAKY_SLOPE = 0.2       # src/model.py:58
DROP_P = 0.2            # src/model.py:59


# ----------------------------------------------------------------------------------------------
# parameter construction (same torch.nn init order as the reference constructors, so that a seed
# re-derives the same weights the reference would get: src/mlp_df_wgan_train.py:150-153)
# ----------------------------------------------------------------------------------------------
def _linear_init(out_f: int, in_f: int) -> Tuple[torch.Tensor, torch.Tensor]:
    lin = torch.nn.Linear(in_f, out_f)
    return lin.weight.detach().clone(), lin.bias.detach().clone()


def _conv_init(out_c: int, in_c: int, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    cv = torch.nn.Conv1d(in_c, out_c, k)
    return cv.weight.detach().clone(), cv.bias.detach().clone()


def _bn_init(n: int) -> Dict[str, torch.Tensor]:
    return {
        "weight": torch.ones(n), "bias": torch.zeros(n),
        "running_mean": torch.zeros(n), "running_var": torch.ones(n),
        "num_batches_tracked": torch.tensor(0, dtype=torch.int64),
    }


def make_generator_state(seq_size: int, class_dim: int, latent_dim: int) -> Dict[str, torch.Tensor]:
    """state_dict of Generator, keys as src/model.py:27-34 produces them."""
    sd: Dict[str, torch.Tensor] = {}
    dims = [(512, latent_dim + class_dim), (1024, 512), (2048, 1024), (seq_size, 2048)]
    order = ["model.0", "model.2", "model.3", "model.5", "model.6", "model.8"]
    li = 0
    for key in order:
        if key in G_BN_KEYS:
            n = 1024 if key == "model.3" else 2048
            for k2, v in _bn_init(n).items():
                sd[f"{key}.{k2}"] = v
        else:
            w, b = _linear_init(*dims[li])
            li += 1
            sd[f"{key}.weight"], sd[f"{key}.bias"] = w, b
    return sd


def make_discriminator_state(seq_size: int, class_dim: int) -> Dict[str, torch.Tensor]:
    """state_dict of Discriminator, src/model.py:56-67."""
    sd: Dict[str, torch.Tensor] = {}
    dims = [(2048, seq_size + class_dim), (1024, 2048), (512, 1024), (1, 512)]
    for key, d in zip(D_LINEAR_KEYS, dims):
        sd[f"{key}.weight"], sd[f"{key}.bias"] = _linear_init(*d)
    return sd


def df_linear_input(length: int) -> int:
    """src/model.py:208-212."""
    res = length
    for _ in range(4):
        res = int(math.ceil(res / 8))
    return res


DF_CONV_SPEC = [  # (state_dict prefix of conv, prefix of its BN, Cin, Cout, activation, pool_after)
    ("layer1.0", "layer1.1", 1, 32, "elu", False),
    ("layer1.3", "layer1.4", 32, 32, "elu", True),
    ("layer2.0", "layer2.1", 32, 64, "relu", False),
    ("layer2.3", "layer2.4", 64, 64, "relu", True),
    ("layer3.0", "layer3.1", 64, 128, "relu", False),
    ("layer3.3", "layer3.4", 128, 128, "relu", True),
    ("layer4.0", "layer4.1", 128, 256, "relu", True),
]
DF_BN_EPS = 1e-5


def make_df_state(length: int, num_classes: int) -> Dict[str, torch.Tensor]:
    """state_dict of DF, src/model.py:136-196 (construction order = RNG consumption order)."""
    sd: Dict[str, torch.Tensor] = {}
    for conv, bn, cin, cout, _act, _pool in DF_CONV_SPEC:
        w, b = _conv_init(cout, cin, 8)
        sd[f"{conv}.conv.weight"], sd[f"{conv}.conv.bias"] = w, b
        for k2, v in _bn_init(cout).items():
            sd[f"{bn}.{k2}"] = v
    fin = 256 * df_linear_input(length)
    sd["layer5.0.weight"], sd["layer5.0.bias"] = _linear_init(512, fin)
    for k2, v in _bn_init(512).items():
        sd[f"layer5.1.{k2}"] = v
    sd["layer5.4.weight"], sd["layer5.4.bias"] = _linear_init(512, 512)
    for k2, v in _bn_init(512).items():
        sd[f"layer5.5.{k2}"] = v
    sd["fc.weight"], sd["fc.bias"] = _linear_init(num_classes, 512)
    return sd


def perturb_df_bn(sd: Dict[str, torch.Tensor], seed: int) -> None:
    """Give the frozen observer non-trivial BN affine / running statistics (a trained DF has them);
    seeded so that fixtures can re-derive it.  Not in the reference: test-data preparation only."""
    g = torch.Generator().manual_seed(seed)
    for k in sorted(sd.keys()):
        if k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape, generator=g) * 1.5 + 0.5
        elif k.endswith(".weight") and sd[k].dim() == 1:
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
        elif k.endswith(".bias") and sd[k].dim() == 1 and (k[:-5] + ".running_mean") in sd:
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1


# ----------------------------------------------------------------------------------------------
# forward passes
# ----------------------------------------------------------------------------------------------
def g_forward(sd: Dict[str, torch.Tensor], z: torch.Tensor, c: torch.Tensor, scaler_min, scaler_max,
              training: bool = True, return_aux: bool = False):
    """Generator.forward, src/model.py:36-48.  BN buffers in ``sd`` are updated in place when training
    (running_mean/var by F.batch_norm, num_batches_tracked here, as nn.BatchNorm1d does)."""
    h = torch.cat([z, c], 1)
    h = F.relu(F.linear(h, sd["model.0.weight"], sd["model.0.bias"]))
    for lin, bn in (("model.2", "model.3"), ("model.5", "model.6")):
        h = F.linear(h, sd[f"{lin}.weight"], sd[f"{lin}.bias"])
        if training:
            sd[f"{bn}.num_batches_tracked"] += 1
        h = F.batch_norm(h, sd[f"{bn}.running_mean"], sd[f"{bn}.running_var"], sd[f"{bn}.weight"],
                         sd[f"{bn}.bias"], training, BN_MOMENTUM, BN_EPS_G)
        h = F.relu(h)
    y = torch.sigmoid(F.linear(h, sd["model.8.weight"], sd["model.8.bias"]))
    # src/model.py:42-47: keep columns 0..trunc(burst_length); zero the tail.  The multiply/add run in the
    # tensor's dtype (the numpy float64 scalars are demoted to Python scalars by torch).
    burst_length = y[:, 0] * (scaler_max - scaler_min) + scaler_min
    idx = burst_length.detach().to(torch.int64) + 1          # .type(LongTensor) truncates toward zero
    if bool((idx >= y.shape[1]).any()) or bool((idx < 0).any()):
        # the reference's advanced-index assignment raises IndexError here
        raise IndexError("generator burst length index out of bounds for dimension 1")
    cols = torch.arange(y.shape[1]).unsqueeze(0)
    mask = (cols < idx.unsqueeze(1)).to(y.dtype)
    out = y * mask
    if return_aux:
        return out, {"y": y, "idx": idx}
    return out


def d_forward(sd: Dict[str, torch.Tensor], trace: torch.Tensor, c: torch.Tensor,
              masks: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
    """Discriminator.forward, src/model.py:69-72.  Dropout is always live (the trainer never calls
    .eval()).  ``masks`` = three tensors with values in {0, 1/(1-p)}; None draws them with F.dropout from
    the global torch generator exactly as nn.Dropout does."""
    h = torch.cat([trace, c], 1)
    for i, key in enumerate(D_LINEAR_KEYS[:3]):
        h = F.leaky_relu(F.linear(h, sd[f"{key}.weight"], sd[f"{key}.bias"]), LEAKY_SLOPE)
        h = F.dropout(h, DROP_P, True) if masks is None else h * masks[i]
    return F.linear(h, sd["model.9.weight"], sd["model.9.bias"])


def _pad_same(x: torch.Tensor, kernel: int, stride: int) -> torch.Tensor:
    """SAME padding arithmetic of src/model.py:96-101 / :123-128 (zeros, left = p//2)."""
    in_dim = x.shape[-1]
    out_dim = (in_dim + stride - 1) // stride
    p = max(0, (out_dim - 1) * stride + kernel - in_dim)
    return F.pad(x, (p // 2, p - p // 2), "constant", 0)


def df_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """DF.forward in eval mode, src/model.py:198-206 (BN from running stats, dropout = identity).
    x: (n, 1, L)."""
    out = x
    for conv, bn, _cin, _cout, act, pool in DF_CONV_SPEC:
        out = F.conv1d(_pad_same(out, 8, 1), sd[f"{conv}.conv.weight"], sd[f"{conv}.conv.bias"])
        out = F.batch_norm(out, sd[f"{bn}.running_mean"], sd[f"{bn}.running_var"], sd[f"{bn}.weight"],
                           sd[f"{bn}.bias"], False, 0.1, DF_BN_EPS)
        out = F.elu(out) if act == "elu" else F.relu(out)
        if pool:
            # MyMaxPool1dPadSame(8, 1): pad computed with stride 1, pooled with stride = kernel = 8
            out = F.max_pool1d(_pad_same(out, 8, 1), 8)
    out = out.reshape(out.size(0), -1)
    for lin, bn in (("layer5.0", "layer5.1"), ("layer5.4", "layer5.5")):
        out = F.linear(out, sd[f"{lin}.weight"], sd[f"{lin}.bias"])
        out = F.relu(F.batch_norm(out, sd[f"{bn}.running_mean"], sd[f"{bn}.running_var"],
                                  sd[f"{bn}.weight"], sd[f"{bn}.bias"], False, 0.1, DF_BN_EPS))
    return F.linear(out, sd["fc.weight"], sd["fc.bias"])


# ----------------------------------------------------------------------------------------------
# penalties
# ----------------------------------------------------------------------------------------------
def wdiv_penalty(validity: torch.Tensor, traces: torch.Tensor) -> torch.Tensor:
    """per-row (||d validity / d traces||^2)^(p/2), src/mlp_df_wgan_train.py:203-207."""
    ones = torch.ones(traces.size(0), 1, dtype=validity.dtype)
    grad = torch.autograd.grad(validity, traces, ones, create_graph=True, retain_graph=True,
                               only_inputs=True)[0]
    return grad.view(grad.size(0), -1).pow(2).sum(1) ** (P_DIV / 2)


def wgan_gp_penalty(d_sd, real: torch.Tensor, fake: torch.Tensor, c: torch.Tensor, alpha: torch.Tensor,
                    masks=None) -> torch.Tensor:
    """compute_gradient_penalty, src/utils.py:64-84 (alpha is the (B,1) U[0,1) draw of :67)."""
    interp = (alpha * real + ((1 - alpha) * fake)).requires_grad_(True)
    d_interp = d_forward(d_sd, interp, c, masks)
    ones = torch.ones(real.shape[0], 1, dtype=real.dtype)
    g = torch.autograd.grad(d_interp, interp, ones, create_graph=True, retain_graph=True,
                            only_inputs=True)[0]
    g = g.view(g.size(0), -1)
    return ((g.norm(2, dim=1) - 1) ** 2).mean(0)


# ----------------------------------------------------------------------------------------------
# the iteration
# ----------------------------------------------------------------------------------------------
@dataclass
class OracleState:
    seq_len: int
    class_dim: int
    latent_dim: int
    scaler_min: float
    scaler_max: float
    G: Dict[str, torch.Tensor]
    D: Dict[str, torch.Tensor]
    DF: Dict[str, torch.Tensor]
    lr: float = 2e-4
    alpha: float = 0.02
    n_critic: int = 3
    penalty: str = "div"
    opt_G: Optional[torch.optim.Optimizer] = None
    opt_D: Optional[torch.optim.Optimizer] = None
    dtype: torch.dtype = torch.float32

    def g_params(self) -> List[torch.Tensor]:
        # nn.Module.parameters() order: registration order, weight then bias
        out = []
        for key in ("model.0", "model.2", "model.3", "model.5", "model.6", "model.8"):
            out += [self.G[f"{key}.weight"], self.G[f"{key}.bias"]]
        return out

    def d_params(self) -> List[torch.Tensor]:
        out = []
        for key in D_LINEAR_KEYS:
            out += [self.D[f"{key}.weight"], self.D[f"{key}.bias"]]
        return out


def make_state(seq_len: int, class_dim: int, latent_dim: int, scaler_min: float, scaler_max: float,
               seed: int, df_seed: Optional[int] = None, df_state: Optional[dict] = None, lr: float = 2e-4,
               alpha: float = 0.02, n_critic: int = 3, penalty: str = "div",
               dtype: torch.dtype = torch.float32) -> OracleState:
    """Seeded construction in the reference's order G -> D -> DF (src/mlp_df_wgan_train.py:150-153); the
    observer weights then come from ``df_state`` (a checkpoint, :154) or a separate seed."""
    torch.manual_seed(seed)
    G = make_generator_state(seq_len, class_dim, latent_dim)
    D = make_discriminator_state(seq_len, class_dim)
    DFs = make_df_state(seq_len - 1, class_dim)
    if df_state is not None:
        DFs = {k: v.clone() for k, v in df_state.items()}
    elif df_seed is not None:
        torch.manual_seed(df_seed)
        DFs = make_df_state(seq_len - 1, class_dim)
        perturb_df_bn(DFs, df_seed + 1)

    def cast(sd):
        return {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}

    G, D, DFs = cast(G), cast(D), cast(DFs)
    st = OracleState(seq_len, class_dim, latent_dim, scaler_min, scaler_max, G, D, DFs, lr, alpha, n_critic,
                     penalty, dtype=dtype)
    for p in st.g_params() + st.d_params():
        p.requires_grad_(True)
    # the reference leaves DF params requires_grad=True but never steps them; their weight-grads are
    # wasted work and do not change any result, so the oracle freezes them.
    st.opt_G = torch.optim.RMSprop(st.g_params(), lr=lr)     # src/mlp_df_wgan_train.py:163
    st.opt_D = torch.optim.RMSprop(st.d_params(), lr=lr)     # :164
    return st


def draw_dropout_masks(batch: int, dtype=torch.float32) -> List[torch.Tensor]:
    """Consume the global torch CPU generator exactly as one Discriminator.forward does
    (three nn.Dropout(0.2) on (B,2048),(B,1024),(B,512)): returns masks with values {0, 1.25}."""
    return [F.dropout(torch.ones(batch, n, dtype=dtype), DROP_P, True) for n in (2048, 1024, 512)]


def iteration(st: OracleState, traces: torch.Tensor, labels: torch.Tensor, z: torch.Tensor, do_g: bool,
              masks_real=None, masks_fake=None, masks_g=None, gp_alpha: Optional[torch.Tensor] = None,
              masks_gp=None, keep: bool = False, after_d_step=None) -> Dict[str, object]:
    """One pass of the loop body src/mlp_df_wgan_train.py:182-275.
    ``after_d_step(st)`` (tests only) is called right after optimizer_D.step(): a parity test that evaluates the G-step on
    the device's own post-step critic parameters loads them into ``st.D`` there.

    traces (B,S) already scaled, labels (B,) int64, z (B,Z).  Masks None -> drawn from the global torch
    generator in the reference's order.  Returns python floats / tensors of everything the parity tests look
    at.  ``keep`` also returns gradients and the fake batch."""
    out: Dict[str, object] = {}
    dt = st.dtype
    real = traces.to(dt).clone().requires_grad_(True)                          # :184
    c_onehot = torch.from_numpy(np.eye(st.class_dim)[labels.numpy()]).to(dt)   # :185
    if c_onehot.dim() == 1:
        c_onehot = c_onehot.unsqueeze(0)
    z = z.to(dt)
    B = real.shape[0]

    # ---- discriminator step, :191-230
    st.opt_D.zero_grad()
    fake, aux = g_forward(st.G, z, c_onehot, st.scaler_min, st.scaler_max, True, return_aux=True)   # :196
    rv = d_forward(st.D, real, c_onehot, masks_real)                       # :198
    fv = d_forward(st.D, fake, c_onehot, masks_fake)                       # :200
    out["idx_d"] = aux["idx"].clone()
    if st.penalty == "div":
        n_r = wdiv_penalty(rv, real)                                        # :203-207
        n_f = wdiv_penalty(fv, fake)                                        # :209-213
        pen = torch.mean(n_r + n_f) * K_DIV / 2                             # :215
        d_loss = -torch.mean(rv) + torch.mean(fv) + pen                     # :218
        out["n_r"], out["n_f"] = n_r.detach().clone(), n_f.detach().clone()
    elif st.penalty == "gp":
        # the commented-out alternative :220-224 (with the one-hot c; the Long label would not cat)
        gpv = wgan_gp_penalty(st.D, real.detach(), fake.detach(), c_onehot, gp_alpha.to(dt), masks_gp)
        pen = gpv
        d_loss = -torch.mean(rv) + torch.mean(fv) + LAMBDA_GP * gpv
    else:
        raise ValueError(st.penalty)
    out["penalty"] = float(pen.detach())
    out["mean_rv"], out["mean_fv"] = float(torch.mean(rv).detach()), float(torch.mean(fv).detach())
    out["w_dist"] = float((torch.mean(rv) - torch.mean(fv)).detach())                   # :226
    out["d_loss"] = float(d_loss.detach())                                             # :227
    if keep:
        out["fake_d"] = fake.detach().clone()
        out["rv"], out["fv"] = rv.detach().clone(), fv.detach().clone()
    d_loss.backward()                                                         # :229
    if keep:
        out["d_grads"] = [p.grad.detach().clone() for p in st.d_params()]
    st.opt_D.step()                                                           # :230
    if after_d_step is not None:
        after_d_step(st)

    st.opt_G.zero_grad()                                                      # :232
    out["did_g"] = bool(do_g)
    if do_g:                                                                  # :234
        fake, aux = g_forward(st.G, z, c_onehot, st.scaler_min, st.scaler_max, True, return_aux=True)  # :240
        fv = d_forward(st.D, fake, c_onehot, masks_g)                         # :246
        g_loss = -torch.mean(fv)                                              # :247
        sel = fv[:, 0] > 0                                                    # :251-252
        n_sel = int(sel.sum())
        if n_sel > 0:
            conv_in = fake[sel][:, 1:]                                        # :257
            conv_in = conv_in.reshape(conv_in.size(0), 1, conv_in.size(1))    # :258-259
            logits = df_forward(st.DF, conv_in)                               # :261
            f_loss = F.cross_entropy(logits, labels[sel])                     # :262 (mean over selected)
            combined = g_loss + st.alpha * f_loss                             # :263
            f_item = float(f_loss.detach())
        else:
            combined = g_loss                                                 # :266
            f_item = 0.0
        out["g_loss"], out["f_loss"], out["g_combined"] = float(g_loss.detach()), f_item, float(combined.detach())
        out["n_sel"] = n_sel
        out["idx_g"] = aux["idx"].clone()
        if keep:
            out["fake_g"] = fake.detach().clone()
            out["fv_g"] = fv.detach().clone()
        combined.backward()                                                   # :273
        if keep:
            out["g_grads"] = [p.grad.detach().clone() for p in st.g_params()]
        st.opt_G.step()                                                       # :275
    return out


# ----------------------------------------------------------------------------------------------
# sampler post-processing
# ----------------------------------------------------------------------------------------------
def scaler_params(data_min: np.ndarray, data_max: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """sklearn MinMaxScaler(feature_range=(0,1)) attributes scale_, min_ from data_min_/data_max_
    (sklearn.preprocessing._data.MinMaxScaler.partial_fit; constant columns get scale 1)."""
    rng = (data_max - data_min).astype(np.float64)
    rng[rng == 0.0] = 1.0           # _handle_zeros_in_scale (exact zeros; float64 data)
    scale = 1.0 / rng
    min_ = 0.0 - data_min * scale
    return scale, min_


def inverse_transform_f32(x32: np.ndarray, scale: np.ndarray, min_: np.ndarray) -> np.ndarray:
    """MinMaxScaler.inverse_transform on a float32 array: check_array keeps float32, then in-place
    ``X -= min_; X /= scale_`` with float64 operands, each result rounded back to float32."""
    x = np.array(x32, dtype=np.float32, copy=True)
    x -= min_
    x /= scale
    return x


def sample_postprocess(row32: np.ndarray, scale: np.ndarray, min_: np.ndarray) -> np.ndarray:
    """src/generate.py:59-63 for one generated row (n = 1, so the mean over axis 0 is the row itself)."""
    x = inverse_transform_f32(row32.reshape(1, -1), scale, min_).flatten()
    x = np.round(np.trim_zeros(x)).astype(int)
    x[x == 0] = 1
    return x


def sample_postprocess_wt(row32: np.ndarray, scale: np.ndarray, min_: np.ndarray) -> np.ndarray:
    """WTGAN.get_ref_trace, src/gan_wt.py:79-87: burst count from element 0 (truncated, capped at S-1, made even),
    then round-half-even of the next ``length`` elements, 0 -> 1.  Element 0 itself is not part of the output."""
    x = inverse_transform_f32(row32.reshape(1, -1), scale, min_).flatten()
    length = min(int(x[0]), len(x) - 1)
    if length % 2 != 0:
        length -= 1
    x = np.round(x[1:1 + length]).astype(int)
    return np.where(x == 0, 1, x)


def sample_postprocess_glue(row32: np.ndarray, scale: np.ndarray, min_: np.ndarray) -> np.ndarray:
    """query_from_gan, src/gan_glue.py:114-124: ``length = int(x0)`` elements after the first, TRUNCATED to int,
    trailing zeros trimmed, odd length drops its last element, values <= 0 become 1, signs alternate +,-."""
    x = inverse_transform_f32(row32.reshape(1, -1), scale, min_).flatten()
    length = int(x[0])
    x = x[1:1 + length].astype(int)
    x = np.trim_zeros(x, trim="b")
    if len(x) % 2 == 1:
        x = x[:-1]
    x[x <= 0] = 1
    return np.tile([1, -1], len(x) // 2) * x


def sample_postprocess_synth(row32: np.ndarray, scale: np.ndarray, min_: np.ndarray) -> np.ndarray:
    """sample_trace, src/synthesize.py:86-90 (n = 1): round-half-even of the whole row to int FIRST, then
    np.trim_zeros at both ends; interior zeros stay 0."""
    x = inverse_transform_f32(row32.reshape(1, -1), scale, min_)
    x = np.round(x).astype(int).flatten()
    return np.trim_zeros(x)


POSTPROCESS = {"generate": sample_postprocess, "wt": sample_postprocess_wt, "glue": sample_postprocess_glue,
               "synth": sample_postprocess_synth}


def sample(st_G: Dict[str, torch.Tensor], c_ind: int, z: np.ndarray, class_dim: int, scaler_min, scaler_max,
           scale: np.ndarray, min_: np.ndarray) -> np.ndarray:
    """sample(), src/generate.py:48-64, with z supplied (the reference draws np.random.randn(1, Z))."""
    with torch.no_grad():
        zt = torch.from_numpy(z.astype("float32"))
        c = torch.zeros(zt.shape[0], class_dim)
        c[:, c_ind] = 1
        y = g_forward(st_G, zt, c, scaler_min, scaler_max, training=False).numpy()
    y = y.mean(axis=0).reshape(1, -1)
    return sample_postprocess(y, scale, min_)


def my_clip(X: np.ndarray, max_len: float, max_out: float, max_in: float) -> np.ndarray:
    """--clip preprocessing, src/mlp_df_wgan_train.py:82-91 applied to every row (vectorised restatement:
    col 0 <= max_len, odd cols <= max_out, even cols >= 2 <= max_in)."""
    X = np.array(X, copy=True)
    X[:, 0] = np.minimum(X[:, 0], max_len)
    X[:, 1::2] = np.minimum(X[:, 1::2], max_out)
    X[:, 2::2] = np.minimum(X[:, 2::2], max_in)
    return X


# ----------------------------------------------------------------------------------------------
# synthetic data (BASELINE.md §3 recipes)
# ----------------------------------------------------------------------------------------------
def synth_dataset(n_cls: int, n_inst: int, L: int, n_lo: int, max_out: int, max_in: int, seed: int = 0):
    """features (n_cls*n_inst, L+1) float64, labels repeat(arange(n_cls), n_inst): row = [n, alternating
    outgoing U{1..max_out} / incoming U{1..max_in} for n columns, zeros]."""
    rng = np.random.default_rng(seed)
    N = n_cls * n_inst
    X = np.zeros((N, L + 1), dtype=np.float64)
    n = rng.integers(n_lo, L + 1, size=N)
    X[:, 0] = n
    out_v = rng.integers(1, max_out + 1, size=(N, (L + 1) // 2 + 1))
    in_v = rng.integers(1, max_in + 1, size=(N, (L + 1) // 2 + 1))
    body = np.zeros((N, L), dtype=np.float64)
    body[:, 0::2] = out_v[:, : body[:, 0::2].shape[1]]
    body[:, 1::2] = in_v[:, : body[:, 1::2].shape[1]]
    keep = np.arange(L)[None, :] < n[:, None]
    X[:, 1:] = body * keep
    y = np.repeat(np.arange(n_cls), n_inst).astype(np.int64)
    return X, y
