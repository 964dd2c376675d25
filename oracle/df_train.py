"""TEST INFRASTRUCTURE ONLY (see oracle/wfdgan_oracle.py): CPU restatement of the DF observer's training,
src/train_df.py:66-161 around src/model.py:135-206 in train mode.  Pinned bit-exact against the reference's own
``DF`` module + ``torch.optim.Adamax`` by tests/golden/make_golden.py (PIN G).

Never imported by the product (wfd_gan_amd/)."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import wfdgan_oracle as O

DROPOUT_P = (0.1, 0.1, 0.1, 0.1, 0.7, 0.5)      # layer1..4 pooled outputs, layer5.3, layer5.7 (src/model.py:152-194)


def df_param_keys(sd: Dict[str, torch.Tensor]) -> List[str]:
    """nn.Module.parameters() order of DF = state_dict order without the BN buffers."""
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]


def dropout_mask_shapes(length: int, batch: int):
    """Shapes of the six nn.Dropout inputs in forward order ((B, C, L) for the conv blocks, (B, 512) for the FCs)."""
    Ls = [length]
    for _ in range(4):
        Ls.append((Ls[-1] + 7) // 8)
    return [(batch, 32, Ls[1]), (batch, 64, Ls[2]), (batch, 128, Ls[3]), (batch, 256, Ls[4]), (batch, 512), (batch, 512)]


def df_forward_train(sd: Dict[str, torch.Tensor], x: torch.Tensor,
                     masks: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
    """DF.forward in TRAIN mode (src/model.py:198-206): batch-statistics BatchNorm (running stats and
    num_batches_tracked in ``sd`` updated in place), dropout active.  ``masks``: six keep-masks with values
    {0, 1/(1-p)} in forward order; None draws them with F.dropout from the global generator exactly as the
    reference's nn.Dropout layers do."""
    def drop(t, i):
        if masks is None:
            return F.dropout(t, DROPOUT_P[i], True)
        return t * masks[i]

    def bn(t, name):
        sd[f"{name}.num_batches_tracked"] += 1
        return F.batch_norm(t, sd[f"{name}.running_mean"], sd[f"{name}.running_var"], sd[f"{name}.weight"],
                            sd[f"{name}.bias"], True, 0.1, O.DF_BN_EPS)

    out, di = x, 0
    for conv, bnn, _cin, _cout, act, pool in O.DF_CONV_SPEC:
        out = F.conv1d(O._pad_same(out, 8, 1), sd[f"{conv}.conv.weight"], sd[f"{conv}.conv.bias"])
        out = bn(out, bnn)
        out = F.elu(out) if act == "elu" else F.relu(out)
        if pool:
            out = F.max_pool1d(O._pad_same(out, 8, 1), 8)
            out = drop(out, di)
            di += 1
    out = out.reshape(out.size(0), -1)
    for lin, bnn in (("layer5.0", "layer5.1"), ("layer5.4", "layer5.5")):
        out = F.linear(out, sd[f"{lin}.weight"], sd[f"{lin}.bias"])
        out = F.relu(bn(out, bnn))
        out = drop(out, di)
        di += 1
    return F.linear(out, sd["fc.weight"], sd["fc.bias"])


class DFTrainer:
    """model + criterion + optimizer of src/train_df.py:113-115 on a functional state dict."""

    def __init__(self, sd: Dict[str, torch.Tensor]):
        self.sd = {k: v.detach().clone() for k, v in sd.items()}
        self.keys = df_param_keys(self.sd)
        for k in self.keys:
            self.sd[k].requires_grad_(True)
        self.opt = torch.optim.Adamax([self.sd[k] for k in self.keys])

    def step(self, x: torch.Tensor, y: torch.Tensor, masks=None, keep: bool = False):
        """One iteration of :121-131.  Returns dict(loss, logits[, grads])."""
        logits = df_forward_train(self.sd, x, masks)
        loss = F.cross_entropy(logits, y)
        self.opt.zero_grad()
        loss.backward()
        out = dict(loss=float(loss.detach()), logits=logits.detach().clone())
        if keep:
            out["grads"] = {k: self.sd[k].grad.detach().clone() for k in self.keys}
        self.opt.step()
        return out

    def accuracy(self, X: torch.Tensor, y: torch.Tensor, bs: int = 128) -> float:
        """:134-157: eval-mode argmax accuracy in batches."""
        correct = 0
        with torch.no_grad():
            sd = {k: v.detach() for k, v in self.sd.items()}
            for a in range(0, X.shape[0], bs):
                pred = O.df_forward(sd, X[a:a + bs]).argmax(1)
                correct += int((pred == y[a:a + bs]).sum())
        return correct / max(1, X.shape[0])

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: v.detach().clone() for k, v in self.sd.items()}


def prepare_dataset(X: np.ndarray, y: np.ndarray):
    """:66-96: re-index labels, MinMax scale, drop the burst-count column, add the channel axis, stratified 90/10
    split with random_state 42."""
    from sklearn import preprocessing
    from sklearn.model_selection import train_test_split
    y = y - y.min()
    Xs = preprocessing.MinMaxScaler().fit_transform(X)[:, 1:]
    Xt = torch.from_numpy(Xs[:, np.newaxis, :]).float()
    yt = torch.from_numpy(y)
    return train_test_split(Xt, yt, test_size=0.1, random_state=42, stratify=yt)
