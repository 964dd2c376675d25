#!/usr/bin/env python3
"""Observed distance between the HIP generator's pre-round burst values and the reference fixture's (fp32 summation order),
and the integer mismatches that follow.  GPU only.   python tools/sampler_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import wfdgan_oracle as O
from wfd_gan_amd.sampler import BulkSampler, minmax_params

for name in ("tiny", "cfg2"):
    fx = np.load(os.path.join("tests", "golden", f"sampler_{name}.npz"))
    meta = json.loads(str(fx["meta"]))
    S, C, Z = meta["seq_len"], meta["class_dim"], meta["latent_dim"]
    torch.manual_seed(meta["seed"])
    gsd = O.make_generator_state(S, C, Z)
    g = torch.Generator().manual_seed(meta["seed"] + 1)
    for k in gsd:
        if k.endswith("running_mean"):
            gsd[k] = torch.randn(gsd[k].shape, generator=g) * 0.05
        elif k.endswith("running_var"):
            gsd[k] = torch.rand(gsd[k].shape, generator=g) * 0.2 + 0.05
    n = fx["z"].shape[0]
    bs = BulkSampler(gsd, S, C, Z, fx["data_min"], fx["data_max"], chunk=n, device="cuda:0")
    out, lens = bs.generate(torch.from_numpy(fx["c"]), torch.from_numpy(fx["z"]))
    y = bs.y[:n, :S].cpu().numpy()
    scale, min_ = minmax_params(fx["data_min"], fx["data_max"])
    rng = fx["data_max"] - fx["data_min"]
    pre_g = O.inverse_transform_f32(y, scale, min_)
    pre_r = O.inverse_transform_f32(fx["y"], scale, min_)
    both = (y != 0) & (fx["y"] != 0)
    err = np.abs(pre_g - pre_r) * both
    rel = err / np.maximum(rng, 1)[None, :]
    out, lens = out.cpu().numpy().astype(np.int64), lens.cpu().numpy()
    nlen = int((lens != fx["out_len"]).sum())
    mism = 0
    rows_exact = 0
    for i in range(n):
        if lens[i] != fx["out_len"][i]:
            continue
        d = out[i][:lens[i]] != fx["out"][i][:lens[i]]
        mism += int(d.sum())
        rows_exact += int(not d.any())
    print(name, "rows", n, "max abs pre-round err", err.max(), "col0 max", err[:, 0].max(), "max rel-to-range", rel.max(),
          "length mismatches", nlen, "element mismatches", mism, "of", int(both.sum()), "exact rows", rows_exact)
