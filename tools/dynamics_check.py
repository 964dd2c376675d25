#!/usr/bin/env python3
"""Training dynamics beside the CPU oracle: the trainer CLI on the dataset / seeds of a recorded oracle run
(100 classes x 200 instances, L = 1400, 5 epochs of 312 + 1 iterations).  The random streams differ (Philox on device vs
NumPy / torch CPU generators), so the trajectories agree statistically, not numerically: per-epoch losses are printed
for comparison with profiles/r01_oracle_dynamics.json."""
import json
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wfdgan_oracle as O  # noqa: E402
from wfd_gan_amd import mlp_df_wgan_train  # noqa: E402

X, y = O.synth_dataset(100, 200, 1400, 50, 75, 179, seed=0)
tmp = tempfile.mkdtemp(prefix="wfd_dyn_")
ddir = os.path.join(tmp, "set", "feature")
os.makedirs(ddir)
npz = os.path.join(ddir, "raw_feature_synth.npz")
np.savez(npz, features=X, labels=y)
torch.manual_seed(0)
ck = os.path.join(tmp, "df.ckpt")
torch.save(O.make_df_state(1400, 100), ck)
out = mlp_df_wgan_train.main(["-d", npz, "--f_model", ck, "--n_epochs", "5", "--freq", "1000", "--seed", "1"] + sys.argv[1:])
ls = np.load(os.path.join(out, "checkpoint", "loss.npz"))
print(json.dumps({k: [float(v) for v in ls[k]] for k in ("generator", "discriminator", "dist")}))
