"""Host-side helpers with the reference's semantics (src/utils.py:25-58,97-103): logger, output directory layout,
npz dataset loader.  Nothing here is on the GPU path."""
import logging
import os
from os.path import join

import numpy as np

from . import common as cm


def init_logger(name, log_dir=None):
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG)
    fmt = logging.Formatter(cm.LOG_FORMAT)
    # a second run in the same process (tests, notebooks) gets its own log file, not the first run's
    for h in list(logger.handlers):
        logger.removeHandler(h)
        if isinstance(h, logging.FileHandler):
            h.close()
    console = logging.StreamHandler()
    console.setFormatter(fmt)
    logger.addHandler(console)
    if log_dir is not None:
        fileh = logging.FileHandler(log_dir, mode="w")
        fileh.setFormatter(fmt)
        logger.addHandler(fileh)
    return logger


def init_directory(dir, tag=""):
    """Outputs go to <grandparent of the dataset file>/<tag>/{model,checkpoint} (src/utils.py:44-52)."""
    basedir = os.path.split(os.path.split(dir)[0])[0]
    modeldir, checkpointdir = join(basedir, tag, "model"), join(basedir, tag, "checkpoint")
    os.makedirs(modeldir, exist_ok=True)
    os.makedirs(checkpointdir, exist_ok=True)
    return join(basedir, tag), modeldir, checkpointdir


def load_dataset(dir):
    """npz with 'features' (N, S) and 'labels' (N,) as extract.py writes it (src/utils.py:97-103)."""
    data = np.load(dir)
    return data["features"], data["labels"]


def fit_transform_device(X: np.ndarray, device):
    """``preprocessing.MinMaxScaler().fit_transform(X)`` followed by the ``.float()`` of the training loop
    (src/mlp_df_wgan_train.py:132-133,184) on the device: returns (fp32 scaled features resident in HBM, a fitted
    sklearn MinMaxScaler with the attributes the reference's scaler would have -- it is what ``model/scaler.gz``
    stores).  Column min / max are exact reductions and the transform is the scaler's two float64 roundings, so the
    result is bit-identical to the host path."""
    import torch
    from sklearn import preprocessing

    from ._lib import lib
    from .sampler import minmax_params
    L = lib()
    Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(device)
    N, S = Xd.shape
    st = torch.cuda.current_stream(device).cuda_stream
    max_chunks = 1024
    part = torch.empty(2 * max_chunks * S, dtype=torch.float64, device=device)
    mn, mx = torch.empty(S, dtype=torch.float64, device=device), torch.empty(S, dtype=torch.float64, device=device)
    L.colminmax_f64(Xd.data_ptr(), N, S, part.data_ptr(), max_chunks, mn.data_ptr(), mx.data_ptr(), st)
    dmin, dmax = mn.cpu().numpy(), mx.cpu().numpy()
    scaler = preprocessing.MinMaxScaler()
    scaler.fit(np.stack([dmin, dmax]))          # two rows carrying the extrema: identical data_min_/max_/scale_/min_
    scaler.n_samples_seen_ = N
    scale, min_ = minmax_params(dmin, dmax)
    assert np.array_equal(scale, scaler.scale_) and np.array_equal(min_, scaler.min_)
    out = torch.empty(N, S, dtype=torch.float32, device=device)
    sd, md = torch.from_numpy(scale).to(device), torch.from_numpy(min_).to(device)
    L.minmax_transform_f64(Xd.data_ptr(), N, S, sd.data_ptr(), md.data_ptr(), out.data_ptr(), st)
    torch.cuda.synchronize(device)
    return out, scaler
