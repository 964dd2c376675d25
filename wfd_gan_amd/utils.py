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
