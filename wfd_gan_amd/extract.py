"""Burst extraction: packet traces -> the npz the trainer consumes (reference: src/extract.py; same flags, same output
files).  The text files are parsed on the host (src/utils.py:126-134 semantics); everything after that -- long-gap
truncation, skipping the leading incoming packets, merging equal-direction packets into bursts, |.|, --norm, padding to
--length + 1 -- runs on the device for all traces at once (wfd_extract_bursts, one workgroup per trace).  The
reference's sanity asserts (src/extract.py:106-107) are evaluated on the host from the device's signed burst sums and
raise AssertionError like the reference does.  There is no CPU path for the extraction itself."""
from __future__ import annotations

import argparse
import configparser
import os
from os.path import join
from typing import List, Sequence, Tuple

import numpy as np

from . import common as cm

CELL_SIZE = 514               # src/common.py:15


def parse_arguments(argv=None):
    """Flags of src/extract.py:20-55 (names, types, defaults)."""
    p = argparse.ArgumentParser(description="Extract burst sequences ipt from raw traces")
    p.add_argument("--dir", metavar="<traces path>", required=True,
                   help="Path to the directory with the traffic traces to be simulated.")
    p.add_argument("--length", type=int, default=1400, help="Pad to length.")
    p.add_argument("--norm", type=bool, default=False,
                   help="Shall we normalize burst sizes by dividing it with cell size?")
    p.add_argument("--norm_cell", type=bool, default=False,
                   help="Shall we ignore the value of a cell (e.g., +-888 -> +-1)?")
    p.add_argument("--format", metavar="<file suffix>", default=".pkt")
    p.add_argument("--log", type=str, dest="log", metavar="<log path>", default="stdout",
                   help="path to the log file. It will print to stdout by default.")
    # additive
    p.add_argument("--conf", default=cm.confdir, help="conf.ini with the [default] site / instance ranges")
    p.add_argument("--out", default=None, help="output directory (default: <repo>/dump/<dir name>/feature)")
    p.add_argument("--cuda_id", type=int, default=0)
    return p.parse_args(argv)


parse_args = parse_arguments


def load_trace(path: str, norm_cell: bool = False) -> np.ndarray:
    """utils.loadTrace (src/utils.py:126-134): tab-separated (time, size) per line -> float64 (n, 2)."""
    with open(path, "r") as f:
        rows = [ln[:-1].split("\t")[:2] for ln in f.readlines()]
    trace = np.array(rows, dtype=np.float64).reshape(-1, 2)
    if norm_cell:
        trace[:, 1] = np.sign(trace[:, 1])
    return trace


def extract_traces(traces: Sequence[np.ndarray], length: int, norm: bool = False, device="cuda:0",
                   names: Sequence[str] = ()) -> Tuple[np.ndarray, List[np.ndarray]]:
    """extract.extract (src/extract.py:113-123) for a list of (n_i, 2) float64 traces: (features (N, length) float64,
    [burst start times of trace i]).  ``length`` is the script's args.length + 1."""
    import torch

    from ._lib import lib
    if not torch.cuda.is_available():
        raise RuntimeError("wfd_gan_amd.extract needs a HIP device: there is no CPU path")
    if any(len(t) == 0 for t in traces):
        raise IndexError("empty trace (the reference fails on it at src/extract.py:87)")
    L = lib()
    dev = torch.device(device)
    n = len(traces)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(t) for t in traces], out=off[1:])
    cat = np.concatenate([np.asarray(t, dtype=np.float64) for t in traces], 0)
    times_h, sizes_h = np.ascontiguousarray(cat[:, 0]), np.ascontiguousarray(cat[:, 1])
    times, sizes = torch.from_numpy(times_h).to(dev), torch.from_numpy(sizes_h).to(dev)
    offd = torch.from_numpy(off).to(dev)
    feat = torch.empty(n, length, dtype=torch.float64, device=dev)
    bsum, btime = torch.empty_like(times), torch.empty_like(times)
    meta = torch.empty(n, 3, dtype=torch.int32, device=dev)
    L.extract_bursts(times.data_ptr(), sizes.data_ptr(), offd.data_ptr(), n, length, float(CELL_SIZE) if norm else 0.0,
                     feat.data_ptr(), bsum.data_ptr(), btime.data_ptr(), meta.data_ptr(),
                     torch.cuda.current_stream(dev).cuda_stream)
    feat_h, bsum_h, btime_h, meta_h = feat.cpu().numpy(), bsum.cpu().numpy(), btime.cpu().numpy(), meta.cpu().numpy()
    out_times = []
    for i in range(n):
        p0, end, nb = (int(v) for v in meta_h[i])
        o = int(off[i])
        runs = bsum_h[o: o + nb]
        kept = sizes_h[o + p0: o + end]
        # src/extract.py:106-107: bursts must alternate outgoing / incoming, starting with an outgoing one
        if not (sum(runs[::2]) == sum(kept[kept > 0]) and sum(runs[1::2]) == sum(kept[kept < 0])):
            raise AssertionError("trace {}: burst directions do not alternate from an outgoing burst "
                                 "(src/extract.py:106-107)".format(names[i] if i < len(names) else i))
        out_times.append(btime_h[o: o + nb].copy())
    return feat_h, out_times


def main(argv=None):
    from . import utils
    args = parse_arguments(argv)
    logger = utils.init_logger("extract")
    length = args.length + 1                      # one more feature: the real number of bursts (src/extract.py:144)
    logger.info("Arguments: %s" % (args))
    outputdir = args.out or join(cm.outputdir, os.path.split(args.dir.rstrip("/"))[1], "feature")
    os.makedirs(outputdir, exist_ok=True)
    cf = configparser.ConfigParser()
    cf.read(args.conf)
    cf = cf["default"]
    site_n, inst_n = int(cf["monitored_site_num"]), int(cf["monitored_inst_num"])
    site0, inst0 = int(cf["monitored_site_start_ind"]), int(cf["monitored_inst_start_ind"])
    flist = []
    for i in range(site0, site0 + site_n):
        for j in range(inst0, inst0 + inst_n):
            f = os.path.join(args.dir, str(i) + "-" + str(j) + args.format)
            if os.path.exists(f):
                flist.append(f)
    logger.info("In total {} files.".format(len(flist)))
    traces, labels = [], []
    for f in flist:
        fname = f.split("/")[-1].split(".")[0]
        traces.append(load_trace(f, norm_cell=args.norm_cell))
        labels.append(int(fname.split("-")[0]) if "-" in fname else site_n)
    bursts, times = extract_traces(traces, length, norm=args.norm, device="cuda:%d" % args.cuda_id, names=flist)
    labels = np.array(labels)
    logger.info("feature sizes:{}, label size:{}".format(bursts.shape, labels.shape))
    tag = "{}-{}x{}-{}".format(site0, site0 + site_n, inst0, inst_n + inst0)
    np.savez_compressed(join(outputdir, "raw_feature_{}.npz".format(tag)), features=bursts, labels=labels)
    logger.info("output to {}".format(join(outputdir, "raw_feature_{}.npz".format(tag))))
    np.savez(join(outputdir, "time_feature_{}.npz".format(tag)), *times)
    return join(outputdir, "raw_feature_{}.npz".format(tag))


if __name__ == "__main__":
    main()
