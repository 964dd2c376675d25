"""TEST INFRASTRUCTURE ONLY (imported by tests/ and tests/golden/ scripts; never by the product path).

CPU restatement of the reference's burst extraction, the step that turns a packet trace into one row of the npz the
trainer consumes: src/extract.py:58-122 (`get_burst`, `extract`) and the text loader
src/utils.py:126-134 (`loadTrace`).  Plain Python / numpy loops in the reference's own order of operations, float64
throughout.  Pinned against the reference's own functions by tests/golden/make_golden_extract.py (bit-exact on seeded
synthetic traces incl. long-gap truncation at both ends, leading incoming packets, zero-size packets, no outgoing
packet at all, truncation to `length`).
"""
import numpy as np

CUT_OFF_THRESHOLD = 10        # src/extract.py:17
CELL_SIZE = 514               # src/common.py:15


def load_trace(path, norm_cell=False):
    """src/utils.py:126-134: tab-separated (time, size) lines -> float64 (n, 2); norm_cell keeps only the sign."""
    rows = []
    with open(path, "r") as f:
        for line in f.readlines():
            a, b = line[:-1].split("\t")[:2]
            rows.append((float(a), float(b)))
    trace = np.array(rows, dtype=np.float64).reshape(-1, 2)
    if norm_cell:
        trace[:, 1] = np.sign(trace[:, 1])
    return trace


def get_burst(trace):
    """src/extract.py:58-110 -> (m, 2) array of [start time, signed size sum] per run of equal-sign packets."""
    n = len(trace)
    start, end = 0, n
    ipt = np.diff(trace[:, 0])
    out = np.where(ipt > CUT_OFF_THRESHOLD)[0]                     # :62-63
    if len(out) > 0:
        if out[0] < 50:                                            # :66-68 head outlier: drop everything up to it
            start = int(out[0]) + 1
        if out[-1] > 50:                                           # :69-71 tail outlier: cut there
            end = int(out[-1]) + 1
    tr = trace[start:end].copy()
    first = len(tr) - 1                                            # :79-84 skip leading incoming packets
    for i in range(len(tr)):
        if tr[i, 1] > 0:
            first = i
            break
    tr = tr[first:].copy()
    tr[:, 0] -= tr[0, 0]                                           # :87
    merged = []
    cnt = 0.0                                                      # :93 (`cnt = 0`; `0 + x` is exact)
    sign = np.sign(tr[0, 1])
    time = tr[0, 0]
    for cur_time, cur_size in tr:                                  # :97-105
        if np.sign(cur_size) == sign:
            cnt += cur_size
        else:
            merged.append([time, cnt])
            sign = np.sign(cur_size)
            cnt = cur_size
            time = cur_time
    merged.append([time, cnt])
    merged = np.array(merged, dtype=np.float64)
    # :106-107 the reference's sanity asserts: runs must alternate outgoing / incoming starting with an outgoing one
    # (they fail on a trace without outgoing packets and, in general, on zero-size packets)
    assert sum(merged[::2, 1]) == sum(tr[tr[:, 1] > 0][:, 1])
    assert sum(merged[1::2, 1]) == sum(tr[tr[:, 1] < 0][:, 1])
    return merged


def extract(trace, length, norm=False):
    """src/extract.py:113-123: (features of `length` entries = [number of bursts, |burst sizes| ..., 0 padding],
    burst start times).  `length` is the script's args.length + 1."""
    bs = get_burst(trace)
    times = bs[:, 0]
    bursts = np.abs(bs[:, 1])
    if norm:
        bursts = bursts / CELL_SIZE
    lst = [float(len(bursts))] + [float(b) for b in bursts]
    lst = lst[:length] + [0.0] * (length - len(lst))
    return np.array(lst, dtype=np.float64), times
