#!/usr/bin/env python3
"""Fixture for the burst-extraction path, generated from the REFERENCE's own functions.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_extract.py

Imports /root/reference/src/extract.py unmodified (torchsummaryX / torchsummary stubbed as in make_golden.py), runs its
`extract()` (-> get_burst) on seeded synthetic packet traces and its `utils.loadTrace` on text files written here, asserts
that oracle/extract_oracle.py reproduces both bit for bit, and writes tests/golden/extract.npz: the ragged inputs
(times, sizes, offsets) and the expected feature rows / burst times for two settings (norm False / True).
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_SRC = "/root/reference/src"
from oracle import extract_oracle as EO      # noqa: E402


def import_reference_extract():
    for name in ("torchsummaryX", "torchsummary"):
        stub = types.ModuleType(name)
        stub.summary = lambda *a, **k: None
        sys.modules.setdefault(name, stub)
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    import extract as ref_extract            # noqa
    return ref_extract


def synth_trace(rng, kind):
    n = int(rng.integers(1, 40)) if kind == "short" else int(rng.integers(60, 900))
    gaps = rng.exponential(0.05, n)
    if kind in ("head", "both") and n > 70:
        gaps[int(rng.integers(1, 49))] = 10.5 + rng.random()
    if kind in ("tail", "both") and n > 70:
        gaps[int(rng.integers(53, n))] = 11.0 + rng.random()
    if kind == "multi" and n > 120:
        for j in rng.integers(1, n, 4):
            gaps[int(j)] = 12.0
    t = np.cumsum(gaps)
    if kind == "cell":
        sz = rng.choice([-1.0, 1.0], n)
    else:
        sz = rng.choice([-1.0, 1.0], n) * rng.integers(1, 1500, n).astype(np.float64)
        # bursty: repeat signs
        runs = np.repeat(rng.choice([-1.0, 1.0], n), rng.integers(1, 9, n))[:n]
        sz = np.abs(sz) * runs
    if kind == "leadin":
        sz[: int(rng.integers(1, min(n, 12) + 1))] *= -np.sign(sz[: 1]) * np.sign(sz[: 1])  # keep; then force negatives
        k = int(rng.integers(1, min(n, 12) + 1))
        sz[:k] = -np.abs(sz[:k])
    if kind == "noout":
        sz = -np.abs(sz)
    if kind == "zeros" and n > 5:
        sz[rng.integers(0, n, max(1, n // 10))] = 0.0
    return np.stack([t, sz], 1)


def main():
    ref = import_reference_extract()
    rng = np.random.default_rng(20240607)
    kinds = ["plain", "head", "tail", "both", "multi", "cell", "leadin", "noout", "zeros", "short"]
    traces = [synth_trace(rng, kinds[i % len(kinds)]) for i in range(120)]
    length = 120 + 1                                   # short enough that long traces are truncated
    out = {}
    for norm in (False, True):
        ref.length, ref.norm = length, norm            # the script's module globals (src/extract.py:143-147)
        feats, times, raises = [], [], []
        for i, tr in enumerate(traces):
            try:
                b, t = ref.extract(tr.copy(), "synthetic-%d" % i)
            except AssertionError:                     # the reference's own sanity asserts (src/extract.py:106-107)
                try:
                    EO.extract(tr.copy(), length, norm)
                    raise SystemExit("oracle accepted trace %d, reference asserted" % i)
                except AssertionError:
                    pass
                raises.append(1)
                feats.append(np.zeros(length))
                times.append(np.zeros(0))
                continue
            raises.append(0)
            ob, ot = EO.extract(tr.copy(), length, norm)
            assert np.array_equal(np.asarray(b, dtype=np.float64), ob), (i, norm)
            assert np.array_equal(np.asarray(t, dtype=np.float64), ot), (i, norm)
            feats.append(ob)
            times.append(ot)
        out["raises"] = np.array(raises, dtype=np.int64)
        tag = "norm" if norm else "raw"
        out["features_" + tag] = np.stack(feats)
        if not norm:
            out["btimes"] = np.concatenate(times)
            out["btimes_off"] = np.cumsum([0] + [len(t) for t in times]).astype(np.int64)
    # text loader
    with tempfile.TemporaryDirectory() as d:
        for i in (0, 5, 17):
            p = os.path.join(d, "%d-0.pkt" % i)
            with open(p, "w") as f:
                for t, s in traces[i]:
                    f.write("%.6f\t%d\n" % (t, int(s)))
            for nc in (False, True):
                assert np.array_equal(ref.utils.loadTrace(p, norm_cell=nc), EO.load_trace(p, norm_cell=nc))
    out["times"] = np.concatenate([t[:, 0] for t in traces])
    out["sizes"] = np.concatenate([t[:, 1] for t in traces])
    out["off"] = np.cumsum([0] + [len(t) for t in traces]).astype(np.int64)
    out["length"] = np.array(length)
    np.savez_compressed(os.path.join(HERE, "extract.npz"), **out)
    print("extract.npz: %d traces, %d packets; oracle == reference (extract, loadTrace)" % (len(traces), len(out["times"])))


if __name__ == "__main__":
    main()
