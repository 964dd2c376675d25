"""CPU: the burst-extraction oracle (oracle/extract_oracle.py, restating src/extract.py:58-123) against the fixture that
tests/golden/make_golden_extract.py generated from the reference's own functions."""
import os

import numpy as np
import pytest

from oracle import extract_oracle as EO

FX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "extract.npz")


def _traces(d):
    off = d["off"]
    return [np.stack([d["times"][a:b], d["sizes"][a:b]], 1) for a, b in zip(off[:-1], off[1:])]


def test_oracle_matches_reference_fixture():
    d = np.load(FX)
    length = int(d["length"])
    for norm, key in ((False, "features_raw"), (True, "features_norm")):
        for i, tr in enumerate(_traces(d)):
            if d["raises"][i]:
                with pytest.raises(AssertionError):
                    EO.extract(tr, length, norm)
                continue
            f, t = EO.extract(tr, length, norm)
            assert np.array_equal(f, d[key][i]), (i, norm)
            if not norm:
                assert np.array_equal(t, d["btimes"][d["btimes_off"][i]: d["btimes_off"][i + 1]]), i


def test_load_trace_text(tmp_path):
    p = tmp_path / "3-7.pkt"
    p.write_text("0.000000\t512\n0.010000\t-1024\n0.500000\t-888\n")
    assert np.array_equal(EO.load_trace(str(p)), np.array([[0.0, 512.0], [0.01, -1024.0], [0.5, -888.0]]))
    assert np.array_equal(EO.load_trace(str(p), norm_cell=True)[:, 1], np.array([1.0, -1.0, -1.0]))
    from wfd_gan_amd.extract import load_trace
    assert np.array_equal(load_trace(str(p)), EO.load_trace(str(p)))
