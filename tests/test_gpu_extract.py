"""GPU: burst extraction on the device (wfd_extract_bursts through wfd_gan_amd.extract) against the reference fixture and
the oracle: bit-exact float64 feature rows and burst times, the reference's AssertionError cases, the CLI's files."""
import os

import numpy as np
import pytest

from oracle import extract_oracle as EO
from tests.test_cpu_extract import FX, _traces

pytestmark = pytest.mark.gpu


def test_device_extraction_matches_reference_fixture():
    from wfd_gan_amd.extract import extract_traces
    d = np.load(FX)
    length = int(d["length"])
    traces = _traces(d)
    good = [i for i in range(len(traces)) if not d["raises"][i]]
    for norm, key in ((False, "features_raw"), (True, "features_norm")):
        feat, times = extract_traces([traces[i] for i in good], length, norm=norm)
        assert feat.dtype == np.float64
        assert np.array_equal(feat, d[key][good])
        if not norm:
            for k, i in enumerate(good):
                assert np.array_equal(times[k], d["btimes"][d["btimes_off"][i]: d["btimes_off"][i + 1]]), i
    for i in np.nonzero(d["raises"])[0][:6]:
        with pytest.raises(AssertionError):
            extract_traces([traces[int(i)]], length)


def test_device_extraction_long_and_ragged():
    """Traces longer than one 256-packet block, one- and two-packet traces, bursts beyond --length.  (Sizes are
    integer-valued as in real traces: with fractional sizes the reference's own float-sum asserts :106-107 fail.)"""
    from wfd_gan_amd.extract import extract_traces
    rng = np.random.default_rng(5)
    traces = []
    for n in (1, 2, 255, 256, 257, 5000, 40000):
        t = np.cumsum(rng.exponential(0.01, n))
        runs = np.repeat(rng.choice([-1.0, 1.0], n), rng.integers(1, 7, n))[:n]
        runs[0] = 1.0
        sz = runs * rng.integers(1, 1500, n).astype(np.float64)
        traces.append(np.stack([t, sz], 1))
    for length in (5, 1401):
        feat, times = extract_traces(traces, length, norm=True)
        for i, tr in enumerate(traces):
            f, t = EO.extract(tr, length, True)
            assert np.array_equal(feat[i], f), (i, length)
            assert np.array_equal(times[i], t), (i, length)


def test_extract_cli_files(tmp_path):
    from wfd_gan_amd import extract as X
    d = np.load(FX)
    traces = _traces(d)
    good = [i for i in range(len(traces)) if not d["raises"][i]][:6]
    src = tmp_path / "traces"
    src.mkdir()
    for site in range(2):
        for inst in range(3):
            tr = traces[good[site * 3 + inst]]
            with open(src / f"{site}-{inst}.pkt", "w") as f:
                for t, s in tr:
                    f.write("%r\t%d\n" % (float(t), int(s)))
    conf = tmp_path / "conf.ini"
    conf.write_text("[default]\nMONITORED_SITE_NUM = 2\nMONITORED_INST_NUM = 3\nMONITORED_SITE_START_IND = 0\n"
                    "MONITORED_INST_START_IND = 0\n")
    out = X.main(["--dir", str(src), "--length", "120", "--conf", str(conf), "--out", str(tmp_path / "feature")])
    assert os.path.basename(out) == "raw_feature_0-2x0-3.npz"
    z = np.load(out)
    assert z["features"].shape == (6, 121) and list(z["labels"]) == [0, 0, 0, 1, 1, 1]
    assert np.array_equal(z["features"], d["features_raw"][good])
    tz = np.load(tmp_path / "feature" / "time_feature_0-2x0-3.npz")
    assert len(tz.files) == 6
    from wfd_gan_amd.utils import load_dataset
    Xf, y = load_dataset(out)                       # the trainer's loader reads it back
    assert Xf.shape == (6, 121) and y.shape == (6,)
