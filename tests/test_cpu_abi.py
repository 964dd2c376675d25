"""CPU: the C-ABI library loads and exports every symbol include/wfdgan.h declares; host-side logic."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from wfd_gan_amd import _lib
    decls = _lib.parse_header()
    assert len(decls) >= 30
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in decls:
        assert hasattr(cdll, name), f"{name} declared in include/wfdgan.h but not exported"
    L = _lib.lib()
    assert L.version() >= 1
    # host-only helper (no GPU needed)
    assert L.gemm_pick_splits(128, 2048, 1501) >= 2
    assert L.gemm_pick_splits(8192, 2048, 1024) == 1


def test_header_cites_reference_for_every_entry_point():
    src = open(os.path.join(ROOT, "include", "wfdgan.h")).read()
    assert src.count("src/") >= 20


def test_module_shells_have_reference_state_dict_layout():
    from oracle import wfdgan_oracle as O
    from wfd_gan_amd.model import DF, Discriminator, Generator
    torch.manual_seed(5)
    G, D, F = Generator(501, 10, 500, 20.0, 500.0), Discriminator(501, 10), DF(500, 10)
    st = O.make_state(501, 10, 500, 20.0, 500.0, 5)
    for mod, sd in ((G, st.G), (D, st.D), (F, st.DF)):
        msd = mod.state_dict()
        assert list(msd.keys()) == list(sd.keys())
        for k in sd:
            assert msd[k].shape == sd[k].shape and msd[k].dtype == sd[k].dtype
            assert torch.equal(msd[k], sd[k].detach()), k
    assert G.model[3].eps == 0.8 and G.model[3].momentum == 0.1
    assert F.linear_input() == 1


def test_no_cpu_fallback():
    from wfd_gan_amd.model import Discriminator, Generator
    G = Generator(41, 5, 16, 0.0, 40.0)
    with pytest.raises(RuntimeError):
        G(torch.zeros(2, 16), torch.zeros(2, 5))
    with pytest.raises(RuntimeError):
        Discriminator(41, 5)(torch.zeros(2, 41), torch.zeros(2, 5))
    if not torch.cuda.is_available():
        from wfd_gan_amd.engine import TrainEngine
        with pytest.raises(RuntimeError):
            TrainEngine(41, 5, 16, 8, 0.0, 40.0, {}, {}, {})


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "wfd_gan_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn


def test_minmax_params_match_sklearn():
    from sklearn import preprocessing
    from wfd_gan_amd.sampler import minmax_params
    rng = np.random.default_rng(1)
    X = rng.integers(0, 200, size=(64, 33)).astype(np.float64)
    X[:, 7] = 3.0
    sc = preprocessing.MinMaxScaler().fit(X)
    scale, min_ = minmax_params(sc.data_min_, sc.data_max_)
    assert np.array_equal(scale, sc.scale_) and np.array_equal(min_, sc.min_)


def test_shard_rows_partitions():
    from wfd_gan_amd.parallel import shard_rows
    for n, w in ((64, 8), (1000003, 8), (5, 8), (100, 3)):
        spans = [shard_rows(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_cli_contract_matches_reference_argparse():
    """tests/golden/cli_contract.json holds the argparse calls of the reference's four scripts on the path (read from
    their source by make_golden.py): every flag must be accepted by the build's parser under every alias, with the
    same type and the same default.  (generate.py's --train / --test are accepted but not required here: the bulk
    sampler reads the scaler from the model directory.)"""
    import json
    from wfd_gan_amd import extract, generate, mlp_df_wgan_train, train_df
    contract = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cli_contract.json")))
    mods = {"mlp_df_wgan_train.py": (mlp_df_wgan_train, ["-d", "X.npz", "--f_model", "df.ckpt"]),
            "generate.py": (generate, ["-m", "model_dir"]),
            "train_df.py": (train_df, ["X.npz"]),
            "extract.py": (extract, ["--dir", "traces"])}
    sample = {"int": ("3", 3), "float": ("0.5", 0.5), "str": ("abc", "abc"), None: ("abc", "abc"), "bool": ("1", True)}
    for script, args in contract.items():
        mod, base = mods[script]
        ns0 = vars(mod.parse_args(base))
        for a in args:
            flags = a["flags"]
            positional = not flags[0].startswith("-")
            dest = flags[0] if positional else [f for f in flags if f.startswith("--")][0][2:]
            assert dest in ns0, (script, dest)
            if not positional and not a.get("required") and dest not in ("dir", "f_model", "model"):
                assert ns0[dest] == a.get("default"), (script, dest, ns0[dest], a.get("default"))
            if positional:
                continue
            for flag in flags:
                if a.get("action") == "store_true":
                    got = vars(mod.parse_args(base + [flag]))[dest]
                    assert got is True, (script, flag)
                else:
                    text, want = sample[a.get("type")]
                    argv = [x for x in base]
                    if flag in argv or any(f in argv for f in flags):     # replace the required placeholder
                        i = next(i for i, x in enumerate(argv) if x in flags)
                        del argv[i:i + 2]
                    got = vars(mod.parse_args(argv + [flag, text]))[dest]
                    assert got == want and type(got) is type(want), (script, flag, got)


def test_mesh_entry_points_reject_bad_arguments_without_touching_a_gpu():
    """wfd_mesh_*: the argument checks come before any HIP call (a null mesh, more than 8 ranks, a rank outside the world)."""
    import ctypes
    from wfd_gan_amd._lib import lib
    L = lib()
    assert L.mesh_flag_bytes() == 2 * 8 * 128
    assert L._raw_wfd_mesh_allreduce_f32(None, 0, 4, None) == 1
    assert L._raw_wfd_mesh_reduce_rmsprop(None, 0, 4, None, 4, 1e-4, 0.99, 1e-8, 1.0, None, 0, 0, None) == 1
    out = ctypes.c_void_p()
    one = (ctypes.c_void_p * 1)(16)
    assert L._raw_wfd_mesh_create(0, 9, one, one, 4, ctypes.addressof(out)) == 1
    assert L._raw_wfd_mesh_create(2, 2, one, one, 4, ctypes.addressof(out)) == 1
    assert L._raw_wfd_mesh_set_max_blocks(None, 4) == 1


def test_gemm_mode_is_per_host_thread():
    """wfd_set_gemm_mode is thread-local state: a thread that selects the bf16 pipe does not switch another thread's plans."""
    import threading
    from wfd_gan_amd._lib import lib
    L = lib()
    assert L.get_gemm_mode() == 0
    seen = {}
    go, done = threading.Event(), threading.Event()

    def other():
        seen["start"] = L.get_gemm_mode()
        L.set_gemm_mode(2)
        go.set()
        done.wait(10)
        seen["end"] = L.get_gemm_mode()
    t = threading.Thread(target=other)
    t.start()
    assert go.wait(10)
    assert L.get_gemm_mode() == 0                 # the other thread's choice is not ours
    L.set_gemm_mode(1)
    done.set()
    t.join(10)
    assert seen == {"start": 0, "end": 2} and L.get_gemm_mode() == 1
    L.set_gemm_mode(0)
