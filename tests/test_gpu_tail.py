"""GPU: the per-epoch tail of the trainer and the module shells against the oracle.

* test_DF_acc_epoch (src/mlp_df_wgan_train.py:57-79)  -> mlp_df_wgan_train.df_accuracy  vs oracle.train_loop.df_acc_epoch
* checkpoint dump (:288-297)                           -> mlp_df_wgan_train.epoch_dump   vs sklearn's MinMaxScaler (bit-equal)
* MyConv1dPadSame / MyMaxPool1dPadSame / DF.forward in train mode (src/model.py:92-105,119-132,198-206) vs torch / the oracle
* the sampler's err flag: IndexError where the reference's Generator.forward raises (src/model.py:44)
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import df_train as DT
from oracle import train_loop as TL
from oracle import wfdgan_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _df(length, C, seed):
    from wfd_gan_amd.model import DF
    torch.manual_seed(seed)
    sd = O.make_df_state(length, C)
    O.perturb_df_bn(sd, seed + 1)
    m = DF(length, C)
    m.load_state_dict(sd)
    return m, sd


@pytest.mark.parametrize("geom", [(500, 10, 333), (1400, 100, 300), (64, 5, 129)])
def test_df_accuracy_matches_oracle(geom):
    """A fixed block of "generated" rows + labels through the trainer's df_accuracy (device-side count, batches of 128
    with a ragged last one) and through the oracle's restatement of test_DF_acc_epoch: the same number of correct rows.
    Labels are the oracle's own prediction on half of the rows, so the accuracy is neither 0 nor 1."""
    from wfd_gan_amd.mlp_df_wgan_train import df_accuracy
    length, C, N = geom
    m, sd = _df(length, C, seed=7)
    m.eval().to(DEV)
    g = torch.Generator().manual_seed(3)
    rows = torch.rand(N, length + 1, generator=g)
    rows = rows * (torch.arange(length + 1) < torch.randint(length // 4, length, (N, 1), generator=g)).float()
    with torch.no_grad():
        pred = O.df_forward(sd, rows[:, 1:].unsqueeze(1)).argmax(1)
    labels = torch.randint(0, C, (N,), generator=g)
    labels[::2] = pred[::2]
    want = TL.df_acc_epoch(sd, list(rows.numpy()), list(labels.numpy()))
    got = df_accuracy(m, rows.to(DEV), labels.to(DEV))
    assert 0.3 < want < 0.9
    assert round(got * N) == round(want * N), (got, want)


def test_epoch_dump_bit_equal_to_sklearn():
    from sklearn import preprocessing
    from wfd_gan_amd.mlp_df_wgan_train import epoch_dump
    X, _ = O.synth_dataset(7, 30, 200, 10, 75, 179, seed=4)
    X[:, 9] = 5.0                                  # a constant column: scale_ = 1
    scaler = preprocessing.MinMaxScaler().fit(X)
    g = torch.Generator().manual_seed(1)
    fake = torch.rand(150, 201, generator=g)
    fake[:, 100:] = 0.0
    sel = np.random.RandomState(0).permutation(X.shape[0])[:150]
    want_real = scaler.inverse_transform(scaler.transform(X[sel]))               # :291, float64
    want_fake = scaler.inverse_transform(fake.numpy().copy())                    # :292, float32
    real, recon = epoch_dump(scaler, X[sel], fake.to(DEV))
    assert real.dtype == np.float64 and recon.dtype == np.float32 == want_fake.dtype
    assert np.array_equal(real, want_real)
    assert np.array_equal(recon, want_fake)


@pytest.mark.parametrize("geom", [(3, 1, 32, 77), (2, 32, 64, 175), (2, 64, 64, 40), (1, 128, 256, 3), (2, 4, 8, 19)])
def test_conv_and_pool_shells_match_torch(geom):
    """MyConv1dPadSame.forward / MyMaxPool1dPadSame.forward on NCL tensors vs F.pad (3, 4) + conv1d / max_pool1d(8)."""
    from wfd_gan_amd.model import MyConv1dPadSame, MyMaxPool1dPadSame
    B, cin, cout, Ln = geom
    torch.manual_seed(B + cin + Ln)
    conv = MyConv1dPadSame(cin, cout, 8, 1)
    x = torch.randn(B, cin, Ln)
    want = F.conv1d(F.pad(x.double(), (3, 4)), conv.conv.weight.detach().double(), conv.conv.bias.detach().double())
    got = conv.to(DEV)(x.to(DEV))
    assert got.shape == want.shape
    assert (got.double().cpu() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())
    pool = MyMaxPool1dPadSame(8, 1)
    y = F.elu(x)                                   # negative values: the ZERO padding must win over them (src/model.py:123-130)
    wantp = F.max_pool1d(F.pad(y, (3, 4)), 8)
    gotp = pool(y.to(DEV))
    assert gotp.shape == wantp.shape and torch.equal(gotp.cpu(), wantp)
    with pytest.raises(RuntimeError):
        conv(x)                                    # CPU tensor: no fallback


def test_df_forward_train_mode_matches_oracle():
    """DF.forward in train mode (batch-statistics BatchNorm, dropout with injected masks) vs oracle.df_forward_train, incl.
    the running statistics and num_batches_tracked the call leaves behind; then eval mode sees the updated statistics."""
    length, C, B = 64, 5, 16
    m, sd = _df(length, C, seed=11)
    m.to(DEV).train()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 1, length, generator=g)
    masks = [((torch.rand(s, generator=g) > p).float() / (1.0 - p)) for s, p in zip(DT.dropout_mask_shapes(length, B), DT.DROPOUT_P)]
    ref_sd = {k: v.clone() for k, v in sd.items()}
    want = DT.df_forward_train(ref_sd, x, masks).detach()
    m.masks = masks
    got = m(x.to(DEV)).cpu()
    assert (got - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())
    msd = m.state_dict()
    for k, v in ref_sd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert (msd[k].cpu() - v).abs().max().item() < 1e-5 * max(1.0, v.abs().max().item()), k
        elif k.endswith("num_batches_tracked"):
            assert int(msd[k]) == int(v), k
    m.masks = None
    m(x.to(DEV))                                   # Philox masks: runs, updates the statistics once more
    assert int(m.state_dict()["layer1.1.num_batches_tracked"]) == int(ref_sd["layer1.1.num_batches_tracked"]) + 1
    m.eval()
    m.refresh()
    sd2 = {k: v.cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        want_eval = O.df_forward(sd2, x)
    assert (m(x.to(DEV)).cpu() - want_eval).abs().max().item() < 1e-4 * max(1.0, want_eval.abs().max().item())


def test_df_eval_after_train_forward_sees_the_new_running_statistics():
    """eval, train, eval with the SAME batch size and no refresh(): the train-mode forward updates the BatchNorm running
    statistics in place, so the packed eval-mode observer cached by the first eval forward is stale and must be rebuilt
    (the cache key (B, fc.weight storage, device) does not change).  The last eval is compared with the oracle on
    the module's state after the train forward, and must differ from the first eval."""
    length, C, B = 64, 5, 16
    m, sd = _df(length, C, seed=13)
    m.to(DEV).eval()
    g = torch.Generator().manual_seed(9)
    x = torch.rand(B, 1, length, generator=g)
    first = m(x.to(DEV)).cpu()
    with torch.no_grad():
        want_first = O.df_forward(sd, x)
    assert (first - want_first).abs().max().item() < 1e-4 * max(1.0, want_first.abs().max().item())
    m.train()
    m(x.to(DEV) * 3.0 + 1.0)                       # moves every running mean / variance
    m.eval()
    last = m(x.to(DEV)).cpu()
    sd2 = {k: v.cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        want_last = O.df_forward(sd2, x)
    assert (last - want_last).abs().max().item() < 1e-4 * max(1.0, want_last.abs().max().item())
    assert (last - first).abs().max().item() > 1e-3          # the statistics did change the logits


def test_sampler_raises_where_reference_raises():
    """A saturated length output (sigmoid -> 1.0, index = seq_size) makes the reference's Generator.forward raise IndexError
    (src/model.py:44); the bulk sampler reports it the same way instead of returning clamped traces."""
    from wfd_gan_amd.sampler import BulkSampler
    S, C, Z = 41, 5, 16
    st = O.make_state(S, C, Z, 0.0, 40.0, seed=1)
    gsd = {k: v.detach().clone() for k, v in st.G.items()}
    dmin, dmax = np.zeros(S), np.full(S, 40.0)
    bs = BulkSampler(gsd, S, C, Z, dmin, dmax, chunk=64, device=DEV, seed=1)
    labels = (torch.arange(64, device=DEV) % C).to(torch.int32)
    bs.run_chunk(labels)
    bs.check_length_index()                        # healthy generator: no error
    gsd["model.8.bias"][0] = 1e4
    bad = BulkSampler(gsd, S, C, Z, dmin, dmax, chunk=64, device=DEV, seed=1)
    bad.run_chunk(labels)
    with pytest.raises(IndexError):
        bad.check_length_index()


@pytest.mark.parametrize("S", [41, 1401, 103, 64])
def test_sampler_wide_access_forms_are_bit_identical(S):
    """From 1024 rows on, the sampler's two row passes (sigmoid + length mask; inverse scaling + trimming + rounding) run
    with four columns per thread and access (ep_sigmoid_lenmask_v4_kernel, sample_post_v4_kernel).  The same rows in one
    1024-row call and in four 256-row calls (the one-column forms): identical fp32 output, length indices, integers and
    lengths -- including rows whose length index cuts a group of four and rows that are all zero."""
    from wfd_gan_amd._lib import lib
    L = lib()
    n, ld = 1024, (S + 3) // 4 * 4
    g = torch.Generator().manual_seed(S)
    P = (torch.randn(n, ld, generator=g) * 2).to(DEV)
    P[:, 0] = torch.linspace(-6, 6, n)                 # length indices from 0 to S - 1
    bias = (torch.randn(ld, generator=g) * 0.1).to(DEV)
    dmin, dmax = np.zeros(S), np.linspace(S - 1.0, 3.0, S)
    from wfd_gan_amd.sampler import minmax_params
    scale, min_ = minmax_params(dmin, dmax)
    sc, mn = torch.from_numpy(scale).to(DEV), torch.from_numpy(min_).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for rows in (n, 256):
        y = torch.full((n, ld), 7.0, device=DEV)
        idx = torch.zeros(n, dtype=torch.int32, device=DEV)
        err = torch.zeros(1, dtype=torch.int32, device=DEV)
        out = torch.zeros(n, S, dtype=torch.int32, device=DEV)
        lens = torch.zeros(n, dtype=torch.int32, device=DEV)
        for a in range(0, n, rows):
            L.ep_sigmoid_lenmask(P[a:].data_ptr(), 1, 0, ld, rows, S, bias.data_ptr(), float(S - 1), 0.0, y[a:].data_ptr(), ld,
                                 idx[a:].data_ptr(), err.data_ptr(), st)
            L.sample_post(y[a:].data_ptr(), ld, rows, S, mn.data_ptr(), sc.data_ptr(), out[a:].data_ptr(), S, lens[a:].data_ptr(),
                          st)
        torch.cuda.synchronize()
        res.append((y, idx, out, lens))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    y, idx, out, lens = res[0]
    assert torch.all(y[:, S:] == 7.0)                  # columns behind S - 1 are not touched
    assert int(idx.min()) <= 8 and int(idx.max()) >= S - 8 and int(lens.max()) > 0      # short and full rows both occur
