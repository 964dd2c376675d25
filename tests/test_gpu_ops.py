"""GPU: op-level checks of the non-GEMM kernels against plain torch (fp32/fp64 on CPU), through the C-ABI, including
ragged / degenerate shapes."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _L():
    from wfd_gan_amd._lib import lib
    return lib(), torch.cuda.current_stream().cuda_stream


def _slabs(ns, M, N, seed=0):
    g = torch.Generator().manual_seed(seed)
    ldp = (N + 3) // 4 * 4
    P = torch.randn(ns, M, ldp, generator=g)
    return P, ldp


@pytest.mark.parametrize("cfg", [(1, 5, 7), (3, 64, 1024), (16, 128, 2048), (5, 33, 1401)])
def test_ep_lrelu_drop_and_mul(cfg):
    L, st = _L()
    ns, M, N = cfg
    P, ldp = _slabs(ns, M, N)
    g = torch.Generator().manual_seed(1)
    bias = torch.randn(N, generator=g)
    mask = (torch.rand(M, N, generator=g) > 0.2).float() * 1.25
    z = P[:, :, :N].double().sum(0) + bias.double()
    Pd, bd, md = P.to(DEV), bias.to(DEV), mask.to(DEV)
    H, D = torch.zeros(M, N, device=DEV), torch.zeros(M, N, device=DEV)
    L.ep_lrelu_drop(Pd.data_ptr(), ns, M * ldp, ldp, M, N, bd.data_ptr(), md.data_ptr(), N, 0.2, H.data_ptr(), N,
                    D.data_ptr(), N, st)
    want_h = F.leaky_relu(z, 0.2) * mask.double()
    want_d = mask.double() * torch.where(z > 0, 1.0, 0.2)
    assert (H.cpu().double() - want_h).abs().max().item() < 1e-5
    assert (D.cpu().double() - want_d).abs().max().item() < 1e-6
    rs = torch.randn(M, generator=g)
    rsd = rs.to(DEV)
    o1, o2 = torch.zeros(M, N, device=DEV), torch.zeros(M, N, device=DEV)
    L.ep_mul(Pd.data_ptr(), ns, M * ldp, ldp, M, N, D.data_ptr(), N, 1, 0, rsd.data_ptr(), o1.data_ptr(), o2.data_ptr(),
             N, st)
    raw = P[:, :, :N].double().sum(0)
    assert (o1.cpu().double() - raw * want_d).abs().max().item() < 1e-4
    assert (o2.cpu().double() - raw * want_d * rs.double()[:, None]).abs().max().item() < 1e-4


@pytest.mark.parametrize("cfg", [(1, 2, 5), (4, 64, 1024), (16, 64, 2048), (3, 37, 70), (2, 100, 70), (2, 200, 70),
                                 (2, 129, 100), (2, 513, 100), (1, 1000, 1024), (2, 4096, 200)])
def test_bn_train_forward_backward(cfg):
    L, st = _L()
    ns, M, N = cfg
    P, ldp = _slabs(ns, M, N, seed=2)
    g = torch.Generator().manual_seed(3)
    bias, gamma, beta = torch.randn(N, generator=g), torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    rm, rv = torch.randn(N, generator=g) * 0.1, torch.rand(N, generator=g) + 0.5
    z = (P[:, :, :N].double().sum(0) + bias.double()).requires_grad_(True)
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    y = F.batch_norm(z, rm_ref, rv_ref, gamma.double(), beta.double(), True, 0.1, 0.8)
    h = F.relu(y)
    dev = lambda t: t.to(DEV).contiguous()
    Pd, bd, gd, bed, rmd, rvd = map(dev, (P, bias, gamma, beta, rm, rv))
    XH, H = torch.zeros(M, N, device=DEV), torch.zeros(M, N, device=DEV)
    mean, m2, istd = (torch.zeros(N, device=DEV) for _ in range(3))
    L.bn_train_relu(Pd.data_ptr(), ns, M * ldp, ldp, M, N, bd.data_ptr(), 0.8, gd.data_ptr(), bed.data_ptr(),
                    XH.data_ptr(), H.data_ptr(), N, mean.data_ptr(), m2.data_ptr(), istd.data_ptr(), rmd.data_ptr(),
                    rvd.data_ptr(), 0.1, 1, st)
    assert (H.cpu().double() - h.detach()).abs().max().item() < 1e-5
    assert (rmd.cpu().double() - rm_ref).abs().max().item() < 1e-6
    assert (rvd.cpu().double() - rv_ref).abs().max().item() < 1e-5
    # split form (data-parallel) agrees with the fused one
    Z2, XH2, H2 = (torch.zeros(M, N, device=DEV) for _ in range(3))
    mean2, m22, istd2 = (torch.zeros(N, device=DEV) for _ in range(3))
    L.bn_stats(Pd.data_ptr(), ns, M * ldp, ldp, M, N, bd.data_ptr(), Z2.data_ptr(), N, mean2.data_ptr(), m22.data_ptr(), st)
    L.bn_apply_relu(Z2.data_ptr(), N, M, N, mean2.data_ptr(), m22.data_ptr(), float(M), 0.8, gd.data_ptr(), bed.data_ptr(),
                    XH2.data_ptr(), H2.data_ptr(), N, istd2.data_ptr(), 0, 0, 0.1, 0, st)
    if M <= 512:
        assert torch.equal(H, H2) and torch.equal(XH, XH2)
    else:       # tall batches: chunk statistics combined pairwise instead of the two-pass sum -- equal to rounding
        assert (H - H2).abs().max().item() < 1e-5 and (XH - XH2).abs().max().item() < 1e-5
    # backward
    dh = torch.randn(1, M, ldp, generator=g)
    (h * dh[0, :, :N].double()).sum().backward()
    dhd = dh.to(DEV)
    sums = torch.zeros(2 * N, device=DEV)
    DZ = torch.zeros(M, N, device=DEV)
    dg, db, dbias = (torch.zeros(N, device=DEV) for _ in range(3))
    L.bn_bwd(dhd.data_ptr(), 1, 0, ldp, M, N, H.data_ptr(), XH.data_ptr(), N, gd.data_ptr(), istd.data_ptr(), float(M),
             sums.data_ptr(), DZ.data_ptr(), N, dg.data_ptr(), db.data_ptr(), dbias.data_ptr(), 3, st)
    scale = max(z.grad.abs().max().item(), 1e-6)
    assert (DZ.cpu().double() - z.grad).abs().max().item() < 2e-5 * max(scale, 1.0)
    # tall form (row chunks in parallel, ordered combine): same dz, dgamma, dbeta, and a bias gradient of rounding size
    sums2, DZ2 = torch.zeros(2 * N, device=DEV), torch.zeros(M, N, device=DEV)
    dg2, db2, dbias2 = (torch.zeros(N, device=DEV) for _ in range(3))
    part = torch.zeros(2 * ((M + 127) // 128) * N, device=DEV)
    L.bn_bwd_tall(dhd.data_ptr(), 1, 0, ldp, M, N, H.data_ptr(), XH.data_ptr(), N, gd.data_ptr(), istd.data_ptr(), float(M),
                  sums2.data_ptr(), DZ2.data_ptr(), N, dg2.data_ptr(), db2.data_ptr(), dbias2.data_ptr(), part.data_ptr(), st)
    assert (DZ2.cpu().double() - z.grad).abs().max().item() < 2e-5 * max(scale, 1.0)
    tol = 1e-4 * max(1.0, dh.abs().max().item()) * (M ** 0.5)
    assert (dg2 - dg).abs().max().item() < tol and (db2 - db).abs().max().item() < tol
    assert (sums2 - sums).abs().max().item() < tol and dbias2.abs().max().item() < tol


@pytest.mark.parametrize("cfg", [(2, 40, 32), (3, 1400, 32), (2, 175, 64), (4, 3, 256), (1, 8, 4), (2, 9, 4)])
def test_pool8_forward_backward(cfg):
    L, st = _L()
    Bn, Ln, C = cfg
    g = torch.Generator().manual_seed(4)
    y = torch.randn(Bn, C, Ln, generator=g)
    y = torch.where(torch.rand(Bn, C, Ln, generator=g) < 0.3, torch.zeros(()), y).double().requires_grad_(True)   # ties at 0
    ypad = F.pad(y, (3, 4))
    po = F.max_pool1d(ypad, 8)
    Lout = po.shape[-1]
    assert Lout == (Ln + 7) // 8
    ycl = y.detach().float().permute(0, 2, 1).contiguous().to(DEV)
    PO = torch.zeros(Bn, Lout, C, device=DEV)
    L.df_pool8_fwd(ycl.data_ptr(), Bn, Ln, C, PO.data_ptr(), st)
    assert torch.equal(PO.cpu().permute(0, 2, 1).double(), po.detach().float().double())
    dpo = torch.randn(Bn, C, Lout, generator=g).double()
    # reference: d/dy of sum(dpo * pool(relu-output y)) times relu'(y) times scale
    scale = torch.rand(C, generator=g) + 0.5
    (po * dpo).sum().backward()
    want = y.grad * (y.detach() > 0) * scale.double()[None, :, None]
    dC = torch.full((Bn, Ln, C), float("nan"), device=DEV)
    dpod, sd = dpo.float().permute(0, 2, 1).contiguous().to(DEV), scale.to(DEV)      # keep the device copies alive
    L.df_pool8_bwd_act(dpod.data_ptr(), ycl.data_ptr(), Bn, Ln, C, sd.data_ptr(), 1, dC.data_ptr(), st)
    got = dC.cpu().permute(0, 2, 1).double()
    assert torch.isfinite(got).all()
    assert (got - want).abs().max().item() < 1e-5


@pytest.mark.parametrize("cfg", [(3, 40, 32), (2, 1400, 32), (1, 7, 32), (2, 65, 16)])
def test_conv1_forward_backward(cfg):
    L, st = _L()
    Bn, Ln, Cout = cfg
    g = torch.Generator().manual_seed(5)
    x = torch.rand(Bn, 1, Ln, generator=g).double().requires_grad_(True)
    w = torch.randn(Cout, 1, 8, generator=g)
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    conv = F.conv1d(F.pad(x, (3, 4)), w.double())
    y = F.elu(conv * scale.double()[None, :, None] + shift.double()[None, :, None])
    ldx = Ln + 5
    X = torch.zeros(Bn, ldx)
    X[:, 1:1 + Ln] = x.detach().float()[:, 0]
    Xd, wd, sd, hd = X.to(DEV), w.reshape(Cout, 8).contiguous().to(DEV), scale.to(DEV), shift.to(DEV)
    Y = torch.zeros(Bn, Ln, Cout, device=DEV)
    L.df_conv1_fwd(Xd.data_ptr(), ldx, 1, Bn, Ln, wd.data_ptr(), sd.data_ptr(), hd.data_ptr(), 2, Y.data_ptr(), Cout, st)
    assert (Y.cpu().permute(0, 2, 1).double() - y.detach()).abs().max().item() < 2e-5
    dc = torch.randn(Bn, Cout, Ln, generator=g).double()
    (conv * dc).sum().backward()
    dX = torch.zeros(Bn, Ln + 3, device=DEV)
    dcd = dc.float().permute(0, 2, 1).contiguous().to(DEV)
    L.df_conv1_bwd(dcd.data_ptr(), Bn, Ln, Cout, wd.data_ptr(), dX.data_ptr(), Ln + 3, st)
    assert (dX.cpu()[:, :Ln].double() - x.grad[:, 0]).abs().max().item() < 1e-4


@pytest.mark.parametrize("cfg", [(8, 5), (64, 100), (3, 257)])
def test_masked_ce(cfg):
    L, st = _L()
    M, C = cfg
    g = torch.Generator().manual_seed(6)
    logits = (torch.randn(M, C, generator=g) * 3).double().requires_grad_(True)
    labels = torch.randint(0, C, (M,), generator=g)
    fv = torch.randn(M, generator=g)
    sel = fv > 0
    for force_none in (False, True):
        if force_none:
            fv = -fv.abs() - 0.1
            sel = fv > 0
        nsel = float(sel.sum())
        ldl = (C + 3) // 4 * 4
        lg = torch.zeros(M, ldl)
        lg[:, :C] = logits.detach().float()
        lgd, labd, fvd = lg.to(DEV), labels.to(torch.int32).to(DEV), fv.to(DEV)
        nd = torch.tensor([nsel], device=DEV)
        ce, dl = torch.zeros(M, device=DEV), torch.zeros(M, ldl, device=DEV)
        L.masked_ce(lgd.data_ptr(), ldl, M, C, labd.data_ptr(), fvd.data_ptr(), nd.data_ptr(), 0.02, ce.data_ptr(),
                    dl.data_ptr(), ldl, st)
        if nsel > 0:
            loss = F.cross_entropy(logits[sel], labels[sel])
            logits.grad = None
            (0.02 * loss).backward()
            assert abs(ce.sum().item() / nsel - loss.item()) < 1e-5 * max(1.0, abs(loss.item()))
            assert (dl.cpu()[:, :C].double() - logits.grad).abs().max().item() < 1e-6
        else:
            assert ce.abs().sum().item() == 0.0 and dl.abs().sum().item() == 0.0


@pytest.mark.parametrize("mode", [0, 1])
def test_penalty_row(mode):
    L, st = _L()
    ns, M, S = 5, 9, 1401
    P, ldp = _slabs(ns, M, S, seed=7)
    P = P * 0.05
    g = P[:, :, :S].double().sum(0).requires_grad_(True)
    n = g.pow(2).sum(1)
    flag = torch.ones(M)
    flag[3] = 0
    pen = (n ** 3 if mode == 0 else (n.sqrt() - 1) ** 2) * flag.double()
    loss = pen.sum() / 4.0 * (1.0 if mode == 0 else 10.0)
    loss.backward()
    Pd, fd = P.to(DEV), flag.to(DEV)
    nsq, pv = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
    GH = torch.zeros(M, ldp, device=DEV)
    L.ep_penalty_row(Pd.data_ptr(), ns, M * ldp, ldp, M, S, mode, 0.25, 10.0, fd.data_ptr(), nsq.data_ptr(), pv.data_ptr(),
                     GH.data_ptr(), ldp, st)
    assert (pv.cpu().double() - pen.detach()).abs().max().item() <= 1e-4 * pen.detach().abs().max().item()
    assert (GH.cpu()[:, :S].double() - g.grad).abs().max().item() <= 1e-4 * g.grad.abs().max().item()


def test_rmsprop_and_colsum_multi():
    L, st = _L()
    g = torch.Generator().manual_seed(8)
    n = 10007 * 4
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    ref_p = p.clone().requires_grad_(True)
    opt = torch.optim.RMSprop([ref_p], lr=2e-4)
    pd, gd, sq = p.to(DEV), gr.to(DEV), torch.zeros(n, device=DEV)
    state = torch.zeros(2, dtype=torch.int64, device=DEV)
    for it in range(3):
        ref_p.grad = gr.clone() * (it + 1)
        opt.step()
        gd.copy_(gr * (it + 1))
        L.rmsprop(pd.data_ptr(), gd.data_ptr(), sq.data_ptr(), n, 2e-4, 0.99, 1e-8, 1.0, state.data_ptr(), 5, 7, st)
    assert (pd.cpu() - ref_p.detach()).abs().max().item() < 1e-6
    assert state.tolist() == [15, 21]
    # column sums with two sources, row weights and an accumulator
    M, N = 37, 70
    X, X2 = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    rs, rs2 = torch.randn(M, generator=g), torch.randn(M, generator=g)
    Xd, X2d, rsd, rs2d = X.to(DEV), X2.to(DEV), rs.to(DEV), rs2.to(DEV)
    out, out2, acc = torch.zeros(N, device=DEV), torch.zeros(1, device=DEV), torch.ones(1, device=DEV)
    v = torch.randn(M, generator=g)
    vd = v.to(DEV)
    desc = torch.tensor([[Xd.data_ptr(), N, rsd.data_ptr(), X2d.data_ptr(), N, rs2d.data_ptr(), M, N, out.data_ptr(), 0, 0],
                         [vd.data_ptr(), 1, rsd.data_ptr(), 0, 0, 0, M, 1, out2.data_ptr(), 3, acc.data_ptr()]],
                        dtype=torch.int64, device=DEV)
    L.colsum_multi(desc.data_ptr(), 2, 4, st)
    want = (X.double() * rs.double()[:, None]).sum(0) + (X2.double() * rs2.double()[:, None]).sum(0)
    assert (out.cpu().double() - want).abs().max().item() < 1e-4
    w2 = float((v.double() * rs.double()).sum())
    assert abs(out2.item() - w2) < 1e-4 and abs(acc.item() - (1.0 + w2)) < 1e-4


def test_bad_arguments_are_rejected():
    from wfd_gan_amd._lib import WfdError
    L, st = _L()
    t = torch.zeros(64, 64, device=DEV)
    with pytest.raises(WfdError):
        L.gemm_f32(0, t.data_ptr(), 64, t.data_ptr(), 64, t.data_ptr(), 32, 0, 64, 64, 64, 1, st)      # ldp < N
    with pytest.raises(WfdError):
        L.gemm_f32(7, t.data_ptr(), 64, t.data_ptr(), 64, t.data_ptr(), 64, 0, 64, 64, 64, 1, st)      # bad layout
    with pytest.raises(WfdError):
        L.gemm_f32(0, 0, 64, t.data_ptr(), 64, t.data_ptr(), 64, 0, 64, 64, 64, 1, st)                 # null pointer
    with pytest.raises(WfdError):
        L.gemm_f32(0, t.data_ptr(), 64, t.data_ptr(), 64, t.data_ptr(), 64, 0, 0, 64, 64, 1, st)       # empty M
    with pytest.raises(WfdError):
        L.gemm_bf16_nt(t.data_ptr(), 64, t.data_ptr(), 64, 64, 64, 60, t.data_ptr(), 64, 0, 0, 0, 0, 0, st)   # K % 8


@pytest.mark.parametrize("shape", [(1000, 41), (70001, 33), (5, 1401)])
def test_minmax_fit_transform_on_device_is_bit_identical(shape):
    """MinMaxScaler().fit_transform + .float() (src/mlp_df_wgan_train.py:132-133,184) on the device vs sklearn."""
    from sklearn import preprocessing
    from wfd_gan_amd import utils
    rng = np.random.default_rng(0)
    N, S = shape
    X = rng.integers(-5, 2000, size=(N, S)).astype(np.float64) + rng.random((N, S)) * (rng.random(S) < 0.3)
    X[:, 3] = 7.0                                   # constant column: scale_ = 1
    X[:, 4] = 0.0
    sk = preprocessing.MinMaxScaler()
    want = torch.from_numpy(sk.fit_transform(X)).float()
    got, scaler = utils.fit_transform_device(X, torch.device(DEV))
    assert torch.equal(got.cpu(), want)
    for attr in ("data_min_", "data_max_", "data_range_", "scale_", "min_"):
        assert np.array_equal(getattr(scaler, attr), getattr(sk, attr)), attr
    assert scaler.n_samples_seen_ == N and scaler.n_features_in_ == S
    y = rng.random((3, S)).astype(np.float32)
    assert np.array_equal(scaler.inverse_transform(y.copy()), sk.inverse_transform(y.copy()))


@pytest.mark.parametrize("shape", [(128, 512, 8), (192, 512, 16), (7, 37, 3), (64, 512, 1)])
def test_d_head_lrelu_equals_the_two_launches(shape):
    """wfd_d_head_lrelu = wfd_ep_lrelu_drop on the layer-3 slabs followed by wfd_d_head, in one launch: bit-identical."""
    from wfd_gan_amd._lib import lib
    L = lib()
    M, K, ns = shape
    dev = "cuda:0"
    g = torch.Generator().manual_seed(17)
    ldp = (K + 3) // 4 * 4
    P = torch.randn(ns, M, ldp, generator=g).to(dev)
    bias, w4, b4 = torch.randn(K, generator=g).to(dev), torch.randn(K, generator=g).to(dev), torch.randn(1, generator=g).to(dev)
    mask = ((torch.rand(M, ldp, generator=g) > 0.2).float() * 1.25).to(dev)
    rs = torch.randn(M, generator=g).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for fused in (False, True):
        H, D = torch.full((M, ldp), 9.0, device=dev), torch.full((M, ldp), 9.0, device=dev)
        v, dl, dls = torch.zeros(M, device=dev), torch.full((M, ldp), 9.0, device=dev), torch.full((M, ldp), 9.0, device=dev)
        if fused:
            L.d_head_lrelu(P.data_ptr(), ns, M * ldp, ldp, bias.data_ptr(), mask.data_ptr(), ldp, 0.2, H.data_ptr(), ldp,
                           D.data_ptr(), ldp, w4.data_ptr(), b4.data_ptr(), M, K, rs.data_ptr(), v.data_ptr(), dl.data_ptr(),
                           dls.data_ptr(), ldp, st)
        else:
            L.ep_lrelu_drop(P.data_ptr(), ns, M * ldp, ldp, M, K, bias.data_ptr(), mask.data_ptr(), ldp, 0.2, H.data_ptr(),
                            ldp, D.data_ptr(), ldp, st)
            L.d_head(H.data_ptr(), ldp, D.data_ptr(), ldp, w4.data_ptr(), b4.data_ptr(), M, K, rs.data_ptr(), v.data_ptr(),
                     dl.data_ptr(), dls.data_ptr(), ldp, st)
        torch.cuda.synchronize()
        outs.append((H, D, v, dl, dls))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    z = P[:, :, :K].double().sum(0).cpu() + bias.double().cpu()
    h = torch.where(z > 0, z, 0.2 * z) * mask[:, :K].double().cpu()
    ref = h @ w4.double().cpu() + b4.double().cpu()
    assert (outs[1][2].double().cpu() - ref).abs().max().item() <= 1e-4 * (1 + ref.abs().max().item())


@pytest.mark.parametrize("B,C", [(8, 5), (64, 100), (200, 128), (33, 64), (64, 65)])
def test_g_tail_equals_its_three_launches(B, C):
    """wfd_g_tail (one workgroup) against wfd_g_head + wfd_masked_ce + wfd_g_loss on the same logits / labels / critic
    scores (src/mlp_df_wgan_train.py:247-271), and against torch's cross_entropy on the selected rows."""
    L, st = _L()
    g = torch.Generator().manual_seed(B * 1000 + C)
    ldl = (C + 3) // 4 * 4
    logits = torch.randn(B, ldl, generator=g) * 3
    labels = torch.randint(0, C, (B,), generator=g, dtype=torch.int32)
    fv = torch.randn(B, generator=g)
    ld, lb, fd = logits.to(DEV), labels.to(DEV), fv.to(DEV)
    outs = []
    for fused in (True, False):
        ce, dl = torch.zeros(B, device=DEV), torch.zeros(B, ldl, device=DEV)
        head, stats, acc = torch.zeros(2, device=DEV), torch.zeros(4, device=DEV), torch.ones(4, device=DEV)
        if fused:
            L.g_tail(ld.data_ptr(), ldl, B, C, lb.data_ptr(), fd.data_ptr(), 1.0 / B, 0.02, ce.data_ptr(), dl.data_ptr(), ldl,
                     head.data_ptr(), stats.data_ptr(), acc.data_ptr(), st)
        else:
            L.g_head(fd.data_ptr(), B, 1.0 / B, head.data_ptr(), st)
            L.masked_ce(ld.data_ptr(), ldl, B, C, lb.data_ptr(), fd.data_ptr(), head.data_ptr() + 4, 0.02, ce.data_ptr(),
                        dl.data_ptr(), ldl, st)
            L.g_loss(ce.data_ptr(), B, head.data_ptr(), 0.02, stats.data_ptr(), acc.data_ptr(), st)
        torch.cuda.synchronize()
        outs.append((ce.cpu(), dl.cpu()[:, :C], head.cpu(), stats.cpu(), acc.cpu()))
    for a, b, what in zip(outs[0], outs[1], ("ce", "dlogits", "head", "stats", "acc")):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max())), what
    sel = fv > 0
    want = torch.nn.functional.cross_entropy(logits[sel][:, :C].double(), labels[sel].long()) if sel.any() else torch.tensor(0.0)
    assert abs(float(outs[0][3][1]) - float(want)) <= 1e-5 * max(1.0, abs(float(want)))
    assert float(outs[0][3][3]) == float(sel.sum())


@pytest.mark.parametrize("Bn,L,Cout,act", [(16, 1400, 32, 2), (13, 1400, 32, 1), (12, 1408, 64, 2), (64, 1400, 32, 2)])
def test_conv8_fused_pool_matches_the_two_launches(Bn, L, Cout, act):
    """wfd_conv8_gemm_f32_fused_pool: the Cin = 32 conv + folded BatchNorm + activation whose launch also writes the
    following max-pool (k = 8, stride 8, zero padding taking part) from its accumulators, against wfd_conv8_gemm_f32_fused +
    wfd_df_pool8_fwd: the activation and the pooled rows must be bit-identical (the row grid is shifted by three rows, the
    arithmetic per element is not touched), including the first window of every sample (left padding) and samples whose
    last three positions lie in no window."""
    from wfd_gan_amd._lib import lib
    L_ = lib()
    g = torch.Generator().manual_seed(Bn * L + Cout)
    X = torch.randn(Bn * L, 32, generator=g).to(DEV)
    W = (torch.randn(Cout, 256, generator=g) * 0.08).to(DEV)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(Cout, generator=g) * 0.3 - 0.4).to(DEV)          # mostly negative: windows whose maximum is the padding
    st = torch.cuda.current_stream().cuda_stream
    Lout = L // 8
    y1 = torch.full((Bn * L, Cout), 9.0, device=DEV); p1 = torch.full((Bn * Lout, Cout), 9.0, device=DEV)
    y2 = torch.full((Bn * L, Cout), 8.0, device=DEV); p2 = torch.full((Bn * Lout, Cout), 8.0, device=DEV)
    L_.conv8_gemm_f32_fused(X.data_ptr(), Bn, L, 32, 3, W.data_ptr(), Cout, y1.data_ptr(), Cout, 1, scale.data_ptr(),
                            shift.data_ptr(), act, 0, 0, 0, st)
    L_.df_pool8_fwd(y1.data_ptr(), Bn, L, Cout, p1.data_ptr(), st)
    L_.conv8_gemm_f32_fused_pool(X.data_ptr(), Bn, L, 32, 3, W.data_ptr(), Cout, y2.data_ptr(), Cout, scale.data_ptr(),
                                 shift.data_ptr(), act, p2.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    assert torch.equal(p1, p2)
    assert float((p1 == 0).float().mean()) > 0.0005 or act != 1      # relu: some windows are won by a zero
    # rejected shapes fall back in the caller: L % 8 != 0, the small form
    with pytest.raises(Exception):
        L_.conv8_gemm_f32_fused_pool(X.data_ptr(), Bn, L - 4, 32, 3, W.data_ptr(), Cout, y2.data_ptr(), Cout, scale.data_ptr(),
                                     shift.data_ptr(), act, p2.data_ptr(), st)
    with pytest.raises(Exception):
        L_.conv8_gemm_f32_fused_pool(X.data_ptr(), 2, L, 32, 3, W.data_ptr(), Cout, y2.data_ptr(), Cout, scale.data_ptr(),
                                     shift.data_ptr(), act, p2.data_ptr(), st)
