"""GPU: the split-inside-the-workgroup contraction (wfd_gemm_f32_wk / wfd_conv8_f32_wk, gemm_wk.hip) against fp64 torch
and against the two-launch form it replaces (wfd_gemm_f32 slabs + wfd_ep_* consumer), through the C-ABI: every layout,
every epilogue, every tile / wave configuration, ragged M / N / K (1401, 1501, 600 are not multiples of anything)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = [(128, 2048, 1501), (64, 512, 600), (192, 1401, 2048), (128, 1024, 512), (5, 7, 33), (33, 100, 512), (64, 100, 40)]


def _L():
    from wfd_gan_amd._lib import lib
    return lib(), torch.cuda.current_stream().cuda_stream


def rup(x, m=4):
    return (x + m - 1) // m * m


def _operands(layout, M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    lda = rup(K) + 4
    A = torch.randn(M, lda, generator=g)
    if layout == 0:
        ldb = rup(K)
        B = torch.randn(N, ldb, generator=g)
        ref = A[:, :K].double() @ B[:, :K].double().t()
    else:
        ldb = rup(N) + 8
        B = torch.randn(K, ldb, generator=g)
        ref = A[:, :K].double() @ B[:, :N].double()
    return A.to(DEV), lda, B.to(DEV), ldb, ref, g


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("shape", SHAPES)
def test_wk_raw_and_epilogues(shape, layout, cfg):
    L, st = _L()
    M, N, K = shape
    A, lda, B, ldb, ref, g = _operands(layout, M, N, K, seed=M + N + K + layout)
    scale_k = ref.abs().max().item()
    ldo = rup(N)
    nul = 0
    # epi 0: raw
    out = torch.full((M, ldo), float("nan"), device=DEV)
    L.gemm_f32_wk(layout, A.data_ptr(), lda, B.data_ptr(), ldb, M, N, K, 0, out.data_ptr(), ldo, nul, 0, nul, nul, 0, nul, 0,
                  0, nul, nul, 0, 0.0, cfg, st)
    torch.cuda.synchronize()
    assert (out[:, :N].double().cpu() - ref).abs().max().item() < 2e-6 * scale_k
    if ldo > N:
        assert torch.isnan(out[:, N:]).all()                      # pad columns untouched
    # epi 1: affine + act
    scale, shift = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    scd, shd = scale.to(DEV), shift.to(DEV)                       # kept alive: the launches below hold raw pointers
    for act, fn in ((0, lambda x: x), (1, F.relu), (2, F.elu), (3, torch.sigmoid)):
        out.fill_(float("nan"))
        L.gemm_f32_wk(layout, A.data_ptr(), lda, B.data_ptr(), ldb, M, N, K, 1, out.data_ptr(), ldo, nul, 0,
                      scd.data_ptr(), shd.data_ptr(), act, nul, 0, 0, nul, nul, 0, 0.0, cfg, st)
        want = fn(ref * scale.double() + shift.double())
        assert (out[:, :N].double().cpu() - want).abs().max().item() < 3e-6 * max(1.0, scale_k)
    # epi 2: chain factor, column scale, second copy with a row scale
    D = torch.randn(M, N, generator=g)
    rs = torch.randn(M, generator=g)
    Dd, rsd = D.to(DEV), rs.to(DEV)
    o1, o2 = torch.full((M, ldo), float("nan"), device=DEV), torch.full((M, ldo), float("nan"), device=DEV)
    for dmode in (0, 1, 2, 3):
        L.gemm_f32_wk(layout, A.data_ptr(), lda, B.data_ptr(), ldb, M, N, K, 2, o1.data_ptr(), ldo, o2.data_ptr(), ldo,
                      scd.data_ptr(), nul, 0, Dd.data_ptr(), N, dmode, rsd.data_ptr(), nul, 0, 0.0, cfg, st)
        d = D.double()
        f = {0: torch.ones_like(d), 1: d, 2: (d > 0).double(), 3: torch.where(d > 0, torch.ones_like(d), d + 1)}[dmode]
        want = ref * f * scale.double()
        assert (o1[:, :N].double().cpu() - want).abs().max().item() < 3e-6 * scale_k * 4
        assert (o2[:, :N].double().cpu() - want * rs.double()[:, None]).abs().max().item() < 3e-6 * scale_k * 16
    # epi 3: bias + LeakyReLU + dropout mask, derivative kept
    mask = (torch.rand(M, N, generator=g) > 0.2).float() * 1.25
    md = mask.to(DEV)
    L.gemm_f32_wk(layout, A.data_ptr(), lda, B.data_ptr(), ldb, M, N, K, 3, o1.data_ptr(), ldo, o2.data_ptr(), ldo, nul,
                  shd.data_ptr(), 0, nul, 0, 0, nul, md.data_ptr(), N, 0.2, cfg, st)
    z = ref + shift.double()
    assert (o1[:, :N].double().cpu() - F.leaky_relu(z, 0.2) * mask.double()).abs().max().item() < 3e-6 * scale_k
    sure = z.abs() > 1e-4 * scale_k                                # away from the kink the derivative is exact
    dd = mask.double() * torch.where(z > 0, 1.0, 0.2)
    assert ((o2[:, :N].double().cpu() - dd).abs() * sure).max().item() < 1e-6


@pytest.mark.parametrize("shape", [(128, 2048, 1501), (64, 1024, 2048), (192, 512, 600)])
def test_wk_agrees_with_two_launch_form(shape):
    """Same product + wfd_ep_lrelu_drop as ONE launch vs slabs + consumer: equal up to fp32 summation order (1e-6 of the
    largest entry), the bar VERDICT r01 #2 sets for replacing the slab round trip."""
    L, st = _L()
    M, N, K = shape
    A, lda, B, ldb, ref, g = _operands(0, M, N, K, seed=9)
    bias = torch.randn(N, generator=g).to(DEV)
    mask = ((torch.rand(M, N, generator=g) > 0.2).float() * 1.25).to(DEV)
    ns = L.gemm_pick_splits(M, N, K)
    ldp = rup(N)
    P = torch.zeros(ns, M, ldp, device=DEV)
    H0, D0, H1, D1 = (torch.zeros(M, N, device=DEV) for _ in range(4))
    L.gemm_f32(0, A.data_ptr(), lda, B.data_ptr(), ldb, P.data_ptr(), ldp, M * ldp, M, N, K, ns, st)
    L.ep_lrelu_drop(P.data_ptr(), ns, M * ldp, ldp, M, N, bias.data_ptr(), mask.data_ptr(), N, 0.2, H0.data_ptr(), N,
                    D0.data_ptr(), N, st)
    L.gemm_f32_wk(0, A.data_ptr(), lda, B.data_ptr(), ldb, M, N, K, 3, H1.data_ptr(), N, D1.data_ptr(), N, 0,
                  bias.data_ptr(), 0, 0, 0, 0, 0, mask.data_ptr(), N, 0.2, 0, st)
    torch.cuda.synchronize()
    assert (H0 - H1).abs().max().item() <= 1e-6 * H0.abs().max().item()
    assert (D0 != D1).float().mean().item() < 1e-5            # a pre-activation within rounding of 0 may flip its slope


@pytest.mark.parametrize("cfg", [0, 1, 5, 3])
@pytest.mark.parametrize("geom", [(3, 175, 32, 64, 3), (2, 22, 64, 128, 3), (5, 3, 128, 256, 3), (2, 40, 64, 32, 4), (1, 7, 32, 48, 4)])
def test_wk_conv8_matches_torch(geom, cfg):
    """k=8 'SAME' conv over channels-last rows (pad 3 = forward, src/model.py:92-105; pad 4 = the transposed form of its
    input gradient) with the folded-BN + activation store."""
    L, st = _L()
    Bn, Ln, Cin, Cout, pad = geom
    g = torch.Generator().manual_seed(Bn * 100 + Ln + Cin)
    X = torch.randn(Bn, Ln, Cin, generator=g)
    W = torch.randn(Cout, Cin, 8, generator=g) * 0.1
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    Wp = W.permute(0, 2, 1).reshape(Cout, 8 * Cin).contiguous()
    xp = F.pad(X.permute(0, 2, 1).double(), (pad, 7 - pad))
    ref = F.conv1d(xp, W.double()).permute(0, 2, 1).reshape(Bn * Ln, Cout)
    want = F.elu(ref * scale.double() + shift.double())
    out = torch.full((Bn * Ln, Cout), float("nan"), device=DEV)
    Xd, Wd, scd, shd = X.to(DEV), Wp.to(DEV), scale.to(DEV), shift.to(DEV)
    L.conv8_f32_wk(Xd.data_ptr(), Bn, Ln, Cin, pad, Wd.data_ptr(), Cout, 1, out.data_ptr(), Cout,
                   scd.data_ptr(), shd.data_ptr(), 2, 0, 0, 0, cfg, st)
    torch.cuda.synchronize()
    assert (out.double().cpu() - want).abs().max().item() < 3e-6 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("shape", [(128, 2048, 1501), (64, 512, 600), (5, 7, 33), (33, 100, 512)])
def test_wk_bf16_pipe_modes(shape, layout, cfg, mode):
    """The same kernel with its fragments converted to bf16 (mode 1) / split-bf16 hi + lo (mode 2, three MFMAs per product)
    on their way out of LDS: gemm_mx.hip's arithmetic.  Stated tolerances against fp64: bf16 operands 2e-2 of the largest
    entry (2^-9 per operand over a random-sign contraction), split-bf16 1e-4 (products ~1e-5 relative); the fused
    LeakyReLU + dropout store is exercised in both."""
    L, st = _L()
    M, N, K = shape
    A, lda, B, ldb, ref, g = _operands(layout, M, N, K, seed=M + N + K + layout)
    scale_k = ref.abs().max().item()
    tol = {1: 2e-2, 2: 1e-4}[mode] * scale_k
    ldo = rup(N)
    out = torch.full((M, ldo), float("nan"), device=DEV)
    shift = torch.randn(N, generator=g)
    mask = (torch.rand(M, N, generator=g) > 0.2).float() * 1.25
    shd, md = shift.to(DEV), mask.to(DEV)
    o1, o2 = torch.full((M, ldo), float("nan"), device=DEV), torch.full((M, ldo), float("nan"), device=DEV)
    L.set_gemm_mode(mode)
    try:
        L.gemm_f32_wk(layout, A.data_ptr(), lda, B.data_ptr(), ldb, M, N, K, 0, out.data_ptr(), ldo, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                      0, 0, 0.0, cfg, st)
        L.gemm_f32_wk(layout, A.data_ptr(), lda, B.data_ptr(), ldb, M, N, K, 3, o1.data_ptr(), ldo, o2.data_ptr(), ldo, 0,
                      shd.data_ptr(), 0, 0, 0, 0, 0, md.data_ptr(), N, 0.2, cfg, st)
        torch.cuda.synchronize()
    finally:
        L.set_gemm_mode(0)
    assert (out[:, :N].double().cpu() - ref).abs().max().item() < tol
    if ldo > N:
        assert torch.isnan(out[:, N:]).all()
    # ... and the bf16 pipe really ran: the same launch in f32 mode gives other bits (f32-input MFMA is ~1e-7 from fp64,
    # the converted operands are not)
    f32 = torch.full((M, ldo), float("nan"), device=DEV)
    L.gemm_f32_wk(layout, A.data_ptr(), lda, B.data_ptr(), ldb, M, N, K, 0, f32.data_ptr(), ldo, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                  0, 0, 0.0, cfg, st)
    torch.cuda.synchronize()
    e_mode = (out[:, :N].double().cpu() - ref).abs().max().item()
    e_f32 = (f32[:, :N].double().cpu() - ref).abs().max().item()
    assert not torch.equal(out[:, :N], f32[:, :N]) and e_mode > 3 * e_f32, (e_mode, e_f32)
    z = ref + shift.double()
    assert (o1[:, :N].double().cpu() - F.leaky_relu(z, 0.2) * mask.double()).abs().max().item() < 1.25 * tol
    sure = z.abs() > 2 * tol
    dd = mask.double() * torch.where(z > 0, 1.0, 0.2)
    assert ((o2[:, :N].double().cpu() - dd).abs() * sure).max().item() < 1e-6


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("geom", [(3, 175, 32, 64, 3), (5, 3, 128, 256, 3), (2, 40, 64, 32, 4)])
def test_wk_conv8_bf16_pipe_modes(geom, mode):
    L, st = _L()
    Bn, Ln, Cin, Cout, pad = geom
    g = torch.Generator().manual_seed(Bn * 100 + Ln + Cin)
    X = torch.randn(Bn, Ln, Cin, generator=g)
    W = torch.randn(Cout, Cin, 8, generator=g) * 0.1
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    Wp = W.permute(0, 2, 1).reshape(Cout, 8 * Cin).contiguous()
    xp = F.pad(X.permute(0, 2, 1).double(), (pad, 7 - pad))
    ref = F.conv1d(xp, W.double()).permute(0, 2, 1).reshape(Bn * Ln, Cout)
    want = F.elu(ref * scale.double() + shift.double())
    out = torch.full((Bn * Ln, Cout), float("nan"), device=DEV)
    Xd, Wd, scd, shd = X.to(DEV), Wp.to(DEV), scale.to(DEV), shift.to(DEV)
    L.set_gemm_mode(mode)
    try:
        L.conv8_f32_wk(Xd.data_ptr(), Bn, Ln, Cin, pad, Wd.data_ptr(), Cout, 1, out.data_ptr(), Cout,
                       scd.data_ptr(), shd.data_ptr(), 2, 0, 0, 0, 1, st)
        torch.cuda.synchronize()
    finally:
        L.set_gemm_mode(0)
    tol = {1: 2e-2, 2: 1e-4}[mode] * 1.5 * max(1.0, ref.abs().max().item())
    assert (out.double().cpu() - want).abs().max().item() < tol


@pytest.mark.parametrize("layout,cfg", [(0, 4), (1, 4), (1, 3), (0, 3)])
def test_wk_multi_equals_separate_launches(layout, cfg):
    """wfd_gemm_f32_wk_multi: three independent products of one layout / configuration (different shapes and epilogues)
    in one launch are bit-identical to their own launches; mismatched configurations are rejected."""
    import ctypes
    import struct
    L, st = _L()
    shapes = [(64, 512, 256), (64, 2048, 1501), (33, 100, 512)]
    keep, calls = [], []
    for k, (M, N, K) in enumerate(shapes):
        A, lda, B, ldb, ref, g = _operands(layout, M, N, K, seed=50 + k)
        ldo = rup(N)
        shift = torch.randn(N, generator=g).to(DEV)
        mask = ((torch.rand(M, N, generator=g) > 0.2).float() * 1.25).to(DEV)
        epi = (1, 3, 0)[k]
        outs = [[torch.full((M, ldo), float("nan"), device=DEV) for _ in range(2)] for _ in range(2)]
        keep += [A, B, shift, mask, outs]
        for o1, o2 in outs:
            calls.append([layout, A.data_ptr(), lda, B.data_ptr(), ldb, M, N, K, epi, o1.data_ptr(), ldo,
                          o2.data_ptr() if epi == 3 else 0, ldo if epi == 3 else 0, 0, shift.data_ptr() if epi else 0,
                          1 if epi == 1 else 0, 0, 0, 0, 0, mask.data_ptr() if epi == 3 else 0, N if epi == 3 else 0, 0.2, cfg])
    for c in calls[0::2]:                                       # separate launches into the first output set
        L.gemm_f32_wk(*c, st)
    words = []
    for c in calls[1::2]:                                       # one launch into the second
        w = list(c)
        w[22] = struct.unpack("<I", struct.pack("<f", 0.2))[0]
        words += [int(v) for v in w]
    arr = (ctypes.c_longlong * len(words))(*words)
    L.gemm_f32_wk_multi(arr, 3, st)
    torch.cuda.synchronize()
    for k in range(3):
        (a1, a2), (b1, b2) = keep[5 * k + 4]
        N = shapes[k][1]
        assert torch.equal(a1[:, :N], b1[:, :N]), k
        if (1, 3, 0)[k] == 3:
            assert torch.equal(a2[:, :N], b2[:, :N]), k
        assert not torch.isnan(b1[:, :N]).any()
    words[23] = 2 if cfg != 2 else 1                             # first product claims another configuration
    bad = (ctypes.c_longlong * len(words))(*words)
    with pytest.raises(Exception):
        L.gemm_f32_wk_multi(bad, 3, st)
