"""GPU: the fp32 MFMA GEMM core and the implicit-GEMM conv against fp64 torch references (through the C-ABI)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run_gemm(layout, M, N, K, lda_pad=0, ldb_pad=0, splits=None, seed=0):
    from wfd_gan_amd._lib import lib
    L = lib()
    g = torch.Generator().manual_seed(seed)
    dev = "cuda:0"
    if layout == 0:
        a_shape, b_shape = (M, K), (N, K)
    elif layout == 1:
        a_shape, b_shape = (M, K), (K, N)
    else:
        a_shape, b_shape = (K, M), (K, N)
    A = torch.randn(a_shape[0], a_shape[1] + lda_pad, generator=g)
    B = torch.randn(b_shape[0], b_shape[1] + ldb_pad, generator=g)
    Ad, Bd = A.to(dev), B.to(dev)
    Av, Bv = A[:, : a_shape[1]].double(), B[:, : b_shape[1]].double()
    ref = {0: lambda: Av @ Bv.T, 1: lambda: Av @ Bv, 2: lambda: Av.T @ Bv}[layout]()
    ns = splits or L.gemm_pick_splits(M, N, K)
    ldp = (N + 3) // 4 * 4
    P = torch.full((ns, M, ldp), float("nan"), device=dev)
    L.gemm_f32(layout, Ad.data_ptr(), A.shape[1], Bd.data_ptr(), B.shape[1], P.data_ptr(), ldp, M * ldp, M, N, K, ns,
               torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = P[:, :, :N].double().sum(0).cpu()
    err = (out - ref).abs().max().item()
    scale = ref.abs().max().item()
    return err / scale


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("shape", [(64, 512, 600), (128, 2048, 1501), (64, 1401, 2048), (8, 41, 21), (128, 1401, 2048),
                                   (2048, 1501, 256), (33, 65, 31), (1, 1, 1)])
def test_gemm_shapes(layout, shape):
    M, N, K = shape
    # tolerance: f32 fma chain, error ~ 1e-7 * sqrt(K) * |a||b| relative to max|C| ~ sqrt(K)
    assert _run_gemm(layout, M, N, K) < 2e-5


@pytest.mark.parametrize("layout", [0, 1, 2])
def test_gemm_unaligned_ld(layout):
    # leading dimensions that are not multiples of 4 take the scalar-load path (W1 is (2048, 1501))
    assert _run_gemm(layout, 70, 130, 1501 if layout != 2 else 77, lda_pad=1, ldb_pad=3) < 2e-5
    assert _run_gemm(layout, 64, 64, 64, lda_pad=0, ldb_pad=1, splits=2) < 2e-5


def test_gemm_asymmetric_identity():
    """A = I with an asymmetric B catches a transposed C/D fragment map."""
    from wfd_gan_amd._lib import lib
    L = lib()
    n = 64
    A = torch.eye(n, device="cuda:0")
    B = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 97).to("cuda:0")
    P = torch.zeros(n, n, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    L.gemm_f32(1, A.data_ptr(), n, B.data_ptr(), n, P.data_ptr(), n, 0, n, n, n, 1, st)
    torch.cuda.synchronize()
    assert torch.equal(P, B)
    L.gemm_f32(0, A.data_ptr(), n, B.data_ptr(), n, P.data_ptr(), n, 0, n, n, n, 1, st)
    torch.cuda.synchronize()
    assert torch.equal(P, B.T.contiguous())
    L.gemm_f32(2, B.data_ptr(), n, A.data_ptr(), n, P.data_ptr(), n, 0, n, n, n, 1, st)
    torch.cuda.synchronize()
    assert torch.equal(P, B.T.contiguous())


@pytest.mark.parametrize("cfg", [(4, 50, 32, 64), (2, 22, 128, 128), (3, 3, 128, 256), (2, 175, 64, 64)])
@pytest.mark.parametrize("pad", [3, 4])
def test_conv8_implicit_gemm(cfg, pad):
    from wfd_gan_amd._lib import lib
    import torch.nn.functional as F
    L = lib()
    Bn, Ln, Cin, Cout = cfg
    g = torch.Generator().manual_seed(1)
    x = torch.randn(Bn, Cin, Ln, generator=g)
    w = torch.randn(Cout, Cin, 8, generator=g)
    ref = F.conv1d(F.pad(x.double(), (pad, 7 - pad)), w.double())        # (Bn, Cout, Ln)
    xcl = x.permute(0, 2, 1).contiguous().to("cuda:0")                   # channels-last
    wp = w.permute(0, 2, 1).reshape(Cout, 8 * Cin).contiguous().to("cuda:0")
    M = Bn * Ln
    ns = L.gemm_pick_splits(M, Cout, 8 * Cin)
    P = torch.zeros(ns, M, Cout, device="cuda:0")
    L.conv8_gemm_f32(xcl.data_ptr(), Bn, Ln, Cin, pad, wp.data_ptr(), Cout, P.data_ptr(), Cout, M * Cout, ns,
                     torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = P.double().sum(0).view(Bn, Ln, Cout).permute(0, 2, 1).cpu()
    assert (out - ref).abs().max().item() / ref.abs().max().item() < 2e-5


@pytest.mark.parametrize("shape", [(500, 32, 256), (70, 17, 100), (64, 1, 64), (4096, 32, 64)])
@pytest.mark.parametrize("layout", [0, 1, 2])
def test_gemm_narrow_n(layout, shape):
    """N <= 32 runs the k-split wave arrangement (both wave columns on the same 32 output columns)."""
    M, N, K = shape
    assert _run_gemm(layout, M, N, K) < 2e-5
    assert _run_gemm(layout, M, N, K, splits=1) < 2e-5


@pytest.mark.parametrize("N", [32, 64, 100])
def test_gemm_fused_epilogues(N):
    from wfd_gan_amd._lib import lib
    L = lib()
    g = torch.Generator().manual_seed(3)
    M, K = 300, 96
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    scale, shift = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    D = torch.randn(M, N, generator=g)
    raw = A.double() @ B.double().T
    Ad, Bd, sd, hd, Dd = (t.to("cuda:0") for t in (A, B, scale, shift, D))
    out = torch.zeros(M, N, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for act, fn in ((1, torch.relu), (2, torch.nn.functional.elu), (3, torch.sigmoid), (0, lambda x: x)):
        L.gemm_f32_fused(0, Ad.data_ptr(), K, Bd.data_ptr(), K, out.data_ptr(), N, M, N, K, 1, sd.data_ptr(),
                         hd.data_ptr(), act, 0, 0, 0, st)
        want = fn(raw * scale.double() + shift.double())
        assert (out.cpu().double() - want).abs().max().item() < 1e-4
    for dmode, f in ((1, lambda r: r * D.double()), (2, lambda r: r * (D > 0)), (3, lambda r: torch.where(D > 0, r, r * (D.double() + 1)))):
        L.gemm_f32_fused(0, Ad.data_ptr(), K, Bd.data_ptr(), K, out.data_ptr(), N, M, N, K, 2, sd.data_ptr(), 0, 0,
                         Dd.data_ptr(), N, dmode, st)
        want = f(raw) * scale.double()
        assert (out.cpu().double() - want).abs().max().item() < 1e-4


@pytest.mark.parametrize("cfg", [(13, 1400, 32), (40, 437, 64), (130, 129, 96)])
@pytest.mark.parametrize("pad", [3, 4])
def test_conv8_lds_resident_kernel(cfg, pad):
    """Cin = 32 on enough 128-row tiles takes the LDS-resident kernel (input rows + weight tile in LDS): samples whose
    length is no multiple of the tile (taps crossing a sample edge inside a tile must read zero), Cout beyond one
    32-column tile and not a multiple of 64, raw and fused epilogues."""
    from wfd_gan_amd._lib import lib
    import torch.nn.functional as F
    L = lib()
    Bn, Ln, Cout = cfg
    Cin = 32
    M = Bn * Ln
    assert L.conv8_direct(M, Cin, Cout) == 1
    g = torch.Generator().manual_seed(2)
    x = torch.randn(Bn, Cin, Ln, generator=g)
    w = torch.randn(Cout, Cin, 8, generator=g)
    ref = F.conv1d(F.pad(x.double(), (pad, 7 - pad)), w.double()).permute(0, 2, 1).reshape(M, Cout)
    xcl = x.permute(0, 2, 1).contiguous().to("cuda:0")
    wp = w.permute(0, 2, 1).reshape(Cout, 8 * Cin).contiguous().to("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    P = torch.full((M, Cout), float("nan"), device="cuda:0")
    L.conv8_gemm_f32(xcl.data_ptr(), Bn, Ln, Cin, pad, wp.data_ptr(), Cout, P.data_ptr(), Cout, 0, 1, st)
    torch.cuda.synchronize()
    scale_ref = ref.abs().max().item()
    assert (P.cpu().double() - ref).abs().max().item() / scale_ref < 2e-5
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    D = torch.randn(M, Cout, generator=g)
    sd, hd, Dd = scale.to("cuda:0"), shift.to("cuda:0"), D.to("cuda:0")
    out = torch.full((M, Cout), float("nan"), device="cuda:0")
    L.conv8_gemm_f32_fused(xcl.data_ptr(), Bn, Ln, Cin, pad, wp.data_ptr(), Cout, out.data_ptr(), Cout, 1, sd.data_ptr(),
                           hd.data_ptr(), 2, 0, 0, 0, st)
    want = F.elu(ref * scale.double() + shift.double())
    assert (out.cpu().double() - want).abs().max().item() / scale_ref < 2e-5
    L.conv8_gemm_f32_fused(xcl.data_ptr(), Bn, Ln, Cin, pad, wp.data_ptr(), Cout, out.data_ptr(), Cout, 2, sd.data_ptr(),
                           0, 0, Dd.data_ptr(), Cout, 3, st)
    want = ref * torch.where(D > 0, torch.ones(()), D + 1).double() * scale.double()
    assert (out.cpu().double() - want).abs().max().item() / scale_ref < 2e-5


@pytest.mark.parametrize("shape", [(2048, 1536, 512), (1100, 1401, 2048), (8192, 512, 600), (1031, 3000, 68)])
def test_gemm_nt_tall_tile(shape):
    """Large-M NT products with one split take the 128x128 tile kernel: ragged M / N / K edges, raw and fused."""
    from wfd_gan_amd._lib import lib
    M, N, K = shape
    assert _run_gemm(0, M, N, K, splits=1) < 2e-5
    L = lib()
    g = torch.Generator().manual_seed(5)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    scale, shift = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    Ad, Bd, sd, hd = (t.to("cuda:0") for t in (A, B, scale, shift))
    ldo = (N + 3) // 4 * 4
    out = torch.full((M, ldo), float("nan"), device="cuda:0")
    L.gemm_f32_fused(0, Ad.data_ptr(), K, Bd.data_ptr(), K, out.data_ptr(), ldo, M, N, K, 1, sd.data_ptr(), hd.data_ptr(), 1,
                     0, 0, 0, torch.cuda.current_stream().cuda_stream)
    want = torch.relu((A.double() @ B.double().T) * scale.double() + shift.double())
    got = out[:, :N].cpu().double()
    assert torch.isfinite(got).all()
    assert (got - want).abs().max().item() / want.abs().max().item() < 2e-5


@pytest.mark.parametrize("lay2", [0, 1])
def test_gemm_pair_equals_separate_launches(lay2):
    """wfd_gemm_f32_pair (a TN weight gradient and an independent NT / NN product in one grid) runs the same tile code as
    the two separate launches: bit-identical outputs, ragged shapes included."""
    from wfd_gan_amd._lib import lib
    L = lib()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(11)
    for (M1, N1, K1), (M2, N2, K2), ns in [((300, 200, 96), (70, 130, 333), 3), ((2048, 1501, 256), (128, 2048, 1401), 8),
                                           ((64, 64, 64), (1, 33, 40), 1)]:
        pad = lambda n: (n + 3) // 4 * 4
        A1, B1 = torch.randn(K1, pad(M1), generator=g).to(dev), torch.randn(K1, pad(N1), generator=g).to(dev)
        A2 = torch.randn(M2, pad(K2), generator=g).to(dev)
        B2 = (torch.randn(N2, pad(K2), generator=g) if lay2 == 0 else torch.randn(K2, pad(N2), generator=g)).to(dev)
        ldp1, ldp2 = pad(N1), pad(N2)
        Pa, Pb = torch.zeros(M1, ldp1, device=dev), torch.zeros(M1, ldp1, device=dev)
        Sa, Sb = torch.zeros(ns, M2, ldp2, device=dev), torch.zeros(ns, M2, ldp2, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        L.gemm_f32(2, A1.data_ptr(), A1.shape[1], B1.data_ptr(), B1.shape[1], Pa.data_ptr(), ldp1, 0, M1, N1, K1, 1, st)
        L.gemm_f32(lay2, A2.data_ptr(), A2.shape[1], B2.data_ptr(), B2.shape[1], Sa.data_ptr(), ldp2, M2 * ldp2, M2, N2, K2,
                   ns, st)
        L.gemm_f32_pair(A1.data_ptr(), A1.shape[1], B1.data_ptr(), B1.shape[1], Pb.data_ptr(), ldp1, M1, N1, K1, lay2,
                        A2.data_ptr(), A2.shape[1], B2.data_ptr(), B2.shape[1], Sb.data_ptr(), ldp2, M2 * ldp2, M2, N2, K2,
                        ns, st)
        torch.cuda.synchronize()
        assert torch.equal(Pa, Pb) and torch.equal(Sa, Sb)
        ref = A1[:, :M1].double().T @ B1[:, :N1].double()
        assert (Pb[:, :N1].double() - ref).abs().max().item() / ref.abs().max().item() < 2e-5
