"""GPU: the bf16-MFMA contraction core over fp32 data (gemm_mx.hip; wfd_set_gemm_mode 1 = bf16 operands, 2 = split-bf16
x3) against fp64 torch references, through the same C-ABI entry points as the f32-operand core.

Tolerances (relative to max|C|): mode 2 carries ~2^-17 per product -> 1e-4 bound here (measured ~1e-5); mode 1 is plain
bf16 rounding of both operands (2^-9 each) -> 2e-2."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = {1: 2e-2, 2: 1e-4}


@pytest.fixture
def mode_guard():
    from wfd_gan_amd._lib import lib
    L = lib()
    yield L
    L.set_gemm_mode(0)


def _gemm(L, layout, M, N, K, splits=None, seed=0, int_data=False):
    g = torch.Generator().manual_seed(seed)
    dev = "cuda:0"
    a_shape, b_shape = {0: ((M, K), (N, K)), 1: ((M, K), (K, N)), 2: ((K, M), (K, N))}[layout]
    pad = lambda n: (n + 3) // 4 * 4
    A = torch.zeros(a_shape[0], pad(a_shape[1]))
    B = torch.zeros(b_shape[0], pad(b_shape[1]))
    if int_data:        # small integers: exact in bf16, so every mode must reproduce the product bit for bit
        A[:, : a_shape[1]] = torch.randint(-8, 9, a_shape, generator=g).float()
        B[:, : b_shape[1]] = torch.randint(-8, 9, b_shape, generator=g).float()
    else:
        A[:, : a_shape[1]] = torch.randn(a_shape, generator=g)
        B[:, : b_shape[1]] = torch.randn(b_shape, generator=g)
    Av, Bv = A[:, : a_shape[1]].double(), B[:, : b_shape[1]].double()
    ref = {0: lambda: Av @ Bv.T, 1: lambda: Av @ Bv, 2: lambda: Av.T @ Bv}[layout]()
    ns = splits or L.gemm_pick_splits(M, N, K)
    ldp = pad(N)
    P = torch.full((ns, M, ldp), float("nan"), device=dev)
    Ad, Bd = A.to(dev), B.to(dev)
    L.gemm_f32(layout, Ad.data_ptr(), A.shape[1], Bd.data_ptr(), B.shape[1], P.data_ptr(), ldp, M * ldp, M, N, K, ns,
               torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = P[:, :, :N].double().sum(0).cpu()
    assert torch.isfinite(out).all()
    return (out - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("shape", [(64, 512, 600), (128, 2048, 1501), (128, 1401, 2048), (8, 41, 21), (2048, 1501, 256),
                                   (33, 65, 31), (1, 1, 1), (500, 32, 256), (300, 100, 64)])
def test_mx_shapes(mode_guard, mode, layout, shape):
    L = mode_guard
    L.set_gemm_mode(mode)
    assert L.get_gemm_mode() == mode
    M, N, K = shape
    assert _gemm(L, layout, M, N, K) < TOL[mode]
    assert _gemm(L, layout, M, N, K, splits=1) < TOL[mode]


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("layout", [0, 1, 2])
def test_mx_exact_on_integers(mode_guard, mode, layout):
    """Operands exactly representable in bf16: the fragment maps (row reads and hardware-transposed reads), the k
    order and the ragged edges are right iff the product is exact."""
    L = mode_guard
    L.set_gemm_mode(mode)
    for shape in [(70, 130, 200), (128, 64, 64), (31, 33, 130)]:
        assert _gemm(L, layout, *shape, int_data=True) == 0.0
        assert _gemm(L, layout, *shape, splits=2, int_data=True) == 0.0


def test_mx_asymmetric_identity(mode_guard):
    L = mode_guard
    n = 64
    A = torch.eye(n, device="cuda:0")
    B = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 97).to("cuda:0")
    P = torch.zeros(n, n, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for mode in (1, 2):
        L.set_gemm_mode(mode)
        L.gemm_f32(1, A.data_ptr(), n, B.data_ptr(), n, P.data_ptr(), n, 0, n, n, n, 1, st)
        torch.cuda.synchronize()
        assert torch.equal(P, B)
        L.gemm_f32(0, A.data_ptr(), n, B.data_ptr(), n, P.data_ptr(), n, 0, n, n, n, 1, st)
        torch.cuda.synchronize()
        assert torch.equal(P, B.T.contiguous())
        L.gemm_f32(2, B.data_ptr(), n, A.data_ptr(), n, P.data_ptr(), n, 0, n, n, n, 1, st)
        torch.cuda.synchronize()
        assert torch.equal(P, B.T.contiguous())


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("cfg", [(4, 50, 32, 64), (2, 22, 128, 128), (3, 3, 128, 256), (2, 175, 64, 64), (3, 1400, 32, 32)])
@pytest.mark.parametrize("pad", [3, 4])
def test_mx_conv8(mode_guard, mode, cfg, pad):
    import torch.nn.functional as F
    L = mode_guard
    L.set_gemm_mode(mode)
    Bn, Ln, Cin, Cout = cfg
    g = torch.Generator().manual_seed(1)
    x = torch.randn(Bn, Cin, Ln, generator=g)
    w = torch.randn(Cout, Cin, 8, generator=g)
    ref = F.conv1d(F.pad(x.double(), (pad, 7 - pad)), w.double())
    xcl = x.permute(0, 2, 1).contiguous().to("cuda:0")
    wp = w.permute(0, 2, 1).reshape(Cout, 8 * Cin).contiguous().to("cuda:0")
    M = Bn * Ln
    st = torch.cuda.current_stream().cuda_stream
    ns = L.gemm_pick_splits(M, Cout, 8 * Cin)
    P = torch.zeros(ns, M, Cout, device="cuda:0")
    L.conv8_gemm_f32(xcl.data_ptr(), Bn, Ln, Cin, pad, wp.data_ptr(), Cout, P.data_ptr(), Cout, M * Cout, ns, st)
    torch.cuda.synchronize()
    out = P.double().sum(0).view(Bn, Ln, Cout).permute(0, 2, 1).cpu()
    assert (out - ref).abs().max().item() / ref.abs().max().item() < TOL[mode]
    # fused store (affine + ReLU)
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    O = torch.full((M, Cout), float("nan"), device="cuda:0")
    sd, hd = scale.to("cuda:0"), shift.to("cuda:0")
    L.conv8_gemm_f32_fused(xcl.data_ptr(), Bn, Ln, Cin, pad, wp.data_ptr(), Cout, O.data_ptr(), Cout, 1, sd.data_ptr(),
                           hd.data_ptr(), 1, 0, 0, 0, st)
    torch.cuda.synchronize()
    want = torch.relu(ref.permute(0, 2, 1).reshape(M, Cout) * scale.double() + shift.double())
    got = O.cpu().double()
    assert (got - want).abs().max().item() / want.abs().max().item() < TOL[mode]


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("cfg", [(24, 1400, 32, 32), (190, 175, 32, 64), (190, 175, 64, 32), (190, 175, 64, 64),
                                 (1100, 33, 32, 32)])
@pytest.mark.parametrize("pad", [3, 4])
def test_mx_conv8_lds_resident(mode_guard, mode, cfg, pad):
    """>= 256 chunks of 128 rows with Cin, Cout in {32, 64}: the LDS-resident conv kernel (weights as bf16 in LDS, input
    rows loaded once); raw product, both fused stores, sample edges inside a chunk (L = 33), ragged last chunk."""
    import torch.nn.functional as F
    L = mode_guard
    L.set_gemm_mode(mode)
    Bn, Ln, Cin, Cout = cfg
    M = Bn * Ln
    assert L.conv8_direct_mode(M, Cin, Cout, mode) == 1
    g = torch.Generator().manual_seed(2)
    x = torch.randn(Bn, Cin, Ln, generator=g)
    w = torch.randn(Cout, Cin, 8, generator=g)
    ref = F.conv1d(F.pad(x.double(), (pad, 7 - pad)), w.double()).permute(0, 2, 1).reshape(M, Cout)
    xcl = x.permute(0, 2, 1).contiguous().to("cuda:0")
    wp = w.permute(0, 2, 1).reshape(Cout, 8 * Cin).contiguous().to("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    P = torch.full((M, Cout), float("nan"), device="cuda:0")
    L.conv8_gemm_f32(xcl.data_ptr(), Bn, Ln, Cin, pad, wp.data_ptr(), Cout, P.data_ptr(), Cout, M * Cout, 1, st)
    torch.cuda.synchronize()
    got = P.cpu().double()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() / ref.abs().max().item() < TOL[mode]
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    sd, hd = scale.to("cuda:0"), shift.to("cuda:0")
    O = torch.full((M, Cout), float("nan"), device="cuda:0")
    L.conv8_gemm_f32_fused(xcl.data_ptr(), Bn, Ln, Cin, pad, wp.data_ptr(), Cout, O.data_ptr(), Cout, 1, sd.data_ptr(),
                           hd.data_ptr(), 2, 0, 0, 0, st)                     # affine + ELU
    torch.cuda.synchronize()
    want = F.elu(ref * scale.double() + shift.double())
    assert (O.cpu().double() - want).abs().max().item() / want.abs().max().item() < TOL[mode]
    D = torch.randn(M, Cout, generator=g)
    Dd = D.to("cuda:0")
    L.conv8_gemm_f32_fused(xcl.data_ptr(), Bn, Ln, Cin, pad, wp.data_ptr(), Cout, O.data_ptr(), Cout, 2, sd.data_ptr(), 0,
                           0, Dd.data_ptr(), Cout, 2, st)                      # * [D > 0] * scale
    torch.cuda.synchronize()
    want = ref * (D > 0).double() * scale.double()
    assert (O.cpu().double() - want).abs().max().item() / ref.abs().max().item() < TOL[mode]
