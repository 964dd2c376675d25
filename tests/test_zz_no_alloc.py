"""GPU: wfd_colsum_tall_static, the allocation-free variant of wfd_colsum_tall (its chunk partials live in a static device
array inside the library), called through the C-ABI and compared bit for bit with wfd_colsum_tall -- two calls in a row, the
second on a smaller input (the array keeps the first call's contents), plus the size limit.  Collected last on purpose."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _both(L, st, X, M, N, ld, rpc):
    nchunk = (M + rpc - 1) // rpc
    part = torch.full((nchunk * N,), float("nan"), device=DEV)
    want, got = torch.zeros(N, device=DEV), torch.full((N,), float("nan"), device=DEV)
    L.colsum_tall(X.data_ptr(), ld, M, N, rpc, part.data_ptr(), want.data_ptr(), st)
    L.colsum_tall_static(X.data_ptr(), ld, M, N, rpc, got.data_ptr(), st)
    torch.cuda.synchronize()
    return want.cpu(), got.cpu()


def test_colsum_tall_static_matches_workspace_form_twice():
    from wfd_gan_amd._lib import WfdError, lib
    L, st = lib(), torch.cuda.current_stream().cuda_stream
    cap = L.static_scratch_floats()
    assert cap == 1 << 20
    g = torch.Generator().manual_seed(11)
    # first call: 70 chunks x 1401 columns (the generator's output width), ragged last chunk, padded rows
    M, N, ld, rpc = 8900, 1401, 1404, 128
    X = torch.randn(M, ld, generator=g).to(DEV)
    want, got = _both(L, st, X, M, N, ld, rpc)
    assert torch.equal(want, got)
    ref = X[:, :N].double().sum(0).cpu()
    assert (got.double() - ref).abs().max().item() < 1e-2 * ref.abs().max().item()
    # second call, smaller input: must not pick up anything the first call left in the array
    M2, N2, ld2, rpc2 = 300, 37, 40, 64
    X2 = torch.randn(M2, ld2, generator=g).to(DEV)
    want2, got2 = _both(L, st, X2, M2, N2, ld2, rpc2)
    assert torch.equal(want2, got2)
    assert (got2.double() - X2[:, :N2].double().sum(0).cpu()).abs().max().item() < 1e-3
    # beyond the static array: refused with the workspace code, nothing launched
    out = torch.zeros(2048, device=DEV)
    big = torch.zeros(8, 2048, device=DEV)
    with pytest.raises(WfdError, match="code 3"):
        L.colsum_tall_static(big.data_ptr(), 2048, 1 << 20, 2048, 1, out.data_ptr(), st)     # 2 Gi partials > 1 Mi
