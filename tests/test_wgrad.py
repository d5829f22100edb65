"""nqa_wgrad (nequip_amd/csrc/wgrad.hip): split-K parameter gradients dW = A^T B against float64 einsum -- the weight-side
backward of ScalarLinearLayer (nequip/nn/mlp.py:262-268), e3nn o3.Linear (interaction_block.py:82-87,129-138) and the
per-type pre-contracted self-connection (:142-146), as restated in oracle/ (the oracle's autograd produces exactly these
einsums)."""
import pytest
import torch


@pytest.mark.gpu
@pytest.mark.parametrize("E,M,N", [(1000, 128, 704), (4133, 128, 192), (777, 8, 128), (63, 40, 72), (5, 64, 64),
                                   (100000, 128, 96)])
def test_wgrad_dense(device, E, M, N):
    from nequip_amd.utils import wgrad as wg

    torch.manual_seed(E + M + N)
    a, b = torch.randn(E, M), torch.randn(E, N)
    ref = a.double().t() @ b.double()
    got = wg.wgrad(a.to(device), b.to(device), wg.WgradTable([(0, 0, M, N, 1, 0)], M * N)).view(M, N)
    torch.testing.assert_close(got.cpu().double(), ref, atol=2e-6 * float(ref.abs().max()) * max(1.0, (E / 1000) ** 0.5),
                               rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("T", [1, 5])
@pytest.mark.parametrize("Z", [1, 300, 8192])
def test_wgrad_irreps_blocks(device, T, Z):
    """Several (input block -> output block) matrices with d = 2l+1 components, per atom type."""
    from nequip_amd.utils import wgrad as wg

    torch.manual_seed(Z + T)
    blocks_in = [(64, 1), (64, 3), (32, 5), (16, 7)]   # (mul, d)
    blocks_out = [(96, 1), (64, 3), (64, 5), (8, 7)]
    din = sum(m * d for m, d in blocks_in)
    dout = sum(m * d for m, d in blocks_out)
    x, g = torch.randn(Z, din), torch.randn(Z, dout)
    types = torch.randint(0, T, (Z,))
    recs, ref_parts = [], []
    io = oo = wo = 0
    onehot = torch.nn.functional.one_hot(types, T).double()
    for (mi, d), (mo, _) in zip(blocks_in, blocks_out):
        recs.append((io, oo, mi, mo, d, wo))
        xb = x[:, io:io + mi * d].double().view(Z, mi, d)
        gb = g[:, oo:oo + mo * d].double().view(Z, mo, d)
        ref_parts.append(torch.einsum("zt,zum,zwm->tuw", onehot, xb, gb).reshape(T, mi * mo))
        io, oo, wo = io + mi * d, oo + mo * d, wo + mi * mo
    ref = torch.cat(ref_parts, dim=1)
    got = wg.wgrad(x.to(device), g.to(device), wg.WgradTable(recs, wo), types.to(device) if T > 1 else None, T)
    assert got.shape == (T, wo)
    torch.testing.assert_close(got.cpu().double(), ref, atol=3e-6 * max(1.0, float(ref.abs().max())), rtol=1e-4)


@pytest.mark.gpu
def test_wgrad_partial_coverage_and_empty(device):
    from nequip_amd.utils import wgrad as wg

    a, b = torch.randn(50, 16, device=device), torch.randn(50, 24, device=device)
    tab = wg.WgradTable([(0, 0, 16, 24, 1, 10)], 16 * 24 + 30)  # gap before and after the matrix stays zero
    got = wg.wgrad(a, b, tab)[0]
    assert torch.count_nonzero(got[:10]) == 0 and torch.count_nonzero(got[10 + 16 * 24:]) == 0
    torch.testing.assert_close(got[10:10 + 16 * 24].view(16, 24), a.t() @ b, atol=1e-4, rtol=1e-4)
    empty = wg.wgrad(a[:0], b[:0], wg.WgradTable([(0, 0, 16, 24, 1, 0)], 16 * 24))
    assert empty.shape == (1, 16 * 24) and torch.count_nonzero(empty) == 0
    with pytest.raises(RuntimeError):
        wg.wgrad(a.cpu(), b.cpu(), tab)
