"""with_edge_vectors_ / ForceStressOutput accept any floating dtype for positions, cell and shifts, as the reference's
``with_edge_vectors_`` does (nequip/nn/utils.py:88-114) -- the float64 kernels must never reinterpret float32 or
integer buffers."""
import pytest
import torch


@pytest.mark.gpu
@pytest.mark.parametrize("pos_dtype,shift_dtype", [(torch.float32, torch.float32), (torch.float64, torch.float32),
                                                   (torch.float32, torch.float64), (torch.float64, torch.int64)])
def test_edge_vectors_any_dtype(device, pos_dtype, shift_dtype):
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.nn.utils import with_edge_vectors_
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=3, seed=4)
    data = syn.make_data(pos, types, 4.5, cell)
    ref = with_edge_vectors_(dict(data))  # CPU: the reference's ATen formulation in float64
    d = K.to_device(dict(data), device)
    d[K.POSITIONS_KEY] = d[K.POSITIONS_KEY].to(pos_dtype).requires_grad_(True)
    d[K.CELL_KEY] = d[K.CELL_KEY].to(pos_dtype)
    d[K.EDGE_CELL_SHIFT_KEY] = d[K.EDGE_CELL_SHIFT_KEY].to(shift_dtype)
    out = with_edge_vectors_(d)
    tol = 1e-12 if pos_dtype == torch.float64 else 2e-5
    torch.testing.assert_close(out[K.EDGE_VECTORS_KEY].detach().cpu(), ref[K.EDGE_VECTORS_KEY], atol=tol, rtol=0)
    g = torch.randn(ref[K.EDGE_VECTORS_KEY].shape, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    (gp,) = torch.autograd.grad(out[K.EDGE_VECTORS_KEY], d[K.POSITIONS_KEY], g.to(device))
    assert gp.dtype == pos_dtype
    ei = data[K.EDGE_INDEX_KEY]
    gp_ref = torch.zeros(len(pos), 3, dtype=torch.float64).index_add_(0, ei[1], g).index_add_(0, ei[0], -g)
    torch.testing.assert_close(gp.cpu().double(), gp_ref, atol=1e-12 if pos_dtype == torch.float64 else 1e-5, rtol=0)


@pytest.mark.gpu
def test_edge_vectors_reject_int32_index(device):
    from nequip_amd.nn.utils import _EdgeVectorsFn

    pos = torch.zeros(3, 3, dtype=torch.float64, device=device)
    ei = torch.tensor([[0, 1], [1, 2]], dtype=torch.int32, device=device)
    with pytest.raises(TypeError):
        _EdgeVectorsFn.apply(pos, None, ei, None, None)


@pytest.mark.gpu
def test_float32_positions_through_the_model(device):
    """float32 positions / cell reach the inference force path through the reference formulation (autograd w.r.t. pos):
    same forces as float64 inputs to float32 accuracy."""
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=3, seed=6)
    data = K.to_device(syn.make_data(pos, types, 4.5, cell), device)
    model = NequIPGNNModel(seed=1, model_dtype="float32", r_max=4.5, type_names=names, num_layers=2, l_max=1,
                           parity=False, num_features=8, radial_mlp_depth=1, radial_mlp_width=16,
                           avg_num_neighbors=38.0).to(device).eval()
    out64 = model(dict(data))
    d32 = dict(data)
    d32[K.POSITIONS_KEY] = data[K.POSITIONS_KEY].float()
    d32[K.CELL_KEY] = data[K.CELL_KEY].float()
    d32[K.EDGE_CELL_SHIFT_KEY] = data[K.EDGE_CELL_SHIFT_KEY].float()
    out32 = model(d32)
    f64, f32 = out64[K.FORCE_KEY].detach(), out32[K.FORCE_KEY].detach().double()
    torch.testing.assert_close(f32, f64, atol=2e-4 * max(1.0, float(f64.abs().max())), rtol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(9,), (3, 3), (1,), (16,)])
def test_frame_sum_and_frame_rows(device, shape):
    """`nqa_frame_sum` (ordered per-frame tree sum) against `index_add_`, its adjoint `per_frame[batch]`, and both once
    more through autograd (the training path differentiates them twice); unsorted frame indices, an empty frame."""
    from nequip_amd.nn.utils import frame_rows, frame_sum

    g = torch.Generator().manual_seed(3)
    N, B = 1000, 7
    batch = torch.randint(0, B - 1, (N,), generator=g)  # frame B-1 stays empty
    rows = torch.randn((N,) + shape, generator=g, dtype=torch.float64)
    ref = torch.zeros((B,) + shape, dtype=torch.float64).index_add_(0, batch, rows)
    rows_d = rows.to(device).requires_grad_(True)
    out = frame_sum(rows_d, batch.to(device), B)
    torch.testing.assert_close(out.cpu(), ref, atol=1e-12, rtol=1e-12)
    assert float(out[B - 1].abs().max()) == 0.0
    c = torch.randn((B,) + shape, generator=g, dtype=torch.float64)
    (g_rows,) = torch.autograd.grad(out, rows_d, c.to(device))
    torch.testing.assert_close(g_rows.cpu(), c[batch], atol=0, rtol=0)

    per_frame = torch.randn((B,) + shape, generator=g, dtype=torch.float64).to(device).requires_grad_(True)
    picked = frame_rows(per_frame, batch.to(device))
    torch.testing.assert_close(picked.cpu(), per_frame.detach().cpu()[batch], atol=0, rtol=0)
    w = torch.randn((N,) + shape, generator=g, dtype=torch.float64).to(device).requires_grad_(True)
    (g_pf,) = torch.autograd.grad(picked, per_frame, w, create_graph=True)  # = frame_sum(w)
    torch.testing.assert_close(g_pf.detach().cpu(), torch.zeros((B,) + shape, dtype=torch.float64).index_add_(0, batch, w.detach().cpu()),
                               atol=1e-12, rtol=1e-12)
    (gg,) = torch.autograd.grad(g_pf, w, c.to(device))  # second order: rows of c again
    torch.testing.assert_close(gg.cpu(), c[batch], atol=0, rtol=0)
