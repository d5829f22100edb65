"""Eval-mode weight handling of the fused modules: cached weight images follow state-dict loads and
mode switches, `invalidate_weight_cache()` covers writes that bypass the version counter, and
`eval_parameter_gradients(True)` restores the reference's behaviour of producing parameter gradients in eval mode."""
import pytest
import torch


def _model(device, seed=0):
    from nequip_amd.model import NequIPGNNModel

    return NequIPGNNModel(seed=seed, model_dtype="float32", r_max=4.5, type_names=["H", "O"], num_layers=2, l_max=2,
                          parity=False, num_features=64, radial_mlp_depth=1, radial_mlp_width=128,
                          avg_num_neighbors=38.0).to(device)


def _data(device):
    from nequip_amd.data import AtomicDataDict
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=3, seed=2)
    return AtomicDataDict.to_device(syn.make_data(pos, types, 4.5, cell), device)


@pytest.mark.gpu
def test_cached_weight_images_follow_state_dict_and_data_writes(device):
    data = _data(device)
    a, b = _model(device, 0).eval(), _model(device, 1).eval()
    e_a0 = a(dict(data))["total_energy"].detach().clone()
    e_b = b(dict(data))["total_energy"].detach().clone()
    assert not torch.allclose(e_a0, e_b)
    a.load_state_dict(b.state_dict())  # caches of `a` were built from the old weights
    torch.testing.assert_close(a(dict(data))["total_energy"].detach(), e_b, rtol=1e-6, atol=1e-5)
    # a write that bypasses the version counter + explicit invalidation
    c = _model(device, 0).eval()
    c(dict(data))
    with torch.no_grad():
        for p_c, p_b in zip(c.parameters(), b.parameters()):
            p_c.data.copy_(p_b.data)
    for m in c.modules():
        if hasattr(m, "invalidate_weight_cache"):
            m.invalidate_weight_cache()
    torch.testing.assert_close(c(dict(data))["total_energy"].detach(), e_b, rtol=1e-6, atol=1e-5)
    # train() / eval() round trip drops the caches too
    c.train()
    with torch.no_grad():
        for p in c.parameters():
            p.data.mul_(0.5)
    c.eval()
    e_half = c(dict(data))["total_energy"].detach()
    assert not torch.allclose(e_half, e_b)


@pytest.mark.gpu
def test_eval_mode_parameter_gradients_on_request(device):
    from nequip_amd.utils.wgrad import eval_parameter_gradients

    data = _data(device)
    full = _model(device, 3)
    m = full.model.func  # the energy model inside ForceStressOutput (whose own autograd.grad would consume the graph)
    m.train()
    m.zero_grad(set_to_none=True)
    m(dict(data))["total_energy"].sum().backward()
    ref = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    assert len(ref) > 6
    m.eval()
    m.zero_grad(set_to_none=True)
    m(dict(data))["total_energy"].sum().backward()  # default: weights of the fused modules are constants in eval mode
    missing = [k for k, p in m.named_parameters() if k in ref and p.grad is None]
    assert missing, "the inference fast path produces no gradients for the fused modules' weights"
    try:
        eval_parameter_gradients(True)
        m.zero_grad(set_to_none=True)
        m(dict(data))["total_energy"].sum().backward()
        for k, p in m.named_parameters():
            if k in ref:
                assert p.grad is not None, k
                scale = max(1e-6, float(ref[k].abs().max()))
                torch.testing.assert_close(p.grad, ref[k], rtol=0, atol=2e-4 * scale, msg=lambda s, k=k: f"{k}: {s}")
    finally:
        eval_parameter_gradients(False)


def test_cell_completion_for_non_periodic_directions():
    from nequip_amd.data._nl import _complete_cell

    slab = torch.tensor([[4.0, 0, 0], [1.0, 3.0, 0], [0, 0, 0]], dtype=torch.float64)
    c = _complete_cell(slab, (True, True, False))
    assert torch.equal(c[:2], slab[:2])
    torch.testing.assert_close(c[2], torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64))
    wire = torch.tensor([[0.0, 0, 0], [0, 0, 0], [0.5, 0.5, 7.0]], dtype=torch.float64)
    c = _complete_cell(wire, (False, False, True))
    assert abs(float(torch.linalg.det(c))) > 1e-6
    assert abs(float(c[0] @ c[2])) < 1e-12 and abs(float(c[1] @ c[2])) < 1e-12 and abs(float(c[0] @ c[1])) < 1e-12
    # tilted, non-orthogonal a / b (neither along an axis): the completed vector is orthogonal to BOTH
    tilted = torch.tensor([[3.0, 1.0, 0.5], [1.0, 4.0, -0.7], [0, 0, 0]], dtype=torch.float64)
    c = _complete_cell(tilted, (True, True, False))
    assert abs(float(c[2] @ c[0])) < 1e-12 and abs(float(c[2] @ c[1])) < 1e-12 and abs(float(c[2].norm()) - 1) < 1e-12
    with pytest.raises(ValueError):
        _complete_cell(slab, (True, True, True))
    with pytest.raises(ValueError):
        _complete_cell(torch.tensor([[1.0, 0, 0], [2.0, 0, 0], [0, 0, 1.0]], dtype=torch.float64), (True, True, True))
    full = torch.eye(3, dtype=torch.float64) * 5
    assert _complete_cell(full, (True, True, True)) is full


@pytest.mark.gpu
def test_neighbor_list_slab_with_zero_cell_vector(device):
    """pbc = (T, T, F) with a zero c vector (ASE slab): same list as with any finite non-periodic c."""
    from nequip_amd.data._nl import _compute_neighborlist_single_frame

    g = torch.Generator().manual_seed(0)
    pos = torch.rand(60, 3, generator=g, dtype=torch.float64) * torch.tensor([6.0, 6.0, 3.0], dtype=torch.float64)
    cell0 = torch.tensor([[6.0, 0, 0], [0, 6.0, 0], [0, 0, 0]], dtype=torch.float64)
    cell1 = torch.tensor([[6.0, 0, 0], [0, 6.0, 0], [0, 0, 50.0]], dtype=torch.float64)
    ei0, sh0 = _compute_neighborlist_single_frame(pos.to(device), 3.0, cell=cell0.to(device), pbc=(True, True, False))
    ei1, sh1 = _compute_neighborlist_single_frame(pos.to(device), 3.0, cell=cell1.to(device), pbc=(True, True, False))
    key0 = sorted(zip(ei0[0].tolist(), ei0[1].tolist(), map(tuple, sh0.round().long().tolist())))
    key1 = sorted(zip(ei1[0].tolist(), ei1[1].tolist(), map(tuple, sh1.round().long().tolist())))
    assert len(key0) > 0 and key0 == key1
