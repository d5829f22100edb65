"""Second-order path (force-matching training): the reference differentiates the forces again
(nequip/nn/grad_output.py:220 create_graph=self.training; nequip/train/lightning.py:239-267).  Parameter gradients of
an energy+force loss from the HIP-backed model must match autograd through the CPU oracle."""

import math

import pytest
import torch

from oracle import model as omodel
from oracle import nn as onn


@pytest.mark.gpu
@pytest.mark.parametrize("lmax", [1, 2, 3])
def test_edge_embed_double_backward(device, lmax):
    from nequip_amd.nn.embedding._edge import _EdgeEmbedFn

    g = torch.Generator().manual_seed(lmax)
    E, r_max, nb = 129, 4.5, 8
    vec = torch.randn(E, 3, generator=g, dtype=torch.float64)
    vec = vec / vec.norm(dim=1, keepdim=True) * (0.8 + 3.5 * torch.rand(E, 1, generator=g, dtype=torch.float64))
    g_sh = torch.randn(E, (lmax + 1) ** 2, generator=g, dtype=torch.float64)
    g_emb = torch.randn(E, nb, generator=g, dtype=torch.float64)
    c = torch.randn(E, 3, generator=g, dtype=torch.float64)
    factor = 2 * math.pi / r_max**2

    def run(fn_sh_emb, to):
        v = to(vec).clone().requires_grad_(True)
        a, b = to(g_sh).clone().requires_grad_(True), to(g_emb).clone().requires_grad_(True)
        sh, emb = fn_sh_emb(v)
        (gv,) = torch.autograd.grad([sh, emb], [v], [a, b], create_graph=True)
        return torch.autograd.grad((gv * to(c)).sum(), [v, a, b])

    ref = run(lambda v: (onn.sh_edge_attrs(v, lmax, torch.float64) + 0.0 * v.sum(),
                         onn.bessel_embedding(v, r_max, nb, 6.0, torch.float64)[0]), lambda t: t)
    bw = torch.linspace(1.0, nb, nb, dtype=torch.float64, device=device)
    cfg = dict(dtype=torch.float64, lmax=lmax, want_sh=True, want_emb=True, nb=nb, rmax_recip=1.0 / r_max, p=6.0,
               factor=factor)
    got = run(lambda v: _EdgeEmbedFn.apply(v, bw, cfg), lambda t: t.to(device))
    for r, k in zip(ref, got):
        torch.testing.assert_close(r, k.cpu(), atol=1e-9 * max(1.0, float(r.abs().max())), rtol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("model_dtype,parity", [("float64", False), ("float64", True), ("float32", False)])
def test_training_step_parameter_gradients(device, model_dtype, parity):
    from nequip_amd.data import AtomicDataDict
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=2, seed=7)
    data = syn.make_data(pos, types, 4.0, cell)
    cfg = dict(r_max=4.0, num_layers=3, l_max=2, parity=parity, num_features=8, radial_mlp_depth=1,
               radial_mlp_width=16, num_bessels=8, polynomial_cutoff_p=6, avg_num_neighbors=25.0,
               model_dtype=model_dtype)
    model = NequIPGNNModel(seed=5, model_dtype=model_dtype, type_names=names,
                           **{k: v for k, v in cfg.items() if k != "model_dtype"})
    gen = torch.Generator().manual_seed(0)
    f_target = torch.randn(len(pos), 3, generator=gen, dtype=torch.float64)
    e_target = torch.randn(1, 1, generator=gen, dtype=torch.float64)

    # oracle: autograd through autograd (create_graph=True)
    param_names = {k for k, _ in model.named_parameters()}
    weights = {k.replace("model.func.", ""): v.detach().clone().requires_grad_(k in param_names)
               for k, v in model.state_dict().items()}
    out = omodel.energy_forces(data, cfg, weights, create_graph=True)
    loss_ref = (out["forces"] - f_target).square().mean() + (out["total_energy"] - e_target).square().mean() / len(pos)
    names_w = [k for k, v in weights.items() if v.requires_grad]
    grads_ref = dict(zip(names_w, torch.autograd.grad(loss_ref, [weights[k] for k in names_w])))

    model = model.to(device).train()
    out = model(AtomicDataDict.to_device(data, device))
    loss = (out["forces"] - f_target.to(device)).square().mean() + (
        out["total_energy"] - e_target.to(device)
    ).square().mean() / len(pos)
    loss.backward()
    tol = 1e-8 if model_dtype == "float64" else 2e-4
    torch.testing.assert_close(loss_ref.detach(), loss.detach().cpu(), atol=tol, rtol=tol)
    for k, p in model.named_parameters():
        key = k.replace("model.func.", "")
        assert p.grad is not None, f"no gradient for {k}"
        r = grads_ref[key]
        torch.testing.assert_close(r, p.grad.cpu(), atol=tol * max(1e-3, float(r.abs().max())), rtol=tol * 10)


@pytest.mark.gpu
@pytest.mark.parametrize("width", [16, 128])
def test_deferred_parameter_gradients_equal_autograd(device, width):
    """``with deferred_parameter_gradients(): loss.backward()`` (parameter gradients of ``o3.Linear`` and the radial MLP's
    last layer launched on a side stream and delivered outside autograd, ``nequip_amd/utils/wgrad.py``) gives every
    parameter the gradient plain ``loss.backward()`` gives it -- twice in a row (accumulation into an existing ``.grad``)."""
    from nequip_amd.data import AtomicDataDict
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.utils import synthetic as syn
    from nequip_amd.utils import wgrad as wg

    pos, types, cell, names = syn.water_box(n_side=3, seed=7)
    data = AtomicDataDict.to_device(syn.make_data(pos, types, 4.0, cell), device)
    model = NequIPGNNModel(seed=5, model_dtype="float32", type_names=names, r_max=4.0, num_layers=3, l_max=2, parity=False,
                           num_features=32, radial_mlp_depth=1, radial_mlp_width=width, num_bessels=8,
                           polynomial_cutoff_p=6, avg_num_neighbors=25.0).to(device).train()
    gen = torch.Generator().manual_seed(0)
    f_target = torch.randn(len(pos), 3, generator=gen, dtype=torch.float64).to(device)

    def loss_of():
        out = model(dict(data))
        return (out["forces"] - f_target).square().mean() + out["total_energy"].square().mean() / len(pos)

    loss_of().backward()
    ref = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    deferred_any = []
    for rep in (1, 2):
        with wg.deferred_parameter_gradients():
            loss_of().backward()
            deferred_any.append(len(wg._deferred) + len(wg._deferred.get("_pending", ())))
        torch.cuda.synchronize()
        for k, p in model.named_parameters():
            assert p.grad is not None, f"no gradient for {k}"
            r = rep * ref[k]
            torch.testing.assert_close(p.grad, r, atol=2e-6 * rep * max(1e-3, float(r.abs().max())), rtol=2e-5,
                                       msg=lambda m, k=k: f"{k}: {m}")
    assert min(deferred_any) >= 6, f"nothing was deferred: {deferred_any}"  # (the o3.Linear weights of three layers at least)
    assert wg._deferred is None
