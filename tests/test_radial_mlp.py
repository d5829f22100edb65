"""Fused MFMA radial MLP (nqa_radial_mlp_fwd/bwd) against the oracle's ScalarMLPFunction restatement
(oracle/nn.py::scalar_mlp following nequip/nn/mlp.py:141-156,262-268).  float32, tolerance 1e-5 relative to
the output scale (exact-fp32 MFMA: only the summation order differs from the CPU mm)."""

import math

import pytest
import torch

from oracle import nn as onn


@pytest.mark.gpu
@pytest.mark.parametrize("E", [1, 127, 128, 1000, 4133])
@pytest.mark.parametrize("H,W", [(128, 704), (128, 192), (64, 64), (64, 160), (128, 2944)])
def test_radial_mlp_fwd_bwd(device, E, H, W):
    from nequip_amd.nn.mlp import ScalarMLPFunction

    torch.manual_seed(E + H + W)
    mlp = ScalarMLPFunction(input_dim=8, output_dim=W, hidden_layers_depth=1, hidden_layers_width=H).eval()
    emb = torch.randn(E, 8) * 0.7
    w0, w1 = mlp.mlp[0].weight.detach(), mlp.mlp[2].weight.detach()

    e_ref = emb.clone().requires_grad_(True)
    ref = onn.scalar_mlp(e_ref, [w0, w1], "silu")
    g = torch.randn(E, W)
    (ge_ref,) = torch.autograd.grad(ref, e_ref, g)

    mlp = mlp.to(device)
    e_dev = emb.to(device).requires_grad_(True)
    assert mlp._fused_ok(e_dev), "fused MFMA path must be taken for this shape"
    out = mlp(e_dev)
    (ge,) = torch.autograd.grad(out, e_dev, g.to(device))
    torch.testing.assert_close(ref.detach(), out.detach().cpu(), atol=1e-5 * float(ref.abs().max()), rtol=1e-5)
    torch.testing.assert_close(ge_ref, ge.cpu(), atol=2e-5 * float(ge_ref.abs().max()), rtol=2e-5)


@pytest.mark.gpu
def test_radial_mlp_training_mode_uses_autograd_path(device):
    """In training mode parameter gradients must exist (mm/SiLU formulation, not the inference kernel)."""
    from nequip_amd.nn.mlp import ScalarMLPFunction

    mlp = ScalarMLPFunction(input_dim=8, output_dim=192, hidden_layers_depth=1, hidden_layers_width=128).to(device)
    mlp.train()
    x = torch.randn(300, 8, device=device)
    assert not mlp._fused_ok(x)
    mlp(x).square().sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in mlp.parameters())
