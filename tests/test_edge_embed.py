"""Parity of the fused edge-embedding HIP kernel (real spherical harmonics + Bessel x polynomial cutoff x factor)
and of its vector-Jacobian product against the oracle (oracle/nn.py, oracle/sh.py), which follows
nequip/nn/embedding/_edge.py:65-80,136-150,193-198 and cutoffs.py:17-27.  Tolerances: 1e-5 (float32 outputs)
/ 1e-9 (float64), as in the reference's tests/unit/nn/test_embed.py:23-49."""

import math

import pytest
import torch

from oracle import nn as onn


def _vectors(E, seed=0, r_lo=0.6, r_hi=5.2):
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(E, 3, generator=g, dtype=torch.float64)
    v = v / v.norm(dim=1, keepdim=True)
    r = r_lo + (r_hi - r_lo) * torch.rand(E, 1, generator=g, dtype=torch.float64)
    return v * r


@pytest.mark.gpu
@pytest.mark.parametrize("lmax", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("trained", [False, True])
def test_sh_and_radial_forward_backward(device, lmax, dtype, trained):
    # trained = True: Bessel weights moved off the roots n + 1 (the kernel's angle-addition path only applies to the
    # untrained roots; any other weights take one sin/cos per basis function)
    from nequip_amd.nn.embedding._edge import _EdgeEmbedFn

    tol = 1e-5 if dtype == torch.float32 else 1e-9
    E, r_max, nb, p = 257, 4.5, 8, 6.0
    vec = _vectors(E, seed=lmax)
    vec[0] = torch.tensor([0.0, 4.5 * 1.2, 0.0])  # beyond the cutoff, on the polar axis
    bw = torch.linspace(1.0, nb, nb, dtype=torch.float64)
    if trained:
        bw = bw + 0.07 * torch.randn(nb, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    factor = 2 * math.pi / (r_max * r_max)

    v_ref = vec.clone().requires_grad_(True)
    sh_ref = onn.sh_edge_attrs(v_ref, lmax, dtype)
    emb_ref, _ = onn.bessel_embedding(v_ref, r_max, nb, p, dtype, bessel_weights=bw.unsqueeze(0))

    v_dev = vec.to(device).requires_grad_(True)
    cfg = dict(dtype=dtype, lmax=lmax, want_sh=True, want_emb=True, nb=nb, rmax_recip=1.0 / r_max, p=p, factor=factor)
    sh, emb = _EdgeEmbedFn.apply(v_dev, bw.to(device), cfg)
    torch.testing.assert_close(sh_ref, sh.cpu(), atol=tol, rtol=tol)
    torch.testing.assert_close(emb_ref, emb.cpu(), atol=tol, rtol=tol)
    assert (emb[0] == 0).all(), "embedding must vanish exactly beyond the cutoff"

    g = torch.Generator().manual_seed(5)
    g_sh = torch.randn(sh_ref.shape, generator=g, dtype=dtype)
    g_emb = torch.randn(emb_ref.shape, generator=g, dtype=dtype)
    sh_ref_d = sh_ref if sh_ref.requires_grad else sh_ref + 0.0 * v_ref.sum()  # lmax = 0: Y_0 = 1 is constant
    (gv_ref,) = torch.autograd.grad([sh_ref_d, emb_ref], [v_ref], [g_sh, g_emb])
    (gv,) = torch.autograd.grad([sh, emb], [v_dev], [g_sh.to(device), g_emb.to(device)])
    gtol = 2e-5 if dtype == torch.float32 else 1e-9
    torch.testing.assert_close(gv_ref, gv.cpu(), atol=gtol * float(gv_ref.abs().max()), rtol=gtol)


@pytest.mark.gpu
def test_sh_properties(device):
    """|Y_l|^2 = 2l+1, Y_1 = sqrt(3) r_hat, closed forms for l = 2."""
    from nequip_amd.nn.embedding._edge import _EdgeEmbedFn

    vec = _vectors(100, seed=11).to(device)
    cfg = dict(dtype=torch.float64, lmax=4, want_sh=True, want_emb=False, nb=0, rmax_recip=1.0, p=6.0, factor=1.0)
    Y = _EdgeEmbedFn.apply(vec, torch.ones(1, dtype=torch.float64, device=device), cfg).cpu()
    for l in range(5):
        n2 = (Y[:, l * l : (l + 1) ** 2] ** 2).sum(1)
        torch.testing.assert_close(n2, torch.full_like(n2, 2 * l + 1.0), atol=1e-12, rtol=1e-12)
    u = torch.nn.functional.normalize(vec.cpu(), dim=1)
    x, y, z = u.T
    torch.testing.assert_close(Y[:, 1:4], math.sqrt(3) * u, atol=1e-13, rtol=0)
    torch.testing.assert_close(Y[:, 4], math.sqrt(15) * x * z, atol=1e-13, rtol=0)
    torch.testing.assert_close(Y[:, 6], math.sqrt(5) * (y * y - 0.5 * (x * x + z * z)), atol=1e-13, rtol=0)
    torch.testing.assert_close(Y[:, 8], math.sqrt(15) / 2 * (z * z - x * x), atol=1e-13, rtol=0)


@pytest.mark.gpu
def test_sh_kernel_equals_standard_real_spherical_harmonics(device):
    """The HIP kernel's real spherical harmonics against sympy's Ynm directly (no oracle in between): sqrt(4 pi) x the
    textbook real harmonics with y as the polar axis, plus sign for every (l, m), l <= 4 (see tests/test_oracle.py)."""
    import numpy as np

    from nequip_amd.nn.embedding._edge import _EdgeEmbedFn
    from tests.test_oracle import _standard_real_sh

    vec = _vectors(80, seed=21)
    cfg = dict(dtype=torch.float64, lmax=4, want_sh=True, want_emb=False, nb=0, rmax_recip=1.0, p=6.0, factor=1.0)
    Y = _EdgeEmbedFn.apply(vec.to(device), torch.ones(1, dtype=torch.float64, device=device), cfg).cpu().numpy()
    u = torch.nn.functional.normalize(vec, dim=1).numpy()
    theta, phi = np.arccos(u[:, 1]), np.arctan2(u[:, 0], u[:, 2])
    for l in range(5):
        for m in range(-l, l + 1):
            ref = math.sqrt(4 * math.pi) * np.broadcast_to(_standard_real_sh(l, m)(theta, phi), theta.shape)
            np.testing.assert_allclose(Y[:, l * l + l + m], ref, atol=1e-12, err_msg=f"l={l} m={m}")
