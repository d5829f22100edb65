"""Real spherical harmonics of edge vectors for the oracle.

Follows SphericalHarmonicEdgeAttrs.forward (nequip/nn/embedding/_edge.py:193-198):
e3nn SphericalHarmonics(irreps 0..lmax, normalize=True, normalization="component") evaluated on float64
edge vectors and *then* cast to the model dtype.  e3nn semantics per SURVEY.md A.4:
u = v / max(|v|, 1e-12) (torch.nn.functional.normalize), Y_0 = 1, Y_1 = sqrt(3) (x, y, z),
Y_l = N_l * C^{(1,l-1,l)} (Y_1 (x) Y_{l-1}) with N_l > 0 such that |Y_l(u)|^2 = 2l+1.
Implemented numerically with differentiable torch ops (no generated polynomials -- the product's HIP
kernel uses sympy-generated polynomials, so the two implementations are independent).
"""

import functools
import math

import torch

from .wigner import wigner_3j


@functools.lru_cache(maxsize=None)
def _norm_const(l):
    # |C^{(1,l-1,l)}(Y_1 (x) Y_{l-1})| is direction independent on the unit sphere: evaluate once
    u = torch.tensor([[0.3, -0.5, math.sqrt(1 - 0.09 - 0.25)]], dtype=torch.float64)
    ys = _sh_list(u, l - 1)
    raw = torch.einsum("ijk,zi,zj->zk", wigner_3j(1, l - 1, l), ys[1], ys[l - 1])
    return math.sqrt(2 * l + 1) / float(raw.norm())


def _sh_list(u, lmax):
    ys = [torch.ones_like(u[:, :1])]
    if lmax >= 1:
        ys.append(math.sqrt(3.0) * u)
    for l in range(2, lmax + 1):
        raw = torch.einsum("ijk,zi,zj->zk", wigner_3j(1, l - 1, l).to(u.dtype), ys[1], ys[l - 1])
        ys.append(_norm_const(l) * raw)
    return ys


def spherical_harmonics(vec, lmax, normalize=True):
    """vec [E,3] (float64) -> [E, (lmax+1)^2], component normalisation."""
    u = torch.nn.functional.normalize(vec, dim=-1) if normalize else vec
    return torch.cat(_sh_list(u, lmax), dim=-1)
