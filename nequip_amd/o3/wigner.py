"""Real Wigner-3j (Clebsch-Gordan) tensors in the e3nn real basis, used to build the kernel plans.

The reference delegates this to ``e3nn.o3.wigner_3j`` (an un-vendored dependency, ``pyproject.toml:22``;
reached through ``e3nn.o3.TensorProduct`` at ``nequip/nn/_tp_scatter_base.py:24-31``).  The construction
is restated from SURVEY.md Appendix A.3:

1. SU(2) Clebsch-Gordan ``<l1 m1 l2 m2 | l3 m3>`` (Racah's closed formula), computed here in *exact*
   rational arithmetic (``fractions.Fraction`` for the squared magnitude) before a single sqrt,
2. the real<->complex change of basis ``Q_l`` (including the ``(-i)^l`` phase that makes the result
   real),
3. ``C = einsum("ij,kl,mn,ikn->jlm", Q1, Q2, conj(Q3^T), CG)`` and Frobenius normalisation to 1.

Component order is ``m = -l..l`` which for ``l = 1`` is ``(x, y, z)``: the polar axis is ``y``.

This module is product code (it feeds the constant tables of the HIP kernels); the test oracle has its
own, separately written float implementation in ``oracle/`` and the two are cross-checked in ``tests/``.
"""

from __future__ import annotations

import functools
from fractions import Fraction
from math import factorial, sqrt

import numpy as np


def _su2_cg_coeff(j1: int, m1: int, j2: int, m2: int, j3: int, m3: int) -> float:
    """<j1 m1 j2 m2 | j3 m3> for integer spins via Racah's formula (exact under the square root)."""
    if m3 != m1 + m2:
        return 0.0
    vmin = max(-j1 + j2 + m3, -j1 + m1, 0)
    vmax = min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3)

    def f(n: int) -> int:
        assert n >= 0
        return factorial(n)

    pref2 = Fraction(
        (2 * j3 + 1) * f(j3 + j1 - j2) * f(j3 - j1 + j2) * f(j1 + j2 - j3) * f(j3 + m3) * f(j3 - m3),
        f(j1 + j2 + j3 + 1) * f(j1 - m1) * f(j1 + m1) * f(j2 - m2) * f(j2 + m2),
    )
    s = Fraction(0)
    for v in range(vmin, vmax + 1):
        s += Fraction(
            (-1) ** (v + j2 + m2) * f(j2 + j3 + m1 - v) * f(j1 - m1 + v),
            f(v) * f(j3 - j1 + j2 - v) * f(j3 + m3 - v) * f(v + j1 - j2 - m3),
        )
    # value = sqrt(pref2) * s ; keep the sign of s, take one sqrt of an exact rational
    val2 = pref2 * s * s
    mag = sqrt(val2.numerator) / sqrt(val2.denominator)
    return mag if s >= 0 else -mag


def su2_clebsch_gordan(l1: int, l2: int, l3: int) -> np.ndarray:
    """Dense ``[2l1+1, 2l2+1, 2l3+1]`` array of SU(2) CG coefficients, index ``l + m``."""
    out = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1), dtype=np.float64)
    if not (abs(l1 - l2) <= l3 <= l1 + l2):
        return out
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            m3 = m1 + m2
            if abs(m3) <= l3:
                out[l1 + m1, l2 + m2, l3 + m3] = _su2_cg_coeff(l1, m1, l2, m2, l3, m3)
    return out


def change_basis_real_to_complex(l: int) -> np.ndarray:
    """``Q_l`` of SURVEY.md A.3(ii): rows = complex ``m``, columns = real component index."""
    q = np.zeros((2 * l + 1, 2 * l + 1), dtype=np.complex128)
    inv_sqrt2 = 1.0 / sqrt(2.0)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = inv_sqrt2
        q[l + m, l - abs(m)] = -1j * inv_sqrt2
    q[l, l] = 1.0
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m * inv_sqrt2
        q[l + m, l - abs(m)] = 1j * (-1) ** m * inv_sqrt2
    return ((-1j) ** l) * q


@functools.lru_cache(maxsize=None)
def _wigner_3j_cached(l1: int, l2: int, l3: int) -> np.ndarray:
    q1 = change_basis_real_to_complex(l1)
    q2 = change_basis_real_to_complex(l2)
    q3 = change_basis_real_to_complex(l3)
    cg = su2_clebsch_gordan(l1, l2, l3).astype(np.complex128)
    c = np.einsum("ij,kl,mn,ikn->jlm", q1, q2, np.conj(q3.T), cg)
    assert np.abs(c.imag).max() < 1e-9, "real-basis Clebsch-Gordan tensor must be real"
    c = np.ascontiguousarray(c.real)
    nrm = np.linalg.norm(c)
    assert nrm > 0
    c = c / nrm
    c[np.abs(c) < 1e-14] = 0.0
    c.setflags(write=False)
    return c


def wigner_3j(l1: int, l2: int, l3: int) -> np.ndarray:
    """Real Wigner-3j tensor ``C[i, j, k]`` (float64, unit Frobenius norm, read-only)."""
    if not (abs(l1 - l2) <= l3 <= l1 + l2):
        raise ValueError(f"triangle rule violated for ({l1}, {l2}, {l3})")
    return _wigner_3j_cached(int(l1), int(l2), int(l3))
