"""Real Wigner-3j tensors for the oracle (float implementation, written separately from the product's
exact-rational generator in nequip_amd/o3/wigner.py; the two are cross-checked in tests/).

Restates e3nn 0.6 ``o3.wigner_3j`` per SURVEY.md A.3: SU(2) Clebsch-Gordan by Racah's formula placed at
[l1+m1, l2+m2, l3+m3]; change of basis real->complex Q_l with the (-i)^l phase;
C = einsum("ij,kl,mn,ikn->jlm", Q1, Q2, conj(Q3^T), CG); real part; Frobenius-normalised.
Reached in the reference through e3nn.o3.TensorProduct at nequip/nn/_tp_scatter_base.py:24-31.
"""

import functools
import math

import torch


def _f(n):
    return math.gamma(n + 1.0)  # float factorial (exact for the small n used here)


def _su2_cg(j1, m1, j2, m2, j3, m3):
    if m3 != m1 + m2:
        return 0.0
    vmin = int(max(-j1 + j2 + m3, -j1 + m1, 0))
    vmax = int(min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3))
    c = math.sqrt(
        (2.0 * j3 + 1.0)
        * _f(j3 + j1 - j2) * _f(j3 - j1 + j2) * _f(j1 + j2 - j3) * _f(j3 + m3) * _f(j3 - m3)
        / (_f(j1 + j2 + j3 + 1) * _f(j1 - m1) * _f(j1 + m1) * _f(j2 - m2) * _f(j2 + m2))
    )
    s = 0.0
    for v in range(vmin, vmax + 1):
        s += (-1.0) ** (v + j2 + m2) * _f(j2 + j3 + m1 - v) * _f(j1 - m1 + v) / (
            _f(v) * _f(j3 - j1 + j2 - v) * _f(j3 + m3 - v) * _f(v + j1 - j2 - m3)
        )
    return c * s


def _real_to_complex(l):
    q = torch.zeros(2 * l + 1, 2 * l + 1, dtype=torch.complex128)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = 1 / math.sqrt(2)
        q[l + m, l - abs(m)] = -1j / math.sqrt(2)
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m / math.sqrt(2)
        q[l + m, l - abs(m)] = 1j * (-1) ** m / math.sqrt(2)
    return (-1j) ** l * q


@functools.lru_cache(maxsize=None)
def wigner_3j(l1, l2, l3):
    assert abs(l1 - l2) <= l3 <= l1 + l2
    cg = torch.zeros(2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1, dtype=torch.complex128)
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            if abs(m1 + m2) <= l3:
                cg[l1 + m1, l2 + m2, l3 + m1 + m2] = _su2_cg(l1, m1, l2, m2, l3, m1 + m2)
    q1, q2, q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    c = torch.einsum("ij,kl,mn,ikn->jlm", q1, q2, torch.conj(q3.T), cg)
    assert c.imag.abs().max() < 1e-9
    c = c.real
    c = c / c.norm()
    return c.contiguous()
