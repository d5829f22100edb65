"""CPU tests of the oracle itself (oracle/): pinned against every check available without e3nn
(SURVEY.md 8(c)): algebraic identities of the real Wigner-3j tensors, sympy's SU(2) Clebsch-Gordan values,
the independently written product generator, closed-form spherical harmonics, equivariance of the tensor product
under the SH-derived Wigner-D matrices, the committed golden fixtures, and the reference's model-level property
suite (rotation / translation / permutation, finite-difference forces, cutoff smoothness;
nequip/utils/unittests/model_tests_basic.py:450-461,631-672,810-843)."""

import math
import os

import numpy as np
import pytest
import torch

from oracle import irreps as oir
from oracle import model as omodel
from oracle import nn as onn
from oracle import tp as otp
from oracle.sh import spherical_harmonics
from oracle.wigner import wigner_3j

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TRIPLES = [(a, b, c) for a in range(4) for b in range(4) for c in range(abs(a - b), min(a + b, 3) + 1)]


def _rand_rotation(seed):
    g = torch.Generator().manual_seed(seed)
    q, r = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    q = q * torch.sign(torch.diagonal(r))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def _wigner_D(l, R):
    """D^l(R) in the oracle's real basis, from the spherical harmonics themselves: Y_l(R v) = D^l(R) Y_l(v)."""
    g = torch.Generator().manual_seed(100 + l)
    v = torch.randn(8 * (2 * l + 1), 3, generator=g, dtype=torch.float64)
    Y = spherical_harmonics(v, l)[:, l * l : (l + 1) ** 2]
    YR = spherical_harmonics(v @ R.T, l)[:, l * l : (l + 1) ** 2]
    return torch.linalg.lstsq(Y, YR).solution.T


@pytest.mark.parametrize("l1,l2,l3", TRIPLES)
def test_wigner_identities(l1, l2, l3):
    C = wigner_3j(l1, l2, l3)
    assert abs(float(C.norm()) - 1.0) < 1e-12
    G = torch.einsum("ijk,ijl->kl", C, C)
    torch.testing.assert_close(G, torch.eye(2 * l3 + 1, dtype=torch.float64) / (2 * l3 + 1), atol=1e-12, rtol=0)
    C2 = wigner_3j(l2, l1, l3)
    torch.testing.assert_close(C2, (-1) ** (l1 + l2 + l3) * C.transpose(0, 1), atol=1e-12, rtol=0)


def test_wigner_known_values_and_golden():
    torch.testing.assert_close(wigner_3j(1, 1, 0)[:, :, 0] * math.sqrt(3), torch.eye(3, dtype=torch.float64))
    eps = torch.zeros(3, 3, 3, dtype=torch.float64)
    eps[0, 1, 2] = eps[1, 2, 0] = eps[2, 0, 1] = 1
    eps[0, 2, 1] = eps[2, 1, 0] = eps[1, 0, 2] = -1
    torch.testing.assert_close(wigner_3j(1, 1, 1) * math.sqrt(6), eps, atol=1e-12, rtol=0)
    gold = np.load(os.path.join(GOLDEN, "wigner_3j.npz"))
    for l1, l2, l3 in TRIPLES:
        np.testing.assert_allclose(wigner_3j(l1, l2, l3).numpy(), gold[f"w3j_{l1}_{l2}_{l3}"], atol=1e-14)


def test_wigner_matches_product_generator_and_sympy():
    from sympy.physics.quantum.cg import CG

    from nequip_amd.o3.wigner import su2_clebsch_gordan
    from nequip_amd.o3.wigner import wigner_3j as product_w3j

    for l1, l2, l3 in [(a, b, c) for a in range(5) for b in range(5) for c in range(abs(a - b), min(a + b, 4) + 1)]:
        np.testing.assert_allclose(wigner_3j(l1, l2, l3).numpy(), product_w3j(l1, l2, l3), atol=1e-13)
    for a, b, c in [(1, 1, 2), (2, 1, 3), (2, 2, 2), (3, 2, 1), (3, 3, 0)]:
        M = su2_clebsch_gordan(a, b, c)
        for m1 in range(-a, a + 1):
            for m2 in range(-b, b + 1):
                if abs(m1 + m2) <= c:
                    assert abs(float(CG(a, m1, b, m2, c, m1 + m2).doit()) - M[a + m1, b + m2, c + m1 + m2]) < 1e-12


def test_spherical_harmonics_closed_forms_and_golden():
    g = torch.Generator().manual_seed(1)
    v = torch.randn(50, 3, generator=g, dtype=torch.float64)
    Y = spherical_harmonics(v, 4)
    for l in range(5):
        n2 = (Y[:, l * l : (l + 1) ** 2] ** 2).sum(1)
        torch.testing.assert_close(n2, torch.full_like(n2, 2 * l + 1.0))
    u = torch.nn.functional.normalize(v, dim=1)
    x, y, z = u.T
    torch.testing.assert_close(Y[:, 1:4], math.sqrt(3) * u)
    y2 = torch.stack([math.sqrt(15) * x * z, math.sqrt(15) * x * y, math.sqrt(5) * (y * y - 0.5 * (x * x + z * z)),
                      math.sqrt(15) * y * z, math.sqrt(15) / 2 * (z * z - x * x)], dim=1)
    torch.testing.assert_close(Y[:, 4:9], y2)
    # zero vector: Y_0 = 1, higher l vanish (no NaN)
    Y0 = spherical_harmonics(torch.zeros(1, 3, dtype=torch.float64), 2)
    assert torch.isfinite(Y0).all() and float(Y0[0, 0]) == 1.0 and float(Y0[0, 1:].abs().max()) == 0.0
    gold = np.load(os.path.join(GOLDEN, "edge_embed.npz"))
    vec = torch.from_numpy(gold["vec"])
    np.testing.assert_allclose(spherical_harmonics(vec, 4).numpy(), gold["sh"], atol=1e-13)
    emb, cut = onn.bessel_embedding(vec, 4.5, 8, 6.0, torch.float64)
    np.testing.assert_allclose(emb.numpy(), gold["emb"], atol=1e-13)
    np.testing.assert_allclose(cut.numpy(), gold["cutoff"], atol=1e-13)
    assert float(emb[0].abs().max()) == 0.0  # r > r_max


def test_radial_basis_closed_form():
    """b_n = sinc(n x) n with torch.sinc(t) = sin(pi t)/(pi t); cutoff polynomial p = 6; factor 2 pi / r_max^2."""
    r = torch.linspace(0.3, 4.4, 9, dtype=torch.float64)
    vec = torch.stack([r, torch.zeros_like(r), torch.zeros_like(r)], dim=1)
    emb, cut = onn.bessel_embedding(vec, 4.5, 8, 6.0, torch.float64)
    x = r / 4.5
    n = torch.arange(1, 9, dtype=torch.float64)
    ref_b = torch.sin(math.pi * x[:, None] * n) / (math.pi * x[:, None])
    ref_c = 1 - 28 * x**6 + 48 * x**7 - 21 * x**8
    torch.testing.assert_close(emb, (2 * math.pi / 4.5**2) * ref_b * ref_c[:, None], atol=1e-12, rtol=1e-12)
    torch.testing.assert_close(cut[:, 0], ref_c, atol=1e-12, rtol=1e-12)


def test_tensor_product_equivariance_and_golden():
    """tp(D1 x, D2 y, w) = D3 tp(x, y, w) with the Wigner-D matrices derived from the spherical harmonics."""
    f_in, e_at, filt = "3x0e + 2x1o + 2x2e + 1x3o", "0e + 1o + 2e + 3o", "0e + 1o + 2e + 3o"
    mid, instr = otp.build_instructions(f_in, e_at, filt)
    R = _rand_rotation(5)
    D = {l: _wigner_D(l, R) for l in range(4)}

    def rot(t, irreps):
        cols, off = [], 0
        for mul, l, _ in irreps:
            blk = t[:, off : off + mul * (2 * l + 1)].reshape(-1, mul, 2 * l + 1)
            cols.append(torch.einsum("ij,zuj->zui", D[l], blk).reshape(t.shape[0], -1))
            off += mul * (2 * l + 1)
        return torch.cat(cols, dim=1)

    g = torch.Generator().manual_seed(9)
    Z = 6
    x = torch.randn(Z, oir.dim(oir.parse(f_in)), generator=g, dtype=torch.float64)
    y = torch.randn(Z, 16, generator=g, dtype=torch.float64)
    w = torch.randn(Z, otp.weight_numel(f_in, e_at, instr), generator=g, dtype=torch.float64)
    out = otp.tensor_product_uvu(x, y, w, f_in, e_at, mid, instr)
    out_r = otp.tensor_product_uvu(rot(x, oir.parse(f_in)), rot(y, oir.parse(e_at)), w, f_in, e_at, mid, instr)
    torch.testing.assert_close(out_r, rot(out, mid), atol=1e-10, rtol=1e-10)

    gold = np.load(os.path.join(GOLDEN, "tp_scatter.npz"))
    t = lambda k: torch.from_numpy(gold[k])  # noqa: E731
    instr_g = [(int(a), int(b), int(c), "uvu", True) for a, b, c in gold["instructions"]]
    x, y, w = t("x").requires_grad_(True), t("y").requires_grad_(True), t("w").requires_grad_(True)
    out = otp.tp_scatter(x, y, w, t("dst"), t("src"), str(gold["feature_irreps_in"]), str(gold["irreps_edge_attr"]),
                         str(gold["irreps_mid"]), instr_g)
    np.testing.assert_allclose(out.detach().numpy(), gold["out"], atol=1e-12)
    gx, gy, gw = torch.autograd.grad(out, [x, y, w], t("go"))
    np.testing.assert_allclose(gx.numpy(), gold["gx"], atol=1e-12)
    np.testing.assert_allclose(gy.numpy(), gold["gy"], atol=1e-12)
    np.testing.assert_allclose(gw.numpy(), gold["gw"], atol=1e-12)


# ---- model-level property suite on the oracle ---------------------------------------------------------------
def _tiny_model(parity=True, l_max=2, seed=0):
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=2, seed=seed)
    cfg = dict(r_max=3.5, num_layers=2, l_max=l_max, parity=parity, num_features=4, radial_mlp_depth=1,
               radial_mlp_width=8, num_bessels=8, polynomial_cutoff_p=6, avg_num_neighbors=12.0, model_dtype="float64")
    model = NequIPGNNModel(seed=1, model_dtype="float64", type_names=names,
                           **{k: v for k, v in cfg.items() if k != "model_dtype"})
    weights = {k.replace("model.func.", ""): v.detach() for k, v in model.state_dict().items()}
    return pos, types, cell, cfg, weights


def _eval(pos, types, cell, cfg, weights, pbc=True):
    from nequip_amd.utils import synthetic as syn

    data = syn.make_data(pos, types, cfg["r_max"], cell if pbc else None, pbc=pbc)
    return omodel.energy_forces(data, cfg, weights)


@pytest.mark.parametrize("parity", [True, False])
def test_oracle_model_e3_invariance(parity):
    pos, types, cell, cfg, weights = _tiny_model(parity=parity)
    base = _eval(pos, types, cell, cfg, weights)
    R = _rand_rotation(3).numpy()
    for Q in (R, -np.eye(3)):  # proper rotation and inversion
        out = _eval(pos @ Q.T, types, cell @ Q.T, cfg, weights)
        assert abs(float(out["total_energy"] - base["total_energy"])) < 1e-9
        np.testing.assert_allclose(out["forces"].numpy(), base["forces"].numpy() @ Q.T, atol=1e-9)
    out = _eval(pos + np.array([0.3, -1.2, 0.7]), types, cell, cfg, weights)  # translation
    assert abs(float(out["total_energy"] - base["total_energy"])) < 1e-9
    np.testing.assert_allclose(out["forces"].numpy(), base["forces"].numpy(), atol=1e-9)
    perm = np.random.default_rng(0).permutation(len(pos))  # permutation
    out = _eval(pos[perm], types[perm], cell, cfg, weights)
    assert abs(float(out["total_energy"] - base["total_energy"])) < 1e-9
    np.testing.assert_allclose(out["forces"].numpy(), base["forces"].numpy()[perm], atol=1e-9)


def test_oracle_model_numeric_gradient_and_golden():
    pos, types, cell, cfg, weights = _tiny_model(parity=False)
    base = _eval(pos, types, cell, cfg, weights)
    eps = 1e-4
    for atom, axis in [(0, 0), (3, 1), (7, 2)]:
        p1, p2 = pos.copy(), pos.copy()
        p1[atom, axis] += eps
        p2[atom, axis] -= eps
        e1 = float(_eval(p1, types, cell, cfg, weights)["total_energy"])
        e2 = float(_eval(p2, types, cell, cfg, weights)["total_energy"])
        assert abs(-(e1 - e2) / (2 * eps) - float(base["forces"][atom, axis])) < 1e-6
    gold = np.load(os.path.join(GOLDEN, "model_si64.npz"))
    cfg = dict(r_max=4.5, num_layers=3, l_max=2, parity=False, num_features=8, radial_mlp_depth=1,
               radial_mlp_width=16, num_bessels=8, polynomial_cutoff_p=6, avg_num_neighbors=20.0, model_dtype="float64")
    weights = {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w::")}
    data = {"pos": torch.from_numpy(gold["pos"]), "atom_types": torch.from_numpy(gold["types"]),
            "edge_index": torch.from_numpy(gold["edge_index"]), "cell": torch.from_numpy(gold["cell"]).view(1, 3, 3),
            "edge_cell_shift": torch.from_numpy(gold["edge_cell_shift"])}
    out = omodel.energy_forces(data, cfg, weights, with_virial=True)
    np.testing.assert_allclose(out["total_energy"].numpy(), gold["total_energy"], atol=1e-10)
    np.testing.assert_allclose(out["forces"].numpy(), gold["forces"], atol=1e-10)
    np.testing.assert_allclose(out["virial"].numpy(), gold["virial"], atol=1e-9)


def test_oracle_force_smoothness_at_cutoff():
    """Two atoms moved through r_max: energy and forces go to the isolated-atom value continuously and are exactly
    constant / zero beyond the cutoff (nequip/utils/unittests/model_tests_basic.py:810-843)."""
    pos, types, cell, cfg, weights = _tiny_model(parity=True)
    e_far = None
    for r in (cfg["r_max"] * 1.01, cfg["r_max"] * 1.5):
        p = np.array([[0.0, 0.0, 0.0], [r, 0.0, 0.0]])
        out = _eval(p, np.array([0, 1]), None, cfg, weights, pbc=False)
        assert float(out["forces"].abs().max()) == 0.0
        e_far = float(out["total_energy"]) if e_far is None else e_far
        assert float(out["total_energy"]) == e_far
    p = np.array([[0.0, 0.0, 0.0], [cfg["r_max"] * (1 - 1e-6), 0.0, 0.0]])
    out = _eval(p, np.array([0, 1]), None, cfg, weights, pbc=False)
    assert abs(float(out["total_energy"]) - e_far) < 1e-8 and float(out["forces"].abs().max()) < 1e-6


# ---------------------------------------------------------------------------------------------------------------
# Outside pins for the e3nn-computed rows (e3nn itself is not installable here): sympy's spherical
# harmonics, real Gaunt coefficients and SU(2) Clebsch-Gordan coefficients are implementations written by other
# people from the published definitions; e3nn documents its conventions as "real spherical harmonics, polar axis y,
# component normalisation" and its 3j tensors as the normalised real-basis coupling coefficients.
# ---------------------------------------------------------------------------------------------------------------
def _standard_real_sh(l, m):
    """The textbook real spherical harmonic S_lm(theta, phi) (no Condon-Shortley phase):
    sqrt2 (-1)^m Re Y_l^|m| for m > 0, Y_l^0, sqrt2 (-1)^m Im Y_l^|m| for m < 0 -- built from sympy's complex Ynm."""
    import sympy as sp

    th, ph = sp.symbols("theta phi", real=True)
    Y = sp.Ynm(l, abs(m), th, ph).expand(func=True)
    if m == 0:
        expr = Y
    elif m > 0:
        expr = sp.sqrt(2) * (-1) ** m * sp.re(Y)
    else:
        expr = sp.sqrt(2) * (-1) ** m * sp.im(Y)
    return sp.lambdify((th, ph), expr, "numpy")


def test_sh_equal_standard_real_spherical_harmonics_l_le_4():
    """Y^{e3nn}_{lm}(x, y, z) = sqrt(4 pi) S_lm evaluated with y as the polar axis ((x,y,z)_std = (z,x,y)), with a plus
    sign for every (l, m), l <= 4: pins signs, component order and normalisation of row a2 to sympy's Ynm."""
    g = torch.Generator().manual_seed(0)
    v = torch.nn.functional.normalize(torch.randn(64, 3, generator=g, dtype=torch.float64), dim=-1)
    Y = spherical_harmonics(v, 4).numpy()
    xs, ys, zs = v[:, 2].numpy(), v[:, 0].numpy(), v[:, 1].numpy()
    theta, phi = np.arccos(zs), np.arctan2(ys, xs)
    for l in range(5):
        for m in range(-l, l + 1):
            ref = math.sqrt(4 * math.pi) * np.broadcast_to(_standard_real_sh(l, m)(theta, phi), theta.shape)
            np.testing.assert_allclose(Y[:, l * l + l + m], ref, atol=1e-13, err_msg=f"l={l} m={m}")


def test_sh_closed_forms_l3():
    """SURVEY.md A.4 closed forms of the l = 3 components (polynomials in x, y, z on the unit sphere)."""
    g = torch.Generator().manual_seed(1)
    v = torch.nn.functional.normalize(torch.randn(32, 3, generator=g, dtype=torch.float64), dim=-1)
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    Y = spherical_harmonics(v, 3)
    y20 = math.sqrt(15) * x * z
    y24 = math.sqrt(15) / 2 * (z * z - x * x)
    s168 = math.sqrt(168) / 8
    ref = torch.stack([
        math.sqrt(42) / 6 * (y20 * z + y24 * x),
        math.sqrt(7) * y20 * y,
        s168 * (4 * y * y - x * x - z * z) * x,
        math.sqrt(7) / 2 * y * (2 * y * y - 3 * (x * x + z * z)),
        s168 * z * (4 * y * y - x * x - z * z),
        math.sqrt(7) * y24 * y,
        math.sqrt(42) / 6 * (y24 * z - y20 * x),
    ], dim=1)
    torch.testing.assert_close(Y[:, 9:16], ref, atol=1e-13, rtol=0)


EVEN_TRIPLES = [t for t in TRIPLES if sum(t) % 2 == 0]


@pytest.mark.parametrize("l1,l2,l3", EVEN_TRIPLES)
def test_real_3j_equal_sympy_real_gaunt(l1, l2, l3):
    """Every even-(l1+l2+l3) tensor -- all paths of the parity=False BASELINE models -- is the Frobenius-normalised
    *real Gaunt* tensor int S_{l1 m1} S_{l2 m2} S_{l3 m3} dOmega with a PLUS sign, entry by entry, against
    sympy.physics.wigner.real_gaunt (Homeier & Steinborn's definition, an outside implementation)."""
    from sympy.physics.wigner import real_gaunt

    G = np.array([[[float(real_gaunt(l1, l2, l3, m1, m2, m3)) for m3 in range(-l3, l3 + 1)]
                   for m2 in range(-l2, l2 + 1)] for m1 in range(-l1, l1 + 1)])
    assert np.linalg.norm(G) > 0
    np.testing.assert_allclose(wigner_3j(l1, l2, l3).numpy(), G / np.linalg.norm(G), atol=1e-13)


def test_real_3j_gaunt_by_quadrature_uses_the_oracle_sh():
    """The same statement through the oracle's own spherical harmonics (ties rows a2 and a8 together with the sign):
    int Y_i Y_j Y_k dOmega = kappa C_ijk with kappa > 0 for even sums and = 0 for odd sums (Gauss-Legendre x uniform
    quadrature, exact for these polynomial degrees)."""
    n, nphi = 16, 32
    xg, wg = np.polynomial.legendre.leggauss(n)
    phis = np.arange(nphi) * 2 * np.pi / nphi
    ct, ph = np.meshgrid(xg, phis, indexing="ij")
    st = np.sqrt(1 - ct**2)
    w = (np.repeat(wg[:, None], nphi, 1) * 2 * np.pi / nphi).ravel()
    pts = np.stack([st * np.sin(ph), ct, st * np.cos(ph)], -1).reshape(-1, 3)
    Y = spherical_harmonics(torch.tensor(pts), 3).numpy()
    for l1, l2, l3 in TRIPLES:
        C = wigner_3j(l1, l2, l3).numpy()
        G = np.einsum("z,zi,zj,zk->ijk", w, Y[:, l1 * l1:(l1 + 1) ** 2], Y[:, l2 * l2:(l2 + 1) ** 2],
                      Y[:, l3 * l3:(l3 + 1) ** 2])
        if (l1 + l2 + l3) % 2:
            assert np.abs(G).max() < 1e-12
        else:
            kappa = float((G * C).sum())
            assert kappa > 0
            np.testing.assert_allclose(G, kappa * C, atol=1e-11)


def test_real_3j_from_sympy_clebsch_gordan_all_triples():
    """All triples l <= 3 (odd sums included): the float Racah evaluation against sympy's exact Clebsch-Gordan
    coefficients pushed through the published real <-> complex basis change.  For the odd-sum tensors
    (parity=True models only) the overall sign rests on that basis change and on C^{111} = +epsilon/sqrt(6)."""
    import sympy as sp
    from sympy.physics.wigner import clebsch_gordan

    def Q(l):
        q = sp.zeros(2 * l + 1, 2 * l + 1)
        r2 = 1 / sp.sqrt(2)
        for m in range(-l, 0):
            q[l + m, l + abs(m)] = r2
            q[l + m, l - abs(m)] = -sp.I * r2
        q[l, l] = 1
        for m in range(1, l + 1):
            q[l + m, l + abs(m)] = (-1) ** m * r2
            q[l + m, l - abs(m)] = sp.I * (-1) ** m * r2
        return (-sp.I) ** l * q

    for l1, l2, l3 in TRIPLES:
        cg = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1), dtype=complex)
        for m1 in range(-l1, l1 + 1):
            for m2 in range(-l2, l2 + 1):
                if abs(m1 + m2) <= l3:
                    cg[l1 + m1, l2 + m2, l3 + m1 + m2] = complex(clebsch_gordan(l1, l2, l3, m1, m2, m1 + m2))
        q1, q2, q3 = (np.array(Q(l).tolist(), dtype=complex) for l in (l1, l2, l3))
        c = np.einsum("ij,kl,mn,ikn->jlm", q1, q2, np.conj(q3.T), cg)
        assert np.abs(c.imag).max() < 1e-12
        c = c.real / np.linalg.norm(c.real)
        np.testing.assert_allclose(wigner_3j(l1, l2, l3).numpy(), c, atol=1e-12, err_msg=f"{(l1, l2, l3)}")


def test_uvu_hand_computed_known_answers():
    """One 'uvu' path at mul = 1, vector (x) vector, worked by hand from SURVEY.md A.2 (component normalisation, each
    path alone in its output slot so c = sqrt(2 l3 + 1)):
        1 (x) 1 -> 0 :  w (a . b) / sqrt(3)
        1 (x) 1 -> 1 :  w (a x b) / sqrt(2)
        1 (x) 1 -> 2 :  w sqrt(5) M_k : (a b^T) / sqrt(5), M_k the orthonormal symmetric traceless basis in the
                        component order (xz, xy, 2y^2-x^2-z^2, yz, z^2-x^2)."""
    a = torch.tensor([[0.3, -1.2, 0.7]], dtype=torch.float64)
    b = torch.tensor([[-0.5, 0.4, 2.0]], dtype=torch.float64)
    w = torch.tensor([[1.7]], dtype=torch.float64)
    ax, ay, az = a[0]
    bx, by, bz = b[0]
    out0 = otp.tensor_product_uvu(a, b, w, "1x1o", "1x1o", "1x0e", [(0, 0, 0, "uvu", True)])
    torch.testing.assert_close(out0, w * (a * b).sum() / math.sqrt(3))
    out1 = otp.tensor_product_uvu(a, b, w, "1x1o", "1x1o", "1x1e", [(0, 0, 0, "uvu", True)])
    torch.testing.assert_close(out1, w * torch.linalg.cross(a, b) / math.sqrt(2))
    out2 = otp.tensor_product_uvu(a, b, w, "1x1o", "1x1o", "1x2e", [(0, 0, 0, "uvu", True)])
    r2, r6 = math.sqrt(2), math.sqrt(6)
    ref2 = torch.stack([
        (ax * bz + az * bx) / r2,
        (ax * by + ay * bx) / r2,
        (2 * ay * by - ax * bx - az * bz) / r6,
        (ay * bz + az * by) / r2,
        (az * bz - ax * bx) / r2,
    ]).view(1, 5) * w
    torch.testing.assert_close(out2, ref2)
    # two paths into one slot: alpha = (2 l3 + 1) / 2 each
    bb = torch.cat([b, 2 * b], dim=1)
    out = otp.tensor_product_uvu(a, bb, torch.tensor([[1.7, -0.4]], dtype=torch.float64), "1x1o", "1x1o+1x1o", "1x0e",
                                 [(0, 0, 0, "uvu", True), (0, 1, 0, "uvu", True)])
    torch.testing.assert_close(out, (1.7 - 0.8) * (a * b).sum().view(1, 1) / math.sqrt(3) / math.sqrt(2))
