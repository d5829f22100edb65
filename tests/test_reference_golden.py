"""Fixtures produced by the REFERENCE's own code (tests/golden/make_reference_golden.py imports mir-group/nequip from
/root/reference with inert stand-ins for the uninstalled e3nn / training stack) for the rows of the hot path that nequip
itself implements: edge vectors (a1), length normaliser + Bessel x polynomial cutoff (a3), ScalarMLPFunction (a4),
AvgNumNeighborsNorm (a6), PerTypeScaleShift / AtomwiseReduce, ForceStressOutput (a12).

CPU tests pin the oracle and the host-side mirrors against them; GPU tests pin the HIP kernels (through the C ABI).
The e3nn-computed rows (SH, tensor product, o3.Linear, Gate) cannot be produced this way and stay "parity unpinned".
"""

import math
import os

import numpy as np
import pytest
import torch

from oracle import nn as onn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(GOLD, name)).items()}


def _close(a, b, tol):
    torch.testing.assert_close(a, b, atol=tol * max(1.0, float(b.abs().max())), rtol=tol)


# ---------------------------------------------------------------------------------------------------- oracle (CPU)
def test_oracle_edge_vectors_match_reference():
    f = _load("ref_edge_vectors.npz")
    for p, batch in (("s", None), ("b", f["b_batch"])):
        pos = f[p + "_pos"].clone().requires_grad_(True)
        cell = f[p + "_cell"].clone().requires_grad_(True)
        vec = onn.edge_vectors(pos, f[p + "_edge_index"], cell, f[p + "_shift"], batch)
        _close(vec.detach(), f[p + "_vec"], 1e-14)
        gp, gc = torch.autograd.grad((vec * f[p + "_coef"]).sum(), [pos, cell])
        _close(gp, f[p + "_gpos"], 1e-13)
        _close(gc, f[p + "_gcell"], 1e-13)
    _close(f["s_vec"].square().sum(1, keepdim=True).sqrt(), f["s_len"], 1e-14)


def test_oracle_radial_basis_matches_reference():
    f = _load("ref_radial_basis.npz")
    r_max = float(f["r_max"])
    factor = 2 * math.pi / (r_max * r_max)
    for name, dt, tol in (("f64", torch.float64, 1e-13), ("f32", torch.float32, 2e-6)):
        v = f["vec"].clone().requires_grad_(True)
        emb, cut = onn.bessel_embedding(v, r_max, 8, 6.0, dt)
        _close(emb.detach() / factor, f["emb_" + name], tol)
        _close(cut.detach(), f["cutoff_" + name], tol)
        (gv,) = torch.autograd.grad((emb / factor * f["cot_" + name]).sum(), [v])
        _close(gv, f["gvec_" + name], tol * 10)
    assert (f["emb_f64"][0] == 0).all() and (f["emb_f64"][1] == 0).all()  # beyond / at the cutoff
    torch.testing.assert_close(f["bessel_weights"].view(-1).double(), torch.linspace(1.0, 8.0, 8, dtype=torch.float64))


def _pt_case(f):
    types, ei = f["pt_types"].long(), f["pt_edge_index"].long()
    return types, ei, torch.stack([types[ei[0]], types[ei[1]]])


def test_per_edge_type_cutoff_and_trained_bessel_roots_match_reference():
    """Fixtures made by the reference's EdgeLengthNormalizer(per_edge_type_cutoff=...) + BesselEdgeLengthEncoding(
    trainable=True) with perturbed roots: the oracle, and the host modules' training-mode (ATen) evaluation incl. the
    gradient w.r.t. the roots (CPU)."""
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.nn.embedding import BesselEdgeLengthEncoding, EdgeLengthNormalizer, PolynomialCutoff

    f = _load("ref_radial_basis.npz")
    r_max = float(f["r_max"])
    factor = 2 * math.pi / (r_max * r_max)
    types, ei, et = _pt_case(f)
    for name, dt, tol in (("f64", torch.float64, 1e-13), ("f32", torch.float32, 2e-6)):
        v = f["vec"].clone().requires_grad_(True)
        w = f["pt_bessel_weights"].clone().requires_grad_(True)
        emb, _ = onn.bessel_embedding(v, r_max, 8, 6.0, dt, bessel_weights=w, per_edge_type_cutoff=f["pt_cutoffs"],
                                      edge_types=et)
        _close(emb.detach() / factor, f["pt_emb_" + name], tol)
        gv, gw = torch.autograd.grad((emb / factor * f["pt_cot"].to(dt)).sum(), [v, w])
        _close(gv, f["pt_gvec_" + name], tol * 10)
        _close(gw, f["pt_gw_" + name], tol * 100)
        # host modules, training mode (differentiable roots -> the ATen formulation)
        old = torch.get_default_dtype()
        torch.set_default_dtype(dt)
        try:
            norm = EdgeLengthNormalizer(r_max=r_max, type_names=["A", "B"], per_edge_type_cutoff={"A": 3.0, "B": {"A": 4.0, "B": 2.5}})
            enc = BesselEdgeLengthEncoding(cutoff=PolynomialCutoff(6), num_bessels=8, trainable=True).train()
        finally:
            torch.set_default_dtype(old)
        assert not norm.symmetric and isinstance(enc.bessel_weights, torch.nn.Parameter)
        with torch.no_grad():
            enc.bessel_weights.copy_(f["pt_bessel_weights"])
        v2 = f["vec"].clone().requires_grad_(True)
        data = enc(norm({K.EDGE_VECTORS_KEY: v2, K.ATOM_TYPE_KEY: types, K.EDGE_INDEX_KEY: ei}))
        torch.testing.assert_close(data["_nqa_rmax_recip_edge"] * v2.detach().norm(dim=1), f["pt_normed_f64"].view(-1),
                                   atol=1e-13, rtol=1e-13)
        e2 = data[K.EDGE_EMBEDDING_KEY]
        _close(e2.detach(), f["pt_emb_" + name], tol)
        gv2, gw2 = torch.autograd.grad((e2 * f["pt_cot"].to(dt)).sum(), [v2, enc.bessel_weights])
        _close(gv2, f["pt_gvec_" + name], tol * 10)
        _close(gw2, f["pt_gw_" + name], tol * 100)


def test_oracle_scalar_mlp_matches_reference():
    f = _load("ref_scalar_mlp.npz")
    for tag, nw in (("d1", 2), ("d2", 3), ("d0", 1), ("d2w", 3)):
        ws = [f[f"{tag}_w{i}"].clone().requires_grad_(True) for i in range(nw)]
        x = f[tag + "_x"].clone().requires_grad_(True)
        y = onn.scalar_mlp(x, ws, "silu")
        _close(y.detach(), f[tag + "_y"], 2e-6)
        grads = torch.autograd.grad((y * f[tag + "_cot"]).sum(), [x] + ws)
        _close(grads[0], f[tag + "_gx"], 5e-6)
        for i in range(nw):
            _close(grads[1 + i], f[f"{tag}_gw{i}"], 5e-6)


def test_oracle_scatter_matches_reference():
    from oracle import tp as otp

    f = _load("ref_scatter.npz")
    out = otp.scatter(f["src"], f["index"], 11)
    torch.testing.assert_close(out, f["out"], atol=1e-6, rtol=1e-6)
    assert (f["out"][[4, 9, 10]] == 0).all()


# ------------------------------------------------------------------------------------------- host mirrors (CPU)
def test_host_modules_match_reference():
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.nn.atomwise import AtomwiseReduce, PerTypeScaleShift
    from nequip_amd.nn.mlp import ScalarMLPFunction
    from nequip_amd.nn.norm import AvgNumNeighborsNorm

    f = _load("ref_atomwise.npz")
    names = ["H", "O"]
    d1 = AvgNumNeighborsNorm(names, 39.5)({K.NODE_FEATURES_KEY: f["feats"].clone(), K.ATOM_TYPE_KEY: f["types"]})
    d2 = AvgNumNeighborsNorm(names, {"H": 31.0, "O": 47.0})({K.NODE_FEATURES_KEY: f["feats"].clone(), K.ATOM_TYPE_KEY: f["types"]})
    _close(d1[K.NODE_FEATURES_KEY], f["norm_scalar"], 1e-7)
    _close(d2[K.NODE_FEATURES_KEY], f["norm_per_type"], 1e-7)
    pts = PerTypeScaleShift(names, field=K.PER_ATOM_ENERGY_KEY, out_field=K.PER_ATOM_ENERGY_KEY,
                            scales={"H": 1.7, "O": 0.6}, shifts={"H": -3.1, "O": 5.5},
                            irreps_in={K.PER_ATOM_ENERGY_KEY: "0e"})
    d3 = pts({K.PER_ATOM_ENERGY_KEY: f["e_atom"].clone(), K.ATOM_TYPE_KEY: f["types"]})
    assert d3[K.PER_ATOM_ENERGY_KEY].dtype == f["e_scaled"].dtype
    _close(d3[K.PER_ATOM_ENERGY_KEY], f["e_scaled"], 1e-14)
    red = AtomwiseReduce(field=K.PER_ATOM_ENERGY_KEY, out_field=K.TOTAL_ENERGY_KEY, irreps_in={K.PER_ATOM_ENERGY_KEY: "0e"})
    d4 = red({K.PER_ATOM_ENERGY_KEY: d3[K.PER_ATOM_ENERGY_KEY], K.BATCH_KEY: f["batch"],
              K.NUM_NODES_KEY: torch.bincount(f["batch"], minlength=3)})
    _close(d4[K.TOTAL_ENERGY_KEY], f["e_total"], 1e-13)

    m = _load("ref_scalar_mlp.npz")
    for tag, (width, depth, dout) in {"d1": (64, 1, 96), "d2": (32, 2, 40), "d0": (None, 0, 24), "d2w": (64, 2, 96)}.items():
        mlp = ScalarMLPFunction(8, dout, depth, width)
        params = list(mlp.parameters())
        with torch.no_grad():
            for i, p in enumerate(params):
                p.copy_(m[f"{tag}_w{i}"])
        x = m[tag + "_x"].clone().requires_grad_(True)
        y = mlp(x)
        _close(y.detach(), m[tag + "_y"], 2e-6)
        grads = torch.autograd.grad((y * m[tag + "_cot"]).sum(), [x] + params)
        _close(grads[0], m[tag + "_gx"], 5e-6)
        for i in range(len(params)):
            _close(grads[1 + i], m[f"{tag}_gw{i}"], 5e-6)


def _pair_energy_module():
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.nn._graph_mixin import GraphModuleMixin
    from nequip_amd.nn.utils import with_edge_vectors_

    class PairEnergy(GraphModuleMixin, torch.nn.Module):
        """E = sum_e w_e |r_e|^2 exp(-|r_e|): same stand-in as in make_reference_golden.py."""

        def __init__(self):
            super().__init__()
            self._init_irreps(irreps_in={K.POSITIONS_KEY: "1o"}, irreps_out={K.TOTAL_ENERGY_KEY: "0e"})

        def forward(self, data):
            data = with_edge_vectors_(data, with_lengths=True)
            r = data[K.EDGE_LENGTH_KEY].view(-1)
            e_edge = data["edge_w"] * r * r * torch.exp(-r)
            n = data[K.POSITIONS_KEY].shape[0]
            per_atom = torch.zeros(n, dtype=r.dtype, device=r.device).index_add_(0, data[K.EDGE_INDEX_KEY][0], e_edge)
            if K.BATCH_KEY in data:
                nb = data[K.NUM_NODES_KEY].shape[0]
                tot = torch.zeros(nb, dtype=r.dtype, device=r.device).index_add_(0, data[K.BATCH_KEY], per_atom)
            else:
                tot = per_atom.sum().view(1)
            data[K.TOTAL_ENERGY_KEY] = tot.view(-1, 1)
            return data

    return PairEnergy()


def _force_stress_case(f, p, device):
    from nequip_amd.data import AtomicDataDict as K

    data = {K.POSITIONS_KEY: f[p + "_pos"].clone(), K.CELL_KEY: f[p + "_cell"].clone().view(-1, 3, 3),
            K.EDGE_INDEX_KEY: f[p + "_edge_index"], K.EDGE_CELL_SHIFT_KEY: f[p + "_shift"], "edge_w": f[p + "_w"]}
    if p == "b":
        data[K.BATCH_KEY] = f["b_batch"]
        data[K.NUM_NODES_KEY] = torch.tensor([7, 9])
    return {k: v.to(device) for k, v in data.items()}


@pytest.mark.parametrize("p", ["s", "b"])
def test_force_stress_output_matches_reference_cpu(p):
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.nn.grad_output import ForceStressOutput

    f = _load("ref_force_stress.npz")
    fso = ForceStressOutput(_pair_energy_module()).eval()
    res = fso(_force_stress_case(f, p, "cpu"))
    _close(res[K.TOTAL_ENERGY_KEY], f[p + "_energy"], 1e-13)
    _close(res[K.FORCE_KEY], f[p + "_forces"], 1e-12)
    _close(res[K.VIRIAL_KEY].view(-1, 3, 3), f[p + "_virial"].view(-1, 3, 3), 1e-12)
    _close(res[K.STRESS_KEY].view(-1, 3, 3), f[p + "_stress"].view(-1, 3, 3), 1e-12)


# ------------------------------------------------------------------------------------------------ HIP kernels (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("p", ["s", "b"])
def test_edge_vector_kernels_match_reference(device, p):
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.nn.utils import with_edge_vectors_

    f = _load("ref_edge_vectors.npz")
    pos = f[p + "_pos"].to(device).requires_grad_(True)
    cell = f[p + "_cell"].to(device).requires_grad_(True)
    data = {K.POSITIONS_KEY: pos, K.CELL_KEY: cell, K.EDGE_INDEX_KEY: f[p + "_edge_index"].to(device),
            K.EDGE_CELL_SHIFT_KEY: f[p + "_shift"].to(device)}
    if p == "b":
        data[K.BATCH_KEY] = f["b_batch"].to(device)
        data[K.NUM_NODES_KEY] = torch.tensor([7, 9], device=device)
    data = with_edge_vectors_(data, with_lengths=False)
    vec = data[K.EDGE_VECTORS_KEY]
    _close(vec.detach().cpu(), f[p + "_vec"], 1e-14)
    gp, gc = torch.autograd.grad((vec * f[p + "_coef"].to(device)).sum(), [pos, cell])
    _close(gp.cpu(), f[p + "_gpos"], 1e-13)
    _close(gc.cpu().view(f[p + "_gcell"].shape), f[p + "_gcell"], 1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("name,dtype,tol", [("f64", torch.float64, 1e-12), ("f32", torch.float32, 2e-6)])
def test_edge_embed_kernel_matches_reference(device, name, dtype, tol):
    from nequip_amd.nn.embedding._edge import _EdgeEmbedFn

    f = _load("ref_radial_basis.npz")
    r_max = float(f["r_max"])
    v = f["vec"].to(device).requires_grad_(True)
    cfg = dict(dtype=dtype, lmax=0, want_sh=False, want_emb=True, nb=8, rmax_recip=1.0 / r_max, p=6.0, factor=1.0)
    emb = _EdgeEmbedFn.apply(v, f["bessel_weights"].view(-1).double().to(device), cfg)
    emb = emb[-1] if isinstance(emb, (tuple, list)) else emb
    _close(emb.detach().cpu(), f["emb_" + name], tol)
    assert (emb[0] == 0).all() and (emb[1] == 0).all()
    (gv,) = torch.autograd.grad((emb * f["cot_" + name].to(device)).sum(), [v])
    _close(gv.cpu(), f["gvec_" + name], tol * 10)


@pytest.mark.gpu
@pytest.mark.parametrize("name,dtype,tol", [("f64", torch.float64, 1e-12), ("f32", torch.float32, 2e-6)])
def test_edge_embed_kernel_per_edge_type_cutoff_and_trained_roots(device, name, dtype, tol):
    """The HIP kernel with an [E] operand of reciprocal cutoffs and non-integer Bessel roots (eval mode of a model trained
    with ``bessel_trainable``) against the reference-generated fixture: embedding and its VJP."""
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.nn.embedding import BesselEdgeLengthEncoding, EdgeLengthNormalizer, PolynomialCutoff

    f = _load("ref_radial_basis.npz")
    types, ei, _ = _pt_case(f)
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        norm = EdgeLengthNormalizer(r_max=float(f["r_max"]), type_names=["A", "B"],
                                    per_edge_type_cutoff={"A": 3.0, "B": {"A": 4.0, "B": 2.5}})
        enc = BesselEdgeLengthEncoding(cutoff=PolynomialCutoff(6), num_bessels=8, trainable=True)
    finally:
        torch.set_default_dtype(old)
    with torch.no_grad():
        enc.bessel_weights.copy_(f["pt_bessel_weights"])
    norm, enc = norm.to(device).eval(), enc.to(device).eval()
    v = f["vec"].to(device).requires_grad_(True)
    data = enc(norm({K.EDGE_VECTORS_KEY: v, K.ATOM_TYPE_KEY: types.to(device), K.EDGE_INDEX_KEY: ei.to(device)}))
    emb = data[K.EDGE_EMBEDDING_KEY]
    _close(emb.detach().cpu(), f["pt_emb_" + name], tol)
    (gv,) = torch.autograd.grad((emb * f["pt_cot"].to(device=device, dtype=dtype)).sum(), [v])
    _close(gv.cpu(), f["pt_gvec_" + name], tol * 10)


def _set_mlp_mode(monkeypatch, mode):
    """f16x3: the default (two-plane fp16 split in both directions); bf16x6: the three-plane bf16 split; fp32: exact MFMA."""
    monkeypatch.setenv("NQA_MLP_EXACT_FP32", "1" if mode == "fp32" else "0")
    for var in ("NQA_MLP_FWD_F16", "NQA_MLP_BWD_F16"):
        if mode == "bf16x6":
            monkeypatch.setenv(var, "0")
        else:
            monkeypatch.delenv(var, raising=False)


@pytest.mark.gpu
def test_deep_radial_mlp_kernels_match_reference(device, monkeypatch):
    """The reference's tutorial radial MLP (depth 2, width 64: configs/tutorial.yaml:222-223) through the HIP path --
    ``nqa_radial_mlp_fwd`` (first two layers) + ``nqa_radial_mlp_last_fwd``, backward ``nqa_radial_mlp_last_bwd`` +
    ``nqa_radial_mlp_bwd`` -- against outputs and input gradients of the reference's own ``ScalarMLPFunction``."""
    from nequip_amd.nn import mlp as M

    _set_mlp_mode(monkeypatch, "f16x3")
    m = _load("ref_scalar_mlp.npz")
    mlp = M.ScalarMLPFunction(8, 96, 2, 64)
    with torch.no_grad():
        for i, p in enumerate(mlp.parameters()):
            p.copy_(m[f"d2w_w{i}"])
    mlp = mlp.to(device).eval()
    x = m["d2w_x"].to(device).requires_grad_(True)
    assert mlp._deep_ok(x), "depth 2 / width 64 must run on the fused kernels"
    calls = []
    real = M._launch_last
    monkeypatch.setattr(M, "_launch_last", lambda *a, **k: (calls.append(k.get("g") is not None), real(*a, **k))[1])
    y = mlp(x)
    _close(y.detach().cpu(), m["d2w_y"], 5e-6)
    (gx,) = torch.autograd.grad((y * m["d2w_cot"].to(device)).sum(), [x])
    _close(gx.cpu(), m["d2w_gx"], 1e-5)
    assert calls == [False, True]
    # ragged row counts, width 128, a third hidden layer: against the ATen formulation of the same module
    for depth, width, dout, rows in ((2, 128, 320, 1), (3, 64, 36, 131), (2, 64, 704, 1000)):
        torch.manual_seed(depth * 100 + width)
        net = M.ScalarMLPFunction(8, dout, depth, width).to(device).eval()
        xx = (torch.randn(rows, 8, device=device) * 0.7).requires_grad_(True)
        assert net._deep_ok(xx)
        yy = net(xx)
        cot = torch.randn_like(yy)
        (g1,) = torch.autograd.grad((yy * cot).sum(), [xx])
        x2 = xx.detach().clone().requires_grad_(True)
        ref = net.mlp(x2)
        (g2,) = torch.autograd.grad((ref * cot).sum(), [x2])
        assert float((yy - ref).abs().max()) <= 5e-6 * max(1.0, float(ref.abs().max()))
        assert float((g1 - g2).abs().max()) <= 1e-5 * max(1.0, float(g2.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f16x3", "bf16x6", "fp32"])
def test_radial_mlp_kernel_matches_reference(device, mode, monkeypatch):
    from nequip_amd.nn.mlp import ScalarMLPFunction

    _set_mlp_mode(monkeypatch, mode)
    m = _load("ref_scalar_mlp.npz")
    mlp = ScalarMLPFunction(8, 96, 1, 64)
    with torch.no_grad():
        for i, p in enumerate(mlp.parameters()):
            p.copy_(m[f"d1_w{i}"])
    mlp = mlp.to(device).eval()
    x = m["d1_x"].to(device).requires_grad_(True)
    assert mlp._fused_ok(x)
    y = mlp(x)
    _close(y.detach().cpu(), m["d1_y"], 5e-6)
    (gx,) = torch.autograd.grad((y * m["d1_cot"].to(device)).sum(), [x])
    _close(gx.cpu(), m["d1_gx"], 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("p", ["s", "b"])
def test_force_stress_inference_path_matches_reference(device, p):
    """ForceStressOutput's eval-mode path (edge vectors as the autograd leaf + one adjoint pass of the HIP kernel)."""
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.nn.grad_output import ForceStressOutput

    f = _load("ref_force_stress.npz")
    fso = ForceStressOutput(_pair_energy_module()).to(device).eval()
    res = fso(_force_stress_case(f, p, device))
    _close(res[K.TOTAL_ENERGY_KEY].cpu(), f[p + "_energy"], 1e-13)
    _close(res[K.FORCE_KEY].cpu(), f[p + "_forces"], 1e-12)
    _close(res[K.VIRIAL_KEY].cpu().view(-1, 3, 3), f[p + "_virial"].view(-1, 3, 3), 1e-12)
    _close(res[K.STRESS_KEY].cpu().view(-1, 3, 3), f[p + "_stress"].view(-1, 3, 3), 1e-12)
