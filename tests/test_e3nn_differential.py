"""Differential tests against e3nn itself -- run only where ``e3nn`` (>= 0.6, < 0.7: the reference's pin,
``/root/reference/pyproject.toml:22``) can be imported; skipped otherwise.

e3nn is absent from the build image and from the GPU boxes, so the e3nn-owned rows of SURVEY.md 8(a) (a2 spherical
harmonics, a5 ``o3.Linear``, a7-a9 the ``uvu`` TensorProduct, a10 FullyConnectedTensorProduct, a11 Gate) are pinned to sympy
and to algebra only ("parity partly unpinned").  This module closes that pin by itself on the day the
environment gains e3nn:

* CPU: ``oracle/`` (the checker of every GPU parity test) against e3nn -- ``wigner_3j`` (incl. the sign of the odd-sum
  tensors), ``spherical_harmonics(normalize=True, normalization="component")``, the ``uvu`` tensor product with external
  weights and its path normalisation, ``Linear``, ``FullyConnectedTensorProduct``, ``Gate`` with its Monte-Carlo
  ``normalize2mom`` constants -- and the host mirrors ``nequip_amd.o3`` against the same modules (state-dict compatible).
* GPU: the reference's own boundary test (``tests/unit/nn/test_tp_scatter_kernel.py:34-179``: irreps matrix, instructions
  built like InteractionBlock, 8 nodes / 15 edges with random repeated indices, forward + gradient w.r.t. x / edge_attr /
  edge_weight, atol = rtol = 1e-5 / 1e-10) with e3nn's ``TensorProduct`` + ``scatter`` as ``tp_base`` and the HIP
  ``TensorProductScatter`` as ``tp_kernel``; with nequip installed as well, the kernel module is the one
  ``enable_NequipAMD`` produces from nequip's own ``TensorProductScatter``.
"""

import math

import pytest
import torch

e3nn = pytest.importorskip("e3nn", reason="e3nn is not installed (the e3nn-owned rows stay pinned to sympy / algebra)")
from e3nn import o3 as eo3  # noqa: E402

from oracle import nn as onn  # noqa: E402
from oracle import sh as osh  # noqa: E402
from oracle import tp as otp  # noqa: E402
from oracle import wigner as owig  # noqa: E402

NUM_NODES = 8
NUM_EDGES = 15
FEATURE_IRREPS = ["4x0e + 3x1o + 2x2e", "2x0e + 2x1o + 2x2e", "8x0e + 8x2e + 8x1o"]
EDGE_ATTR_IRREPS = ["0e + 1o", "0e + 1o + 2e"]
MID_IRREPS = ["0e + 1o + 2e", "2x0e + 2x1o + 2x2e", "24x0e + 32x1o + 16x1e + 16x2o + 32x2e"]


@pytest.mark.parametrize("l1,l2,l3", [(a, b, c) for a in range(5) for b in range(5) for c in range(abs(a - b), min(a + b, 4) + 1)])
def test_wigner_3j_equals_e3nn(l1, l2, l3):
    """Values AND sign (the overall sign of an odd-sum tensor is invisible to the algebraic checks of test_oracle.py)."""
    ref = eo3.wigner_3j(l1, l2, l3).to(torch.float64)
    torch.testing.assert_close(owig.wigner_3j(l1, l2, l3).to(torch.float64), ref, atol=1e-12, rtol=0)
    from nequip_amd.o3.wigner import wigner_3j as host_3j

    torch.testing.assert_close(torch.as_tensor(host_3j(l1, l2, l3), dtype=torch.float64), ref, atol=1e-12, rtol=0)


@pytest.mark.parametrize("lmax", [1, 2, 3, 4])
def test_spherical_harmonics_equal_e3nn(lmax):
    g = torch.Generator().manual_seed(lmax)
    vec = torch.randn(257, 3, generator=g, dtype=torch.float64) * 2.0
    ref = eo3.spherical_harmonics(list(range(lmax + 1)), vec, normalize=True, normalization="component")
    torch.testing.assert_close(osh.spherical_harmonics(vec, lmax, normalize=True), ref, atol=1e-12, rtol=1e-12)


def _instructions(feature_irreps_in, irreps_edge_attr, irreps_mid_filter):
    """InteractionBlock's construction with e3nn objects (nequip/nn/interaction_block.py:89-109)."""
    f_in, e_at, mid_f = eo3.Irreps(feature_irreps_in), eo3.Irreps(irreps_edge_attr), eo3.Irreps(irreps_mid_filter)
    mids, ins = [], []
    for i, (mul, ir_in) in enumerate(f_in):
        for j, (_, ir_edge) in enumerate(e_at):
            for ir_out in ir_in * ir_edge:
                if ir_out in mid_f:
                    k = len(mids)
                    mids.append((mul, ir_out))
                    ins.append((i, j, k, "uvu", True))
    if not ins:
        return None
    mid, p, _ = eo3.Irreps(mids).sort()
    return f_in, e_at, mid, [(a, b, p[c], m, t) for a, b, c, m, t in ins]


@pytest.mark.parametrize("feature_irreps_in", FEATURE_IRREPS + ["16x0e + 16x1o + 16x2e + 16x3o", "4x0e + 4x0o + 4x1e + 4x1o + 4x2e + 4x2o"])
@pytest.mark.parametrize("irreps_edge_attr", EDGE_ATTR_IRREPS + ["0e + 1o + 2e + 3o"])
@pytest.mark.parametrize("irreps_mid", MID_IRREPS + ["0e + 0o + 1e + 1o + 2e + 2o + 3e + 3o"])
def test_oracle_uvu_tensor_product_equals_e3nn(feature_irreps_in, irreps_edge_attr, irreps_mid):
    built = _instructions(feature_irreps_in, irreps_edge_attr, irreps_mid)
    if built is None:
        pytest.skip("No valid tensor product instructions generated")
    f_in, e_at, mid, ins = built
    tp = eo3.TensorProduct(f_in, e_at, mid, ins, shared_weights=False, internal_weights=False).double()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(NUM_EDGES, f_in.dim, generator=g, dtype=torch.float64)
    y = torch.randn(NUM_EDGES, e_at.dim, generator=g, dtype=torch.float64)
    w = torch.randn(NUM_EDGES, tp.weight_numel, generator=g, dtype=torch.float64)
    ref = tp(x, y, w)
    o_mid, o_ins = otp.build_instructions(feature_irreps_in, irreps_edge_attr, irreps_mid)
    assert [tuple(i)[:3] for i in o_ins] == [tuple(i)[:3] for i in ins]
    assert otp.weight_numel(str(f_in), str(e_at), o_ins) == tp.weight_numel
    out = otp.tensor_product_uvu(x, y, w, str(f_in), str(e_at), str(mid), o_ins)
    torch.testing.assert_close(out, ref, atol=1e-11, rtol=1e-11)


@pytest.mark.parametrize("irreps_in,irreps_out", [("64x0e+64x1o+64x2e", "64x0e+64x1o+64x2e"),
                                                   ("16x0e+16x0o+16x1e+16x1o", "32x0e+8x1o"),
                                                   ("8x0e+8x1o+8x2e+8x3o", "24x0e+8x1o+8x2e+8x3o")])
def test_linear_equals_e3nn(irreps_in, irreps_out):
    from nequip_amd.o3.modules import Linear

    torch.manual_seed(0)
    ref_mod = eo3.Linear(irreps_in, irreps_out).double()
    x = torch.randn(19, eo3.Irreps(irreps_in).dim, dtype=torch.float64)
    ref = ref_mod(x)
    torch.testing.assert_close(onn.o3_linear(x, ref_mod.weight.detach(), irreps_in, irreps_out), ref, atol=1e-11, rtol=1e-11)
    mine = Linear(irreps_in, irreps_out).double()
    assert mine.weight.shape == ref_mod.weight.shape
    mine.load_state_dict({"weight": ref_mod.weight.detach().clone()}, strict=False)
    torch.testing.assert_close(mine(x), ref, atol=1e-11, rtol=1e-11)


@pytest.mark.parametrize("ir1,ir2,iro", [("16x0e+16x1o+16x2e", "12x0e", "48x0e+16x1o+16x2e"),
                                         ("8x0e+8x0o+8x1e+8x1o", "5x0e", "16x0e+8x0o+8x1e+8x1o")])
def test_fully_connected_tensor_product_equals_e3nn(ir1, ir2, iro):
    from nequip_amd.o3.modules import FullyConnectedTensorProduct

    torch.manual_seed(1)
    ref_mod = eo3.FullyConnectedTensorProduct(ir1, ir2, iro).double()
    x = torch.randn(13, eo3.Irreps(ir1).dim, dtype=torch.float64)
    a = torch.randn(13, eo3.Irreps(ir2).dim, dtype=torch.float64)
    ref = ref_mod(x, a)
    torch.testing.assert_close(onn.fully_connected_tp(x, a, ref_mod.weight.detach().view(-1), ir1, ir2, iro), ref,
                               atol=1e-11, rtol=1e-11)
    mine = FullyConnectedTensorProduct(ir1, ir2, iro).double()
    assert mine.weight.numel() == ref_mod.weight.numel()
    with torch.no_grad():
        mine.weight.copy_(ref_mod.weight.detach().view_as(mine.weight))
    torch.testing.assert_close(mine(x, a), ref, atol=1e-11, rtol=1e-11)


@pytest.mark.parametrize("case", [("64x0e", ["silu"], "128x0e", ["silu"], "64x1o+64x2e"),
                                  ("8x0e+8x0o", ["silu", "tanh"], "16x0e", ["silu"], "4x1e+4x1o+4x2e+4x2o")])
def test_gate_and_normalize2mom_equal_e3nn(case):
    from e3nn.nn import Gate as EGate

    from nequip_amd.o3.modules import Gate

    sc, acts, ga, actg, gd = case
    fns = {"silu": torch.nn.functional.silu, "tanh": torch.tanh}
    ref_mod = EGate(sc, [fns[a] for a in acts], ga, [fns[a] for a in actg], gd)
    x = torch.randn(21, ref_mod.irreps_in.dim, dtype=torch.float64)
    ref = ref_mod(x)
    torch.testing.assert_close(onn.gate(x, sc, acts, ga, actg, gd), ref, atol=1e-9, rtol=1e-9)
    mine = Gate(sc, [fns[a] for a in acts], ga, [fns[a] for a in actg], gd)
    assert str(mine.irreps_in) == str(ref_mod.irreps_in) and str(mine.irreps_out) == str(ref_mod.irreps_out)
    torch.testing.assert_close(mine(x), ref, atol=1e-9, rtol=1e-9)
    for name in ("silu", "tanh"):  # the constants e3nn-trained checkpoints assume (Monte-Carlo second moment, seed 0)
        from e3nn.math import normalize2mom

        assert math.isclose(onn.normalize2mom_const(name), float(normalize2mom(fns[name]).cst), rel_tol=1e-7)


# ---- the reference's boundary test, verbatim recipe, with real e3nn as tp_base --------------------------------------------
class _E3nnTPScatter(torch.nn.Module):
    """nequip/nn/_tp_scatter_base.py:9-38 (e3nn TensorProduct + scatter), used when nequip itself is not importable."""

    def __init__(self, feature_irreps_in, irreps_edge_attr, irreps_mid, instructions):
        super().__init__()
        self.tp = eo3.TensorProduct(feature_irreps_in, irreps_edge_attr, irreps_mid, instructions, shared_weights=False,
                                    internal_weights=False)

    def forward(self, x, edge_attr, edge_weight, edge_dst, edge_src):
        edge_features = self.tp(x[edge_src], edge_attr, edge_weight)
        out = edge_features.new_zeros(x.size(0), edge_features.size(1))
        return out.index_add_(0, edge_dst, edge_features)


@pytest.mark.gpu
@pytest.mark.parametrize("feature_irreps_in", FEATURE_IRREPS)
@pytest.mark.parametrize("irreps_edge_attr", EDGE_ATTR_IRREPS)
@pytest.mark.parametrize("irreps_mid", MID_IRREPS)
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_tp_scatter_kernel_against_e3nn(device, feature_irreps_in, irreps_edge_attr, irreps_mid, dtype):
    from nequip_amd.model.nequip_models import torch_default_dtype

    built = _instructions(feature_irreps_in, irreps_edge_attr, irreps_mid)
    if built is None:
        pytest.skip("No valid tensor product instructions generated")
    f_in, e_at, mid, instructions = built
    tdtype = {"float32": torch.float32, "float64": torch.float64}[dtype]
    with torch_default_dtype(tdtype):
        try:  # nequip importable: its own module, and the modifier the extension registers on it
            from nequip.nn._tp_scatter_base import TensorProductScatter as RefTPS

            from nequip_amd.integrations import nequip_extension

            nequip_extension.register()
            tp_base = RefTPS(feature_irreps_in=f_in, irreps_edge_attr=e_at, irreps_mid=mid, instructions=instructions).to(device)
            holder = torch.nn.ModuleDict({"tp_scatter": RefTPS(feature_irreps_in=f_in, irreps_edge_attr=e_at, irreps_mid=mid,
                                                               instructions=instructions)})
            holder.is_compile_graph_model = False
            tp_kernel = RefTPS.enable_NequipAMD(holder)["tp_scatter"].to(device)
        except ImportError:
            from nequip_amd.nn import TensorProductScatter

            tp_base = _E3nnTPScatter(f_in, e_at, mid, instructions).to(device)
            tp_kernel = TensorProductScatter(feature_irreps_in=f_in, irreps_edge_attr=e_at, irreps_mid=mid,
                                             instructions=instructions).to(device)
        tp_kernel.eval()
        g = torch.Generator(device="cpu").manual_seed(0)
        x = f_in.randn(NUM_NODES, -1).to(device)
        edge_attr = e_at.randn(NUM_EDGES, -1).to(device)
        edge_weight = torch.randn(NUM_EDGES, tp_base.tp.weight_numel, generator=g).to(device)
        edge_src = torch.randint(0, NUM_NODES, (NUM_EDGES,), generator=g).to(device)
        edge_dst = torch.randint(0, NUM_NODES, (NUM_EDGES,), generator=g).to(device)
        tol = {torch.float32: 1e-5, torch.float64: 1e-10}[torch.get_default_dtype()]
        with torch.no_grad():
            torch.testing.assert_close(tp_base(x, edge_attr, edge_weight, edge_dst, edge_src),
                                       tp_kernel(x, edge_attr, edge_weight, edge_dst, edge_src), atol=tol, rtol=tol)
        for name in ["x", "edge_attr", "edge_weight"]:
            inputs = {"x": x, "edge_attr": edge_attr, "edge_weight": edge_weight}
            inputs[name].requires_grad_(True)
            out_base = tp_base(**inputs, edge_dst=edge_dst, edge_src=edge_src)
            out_kernel = tp_kernel(**inputs, edge_dst=edge_dst, edge_src=edge_src)
            grad_output = torch.randn_like(out_base)
            grad_base = torch.autograd.grad(out_base, inputs[name], grad_output, retain_graph=True)[0]
            grad_kernel = torch.autograd.grad(out_kernel, inputs[name], grad_output, retain_graph=True)[0]
            torch.testing.assert_close(grad_base, grad_kernel, atol=tol, rtol=tol)
            inputs[name].requires_grad_(False)
