"""Fused node kernels (nqa_node_linear / nqa_gate) against the oracle's e3nn restatements
(oracle/nn.py: o3_linear, fully_connected_tp, gate -- SURVEY.md A.5-A.7)."""

import pytest
import torch

from oracle import nn as onn

CASES = [
    ("64x0e+64x1o+64x2e", "64x0e+64x1o+64x2e"),
    ("192x0e+256x1o+256x2e", "192x0e+64x1o+64x2e"),
    ("8x0e+8x0e+8x1o+4x1o+8x2e", "12x0e+6x1o+10x2e+3x3o"),
    ("7x0e+13x1e+5x2o", "7x0e+13x1e+5x2o"),
    ("128x0e+128x1o+128x2e+128x3o", "512x0e+128x1o+128x2e+128x3o"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("irreps_in,irreps_out", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_linear_kernel(device, irreps_in, irreps_out, dtype):
    from nequip_amd.model.nequip_models import torch_default_dtype
    from nequip_amd.o3.modules import Linear

    tol = 2e-5 if dtype == torch.float32 else 1e-11
    with torch_default_dtype(dtype):
        lin = Linear(irreps_in, irreps_out)
    g = torch.Generator().manual_seed(1)
    Z = 37
    x = torch.randn(Z, lin.irreps_in.dim, generator=g, dtype=dtype)
    add = torch.randn(Z, lin.irreps_out.dim, generator=g, dtype=dtype)
    go = torch.randn(Z, lin.irreps_out.dim, generator=g, dtype=dtype)

    xr = x.clone().requires_grad_(True)
    wr = lin.weight.detach().clone().requires_grad_(True)
    ref = onn.o3_linear(xr, wr, str(lin.irreps_in), str(lin.irreps_out)) * 0.37 + add
    gx_ref, gw_ref = torch.autograd.grad(ref, [xr, wr], go)

    lin = lin.to(device)
    xd = x.to(device).requires_grad_(True)
    out = lin(xd, addend=add.to(device), scale=0.37)
    gx, gw = torch.autograd.grad(out, [xd, lin.weight], go.to(device))
    sc = float(ref.abs().max())
    torch.testing.assert_close(ref.detach(), out.detach().cpu(), atol=tol * sc, rtol=tol)
    torch.testing.assert_close(gx_ref, gx.cpu(), atol=tol * float(gx_ref.abs().max()), rtol=tol)
    torch.testing.assert_close(gw_ref, gw.cpu(), atol=tol * float(gw_ref.abs().max()), rtol=tol)


@pytest.mark.gpu
@pytest.mark.parametrize("n_types", [1, 2, 5])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_self_connection_typed_kernel(device, n_types, dtype):
    """sc(x, table[types]) with per-type pre-contracted weights == e3nn FullyConnectedTensorProduct (oracle)."""
    from nequip_amd.model.nequip_models import torch_default_dtype
    from nequip_amd.o3.modules import FullyConnectedTensorProduct

    tol = 3e-5 if dtype == torch.float32 else 1e-11
    ir_in, ir_attr, ir_out = "16x0e+16x1o+16x2e", "12x0e", "48x0e+16x1o+16x2e"
    with torch_default_dtype(dtype):
        sc = FullyConnectedTensorProduct(ir_in, ir_attr, ir_out)
    g = torch.Generator().manual_seed(2)
    Z = 29
    x = torch.randn(Z, sc.irreps_in1.dim, generator=g, dtype=dtype)
    table = torch.randn(n_types, 12, generator=g, dtype=dtype)
    types = torch.randint(0, n_types, (Z,), generator=g)
    go = torch.randn(Z, sc.irreps_out.dim, generator=g, dtype=dtype)

    xr, tr = x.clone().requires_grad_(True), table.clone().requires_grad_(True)
    wr = sc.weight.detach().clone().requires_grad_(True)
    ref = onn.fully_connected_tp(xr, tr[types], wr, ir_in, ir_attr, ir_out)
    gx_ref, gw_ref, gt_ref = torch.autograd.grad(ref, [xr, wr, tr], go)

    sc = sc.to(device)
    xd, td = x.to(device).requires_grad_(True), table.to(device).requires_grad_(True)
    out = sc.forward_typed(xd, types.to(device), td)
    gx, gw, gt = torch.autograd.grad(out, [xd, sc.weight, td], go.to(device))
    torch.testing.assert_close(ref.detach(), out.detach().cpu(), atol=tol * float(ref.abs().max()), rtol=tol)
    torch.testing.assert_close(gx_ref, gx.cpu(), atol=tol * float(gx_ref.abs().max()), rtol=tol)
    torch.testing.assert_close(gw_ref, gw.cpu(), atol=tol * float(gw_ref.abs().max()), rtol=tol)
    torch.testing.assert_close(gt_ref, gt.cpu(), atol=tol * float(gt_ref.abs().max()), rtol=tol)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_gate_kernel(device, dtype):
    from nequip_amd.o3.modules import Gate

    tol = 1e-5 if dtype == torch.float32 else 1e-12
    silu, tanh = torch.nn.functional.silu, torch.tanh
    for sc, acts, ga, actg, gd in [
        ("64x0e", [silu], "128x0e", [silu], "64x1o+64x2e"),
        ("8x0e+8x0o", [silu, tanh], "16x0e", [silu], "4x1e+4x1o+4x2e+4x2o"),
        ("16x0e", [silu], "", [], ""),
    ]:
        gate = Gate(sc, acts, ga, actg, gd).eval()
        g = torch.Generator().manual_seed(4)
        Z = 33
        x = torch.randn(Z, gate.irreps_in.dim, generator=g, dtype=dtype)
        go = torch.randn(Z, gate.irreps_out.dim, generator=g, dtype=dtype)
        names = {silu: "silu", tanh: "tanh"}
        xr = x.clone().requires_grad_(True)
        ref = onn.gate(xr, str(gate.irreps_scalars), [names[a] for a in acts], str(gate.irreps_gates) if ga else [],
                       [names[a] for a in actg], str(gate.irreps_gated) if gd else [])
        (gx_ref,) = torch.autograd.grad(ref, xr, go)
        xd = x.to(device).requires_grad_(True)
        out = gate.to(device)(xd)
        (gx,) = torch.autograd.grad(out, xd, go.to(device))
        torch.testing.assert_close(ref.detach(), out.detach().cpu(), atol=tol, rtol=tol)
        torch.testing.assert_close(gx_ref, gx.cpu(), atol=tol * max(1.0, float(gx_ref.abs().max())), rtol=tol)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_gate_kernel_double_backward(device, dtype):
    """Second order of the gate kernels (what force-matching training differentiates: grad_in as a function of x and
    grad_out) against autograd through the oracle's Gate restatement."""
    from nequip_amd.o3.modules import Gate

    tol = 2e-5 if dtype == torch.float32 else 1e-11
    silu, tanh = torch.nn.functional.silu, torch.tanh
    names = {silu: "silu", tanh: "tanh"}
    for sc, acts, ga, actg, gd in [
        ("64x0e", [silu], "128x0e", [silu], "64x1o+64x2e"),
        ("8x0e+8x0o", [silu, tanh], "16x0e", [silu], "4x1e+4x1o+4x2e+4x2o"),
        ("16x0e", [silu], "", [], ""),
    ]:
        gate = Gate(sc, acts, ga, actg, gd).train()  # kernels in training mode too
        g = torch.Generator().manual_seed(9)
        Z = 21
        x = torch.randn(Z, gate.irreps_in.dim, generator=g, dtype=dtype)
        go = torch.randn(Z, gate.irreps_out.dim, generator=g, dtype=dtype)
        cot = torch.randn(Z, gate.irreps_in.dim, generator=g, dtype=dtype)

        def second_order(fn, x0, go0, cot0):
            xr = x0.clone().requires_grad_(True)
            gr = go0.clone().requires_grad_(True)
            (gin,) = torch.autograd.grad(fn(xr), xr, gr, create_graph=True)
            gx2, gg2 = torch.autograd.grad(gin, [xr, gr], cot0)
            return gin.detach(), gx2, gg2

        ref = second_order(lambda t: onn.gate(t, str(gate.irreps_scalars), [names[a] for a in acts],
                                              str(gate.irreps_gates) if ga else [], [names[a] for a in actg],
                                              str(gate.irreps_gated) if gd else []), x, go, cot)
        gate_d = gate.to(device)
        out = second_order(gate_d, x.to(device), go.to(device), cot.to(device))
        for r, o in zip(ref, out):
            torch.testing.assert_close(r, o.cpu(), atol=tol * max(1.0, float(r.abs().max())), rtol=tol)
