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


@pytest.mark.gpu
@pytest.mark.parametrize("typed", [False, True])
def test_node_linear_constant_weight_modes_over_a_wide_dynamic_range(device, monkeypatch, typed):
    """Constant weights (eval mode) run on 16-bit MFMA operands: two fp16 planes with exact power-of-two scales -- per K
    block of the weights, and a RUNNING one per output column that is lowered, with the column's accumulators, when a
    later block outgrows it -- (default), three bf16 planes (NQA_NODE_F16=0), or exact fp32 (NQA_NODE_EXACT_FP32=1).
    Atoms from 1e-12 to 1e+8, irrep blocks six orders apart, channel groups inside a block twenty orders apart in
    either order, weight matrices of 1e-5 and 1e+4, non-multiple-of-16 multiplicities: every output element within the
    fp32 level of its magnitude sum  sum |x| |W|, forward and transposed."""
    from nequip_amd.o3._node_kernels import NodeLinearMeta, _launch_linear, _transposed, meta_transposed_weights
    from nequip_amd.o3.irreps import Irreps

    torch.manual_seed(2)
    s_in = "64x0e+64x0e+40x1o+64x1o+70x2e+64x3o"
    s_out = "128x0e+96x1o+64x2e+16x3o"
    ins = [(0, 0), (1, 0), (2, 1), (3, 1), (4, 2), (5, 3)]
    i_in, i_out = Irreps(s_in), Irreps(s_out)
    meta = NodeLinearMeta(i_in, i_out, ins)
    N, T = 203, (3 if typed else 1)
    x = torch.randn(N, i_in.dim)
    atom = torch.ones(N)
    atom[20:60] = 1e-12
    atom[60:90] = 1e8
    x *= atom[:, None]
    off = i_in.offsets()
    x[:, off[1]:off[2]] *= 1e6                      # a whole block far above its neighbour feeding the same output
    x[100:140, off[3]:off[3] + 16 * 3] *= 1e-20     # small channels first, then O(1): rescale inside the block
    x[140:180, off[3] + 16 * 3:off[4]] *= 1e-20     # large channels first
    x[7] = 0.0
    wp = torch.randn(T, meta.wstride)
    for k, (i, o) in enumerate(ins):
        n = i_in[i].mul * i_out[o].mul
        if k == 2:
            wp[:, meta.w_off[k]:meta.w_off[k] + n] *= 1e-5
        if k == 4:
            wp[:, meta.w_off[k]:meta.w_off[k] + n] *= 1e4
    types = torch.randint(0, T, (N,)) if typed else None

    def reference(xx, w, m):
        out = torch.zeros(xx.shape[0], m.dout, dtype=torch.float64)
        mag = torch.zeros_like(out)
        io, oo = m.irreps_in.offsets(), m.irreps_out.offsets()
        for k, (i, o) in enumerate(m.instructions):
            mi, ir = m.irreps_in[i]
            mo = m.irreps_out[o].mul
            d = ir.dim
            W = w[:, m.w_off[k]:m.w_off[k] + mi * mo].view(-1, mi, mo).double()
            Wz = W[types] if typed else W[0].expand(xx.shape[0], mi, mo)
            xb = xx[:, io[i]:io[i] + mi * d].view(-1, mi, d).double()
            out[:, oo[o]:oo[o] + mo * d] += torch.einsum("zum,zuw->zwm", xb, Wz).reshape(-1, mo * d)
            mag[:, oo[o]:oo[o] + mo * d] += torch.einsum("zum,zuw->zwm", xb.abs(), Wz.abs()).reshape(-1, mo * d)
        return out, mag.clamp_min(1e-300)

    cases = [("fwd", x, wp, meta)]
    xt = torch.randn(N, i_out.dim) * atom[:, None]
    xt[7] = 0.0
    cases.append(("transposed", xt, meta_transposed_weights(meta, wp), _transposed(meta)))
    for label, xx, w, m in cases:
        ref, mag = reference(xx, w, m)
        errs = {}
        for mode, env in (("f16", {"NQA_NODE_F16": "1", "NQA_NODE_EXACT_FP32": "0"}),
                          ("bf16", {"NQA_NODE_F16": "0", "NQA_NODE_EXACT_FP32": "0"}),
                          ("fp32", {"NQA_NODE_F16": "1", "NQA_NODE_EXACT_FP32": "1"})):
            for k_, v_ in env.items():
                monkeypatch.setenv(k_, v_)
            out = _launch_linear(xx.to(device), w.to(device).contiguous(), None, None if types is None else types.to(device),
                                 m, "fwd", 1.0).cpu().double()
            assert torch.isfinite(out).all(), (label, mode)
            assert torch.equal(out[7], torch.zeros(m.dout, dtype=torch.float64)), (label, mode)
            errs[mode] = float(((out - ref).abs() / mag).max())
        print(label, "max error / magnitude sum:", errs)
        assert all(e < 1e-6 for e in errs.values()), (label, errs)
        assert errs["f16"] < 3 * max(errs["bf16"], errs["fp32"]) + 2e-7, (label, errs)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f16", "fp32"])
def test_typed_node_linear_atom_order_is_a_pure_schedule(device, monkeypatch, mode):
    """``nqa_node_linear_ordered`` / ``nqa_node_linear_packed_ordered``: the atom order only decides which atoms share a
    work unit (and which typed stages that unit can skip) -- natural order, grouped by type, or a random permutation give
    the same output bit for bit, including a type that no atom has and N not a multiple of the unit."""
    import ctypes

    from nequip_amd import _lib
    from nequip_amd.o3._node_kernels import NodeLinearMeta, _launch_linear, _ptr, _stream, packed_weights
    from nequip_amd.o3.irreps import Irreps

    monkeypatch.setenv("NQA_NODE_EXACT_FP32", "1" if mode == "fp32" else "0")
    monkeypatch.setenv("NQA_NODE_F16", "1")
    torch.manual_seed(5)
    i_in, i_out = Irreps("64x0e+40x1o+24x2e"), Irreps("96x0e+64x1o+16x2e")
    meta = NodeLinearMeta(i_in, i_out, [(0, 0), (1, 1), (2, 2)])
    N, T = 1237, 6
    x = torch.randn(N, i_in.dim, device=device)
    wp = torch.randn(T, meta.wstride, device=device)
    types = torch.randint(0, T - 1, (N,), device=device)  # (type T-1 is absent)
    types[:70] = 3                                         # (a run of units holding one type only)

    monkeypatch.setenv("NQA_NODE_TYPE_ORDER", "0")
    natural = _launch_linear(x, wp, None, types, meta, "fwd", 0.5)
    monkeypatch.setenv("NQA_NODE_TYPE_ORDER", "1")
    grouped = _launch_linear(x, wp, None, types, meta, "fwd", 0.5)
    assert torch.equal(natural, grouped)

    lib = _lib.load()
    ct, nchunks, it, ninstr = meta.host_tables("fwd")
    perm = torch.randperm(N, device=device).to(torch.int32)
    out = torch.empty_like(natural)
    if mode == "fp32":
        rc = lib.nqa_node_linear_ordered(_lib.NQA_F32, _ptr(x), _ptr(wp), None, _ptr(out), _ptr(types), _ptr(perm),
                                         ctypes.cast(ct, ctypes.c_void_p), nchunks, ctypes.cast(it, ctypes.c_void_p),
                                         ninstr, T, wp.shape[1], meta.din, meta.dout, N, 0.5, 64, _stream(x.device))
    else:
        wf = packed_weights(wp, meta, "fwd")
        rc = lib.nqa_node_linear_packed_ordered(_ptr(x), _ptr(wf), None, _ptr(out), _ptr(types), _ptr(perm),
                                                ctypes.cast(ct, ctypes.c_void_p), nchunks,
                                                ctypes.cast(it, ctypes.c_void_p), ninstr, T, meta.din, meta.dout, N, 0.5,
                                                _stream(x.device))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(natural, out)
