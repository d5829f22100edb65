"""Dispatcher-op form of the tensor-product scatter (nequip_amd/nn/_tp_scatter_ops.py): schema, fake (meta) kernels and
autograd registration are checked here without a GPU by tracing through FakeTensor / make_fx, the way the reference's
compile path does (nequip/nn/compile.py:176-191); the numbers are compared with the autograd-Function form on the GPU."""
import pytest
import torch

from nequip_amd.nn import _tp_scatter_ops as ops
from nequip_amd.o3.irreps import Irreps

IN1, IN2, OUT = "8x0e+8x1o", "1x0e+1x1o", "8x0e+8x1o+8x1o+8x0e"
INSTR = [(0, 0, 0, "uvu", True), (0, 1, 1, "uvu", True), (1, 0, 2, "uvu", True), (1, 1, 3, "uvu", True)]


def test_plan_key_round_trip():
    key = ops.plan_key(IN1, IN2, OUT, INSTR)
    i1, i2, io, ins = ops._parse(key)
    assert (str(i1), str(i2), str(io)) == (str(Irreps(IN1)), str(Irreps(IN2)), str(Irreps(OUT)))
    assert [(i.i_in1, i.i_in2, i.i_out) for i in ins] == [t[:3] for t in INSTR]
    assert ops.plan_dims(key) == (Irreps(IN1).dim, Irreps(IN2).dim, Irreps(OUT).dim, 32)
    assert ops.plan_key(i1, i2, io, ins) == key


def test_ops_are_registered_with_schema():
    fwd = torch.ops.nequip_amd.tp_scatter_fwd.default
    bwd = torch.ops.nequip_amd.tp_scatter_bwd.default
    assert [a.name for a in fwd._schema.arguments] == ["x", "edge_attr", "edge_weight", "edge_dst", "edge_src", "plan"]
    assert len(bwd._schema.returns) == 3


def test_cpu_tensors_raise():
    key = ops.plan_key(IN1, IN2, OUT, INSTR)
    d1, d2, do, wn = ops.plan_dims(key)
    idx = torch.zeros(3, dtype=torch.long)
    with pytest.raises((NotImplementedError, RuntimeError)):
        ops.tp_scatter(torch.randn(2, d1), torch.randn(3, d2), torch.randn(3, wn), idx, idx, key)


def _fake_inputs(mode, N, E, key, device="cpu"):
    d1, d2, do, wn = ops.plan_dims(key)
    with mode:
        x = torch.randn(N, d1, device=device, requires_grad=True)
        y = torch.randn(E, d2, device=device, requires_grad=True)
        w = torch.randn(E, wn, device=device, requires_grad=True)
        dst = torch.zeros(E, dtype=torch.long, device=device)
        src = torch.zeros(E, dtype=torch.long, device=device)
    return x, y, w, dst, src


def test_fake_forward_and_double_backward_shapes():
    from torch._subclasses.fake_tensor import FakeTensorMode

    key = ops.plan_key(IN1, IN2, OUT, INSTR)
    d1, d2, do, wn = ops.plan_dims(key)
    mode = FakeTensorMode()
    x, y, w, dst, src = _fake_inputs(mode, 5, 11, key)
    with mode:
        out = ops.tp_scatter(x, y, w, dst, src, key)
        assert out.shape == (5, do)
        gx, gy, gw = torch.autograd.grad(out.sum(), [x, y, w], create_graph=True)
        assert gx.shape == x.shape and gy.shape == y.shape and gw.shape == w.shape
        # second order: every input receives a gradient of a scalar built from the first derivatives
        s = (gx * gx).sum() + (gy * gy).sum() + (gw * gw).sum()
        hx, hy, hw = torch.autograd.grad(s, [x, y, w])
        assert hx.shape == x.shape and hy.shape == y.shape and hw.shape == w.shape
        with pytest.raises(RuntimeError):
            ops.tp_scatter(x[:, :-1], y, w, dst, src, key)


def test_make_fx_symbolic_trace_contains_only_our_ops():
    from torch.fx.experimental.proxy_tensor import make_fx

    key = ops.plan_key(IN1, IN2, OUT, INSTR)
    d1, d2, do, wn = ops.plan_dims(key)

    def energy_and_forces(x, y, w, dst, src):
        out = ops.tp_scatter(x, y, w, dst, src, key)
        (gy,) = torch.autograd.grad(out.square().sum(), [y], create_graph=True)
        return out, gy

    N, E = 4, 9
    args = (torch.randn(N, d1), torch.randn(E, d2, requires_grad=True), torch.randn(E, wn),
            torch.zeros(E, dtype=torch.long), torch.zeros(E, dtype=torch.long))
    gm = make_fx(energy_and_forces, tracing_mode="symbolic")(*args)
    targets = {str(n.target) for n in gm.graph.nodes if n.op == "call_function"}
    assert "nequip_amd.tp_scatter_fwd.default" in targets and "nequip_amd.tp_scatter_bwd.default" in targets


def test_module_selects_dispatcher_form():
    from nequip_amd.nn import TensorProductScatter

    m = TensorProductScatter(IN1, IN2, OUT, INSTR, use_dispatcher_ops=True)
    assert m.use_dispatcher_ops and m._plan_key == ops.plan_key(IN1, IN2, OUT, INSTR)
    assert not TensorProductScatter(IN1, IN2, OUT, INSTR).use_dispatcher_ops


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_dispatcher_form_matches_function_form(device, dtype):
    """Same kernels behind both forms: outputs, first and second derivatives agree bit for bit."""
    from nequip_amd.model.nequip_models import torch_default_dtype
    from nequip_amd.nn import TensorProductScatter

    torch.manual_seed(0)
    N, E = 40, 300
    with torch_default_dtype(dtype):
        ref_m = TensorProductScatter(IN1, IN2, OUT, INSTR).to(device)
        ops_m = TensorProductScatter(IN1, IN2, OUT, INSTR, use_dispatcher_ops=True).to(device)
    d1, d2, do, wn = ops.plan_dims(ops_m._plan_key)
    dst = torch.randint(0, N, (E,), device=device)
    src = torch.randint(0, N, (E,), device=device)
    base = [torch.randn(N, d1, dtype=dtype, device=device), torch.randn(E, d2, dtype=dtype, device=device),
            torch.randn(E, wn, dtype=dtype, device=device)]
    v = torch.randn(N, do, dtype=dtype, device=device)

    def run(m):
        x, y, w = [t.clone().requires_grad_(True) for t in base]
        out = m(x, y, w, dst, src)
        g = torch.autograd.grad((out * v).sum(), [x, y, w], create_graph=True)
        s = sum((t * t).sum() for t in g)
        h = torch.autograd.grad(s, [x, y, w])
        return [out.detach()] + [t.detach() for t in g] + list(h)

    for a, b in zip(run(ref_m), run(ops_m)):
        torch.testing.assert_close(a, b, rtol=0, atol=0)


@pytest.mark.gpu
def test_opcheck(device):
    key = ops.plan_key(IN1, IN2, OUT, INSTR)
    d1, d2, do, wn = ops.plan_dims(key)
    N, E = 6, 20
    x = torch.randn(N, d1, device=device, requires_grad=True)
    y = torch.randn(E, d2, device=device, requires_grad=True)
    w = torch.randn(E, wn, device=device, requires_grad=True)
    dst = torch.randint(0, N, (E,), device=device)
    src = torch.randint(0, N, (E,), device=device)
    torch.library.opcheck(torch.ops.nequip_amd.tp_scatter_fwd.default, (x, y, w, dst, src, key),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
