"""Dispatcher-op form of the tensor-product scatter: ``torch.ops.nequip_amd.tp_scatter_fwd`` / ``tp_scatter_bwd``.

The reference's accelerated adapters choose between a traceable and an *opaque* custom-op form of their kernel when the
model is compiled (``nequip/nn/_tp_scatter_oeq.py:13,26-47`` ``use_opaque``; factories read
``model.is_compile_graph_model``, ``nequip/nn/_tp_scatter_base.py:60-69``) because ``torch.compile(dynamic=True)`` /
``make_fx`` symbolic tracing (``nequip/nn/compile.py:176-191``) cannot look inside a Python autograd Function that
calls a C library through raw pointers.  These two ops are that form for ``libnequip_amd.so``:

* ``tp_scatter_fwd(x, edge_attr, edge_weight, edge_dst, edge_src, plan) -> out``
* ``tp_scatter_bwd(grad_out, x, edge_attr, edge_weight, edge_dst, edge_src, plan, need_x, need_y, need_w)
  -> (grad_x, grad_edge_attr, grad_edge_weight)``  (an operand that is not needed comes back as an empty tensor)

with fake (meta) implementations for shape propagation and autograd formulas registered on both, so that first and
second derivatives trace into the same two ops: the op is trilinear, hence the family {fwd, bwd} is closed under
differentiation (same algebra as ``_TPScatterBwdFn.backward``).  ``plan`` is the canonical text of the
four constructor arguments; the native plan is rebuilt from it on first use per device, so a traced / exported graph
carries no Python object.  There is no CPU kernel: the CPU dispatch key is not registered, calling the ops with CPU
tensors raises.  Eager execution keeps using the autograd Functions of ``_tp_scatter_base`` (fewer dispatcher hops);
``TensorProductScatter(..., use_dispatcher_ops=True)`` or tracing selects this form.
"""

from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from ..o3.irreps import Irreps
from ..o3.tensor_product import NativePlan, _normalize_instructions

_NS = "nequip_amd"
_lib_def = torch.library.Library(_NS, "FRAGMENT")
_lib_def.define("tp_scatter_fwd(Tensor x, Tensor edge_attr, Tensor edge_weight, Tensor edge_dst, Tensor edge_src, "
                "str plan) -> Tensor")
_lib_def.define("tp_scatter_bwd(Tensor grad_out, Tensor x, Tensor edge_attr, Tensor edge_weight, Tensor edge_dst, "
                "Tensor edge_src, str plan, bool need_x, bool need_y, bool need_w) -> (Tensor, Tensor, Tensor)")


def plan_key(irreps_in1, irreps_in2, irreps_out, instructions) -> str:
    """Canonical text of the TensorProductScatter constructor arguments (``_tp_scatter_base.py:10-16``)."""
    ins = ";".join(f"{i.i_in1},{i.i_in2},{i.i_out},{i.path_weight!r}" for i in _normalize_instructions(instructions))
    return f"{Irreps(str(irreps_in1))}|{Irreps(str(irreps_in2))}|{Irreps(str(irreps_out))}|{ins}"


def _parse(plan: str):
    s1, s2, so, ins = plan.split("|")
    instructions = []
    for rec in ins.split(";"):
        if rec:
            i1, i2, io, pw = rec.split(",")
            instructions.append((int(i1), int(i2), int(io), "uvu", True, float(pw)))
    return Irreps(s1), Irreps(s2), Irreps(so), _normalize_instructions(instructions)


_DIMS: Dict[str, Tuple[int, int, int, int]] = {}
_KERNELS: Dict[Tuple[str, str], object] = {}


def plan_dims(plan: str) -> Tuple[int, int, int, int]:
    """(dim_in1, dim_in2, dim_out, weight_numel) from the text alone (no native library: usable while tracing)."""
    if plan not in _DIMS:
        i1, i2, io, ins = _parse(plan)
        _DIMS[plan] = (i1.dim, i2.dim, io.dim, sum(i1[i.i_in1].mul for i in ins if i.has_weight))
    return _DIMS[plan]


def _kernels(plan: str, device: torch.device):
    from ._tp_scatter_base import _Kernels

    key = (plan, str(device))
    if key not in _KERNELS:
        i1, i2, io, ins = _parse(plan)
        native = NativePlan(i1, i2, io, ins)
        _KERNELS[key] = _Kernels(native, native.image.to(device))
    return _KERNELS[key]


def _topology(edge_dst, edge_src, num_nodes: int):
    from ._topology import topology_cache

    return topology_cache.get(edge_dst, edge_src, num_nodes)


# ---- device implementations ------------------------------------------------------------------------------------------
def _fwd_cuda(x, edge_attr, edge_weight, edge_dst, edge_src, plan: str):
    k = _kernels(plan, x.device)
    return k.fwd(x.contiguous(), edge_attr.contiguous(), edge_weight.contiguous(), _topology(edge_dst, edge_src, x.size(0)))


def _bwd_cuda(grad_out, x, edge_attr, edge_weight, edge_dst, edge_src, plan: str, need_x: bool, need_y: bool,
              need_w: bool):
    k = _kernels(plan, x.device)
    topo = _topology(edge_dst, edge_src, x.size(0))
    g, x, y, w = grad_out.contiguous(), x.contiguous(), edge_attr.contiguous(), edge_weight.contiguous()
    fused = None
    if need_x and need_y and need_w and k.prefer_fused_bwd and k.fused_rows_ok:
        fused = k.bwd_fused(x, y, w, g, topo)
    if fused is not None:
        gx, gw, gy = fused
    else:
        gx = k.bwd_x(y, w, g, topo) if need_x else None
        gw, gy = k.bwd_edge(x, y, w, g, topo, need_gw=need_w, need_gy=need_y)
    empty = x.new_empty(0)
    return (gx if gx is not None else empty, gy if gy is not None else empty, gw if gw is not None else empty)


_lib_def.impl("tp_scatter_fwd", _fwd_cuda, "CUDA")
_lib_def.impl("tp_scatter_bwd", _bwd_cuda, "CUDA")


# ---- fake (meta) implementations: shapes only --------------------------------------------------------------------------
@torch.library.register_fake(f"{_NS}::tp_scatter_fwd")
def _fwd_fake(x, edge_attr, edge_weight, edge_dst, edge_src, plan: str):
    d1, d2, do, wn = plan_dims(plan)
    torch._check(x.dim() == 2 and x.shape[1] == d1, lambda: f"x must be [N, {d1}]")
    torch._check(edge_attr.dim() == 2 and edge_attr.shape[1] == d2, lambda: f"edge_attr must be [E, {d2}]")
    torch._check(edge_weight.dim() == 2 and edge_weight.shape[1] == wn, lambda: f"edge_weight must be [E, {wn}]")
    return x.new_empty((x.shape[0], do))


@torch.library.register_fake(f"{_NS}::tp_scatter_bwd")
def _bwd_fake(grad_out, x, edge_attr, edge_weight, edge_dst, edge_src, plan: str, need_x: bool, need_y: bool,
              need_w: bool):
    return (torch.empty_like(x) if need_x else x.new_empty(0),
            torch.empty_like(edge_attr) if need_y else x.new_empty(0),
            torch.empty_like(edge_weight) if need_w else x.new_empty(0))


# ---- autograd: fwd -> bwd, bwd -> {fwd, bwd} ---------------------------------------------------------------------------
def _fwd_setup(ctx, inputs, output):
    x, y, w, dst, src, plan = inputs
    ctx.save_for_backward(x, y, w, dst, src)
    ctx.plan = plan
    ctx.set_materialize_grads(False)


def _fwd_backward(ctx, g):
    if g is None:
        return (None,) * 6
    x, y, w, dst, src = ctx.saved_tensors
    nx, ny, nw = ctx.needs_input_grad[:3]
    gx, gy, gw = torch.ops.nequip_amd.tp_scatter_bwd(g, x, y, w, dst, src, ctx.plan, nx, ny, nw)
    return (gx if nx else None, gy if ny else None, gw if nw else None, None, None, None)


torch.library.register_autograd(f"{_NS}::tp_scatter_fwd", _fwd_backward, setup_context=_fwd_setup)


def _bwd_setup(ctx, inputs, output):
    g, x, y, w, dst, src, plan, nx, ny, nw = inputs
    ctx.save_for_backward(g, x, y, w, dst, src)
    ctx.plan, ctx.need = plan, (nx, ny, nw)
    ctx.set_materialize_grads(False)


def _bwd_backward(ctx, c_x, c_y, c_w):
    """Cotangents (c_x, c_y, c_w) of (grad_x, grad_y, grad_w) -> gradients w.r.t. (grad_out, x, y, w).  With
    F = fwd and B(g, X, Y, W) = (Bx(Y, W, g), By(X, W, g), Bw(X, Y, g)):
      d grad_out = F(c_x, y, w) + F(x, c_y, w) + F(x, y, c_w)
      d x = Bx(c_y, w, g) + Bx(y, c_w, g);  d y = By(c_x, w, g) + By(x, c_w, g);  d w = Bw(c_x, y, g) + Bw(x, c_y, g)."""
    g, x, y, w, dst, src = ctx.saved_tensors
    plan = ctx.plan
    nx, ny, nw = ctx.need
    need_g, need_x, need_y, need_w = ctx.needs_input_grad[:4]
    F = torch.ops.nequip_amd.tp_scatter_fwd
    B = torch.ops.nequip_amd.tp_scatter_bwd
    # an output that was not requested has no meaningful cotangent
    c_x = c_x if nx else None
    c_y = c_y if ny else None
    c_w = c_w if nw else None

    def add(a, b):
        return b if a is None else (a if b is None else a + b)

    gg = gxx = gyy = gww = None
    if need_g:
        if c_x is not None:
            gg = add(gg, F(c_x, y, w, dst, src, plan))
        if c_y is not None:
            gg = add(gg, F(x, c_y, w, dst, src, plan))
        if c_w is not None:
            gg = add(gg, F(x, y, c_w, dst, src, plan))
    if c_x is not None and (need_y or need_w):
        _, a_y, a_w = B(g, c_x, y, w, dst, src, plan, False, need_y, need_w)
        gyy, gww = add(gyy, a_y if need_y else None), add(gww, a_w if need_w else None)
    if c_y is not None and (need_x or need_w):
        a_x, _, a_w = B(g, x, c_y, w, dst, src, plan, need_x, False, need_w)
        gxx, gww = add(gxx, a_x if need_x else None), add(gww, a_w if need_w else None)
    if c_w is not None and (need_x or need_y):
        a_x, a_y, _ = B(g, x, y, c_w, dst, src, plan, need_x, need_y, False)
        gxx, gyy = add(gxx, a_x if need_x else None), add(gyy, a_y if need_y else None)
    return gg, gxx, gyy, gww, None, None, None, None, None, None


torch.library.register_autograd(f"{_NS}::tp_scatter_bwd", _bwd_backward, setup_context=_bwd_setup)


def tp_scatter(x, edge_attr, edge_weight, edge_dst, edge_src, plan: str) -> torch.Tensor:
    return torch.ops.nequip_amd.tp_scatter_fwd(x, edge_attr, edge_weight, edge_dst, edge_src, plan)


__all__: List[str] = ["plan_key", "plan_dims", "tp_scatter"]
