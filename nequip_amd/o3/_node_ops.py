"""Dispatcher-op forms of the node kernels (inference grade): ``torch.ops.nequip_amd.node_linear`` and
``gate / gate_bwd / gate_bwd_bwd``.

``nqa_node_linear`` (e3nn ``o3.Linear`` as ``linear_1`` / ``linear_2`` and the type-pre-contracted self-connection,
``nequip/nn/interaction_block.py:82-87,129-146,175-177,201-204``) and ``nqa_gate`` (``nequip/nn/convnetlayer.py:104-112``)
in a form a tracer keeps (``utils/tracing.py``).  The irreps tables travel as a text key (rebuilt on first use, like the
tensor-product plan).  ``node_linear`` is linear in ``x``: its derivative is the same op on the transposed tables
(``transposed`` flag), so the family is closed under differentiation; the packed weights are constants of the op (modules
with differentiable parameters keep the ATen formulation while tracing).  The gate's derivative ops are the kernels of its
autograd Functions (second order included).
"""

from __future__ import annotations

from typing import Dict, Optional

import torch

from .irreps import Irreps

_NS = "nequip_amd"
_lib_def = torch.library.Library(_NS, "FRAGMENT")
_lib_def.define("node_linear(Tensor x, Tensor wp, Tensor? addend, Tensor? types, str key, float scale, bool transposed) "
                "-> Tensor")
_lib_def.define("gate(Tensor x, str key) -> Tensor")
_lib_def.define("gate_bwd(Tensor x, Tensor g, str key) -> Tensor")
_lib_def.define("gate_bwd_bwd(Tensor x, Tensor g, Tensor c, str key, bool need_x, bool need_g) -> (Tensor, Tensor)")
# Gate + linear_1 + typed self-connection of a layer boundary as one launch per direction (`nqa_node_fused`, inference:
# constant weights, first order).  h: the previous layer's PRE-gate rows; x1 = scale * linear_1(Gate(h)); sc = sc(Gate(h), types)
_lib_def.define("node_stage_fwd(Tensor h, Tensor types, Tensor wp1, Tensor wps, str gate_key, str lin_key, str sc_key, "
                "float scale) -> (Tensor, Tensor)")
_lib_def.define("node_stage_bwd(Tensor g_x1, Tensor g_sc, Tensor h, Tensor types, Tensor wp1, Tensor wps, str gate_key, "
                "str lin_key, str sc_key, float scale) -> Tensor")

_LINEAR: Dict[str, object] = {}
_GATE: Dict[str, object] = {}


# ---- keys ---------------------------------------------------------------------------------------------------------------
def linear_key(irreps_in, irreps_out, instructions) -> str:
    return f"{Irreps(str(irreps_in))}|{Irreps(str(irreps_out))}|" + ",".join(f"{i}-{o}" for i, o in instructions)


def _linear_meta(key: str):
    meta = _LINEAR.get(key)
    if meta is None:
        from ._node_kernels import NodeLinearMeta

        s_in, s_out, ins = key.split("|")
        meta = _LINEAR[key] = NodeLinearMeta(Irreps(s_in), Irreps(s_out),
                                             [tuple(int(v) for v in rec.split("-")) for rec in ins.split(",") if rec])
    return meta


def linear_dims(key: str):
    s_in, s_out, _ = key.split("|")
    return Irreps(s_in).dim, Irreps(s_out).dim


def gate_key(irreps_scalars, act_scalars, irreps_gates, act_gates, irreps_gated) -> str:
    acts = lambda a: ",".join(f"{n}:{float(c)!r}" for n, c in a)  # noqa: E731
    return (f"{Irreps(str(irreps_scalars))}|{acts(act_scalars)}|{Irreps(str(irreps_gates))}|{acts(act_gates)}|"
            f"{Irreps(str(irreps_gated))}")


def _parse_acts(text: str):
    return [(rec.split(":")[0], float(rec.split(":")[1])) for rec in text.split(",") if rec]


def _gate_meta(key: str):
    meta = _GATE.get(key)
    if meta is None:
        from ._node_kernels import GateMeta

        s_s, a_s, s_g, a_g, s_d = key.split("|")
        meta = _GATE[key] = GateMeta(Irreps(s_s), _parse_acts(a_s), Irreps(s_g), _parse_acts(a_g), Irreps(s_d))
    return meta


def gate_dims(key: str):
    s_s, _, s_g, _, s_d = key.split("|")
    ns, ng, nd = Irreps(s_s).dim, Irreps(s_g).dim, Irreps(s_d).dim
    return ns + ng + nd, ns + nd


# ---- device implementations ------------------------------------------------------------------------------------------
def _node_linear_cuda(x, wp, addend, types, key, scale, transposed):
    from ._node_kernels import _launch_linear, _transposed, meta_transposed_weights

    meta = _linear_meta(key)
    x, wp = x.contiguous(), wp.contiguous()
    if addend is not None:
        addend = addend.contiguous()
    if transposed:
        return _launch_linear(x, meta_transposed_weights(meta, wp).contiguous(), addend, types, _transposed(meta), "fwd", scale)
    return _launch_linear(x, wp, addend, types, meta, "fwd", scale)


def _gate_cuda(x, key):
    from ._node_kernels import _launch_gate

    return _launch_gate(x.contiguous(), None, _gate_meta(key), 0)


def _gate_bwd_cuda(x, g, key):
    from ._node_kernels import _launch_gate

    return _launch_gate(x.contiguous(), g.contiguous(), _gate_meta(key), 1)


def _gate_bwd_bwd_cuda(x, g, c, key, need_x, need_g):
    from ._node_kernels import _launch_gate

    meta = _gate_meta(key)
    x, g, c = x.contiguous(), g.contiguous(), c.contiguous()
    gx = _launch_gate(x, g, meta, 3, cot=c) if need_x else x.new_empty(0)
    gg = _launch_gate(x, None, meta, 2, cot=c) if need_g else x.new_empty(0)
    return gx, gg


def _stage_order(types, wps):
    import os

    from ._node_kernels import type_order

    typed = wps.shape[0] > 1
    return typed, (type_order(types) if (typed and os.environ.get("NQA_NODE_TYPE_ORDER", "") != "0") else None)


def _node_stage_fwd_cuda(h, types, wp1, wps, gate_key, lin_key, sc_key, scale):
    from ._node_kernels import FusedPart, launch_fused

    gm, m1, ms = _gate_meta(gate_key), _linear_meta(lin_key), _linear_meta(sc_key)
    h, types = h.contiguous(), types.contiguous()
    typed, order = _stage_order(types, wps)
    parts = [FusedPart(h, wp1.contiguous(), m1, scale, in_gate=gm), FusedPart(h, wps.contiguous(), ms, 1.0, in_gate=gm)]
    x1, sc = launch_fused(parts, types if typed else None, order=order)
    return x1, sc


def _node_stage_bwd_cuda(g_x1, g_sc, h, types, wp1, wps, gate_key, lin_key, sc_key, scale):
    from ._node_kernels import FusedPart, _scaled, _transposed, launch_fused, meta_transposed_weights

    gm, m1, ms = _gate_meta(gate_key), _linear_meta(lin_key), _linear_meta(sc_key)
    h, types = h.contiguous(), types.contiguous()
    typed, order = _stage_order(types, wps)
    parts = [FusedPart(g_x1.contiguous(), _scaled(meta_transposed_weights(m1, wp1.contiguous()), scale), _transposed(m1)),
             FusedPart(g_sc.contiguous(), meta_transposed_weights(ms, wps.contiguous()), _transposed(ms), accumulate=True)]
    (gh,) = launch_fused(parts, types if typed else None, out_gate=gm, gate_h=h, order=order)
    return gh


_lib_def.impl("node_stage_fwd", _node_stage_fwd_cuda, "CUDA")
_lib_def.impl("node_stage_bwd", _node_stage_bwd_cuda, "CUDA")
_lib_def.impl("node_linear", _node_linear_cuda, "CUDA")
_lib_def.impl("gate", _gate_cuda, "CUDA")
_lib_def.impl("gate_bwd", _gate_bwd_cuda, "CUDA")
_lib_def.impl("gate_bwd_bwd", _gate_bwd_bwd_cuda, "CUDA")


# ---- fake kernels ---------------------------------------------------------------------------------------------------
@torch.library.register_fake(f"{_NS}::node_linear")
def _node_linear_fake(x, wp, addend, types, key, scale, transposed):
    din, dout = linear_dims(key)
    if transposed:
        din, dout = dout, din
    torch._check(x.dim() == 2 and x.shape[1] == din, lambda: f"x must be [N, {din}]")
    return x.new_empty((x.shape[0], dout))


@torch.library.register_fake(f"{_NS}::gate")
def _gate_fake(x, key):
    din, dout = gate_dims(key)
    torch._check(x.dim() == 2 and x.shape[1] == din, lambda: f"x must be [N, {din}]")
    return x.new_empty((x.shape[0], dout))


@torch.library.register_fake(f"{_NS}::gate_bwd")
def _gate_bwd_fake(x, g, key):
    return torch.empty_like(x)


@torch.library.register_fake(f"{_NS}::gate_bwd_bwd")
def _gate_bwd_bwd_fake(x, g, c, key, need_x, need_g):
    return (torch.empty_like(x) if need_x else x.new_empty(0), torch.empty_like(g) if need_g else x.new_empty(0))


@torch.library.register_fake(f"{_NS}::node_stage_fwd")
def _node_stage_fwd_fake(h, types, wp1, wps, gate_key, lin_key, sc_key, scale):
    din, dgated = gate_dims(gate_key)
    l_in, l_out = linear_dims(lin_key)
    s_in, s_out = linear_dims(sc_key)
    torch._check(h.dim() == 2 and h.shape[1] == din, lambda: f"h must be [N, {din}] (pre-gate rows)")
    torch._check(l_in == dgated and s_in == dgated, lambda: "linear_1 / self-connection do not take the gate's output")
    return h.new_empty((h.shape[0], l_out)), h.new_empty((h.shape[0], s_out))


@torch.library.register_fake(f"{_NS}::node_stage_bwd")
def _node_stage_bwd_fake(g_x1, g_sc, h, types, wp1, wps, gate_key, lin_key, sc_key, scale):
    return torch.empty_like(h)


# ---- autograd --------------------------------------------------------------------------------------------------------
def _ns_setup(ctx, inputs, output):
    h, types, wp1, wps, gate_key, lin_key, sc_key, scale = inputs
    ctx.save_for_backward(h, types, wp1, wps)
    ctx.keys, ctx.scale = (gate_key, lin_key, sc_key), scale
    ctx.dims = (output[0].shape[1], output[1].shape[1])


def _ns_backward(ctx, g1, gs):
    h, types, wp1, wps = ctx.saved_tensors
    if not ctx.needs_input_grad[0]:
        return (None,) * 8
    if g1 is None:
        g1 = h.new_zeros((h.shape[0], ctx.dims[0]))
    if gs is None:
        gs = h.new_zeros((h.shape[0], ctx.dims[1]))
    gh = torch.ops.nequip_amd.node_stage_bwd(g1, gs, h, types, wp1, wps, *ctx.keys, ctx.scale)
    return (gh,) + (None,) * 7


torch.library.register_autograd(f"{_NS}::node_stage_fwd", _ns_backward, setup_context=_ns_setup)


def _nl_setup(ctx, inputs, output):
    x, wp, addend, types, key, scale, transposed = inputs
    ctx.save_for_backward(wp, types)
    ctx.key, ctx.scale, ctx.transposed, ctx.has_addend = key, scale, transposed, addend is not None


def _nl_backward(ctx, g):
    wp, types = ctx.saved_tensors
    gx = None
    if ctx.needs_input_grad[0]:
        gx = torch.ops.nequip_amd.node_linear(g, wp, None, types, ctx.key, ctx.scale, not ctx.transposed)
    gadd = g if (ctx.has_addend and ctx.needs_input_grad[2]) else None
    return gx, None, gadd, None, None, None, None


torch.library.register_autograd(f"{_NS}::node_linear", _nl_backward, setup_context=_nl_setup)


def _gate_setup(ctx, inputs, output):
    x, key = inputs
    ctx.save_for_backward(x)
    ctx.key = key


def _gate_backward(ctx, g):
    (x,) = ctx.saved_tensors
    return torch.ops.nequip_amd.gate_bwd(x, g, ctx.key), None


torch.library.register_autograd(f"{_NS}::gate", _gate_backward, setup_context=_gate_setup)


def _gate_bwd_setup(ctx, inputs, output):
    x, g, key = inputs
    ctx.save_for_backward(x, g)
    ctx.key = key


def _gate_bwd_backward(ctx, c):
    x, g = ctx.saved_tensors
    need_x, need_g = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    gx, gg = torch.ops.nequip_amd.gate_bwd_bwd(x, g, c, ctx.key, need_x, need_g)
    return (gx if need_x else None, gg if need_g else None, None)


torch.library.register_autograd(f"{_NS}::gate_bwd", _gate_bwd_backward, setup_context=_gate_bwd_setup)


def node_linear_op(x, wp, key: str, addend: Optional[torch.Tensor] = None, types: Optional[torch.Tensor] = None,
                   scale: float = 1.0) -> torch.Tensor:
    return torch.ops.nequip_amd.node_linear(x, wp, addend, types, key, float(scale), False)


def gate_op(x, key: str) -> torch.Tensor:
    return torch.ops.nequip_amd.gate(x, key)


def node_stage(h, types, wp1, wps, gate_key: str, lin_key: str, sc_key: str, scale: float):
    return torch.ops.nequip_amd.node_stage_fwd(h, types, wp1, wps, gate_key, lin_key, sc_key, float(scale))
