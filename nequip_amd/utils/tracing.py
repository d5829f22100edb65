"""Traceable forms of the modules.

The reference compiles a model by symbolic tracing (``make_fx`` over ``torch.compile(dynamic=True)`` graphs,
``nequip/nn/compile.py:176-191``; ``nequip/scripts/compile.py:248-344`` for AOTInductor), which cannot look inside a Python
``autograd.Function`` that calls a C library through raw pointers.  While :func:`traceable` is true every module of this
package chooses a form a tracer can follow:

* every fused kernel family goes through dispatcher ops (``torch.ops.nequip_amd.*``, with fake kernels for shape
  propagation and autograd formulas that stay inside the op family).  An inference graph (eval mode, constant weights) is
  made of the same launches as the eager evaluation: ``radial_tp_fwd / _bwd`` (radial MLP + tensor-product scatter of one
  convolution; the reverse-edge pairing decision is taken INSIDE the op, at run time, from the topology cache),
  ``node_stage_fwd / _bwd`` (Gate + linear_1 + self-connection of a layer boundary), ``node_linear``, ``energy_head_fwd /
  _bwd``, ``edge_embed_*``, ``edge_vectors`` and ``force_virial`` (the eval-mode force / virial tail);
* modules with differentiable parameters, graphs that differentiate the forces again, deep radial MLPs and irreps outside the
  fused kernels take the separate, twice-differentiable ops (``radial_mlp_*``, ``tp_scatter_*``, ``gate*``,
  ``edge_vectors_adj``) or their ATen formulations; ``NQA_NO_RADIAL_TP_OP=1``, ``NQA_TRACE_NO_NODE_FUSION=1`` and
  ``NQA_TRACE_REFERENCE_TAIL=1`` select those forms for an inference graph too;
* nothing data dependent is decided on the host OF THE GRAPH (no side streams; what depends on the edge list lives behind
  the dispatcher).

It is true while ``torch.compile`` is tracing and inside :func:`traceable_forms`.  Eager evaluation keeps the fused kernels.
"""

from __future__ import annotations

import contextlib

import torch

_forced = 0


def traceable() -> bool:
    return _forced > 0 or torch.compiler.is_compiling()


@contextlib.contextmanager
def traceable_forms(enabled: bool = True):
    """``with traceable_forms(): gm = make_fx(f, tracing_mode="fake")(...)`` -- also usable around plain eager calls to
    evaluate the traceable forms themselves (what the parity test of the traced graph compares against)."""
    global _forced
    if enabled:
        _forced += 1
    try:
        yield
    finally:
        if enabled:
            _forced -= 1


def trace_model(model: torch.nn.Module, example_inputs, tracing_mode: str = "symbolic", fold: bool = False):
    """``make_fx`` graph of ``model(dict(inputs))`` with the weights as graph inputs, the way the reference's compile path
    does it (``nequip/nn/compile.py:150-191``).  Returns ``(graph_module, params, buffers)``; call the graph as
    ``graph_module(params, buffers, inputs)`` -- it returns the model's output dict.  ``fold``: see ``fold_constants``."""
    from torch.fx.experimental.proxy_tensor import make_fx

    params, buffers = dict(model.named_parameters()), dict(model.named_buffers())

    def f(params, buffers, inputs):
        return torch.func.functional_call(model, (params, buffers), (dict(inputs),))

    with traceable_forms():
        gm = make_fx(f, tracing_mode=tracing_mode)(params, buffers, example_inputs)
    if fold:
        fold_constants(gm, params, buffers)
    return gm, params, buffers


def fold_constants(gm: torch.fx.GraphModule, params, buffers) -> int:
    """Deployment-time freezing of a ``trace_model`` graph: every node that depends on the weights alone -- the path
    normalisation folded into ``o3.Linear`` weights, the per-type contraction of the self-connection weights, the scaled
    readout weight, dtype casts of the scale / shift tables: two dozen small kernels per evaluation otherwise -- is evaluated
    ONCE, here, on the current parameter values, and enters the graph as a constant buffer (``_folded_<k>``).  The graph keeps
    its signature (the weight inputs stay, mostly unused); it is then a snapshot of the weights it was folded with, which is
    what an exported package is anyway.  Constant buffers are also what lets the dispatcher ops keep their packed /
    transposed / split weight images across calls (``utils/constcache.py``).  Returns the number of buffers created."""
    import torch.utils._pytree as pytree

    leaves = pytree.tree_leaves((params, buffers))
    placeholders = [n for n in gm.graph.nodes if n.op == "placeholder"]
    assert len(placeholders) >= len(leaves), "not a trace_model graph (weights first)"
    env = {n: v.detach() for n, v in zip(placeholders, leaves)}
    const_inputs = set(env)

    def known(a) -> bool:
        ok = True

        def visit(x):
            nonlocal ok
            if isinstance(x, torch.fx.Node) and x not in env:
                ok = False
            return x

        torch.fx.node.map_arg(a, visit)
        return ok

    with torch.no_grad():
        for n in gm.graph.nodes:
            if n.op != "call_function" or not known(n.args) or not known(n.kwargs):
                continue
            if not any(isinstance(a, torch.fx.Node) for a in pytree.tree_leaves((n.args, n.kwargs))):
                continue  # (factory calls: nothing to gain, and their sizes may be symbolic)
            args = torch.fx.node.map_arg(n.args, lambda x: env[x])
            kwargs = torch.fx.node.map_arg(n.kwargs, lambda x: env[x])
            env[n] = n.target(*args, **kwargs)
    first = next(n for n in gm.graph.nodes if n.op != "placeholder")
    made = 0
    for n, val in list(env.items()):
        if n in const_inputs or not isinstance(val, torch.Tensor):
            continue
        outside = [u for u in n.users if u not in env]
        if not outside:
            continue
        name = f"_folded_{made}"
        made += 1
        gm.register_buffer(name, val.contiguous().clone())
        with gm.graph.inserting_before(first):
            const = gm.graph.get_attr(name)
        const.meta = dict(n.meta)
        for u in outside:
            u.replace_input_with(n, const)
    gm.graph.eliminate_dead_code()
    gm.recompile()
    return made
