"""Traceable forms of the modules.

The reference compiles a model by symbolic tracing (``make_fx`` over ``torch.compile(dynamic=True)`` graphs,
``nequip/nn/compile.py:176-191``; ``nequip/scripts/compile.py:248-344`` for AOTInductor), which cannot look inside a Python
``autograd.Function`` that calls a C library through raw pointers.  While :func:`traceable` is true every module of this
package chooses a form a tracer can follow:

* the tensor-product scatter and the edge embedding go through dispatcher ops (``torch.ops.nequip_amd.*``, with fake
  kernels for shape propagation and autograd formulas that stay inside the op family),
* the radial MLP, ``o3.Linear``, the self-connection, ``Gate``, the edge vectors and the force / virial tail use their ATen
  formulations (the ones they keep for shapes outside the fused kernels),
* nothing data dependent is decided on the host (no reverse-edge pairing, no side streams).

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


def trace_model(model: torch.nn.Module, example_inputs, tracing_mode: str = "symbolic"):
    """``make_fx`` graph of ``model(dict(inputs))`` with the weights as graph inputs, the way the reference's compile path
    does it (``nequip/nn/compile.py:150-191``).  Returns ``(graph_module, params, buffers)``; call the graph as
    ``graph_module(params, buffers, inputs)`` -- it returns the model's output dict."""
    from torch.fx.experimental.proxy_tensor import make_fx

    params, buffers = dict(model.named_parameters()), dict(model.named_buffers())

    def f(params, buffers, inputs):
        return torch.func.functional_call(model, (params, buffers), (dict(inputs),))

    with traceable_forms():
        gm = make_fx(f, tracing_mode=tracing_mode)(params, buffers, example_inputs)
    return gm, params, buffers
