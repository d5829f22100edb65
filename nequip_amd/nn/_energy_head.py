"""Per-atom energy head in one launch per direction (``nqa_energy_head``, csrc/energy_head.hip).

In eval mode on the GPU the tail of the energy model -- the last layer's Gate (scalars only), the depth-0 ``ScalarMLP``
readout and ``PerTypeScaleShift`` (``nequip/model/nequip_models.py:371-399``, ``nequip/nn/atomwise.py:116-284``) -- and its
backward are a dozen launches on ``[N, 64]`` / ``[N, 1]`` tensors; here they are two.  The modules, their parameters and
their state-dict keys stay where the reference has them: the last ``ConvNetLayer`` defers its gate (``_nqa_pregate``), the
readout module consumes it and marks the per-atom energies as already scaled, ``PerTypeScaleShift`` then passes them on.
"""

from __future__ import annotations

import ctypes
from typing import Optional

import torch

from .. import _lib
from ..utils import ktimer


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()


def _launch(backward: int, h, w, scales, shifts, types, g_e, out, act: int, cst: float):
    lib = _lib.load()
    N, D = h.shape
    ns = 0 if scales is None else scales.numel()
    nh = 0 if shifts is None else shifts.numel()
    with torch.cuda.device(h.device), ktimer.region("energy_head", 4.0 * N * D * (2 if backward else 1) + 8.0 * N):
        rc = lib.nqa_energy_head(backward, _ptr(h), _ptr(w), _ptr(scales), ns, _ptr(shifts), nh, _ptr(types), _ptr(g_e),
                                 _ptr(out), D, act, float(cst), N,
                                 ctypes.c_void_p(torch.cuda.current_stream(h.device).cuda_stream))
    _lib.check(rc, "nqa_energy_head")


class _EnergyHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, w, scales, shifts, types, act: int, cst: float):
        h = h.contiguous()
        e = torch.empty((h.shape[0], 1), dtype=torch.float64, device=h.device)
        _launch(0, h, w, scales, shifts, types, None, e, act, cst)
        ctx.save_for_backward(h, w, scales, types)
        ctx.act, ctx.cst = act, cst
        return e

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        h, w, scales, types = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return (None,) * 7
        g = g.to(torch.float64)
        if not g.is_contiguous() or g.stride(0) != 1:  # (the expanded gradient of a sum: stride 0)
            g = g.contiguous()
        gh = torch.empty_like(h)
        _launch(1, h, w, scales, None, types, g.view(-1), gh, ctx.act, ctx.cst)
        return gh, None, None, None, None, None, None


def energy_head(h, w, scales, shifts, types, act: int, cst: float) -> torch.Tensor:
    """``[N, 1]`` float64 per-atom energies ``shift[t] + scale[t] * double(sum_c w[c] cst act(h[:, c]))``; the gradient
    w.r.t. ``h`` is one launch (constant weights: eval mode)."""
    return _EnergyHeadFn.apply(h, w, scales, shifts, types, act, cst)


# ---- the same two launches as dispatcher ops (a tracer keeps them: utils/tracing.py) ----------------------------------------
_NS = "nequip_amd"
_lib_def = torch.library.Library(_NS, "FRAGMENT")
_lib_def.define("energy_head_fwd(Tensor h, Tensor w, Tensor? scales, Tensor? shifts, Tensor types, int act, float cst) "
                "-> Tensor")
_lib_def.define("energy_head_bwd(Tensor g_e, Tensor h, Tensor w, Tensor? scales, Tensor types, int act, float cst) -> Tensor")


def _fwd_cuda(h, w, scales, shifts, types, act, cst):
    h = h.contiguous()
    e = torch.empty((h.shape[0], 1), dtype=torch.float64, device=h.device)
    _launch(0, h, w.contiguous(), scales, shifts, types.contiguous(), None, e, int(act), float(cst))
    return e


def _bwd_cuda(g_e, h, w, scales, types, act, cst):
    h = h.contiguous()
    g = g_e.to(torch.float64).contiguous().view(-1)
    gh = torch.empty_like(h)
    _launch(1, h, w.contiguous(), scales, None, types.contiguous(), g, gh, int(act), float(cst))
    return gh


_lib_def.impl("energy_head_fwd", _fwd_cuda, "CUDA")
_lib_def.impl("energy_head_bwd", _bwd_cuda, "CUDA")


@torch.library.register_fake(f"{_NS}::energy_head_fwd")
def _fwd_fake(h, w, scales, shifts, types, act, cst):
    torch._check(h.dim() == 2 and w.dim() == 1 and w.shape[0] == h.shape[1], lambda: "h [N, D], w [D]")
    return h.new_empty((h.shape[0], 1), dtype=torch.float64)


@torch.library.register_fake(f"{_NS}::energy_head_bwd")
def _bwd_fake(g_e, h, w, scales, types, act, cst):
    return torch.empty_like(h)


def _op_setup(ctx, inputs, output):
    h, w, scales, shifts, types, act, cst = inputs
    ctx.save_for_backward(h, w, scales, types)
    ctx.act, ctx.cst = act, cst


def _op_backward(ctx, g):
    h, w, scales, types = ctx.saved_tensors
    gh = None
    if ctx.needs_input_grad[0]:
        gh = torch.ops.nequip_amd.energy_head_bwd(g, h, w, scales, types, ctx.act, ctx.cst)
    return gh, None, None, None, None, None, None


torch.library.register_autograd(f"{_NS}::energy_head_fwd", _op_backward, setup_context=_op_setup)


def energy_head_op(h, w, scales, shifts, types, act: int, cst: float) -> torch.Tensor:
    return torch.ops.nequip_amd.energy_head_fwd(h, w, scales, shifts, types, int(act), float(cst))
