"""Dispatcher-op form of the fused radial MLP (inference grade): ``torch.ops.nequip_amd.radial_mlp_fwd / _bwd / _bwd_bwd``.

``ScalarMLPFunction`` as ``InteractionBlock.edge_mlp`` (``nequip/nn/mlp.py:194-196,262-268``) on the fused MFMA kernels,
in a form a tracer can keep (``utils/tracing.py``): ``radial_mlp_fwd(emb, w0, w1, alpha0, alpha1)`` and the gradient
w.r.t. the embedding ``radial_mlp_bwd(emb, w0, w1, g, ...)``, whose own derivatives (w.r.t. ``emb`` and ``g``) are
``radial_mlp_bwd_bwd`` -- the second-order kernels of the training path.  The weights are constants of these ops (no
gradient flows to ``w0`` / ``w1``): modules with differentiable parameters keep the ATen formulation while tracing.
"""

from __future__ import annotations

import torch

_NS = "nequip_amd"
_lib_def = torch.library.Library(_NS, "FRAGMENT")
_lib_def.define("radial_mlp_fwd(Tensor emb, Tensor w0, Tensor w1, float alpha0, float alpha1) -> Tensor")
_lib_def.define("radial_mlp_bwd(Tensor emb, Tensor w0, Tensor w1, Tensor g, float alpha0, float alpha1) -> Tensor")
_lib_def.define("radial_mlp_bwd_bwd(Tensor emb, Tensor w0, Tensor w1, Tensor g, Tensor c, float alpha0, float alpha1, "
                "bool need_emb, bool need_g) -> (Tensor, Tensor)")

def _cache_for(w1: torch.Tensor, alpha1: float = 1.0):
    """The split / re-laid-out images of the second-layer weights, once per constant: the op sees the weights as plain
    tensors, so the cache is keyed on the identity of their storage (``utils/constcache.py``) -- a compiled graph's constant
    buffer hits at every call, a weight tensor that is recomputed per call runs the ~7 us prepass kernel each time."""
    from ..utils import constcache
    from .mlp import _WeightImages

    return constcache.get(w1, ("radial_mlp_images", float(alpha1)), _WeightImages)  # (the prepass folds alpha1 in)


def _fwd_cuda(emb, w0, w1, alpha0, alpha1):
    from . import mlp as m

    return m._launch_fwd(emb.contiguous(), w0, w1, alpha0, alpha1, m.radial_mlp_mode(), _cache_for(w1, alpha1))


def _bwd_cuda(emb, w0, w1, g, alpha0, alpha1):
    from . import mlp as m

    return m._launch_bwd(emb.contiguous(), w0, w1, alpha0, alpha1, g.contiguous(), m.radial_mlp_mode(), _cache_for(w1, alpha1))


def _bwd_bwd_cuda(emb, w0, w1, g, c, alpha0, alpha1, need_emb, need_g):
    from . import mlp as m

    mode, cache = m.radial_mlp_mode(), _cache_for(w1, alpha1)
    emb, g, c = emb.contiguous(), g.contiguous(), c.contiguous()
    g_emb2 = emb.new_empty(0)
    gg = emb.new_empty(0)
    if need_emb:
        g_emb2 = m._launch_bwd_train(emb, w0, w1, alpha0, alpha1, g, c, mode, cache)[0]
    if need_g:
        gg = m._launch_fwd_tangent(emb, c, w0, w1, alpha0, alpha1, mode, cache)
    return g_emb2, gg


_lib_def.impl("radial_mlp_fwd", _fwd_cuda, "CUDA")
_lib_def.impl("radial_mlp_bwd", _bwd_cuda, "CUDA")
_lib_def.impl("radial_mlp_bwd_bwd", _bwd_bwd_cuda, "CUDA")


@torch.library.register_fake(f"{_NS}::radial_mlp_fwd")
def _fwd_fake(emb, w0, w1, alpha0, alpha1):
    torch._check(emb.dim() == 2 and w0.dim() == 2 and w1.dim() == 2, lambda: "emb [E, nb], w0 [nb, H], w1 [H, W]")
    return emb.new_empty((emb.shape[0], w1.shape[1]))


@torch.library.register_fake(f"{_NS}::radial_mlp_bwd")
def _bwd_fake(emb, w0, w1, g, alpha0, alpha1):
    return torch.empty_like(emb)


@torch.library.register_fake(f"{_NS}::radial_mlp_bwd_bwd")
def _bwd_bwd_fake(emb, w0, w1, g, c, alpha0, alpha1, need_emb, need_g):
    return (torch.empty_like(emb) if need_emb else emb.new_empty(0), torch.empty_like(g) if need_g else emb.new_empty(0))


def _fwd_setup(ctx, inputs, output):
    emb, w0, w1, alpha0, alpha1 = inputs
    ctx.save_for_backward(emb, w0, w1)
    ctx.alphas = (alpha0, alpha1)


def _fwd_backward(ctx, g):
    emb, w0, w1 = ctx.saved_tensors
    g_emb = torch.ops.nequip_amd.radial_mlp_bwd(emb, w0, w1, g, *ctx.alphas) if ctx.needs_input_grad[0] else None
    return g_emb, None, None, None, None


torch.library.register_autograd(f"{_NS}::radial_mlp_fwd", _fwd_backward, setup_context=_fwd_setup)


def _bwd_setup(ctx, inputs, output):
    emb, w0, w1, g, alpha0, alpha1 = inputs
    ctx.save_for_backward(emb, w0, w1, g)
    ctx.alphas = (alpha0, alpha1)


def _bwd_backward(ctx, c):
    emb, w0, w1, g = ctx.saved_tensors
    need_emb, need_g = ctx.needs_input_grad[0], ctx.needs_input_grad[3]
    g_emb2, gg = torch.ops.nequip_amd.radial_mlp_bwd_bwd(emb, w0, w1, g, c, *ctx.alphas, need_emb, need_g)
    return (g_emb2 if need_emb else None, None, None, gg if need_g else None, None, None)


torch.library.register_autograd(f"{_NS}::radial_mlp_bwd", _bwd_backward, setup_context=_bwd_setup)


def radial_mlp(emb, w0, w1, alpha0: float, alpha1: float) -> torch.Tensor:
    return torch.ops.nequip_amd.radial_mlp_fwd(emb, w0, w1, float(alpha0), float(alpha1))
