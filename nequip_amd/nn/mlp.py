"""Scalar MLPs (mirror of ``nequip/nn/mlp.py:34-271``): bias-free ``x @ (W * alpha)`` layers with
``alpha = gain / sqrt(fan_in)`` and SiLU in between.  ``ScalarMLPFunction`` as the per-edge radial network
(``InteractionBlock.edge_mlp``, ``nequip/nn/interaction_block.py:119-127``) is the dense-GEMM part of the hot
path: on the GPU it runs through the fused MFMA kernel of ``nequip_amd/csrc/radial_mlp.hip`` when available
for its shape, otherwise through hipBLASLt ``mm``."""

import contextlib
import os
from math import sqrt
from typing import Optional

import torch

from .. import _lib
from ..data import AtomicDataDict
from ..o3.irreps import Irreps
from ..utils import ktimer
from ..utils.wgrad import _WeightCacheMixin, differentiable_parameters
from ._graph_mixin import GraphModuleMixin


def radial_mlp_mode() -> int:
    """GEMM mode of the fused radial MLP: split-bf16 (fp32-accurate, default) or exact fp32 MFMA
    (``NQA_MLP_EXACT_FP32=1``)."""
    return _lib.NQA_MLP_FP32 if os.environ.get("NQA_MLP_EXACT_FP32", "") not in ("", "0") else _lib.NQA_MLP_BF16X6


def forward_mode(mode: int) -> int:
    """Mode of the plain forward launch: the split-bf16 default runs its forward GEMM on the two-plane fp16 split
    (``NQA_MLP_F16X3``: three matrix instructions per k-step instead of six, operands scaled by powers of two;
    ``NQA_MLP_FWD_F16=0`` keeps the bf16 split)."""
    if mode == _lib.NQA_MLP_BF16X6 and os.environ.get("NQA_MLP_FWD_F16", "") not in ("0",):
        return _lib.NQA_MLP_F16X3
    return mode


def backward_mode(mode: int) -> int:
    """Mode of the backward launches (``nqa_radial_mlp_bwd`` / ``_bwd_paired`` / ``_bwd_train``): the split-bf16 default
    runs them on the two-plane fp16 split with a running per-row scale (``NQA_MLP_BWD_F16=0`` keeps the bf16 split).
    ``_fwd_tangent`` stays on the bf16 split (its kernel keeps the three-plane weight image)."""
    if mode == _lib.NQA_MLP_BF16X6 and os.environ.get("NQA_MLP_BWD_F16", "") not in ("0",):
        return _lib.NQA_MLP_F16X3
    return mode


from ..utils.tracing import traceable

class _WeightImages:
    """Per-module cache of the split / re-laid-out second-layer weights (``workspace`` of ``nqa_radial_mlp_fwd/bwd``):
    in eval mode the weights are constants, so the prepass kernel runs once per parameter version, not per call."""

    def __init__(self):
        self.key = None
        self.images = {}

    def validate(self, param: torch.Tensor) -> None:
        """Drop the images if the parameter they were made from changed (new object, storage, version or device)."""
        key = (id(param), param.data_ptr(), param._version, param.device)
        if key != self.key:
            self.key, self.images = key, {}

    def get(self, w1: torch.Tensor, mode: int, backward: int, nbytes: int, rows: int = 1):
        """(workspace, filled?).  ``rows``: the rows of the launch -- one over zero rows returns before the prepass, so it
        gets a scratch buffer and leaves no image behind that a later call would take for a filled one."""
        if rows == 0:
            return torch.empty(max(nbytes, 1), dtype=torch.uint8, device=w1.device), False
        hit = (mode, backward) in self.images
        if not hit:
            self.images[(mode, backward)] = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=w1.device)
        return self.images[(mode, backward)], hit


def _launch_fwd(emb, w0, w1, alpha0: float, alpha1: float, mode: int, cache: _WeightImages):
    """``silu(emb @ (w0 alpha0)) @ (w1 alpha1)`` in one fused MFMA launch (``nqa_radial_mlp_fwd``)."""
    from ._topology import _ptr, current_stream_ptr

    lib = _lib.load()
    E, nb = emb.shape
    H, W = w1.shape
    out = torch.empty((E, W), dtype=emb.dtype, device=emb.device)
    flops = 2.0 * E * (nb * H + H * W)
    mode = forward_mode(mode)
    ws_bytes = lib.nqa_radial_mlp_workspace_bytes(mode, 0, H, W)
    ws, ready = cache.get(w1, mode, 0, ws_bytes, E)
    with torch.cuda.device(emb.device), ktimer.region("radial_mlp_fwd", 4.0 * E * (nb + W), flops):
        rc = lib.nqa_radial_mlp_fwd(_lib.NQA_F32, mode, _ptr(emb), _ptr(w0), alpha0, _ptr(w1), alpha1, nb, H, W, E,
                                    _ptr(out), _ptr(ws), ws_bytes, int(ready), current_stream_ptr(emb.device))
    _lib.check(rc, "nqa_radial_mlp_fwd")
    return out


def _launch_bwd(emb, w0, w1, alpha0: float, alpha1: float, g_w, mode: int, cache: _WeightImages, device_idle: bool = False):
    """Gradient of the fused MLP w.r.t. the edge embedding (``nqa_radial_mlp_bwd``; hidden layer recomputed on chip).
    ``device_idle``: the launch has the device to itself (``NQA_MLP_HINT_DEVICE_IS_IDLE``, include/nequip_amd.h)."""
    from ._topology import _ptr, current_stream_ptr

    lib = _lib.load()
    E, nb = emb.shape
    H, W = w1.shape
    g_emb = torch.empty_like(emb)
    flops = 2.0 * E * (nb * H * 2 + H * W)
    mode = backward_mode(mode)
    ws_bytes = lib.nqa_radial_mlp_workspace_bytes(mode, 1, H, W)
    ws, ready = cache.get(w1, mode, 1, ws_bytes, E)
    with torch.cuda.device(emb.device), ktimer.region("radial_mlp_bwd", 4.0 * E * (2 * nb + W), flops):
        rc = lib.nqa_radial_mlp_bwd(_lib.NQA_F32, mode | (_lib.NQA_MLP_HINT_DEVICE_IS_IDLE if device_idle else 0), _ptr(emb),
                                    _ptr(w0), alpha0, _ptr(w1), alpha1, _ptr(g_w), nb, H, W, E, _ptr(g_emb), _ptr(ws), ws_bytes,
                                    int(ready), current_stream_ptr(emb.device))
    _lib.check(rc, "nqa_radial_mlp_bwd")
    return g_emb


def _launch_bwd_train(emb, w0, w1, alpha0: float, alpha1: float, g_w, cot, mode: int, cache: _WeightImages):
    """``nqa_radial_mlp_bwd_train``: the embedding-side gradient plus the on-chip pieces of the parameter gradients.
    ``cot is None``: (g_emb, silu(P), dW0 / alpha0); second order (``cot`` = cotangent of g_emb): (d<cot,g_emb>/d emb,
    Q silu'(P), d<cot,g_emb>/dW0 / alpha0) -- see include/nequip_amd.h."""
    from ._topology import _ptr, current_stream_ptr

    lib = _lib.load()
    E, nb = emb.shape
    H, W = w1.shape
    g_emb = torch.empty_like(emb)
    hid = torch.empty((E, H), dtype=emb.dtype, device=emb.device)
    tiles = lib.nqa_radial_mlp_train_tiles(E)
    parts = torch.empty((tiles, nb, H), dtype=emb.dtype, device=emb.device)
    flops = 2.0 * E * (nb * H * (3 if cot is None else 5) + H * W)
    mode = backward_mode(mode)
    ws_bytes = lib.nqa_radial_mlp_workspace_bytes(mode, 1, H, W)
    ws, ready = cache.get(w1, mode, 1, ws_bytes, E)
    with torch.cuda.device(emb.device), ktimer.region("radial_mlp_bwd_train", 4.0 * E * (3 * nb + W + H), flops):
        rc = lib.nqa_radial_mlp_bwd_train(_lib.NQA_F32, mode, _ptr(emb), _ptr(cot), _ptr(w0), alpha0, _ptr(w1), alpha1,
                                          _ptr(g_w), nb, H, W, E, _ptr(g_emb), _ptr(hid), _ptr(parts), _ptr(ws),
                                          ws_bytes, int(ready), current_stream_ptr(emb.device))
    _lib.check(rc, "nqa_radial_mlp_bwd_train")
    return g_emb, hid, parts.sum(0)


def _launch_fwd_tangent(emb, cot, w0, w1, alpha0: float, alpha1: float, mode: int, cache: _WeightImages):
    """``((cot W0) silu'(emb W0)) W1`` (``nqa_radial_mlp_fwd_tangent``): gradient of <cot, g_emb> w.r.t. grad_out."""
    from ._topology import _ptr, current_stream_ptr

    lib = _lib.load()
    E, nb = emb.shape
    H, W = w1.shape
    out = torch.empty((E, W), dtype=emb.dtype, device=emb.device)
    ws_bytes = lib.nqa_radial_mlp_workspace_bytes(mode, 0, H, W)
    ws, ready = cache.get(w1, mode, 0, ws_bytes, E)
    with torch.cuda.device(emb.device), ktimer.region("radial_mlp_fwd", 4.0 * E * (2 * nb + W), 2.0 * E * (2 * nb * H + H * W)):
        rc = lib.nqa_radial_mlp_fwd_tangent(_lib.NQA_F32, mode, _ptr(emb), _ptr(cot), _ptr(w0), alpha0, _ptr(w1),
                                            alpha1, nb, H, W, E, _ptr(out), _ptr(ws), ws_bytes, int(ready),
                                            current_stream_ptr(emb.device))
    _lib.check(rc, "nqa_radial_mlp_fwd_tangent")
    return out


def _launch_bwd_paired(emb, w0, w1, alpha0: float, alpha1: float, g_a, g_b, mode: int, cache: _WeightImages):
    """``_launch_bwd`` for a gradient given as two row streams ``g_a + g_b`` (``nqa_radial_mlp_bwd_paired``)."""
    from ._topology import _ptr, current_stream_ptr

    lib = _lib.load()
    E, nb = emb.shape
    H, W = w1.shape
    assert g_a.shape == (E, W) and g_b.shape == (E, W) and g_a.is_contiguous() and g_b.is_contiguous()
    g_emb = torch.empty_like(emb)
    flops = 2.0 * E * (nb * H * 2 + H * W)
    mode = backward_mode(mode)
    ws_bytes = lib.nqa_radial_mlp_workspace_bytes(mode, 1, H, W)
    ws, ready = cache.get(w1, mode, 1, ws_bytes, E)
    with torch.cuda.device(emb.device), ktimer.region("radial_mlp_bwd", 4.0 * E * (2 * nb + 2 * W), flops):
        rc = lib.nqa_radial_mlp_bwd_paired(_lib.NQA_F32, mode, _ptr(emb), _ptr(w0), alpha0, _ptr(w1), alpha1, _ptr(g_a),
                                           _ptr(g_b), nb, H, W, E, _ptr(g_emb), _ptr(ws), ws_bytes, int(ready),
                                           current_stream_ptr(emb.device))
    _lib.check(rc, "nqa_radial_mlp_bwd_paired")
    return g_emb


def _launch_wgrad(a, b):
    """``a^T b`` reduced over the edge rows (``nqa_wgrad``): [E, M], [E, N] -> [M, N]."""
    from ..utils import wgrad as _wg

    M, N = a.shape[1], b.shape[1]
    return _wg.wgrad(a, b, _wg.WgradTable([(0, 0, M, N, 1, 0)], M * N)).view(M, N)


class _RadialMLPFn(torch.autograd.Function):
    """Fused two-layer radial MLP on the matrix cores (``nqa_radial_mlp_fwd/bwd``); inference path: differentiable
    w.r.t. the edge embedding only (that is what the force backward needs)."""

    @staticmethod
    def forward(ctx, emb, w0, w1, alpha0: float, alpha1: float, mode: int, cache: _WeightImages):
        emb = emb.contiguous()
        out = _launch_fwd(emb, w0, w1, alpha0, alpha1, mode, cache)
        ctx.save_for_backward(emb, w0, w1)
        ctx.alphas = (alpha0, alpha1)
        ctx.mode, ctx.cache = mode, cache
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_w):
        emb, w0, w1 = ctx.saved_tensors
        g_emb = _launch_bwd(emb, w0, w1, ctx.alphas[0], ctx.alphas[1], g_w.contiguous(), ctx.mode, ctx.cache)
        return g_emb, None, None, None, None, None, None


def _launch_last(pre, w, alpha: float, cache: _WeightImages, g=None):
    """The last layer of a deeper MLP on the fused kernels' GEMM cores: ``silu(pre) @ (w alpha)`` (``g is None``,
    ``nqa_radial_mlp_last_fwd``) or ``(g @ (w alpha)^T) * silu'(pre)`` (``nqa_radial_mlp_last_bwd``)."""
    from ._topology import _ptr, current_stream_ptr

    lib = _lib.load()
    E, H = pre.shape
    W = w.shape[1]
    mode = _lib.NQA_MLP_F16X3
    backward = 0 if g is None else 1
    ws_bytes = lib.nqa_radial_mlp_workspace_bytes(mode, backward, H, W)
    ws, ready = cache.get(w, mode, backward, ws_bytes, E)
    flops = 2.0 * E * H * W
    if g is None:
        out = torch.empty((E, W), dtype=pre.dtype, device=pre.device)
        with torch.cuda.device(pre.device), ktimer.region("radial_mlp_fwd", 4.0 * E * (H + W), flops):
            rc = lib.nqa_radial_mlp_last_fwd(_lib.NQA_F32, mode, _ptr(pre), _ptr(w), alpha, H, W, E, _ptr(out), _ptr(ws),
                                             ws_bytes, int(ready), current_stream_ptr(pre.device))
        _lib.check(rc, "nqa_radial_mlp_last_fwd")
        return out
    out = torch.empty_like(pre)
    with torch.cuda.device(pre.device), ktimer.region("radial_mlp_bwd", 4.0 * E * (2 * H + W), flops):
        rc = lib.nqa_radial_mlp_last_bwd(_lib.NQA_F32, mode, _ptr(pre), _ptr(w), alpha, _ptr(g), None, H, W, E, _ptr(out),
                                         _ptr(ws), ws_bytes, int(ready), current_stream_ptr(pre.device))
    _lib.check(rc, "nqa_radial_mlp_last_bwd")
    return out


class _RadialMLPLastFn(torch.autograd.Function):
    """``pre -> silu(pre) @ (w alpha)``: one more layer on top of the fused two-layer kernel (depth >= 2 MLPs, inference:
    differentiable w.r.t. ``pre`` only)."""

    @staticmethod
    def forward(ctx, pre, w, alpha: float, cache: _WeightImages):
        pre = pre.contiguous()
        out = _launch_last(pre, w, alpha, cache)
        ctx.save_for_backward(pre, w)
        ctx.alpha, ctx.cache = alpha, cache
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        pre, w = ctx.saved_tensors
        return _launch_last(pre, w, ctx.alpha, ctx.cache, g=g.contiguous()), None, None, None


def _silu_derivs(p):
    """silu'(p), silu''(p)."""
    sig = torch.sigmoid(p)
    one_m = 1.0 - sig
    return sig * (1.0 + p * one_m), sig * one_m * (2.0 + p * (1.0 - 2.0 * sig))


class _RadialMLPTrainFn(torch.autograd.Function):
    """Training-mode radial MLP ``(emb, w0, w1) -> silu(emb W0) W1`` (``W_i = w_i alpha_i``): the same fused forward
    kernel as inference; its backward is ``_RadialMLPTrainBwdFn`` so that force-matching training
    (``nequip/nn/grad_output.py:220``, ``create_graph=True``) can differentiate the force pass once more."""

    @staticmethod
    def forward(ctx, emb, w0, w1, alpha0: float, alpha1: float, mode: int, cache: _WeightImages):
        emb = emb.contiguous()
        out = _launch_fwd(emb, w0.detach(), w1.detach(), alpha0, alpha1, mode, cache)
        ctx.save_for_backward(emb, w0, w1)
        from ..utils import wgrad as _wg

        ctx.args = (alpha0, alpha1, mode, cache, _wg.is_param_side(w1))
        return out

    @staticmethod
    def backward(ctx, g):
        emb, w0, w1 = ctx.saved_tensors
        g_emb, g_w0, g_w1 = _RadialMLPTrainBwdFn.apply(emb, w0, w1, g, *ctx.args)
        return g_emb, g_w0, g_w1, None, None, None, None


class _RadialMLPTrainBwdFn(torch.autograd.Function):
    """``(emb, w0, w1, g) -> (g_emb, g_w0, g_w1)``, the vector-Jacobian product of the MLP, itself differentiable once
    (second order of force matching).  With P = emb W0, h = silu(P), G_h = g W1^T, G_P = G_h silu'(P):
    ``g_emb = G_P W0^T``, ``g_W0 = emb^T G_P`` (both inside the fused kernel: ``nqa_radial_mlp_bwd_train``),
    ``g_W1 = h^T g`` (split-K ``nqa_wgrad`` on the kernel's ``h``).  Parameter gradients are skipped when the caller
    only wants data gradients (``utils/wgrad.py``).  Its own backward, for a cotangent c of g_emb and Q = c W0:
    d emb = (Q G_h silu''(P)) W0^T, dW0 (same launch, second-order mode), dW1 = (Q silu'(P))^T g (``nqa_wgrad``),
    d g = (Q silu'(P)) W1 (``nqa_radial_mlp_fwd_tangent``).  The exact-fp32 GEMM mode (``NQA_MLP_EXACT_FP32``) has no
    training epilogues: there the pieces are composed from ATen ops around the first-order kernels."""

    @staticmethod
    def forward(ctx, emb, w0, w1, g, alpha0: float, alpha1: float, mode: int, cache: _WeightImages,
                param_side: bool = False):
        from ..utils import wgrad as _wg

        g = g.contiguous()
        w0d, w1d = w0.detach(), w1.detach()
        g_w0 = g_w1 = None
        want_params = _wg.param_grads_wanted() and (w0.requires_grad or w1.requires_grad)
        if not want_params:
            g_emb = _launch_bwd(emb, w0d, w1d, alpha0, alpha1, g, mode, cache)
        elif mode == _lib.NQA_MLP_BF16X6:
            g_emb, h, p0 = _launch_bwd_train(emb, w0d, w1d, alpha0, alpha1, g, None, mode, cache)
            g_w0 = p0 * alpha0
            # (parameter-side weights, first order: the split-K product, its sum and the scaling next to the data chain)
            side = param_side and not torch.is_grad_enabled()
            if w1.is_leaf and w1.requires_grad and _wg.deferring():
                _wg.defer(w1, lambda: _launch_wgrad(h, g) * alpha1, h, g)  # (deferred parameter gradients, utils/wgrad.py)
            else:
                with (_wg.parameter_side(g.device, h, g) if side else contextlib.nullcontext()):
                    g_w1 = _launch_wgrad(h, g) * alpha1
        else:
            g_emb = _launch_bwd(emb, w0d, w1d, alpha0, alpha1, g, mode, cache)
            P = emb @ (w0d * alpha0)
            s1, _ = _silu_derivs(P)
            g_w1 = _launch_wgrad(P * torch.sigmoid(P), g) * alpha1
            g_w0 = _launch_wgrad(emb, (g @ (w1d * alpha1).t()) * s1) * alpha0
        ctx.save_for_backward(emb, w0, w1, g)
        ctx.args = (alpha0, alpha1, mode, cache)
        ctx.param_side = param_side
        ctx.mark_non_differentiable(*[t for t in (g_w0, g_w1) if t is not None])
        return g_emb, g_w0, g_w1

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, c, c_w0, c_w1):
        # cotangent c of g_emb only (parameter gradients are leaves of the training step, nobody differentiates them)
        emb, w0, w1, g = ctx.saved_tensors
        alpha0, alpha1, mode, cache = ctx.args
        w0d, w1d = w0.detach(), w1.detach()
        c = c.contiguous()
        need_emb, need_w0, need_w1, need_g = ctx.needs_input_grad[:4]
        d_emb = d_w0 = d_w1 = d_g = None
        if mode == _lib.NQA_MLP_BF16X6:
            if need_emb or need_w0 or need_w1:
                d_emb, R, p0 = _launch_bwd_train(emb, w0d, w1d, alpha0, alpha1, g, c, mode, cache)
                d_w0 = p0 * alpha0 if need_w0 else None
                if need_w1:
                    from ..utils import wgrad as _wg

                    if w1.is_leaf and w1.requires_grad and _wg.deferring():
                        _wg.defer(w1, lambda: _launch_wgrad(R, g) * alpha1, R, g)
                    else:
                        with (_wg.parameter_side(g.device, R, g) if ctx.param_side else contextlib.nullcontext()):
                            d_w1 = _launch_wgrad(R, g) * alpha1
            if need_g:
                d_g = _launch_fwd_tangent(emb, c, w0d, w1d, alpha0, alpha1, mode, cache)
            return d_emb, d_w0, d_w1, d_g, None, None, None, None, None
        W0, W1 = w0d * alpha0, w1d * alpha1
        P = emb @ W0
        s1, s2 = _silu_derivs(P)
        Q = c @ W0
        R = Q * s1
        G_h = g @ W1.t()
        cotP = Q * G_h * s2
        d_emb = cotP @ W0.t() if need_emb else None
        d_g = R @ W1 if need_g else None
        if need_w1:
            d_w1 = _launch_wgrad(R, g) * alpha1
        if need_w0:
            d_w0 = (_launch_wgrad(emb, cotP) + _launch_wgrad(c, G_h * s1)) * alpha0
        return d_emb, d_w0, d_w1, d_g, None, None, None, None, None


class ScalarLinearLayer(torch.nn.Module):
    def __init__(self, in_features: int, out_features: int, alpha: float = 1.0, bias: bool = False,
                 init_mode: str = "uniform") -> None:
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.register_buffer("alpha", torch.tensor(alpha), persistent=False)
        self.alpha_value = float(alpha)  # the same number as a host constant (kernel arguments, tracing)
        self.weight = torch.nn.Parameter(torch.empty((in_features, out_features)))
        if init_mode == "uniform":
            torch.nn.init.uniform_(self.weight, -sqrt(3), sqrt(3))
        elif init_mode == "normal":
            torch.nn.init.normal_(self.weight, mean=0.0, std=1.0)
        else:
            raise ValueError(f"Unknown init_mode: {init_mode}. Must be 'uniform' or 'normal'.")
        # bias (zeros) as in the reference (mlp.py:252-256); the hot-path MLPs are bias-free (interaction_block.py:125)
        if bias:
            self.bias = torch.nn.Parameter(torch.zeros(out_features))
        else:
            self.register_parameter("bias", None)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        w = self.weight * self.alpha
        if self.bias is not None:
            return torch.addmm(self.bias, input, w)
        if self.out_features == 1 and self.training and input.is_cuda and not traceable():
            # single-column layer (the per-atom energy readout): its weight-side backward as a library GEMM is a
            # [in, N] x [N, 1] product that runs on one workgroup (0.12 ms at 8k atoms); as multiply + row sum both
            # directions are plain streaming kernels
            return (input * w.view(1, -1)).sum(dim=1, keepdim=True)
        return torch.mm(input, w)

    def extra_repr(self) -> str:
        return (f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}, "
                f"alpha={float(self.alpha):.6f}")


class ScalarMLPFunction(_WeightCacheMixin, torch.nn.Module):
    def __init__(self, input_dim: int, output_dim: int, hidden_layers_depth: int = 0,
                 hidden_layers_width: Optional[int] = None, nonlinearity: Optional[str] = "silu", bias: bool = False,
                 forward_weight_init: bool = True, init_mode: str = "uniform"):
        super().__init__()
        assert nonlinearity in ("silu", None), "the hot path hard-codes SiLU (interaction_block.py:124)"
        assert forward_weight_init
        if hidden_layers_depth != 0:
            assert hidden_layers_depth > 0 and hidden_layers_width > 0
        self.has_bias = bool(bias)
        self.dims = [input_dim] + hidden_layers_depth * [hidden_layers_width] + [output_dim]
        self.num_layers = len(self.dims) - 1
        self.is_nonlinear = False
        mlp = torch.nn.Sequential()
        for layer, (h_in, h_out) in enumerate(zip(self.dims, self.dims[1:])):
            gain = 1.0 if nonlinearity is None or (layer == 0) else sqrt(2)
            mlp.append(ScalarLinearLayer(h_in, h_out, alpha=gain / sqrt(h_in), bias=bias, init_mode=init_mode))
            if (layer != self.num_layers - 1) and (nonlinearity is not None):
                mlp.append(torch.nn.SiLU())
                self.is_nonlinear = True
        self.mlp = mlp

    def _fused_ok(self, x: torch.Tensor, tracing_ok: bool = False) -> bool:
        if not x.is_cuda or x.dtype != torch.float32 or self.num_layers != 2 or not self.is_nonlinear or self.has_bias:
            return False
        if traceable() and not tracing_ok:
            return False  # (callers that cannot use the dispatcher-op form take the mm / SiLU form, which traces)
        ok = getattr(self, "_fused_supported", None)
        if ok is None:
            lib = _lib.load()
            ok = bool(lib.nqa_radial_mlp_supported(_lib.NQA_F32, self.dims[0], self.dims[1], self.dims[2]))
            self._fused_supported = ok
            self._alphas = (self.mlp[0].alpha_value, self.mlp[2].alpha_value)
        return ok

    def _deep_ok(self, x: torch.Tensor) -> bool:
        """Depth >= 2 (three or more weight matrices, e.g. the tutorial's 8-64-64-W): the first two layers on the fused
        two-layer kernel (its output = the pre-activations of the second hidden layer), every further layer on
        ``nqa_radial_mlp_last_fwd / _bwd``.  Inference, float32, no bias, hidden widths 64 / 128, fp16-split modes."""
        if (not x.is_cuda or x.dtype != torch.float32 or self.num_layers < 3 or not self.is_nonlinear or self.has_bias
                or traceable() or os.environ.get("NQA_MLP_DEEP_ATEN", "") not in ("", "0")):
            return False
        ok = getattr(self, "_deep_supported", None)
        if ok is None:
            lib = _lib.load()
            ok = (all(h in (64, 128) for h in self.dims[1:-1]) and self.dims[-1] % 4 == 0
                  and bool(lib.nqa_radial_mlp_supported(_lib.NQA_F32, self.dims[0], self.dims[1], self.dims[2])))
            self._deep_supported = ok
            self._deep_alphas = [self.mlp[2 * k].alpha_value for k in range(self.num_layers)]
        mode = radial_mlp_mode()
        return (ok and forward_mode(mode) == _lib.NQA_MLP_F16X3 and backward_mode(mode) == _lib.NQA_MLP_F16X3
                and not differentiable_parameters(self.training, *[self.mlp[2 * k].weight for k in range(self.num_layers)]))

    def _forward_deep(self, x):
        caches = self.__dict__.get("_deep_images")
        if caches is None:
            caches = self.__dict__["_deep_images"] = [_WeightImages() for _ in range(self.num_layers)]
        ws = [self.mlp[2 * k].weight for k in range(self.num_layers)]
        for k in range(1, self.num_layers):
            caches[k].validate(ws[k])
        al = self._deep_alphas
        pre = _RadialMLPFn.apply(x, ws[0].detach(), ws[1].detach(), al[0], al[1], radial_mlp_mode(), caches[1])
        for k in range(2, self.num_layers):
            pre = _RadialMLPLastFn.apply(pre, ws[k].detach(), al[k], caches[k])
        return pre

    def forward(self, x):
        if self._deep_ok(x):
            return self._forward_deep(x)
        # GPU, float32, two layers of a supported shape: fused MFMA kernels (the hidden layer stays on chip).  Eval mode:
        # weights are constants (inference Function, gradient w.r.t. the embedding only).  Training: the same kernels
        # inside a twice-differentiable Function pair that also produces the parameter gradients.
        if traceable():
            # while tracing: the fused kernels as dispatcher ops when the weights are constants (nn/_mlp_ops.py), else ATen
            if self._fused_ok(x, tracing_ok=True) and not differentiable_parameters(
                    self.training, self.mlp[0].weight, self.mlp[2].weight):
                from ._mlp_ops import radial_mlp

                return radial_mlp(x, self.mlp[0].weight.detach(), self.mlp[2].weight.detach(), self._alphas[0],
                                  self._alphas[1])
            return self.mlp(x)
        if self._fused_ok(x):
            cache = getattr(self, "_weight_images", None)
            if cache is None:
                cache = self._weight_images = _WeightImages()
            cache.validate(self.mlp[2].weight)
            diff = differentiable_parameters(self.training, self.mlp[0].weight, self.mlp[2].weight)
            if diff and os.environ.get("NQA_MLP_TRAIN_ATEN", "") in ("", "0"):
                # the parameters enter through views made on the parameter-side stream: the consumers of their gradients --
                # the views' adjoints, AccumulateGrad -- are then nodes of that stream, and the parameter-gradient launches
                # of the backward pass run there, next to the data chain (utils/wgrad.py)
                from ..utils import wgrad as _wg

                w0, w1 = self.mlp[0].weight, self.mlp[2].weight
                with _wg.parameter_side(x.device) as side:
                    if side is not None:
                        w0, w1 = w0.view_as(w0), w1.view_as(w1)
                if side is not None:
                    _wg.publish(x.device, w0, w1)
                return _RadialMLPTrainFn.apply(x, w0, w1, self._alphas[0], self._alphas[1], radial_mlp_mode(), cache)
            if not diff:
                return _RadialMLPFn.apply(x, self.mlp[0].weight.detach(), self.mlp[2].weight.detach(),
                                          self._alphas[0], self._alphas[1], radial_mlp_mode(), cache)
        return self.mlp(x)


class ScalarMLP(_WeightCacheMixin, GraphModuleMixin, torch.nn.Module):
    """Apply an MLP to a scalar node field (the per-atom energy readout, ``nequip_models.py:371-381``)."""

    def __init__(self, output_dim: int, hidden_layers_depth: int = 0, hidden_layers_width: Optional[int] = None,
                 nonlinearity: Optional[str] = "silu", bias: bool = False, forward_weight_init: bool = True,
                 field: str = AtomicDataDict.NODE_FEATURES_KEY, out_field: Optional[str] = None, irreps_in=None):
        super().__init__()
        self.field = field
        self.out_field = out_field if out_field is not None else field
        self._init_irreps(irreps_in=irreps_in, required_irreps_in=[self.field])
        assert len(self.irreps_in[self.field]) == 1 and self.irreps_in[self.field][0].ir == (0, 1)
        self.mlp_module = ScalarMLPFunction(
            input_dim=self.irreps_in[self.field][0].mul, output_dim=output_dim,
            hidden_layers_depth=hidden_layers_depth, hidden_layers_width=hidden_layers_width,
            nonlinearity=nonlinearity, bias=bias, forward_weight_init=forward_weight_init,
        )
        self.irreps_out[self.out_field] = Irreps([(self.mlp_module.dims[-1], (0, 1))])

    def _energy_head(self, data, h: torch.Tensor, gate_meta):
        """Gate (scalars only) + this depth-0 readout + the following PerTypeScaleShift as one launch (nn/_energy_head.py);
        None when the shapes / modes do not fit (the caller then applies the gate and the modules run one by one)."""
        tail = self.__dict__.get("_scale_shift")
        fn = self.mlp_module
        if (tail is None or fn.num_layers != 1 or fn.has_bias or fn.dims[-1] != 1 or not h.is_cuda or h.dtype != torch.float32
                or self.training or h.shape[1] % 4 != 0 or len(gate_meta.blocks) != 1
                or gate_meta.blocks[0][4] >= 0 or os.environ.get("NQA_NO_ENERGY_HEAD", "") not in ("", "0")
                or differentiable_parameters(self.training, fn.mlp[0].weight)):
            return None
        ss = tail[0]
        if ss.field != self.out_field or ss.out_field != self.out_field:
            return None
        # the head reads scales / shifts as float64 and the types as int64 (a `model.float()` must not be reinterpreted)
        if ((ss.has_scales and ss.scales.dtype != torch.float64) or (ss.has_shifts and ss.shifts.dtype != torch.float64)
                or data[AtomicDataDict.ATOM_TYPE_KEY].dtype != torch.int64):
            return None
        lin = fn.mlp[0]
        types = data[AtomicDataDict.ATOM_TYPE_KEY].view(-1)[: h.shape[0]].contiguous()
        _, _, _, _, _, act, cst = gate_meta.blocks[0]
        if traceable():  # the same launch as a dispatcher-op pair (nn/_energy_head.py)
            from ._energy_head import energy_head_op

            w = (lin.weight.detach().view(-1) * lin.alpha_value).to(torch.float32)
            return energy_head_op(h, w, ss.scales.view(-1) if ss.has_scales else None,
                                  ss.shifts.view(-1) if ss.has_shifts else None, types, act, cst)
        if torch.is_grad_enabled() and lin.weight.requires_grad and not h.requires_grad:
            # nothing upstream asks for a gradient but the readout weight does (an energy-only backward in eval mode): the
            # module chain, whose `mm` gives that gradient -- the fused head treats the weight as a constant
            return None
        key = (lin.weight._version, lin.weight.data_ptr(), h.device)
        cached = self.__dict__.get("_head_w")
        if cached is None or cached[0] != key:
            cached = (key, (lin.weight.detach().view(-1) * lin.alpha_value).to(torch.float32).contiguous())
            self.__dict__["_head_w"] = cached
        from ._energy_head import energy_head

        return energy_head(h, cached[1], ss.scales.view(-1) if ss.has_scales else None,
                           ss.shifts.view(-1) if ss.has_shifts else None, types, act, cst)

    def forward(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        pregate = data.pop("_nqa_pregate", None)
        if pregate is not None:  # the last convolution layer left its Gate to this module (ConvNetLayer.defer_gate)
            h, gate_meta = pregate[0], pregate[1]
            e = self._energy_head(data, h, gate_meta)
            if e is not None:
                data[self.out_field] = e
                # PerTypeScaleShift has been applied by THAT module's parameters (it passes the field on; a different
                # module in its place -- the Sequential entry was replaced after the model was built -- raises)
                data["_nqa_energy_scaled"] = id(self.__dict__["_scale_shift"][0])
                # the gated last-layer features were never formed: the field must not keep the PRE-gate rows
                if self.field != self.out_field:
                    data.pop(self.field, None)
                return data
            from ..o3 import _node_kernels

            data[self.field] = _node_kernels.apply_deferred_gate(pregate)
        data[self.out_field] = self.mlp_module(data[self.field])
        return data
