"""Scalar MLPs (mirror of ``nequip/nn/mlp.py:34-271``): bias-free ``x @ (W * alpha)`` layers with
``alpha = gain / sqrt(fan_in)`` and SiLU in between.  ``ScalarMLPFunction`` as the per-edge radial network
(``InteractionBlock.edge_mlp``, ``nequip/nn/interaction_block.py:119-127``) is the dense-GEMM part of the hot
path: on the GPU it runs through the fused MFMA kernel of ``nequip_amd/csrc/radial_mlp.hip`` when available
for its shape, otherwise through hipBLASLt ``mm``."""

from math import sqrt
from typing import Optional

import torch

from .. import _lib
from ..data import AtomicDataDict
from ..o3.irreps import Irreps
from ..utils import ktimer
from ._graph_mixin import GraphModuleMixin


def radial_mlp_mode() -> int:
    """GEMM mode of the fused radial MLP: split-bf16 (fp32-accurate, default) or exact fp32 MFMA
    (``NQA_MLP_EXACT_FP32=1``)."""
    import os

    return _lib.NQA_MLP_FP32 if os.environ.get("NQA_MLP_EXACT_FP32", "") not in ("", "0") else _lib.NQA_MLP_BF16X6


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

    def get(self, w1: torch.Tensor, mode: int, backward: int, nbytes: int):
        hit = (mode, backward) in self.images
        if not hit:
            self.images[(mode, backward)] = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=w1.device)
        return self.images[(mode, backward)], hit


class _RadialMLPFn(torch.autograd.Function):
    """Fused two-layer radial MLP on the matrix cores (``nqa_radial_mlp_fwd/bwd``); inference path: differentiable
    w.r.t. the edge embedding only (that is what the force backward needs)."""

    @staticmethod
    def forward(ctx, emb, w0, w1, alpha0: float, alpha1: float, mode: int, cache: _WeightImages):
        from ._topology import _ptr, current_stream_ptr

        lib = _lib.load()
        emb = emb.contiguous()
        E, nb = emb.shape
        H, W = w1.shape
        out = torch.empty((E, W), dtype=emb.dtype, device=emb.device)
        flops = 2.0 * E * (nb * H + H * W)
        ws_bytes = lib.nqa_radial_mlp_workspace_bytes(mode, 0, H, W)
        ws, ready = cache.get(w1, mode, 0, ws_bytes)
        with torch.cuda.device(emb.device), ktimer.region("radial_mlp_fwd", 4.0 * E * (nb + W), flops):
            rc = lib.nqa_radial_mlp_fwd(_lib.NQA_F32, mode, _ptr(emb), _ptr(w0), alpha0, _ptr(w1), alpha1, nb, H, W, E,
                                        _ptr(out), _ptr(ws), ws_bytes, int(ready), current_stream_ptr(emb.device))
        _lib.check(rc, "nqa_radial_mlp_fwd")
        ctx.save_for_backward(emb, w0, w1)
        ctx.alphas = (alpha0, alpha1)
        ctx.mode, ctx.cache = mode, cache
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_w):
        from ._topology import _ptr, current_stream_ptr

        emb, w0, w1 = ctx.saved_tensors
        lib = _lib.load()
        g_w = g_w.contiguous()
        E, nb = emb.shape
        H, W = w1.shape
        g_emb = torch.empty_like(emb)
        flops = 2.0 * E * (nb * H * 2 + H * W)
        ws_bytes = lib.nqa_radial_mlp_workspace_bytes(ctx.mode, 1, H, W)
        ws, ready = ctx.cache.get(w1, ctx.mode, 1, ws_bytes)
        with torch.cuda.device(emb.device), ktimer.region("radial_mlp_bwd", 4.0 * E * (2 * nb + W), flops):
            rc = lib.nqa_radial_mlp_bwd(_lib.NQA_F32, ctx.mode, _ptr(emb), _ptr(w0), ctx.alphas[0], _ptr(w1),
                                        ctx.alphas[1], _ptr(g_w), nb, H, W, E, _ptr(g_emb), _ptr(ws), ws_bytes,
                                        int(ready), current_stream_ptr(emb.device))
        _lib.check(rc, "nqa_radial_mlp_bwd")
        return g_emb, None, None, None, None, None, None


class ScalarLinearLayer(torch.nn.Module):
    def __init__(self, in_features: int, out_features: int, alpha: float = 1.0, bias: bool = False,
                 init_mode: str = "uniform") -> None:
        super().__init__()
        assert not bias, "the hot-path MLPs are bias-free (nequip/nn/interaction_block.py:125)"
        self.in_features = in_features
        self.out_features = out_features
        self.register_buffer("alpha", torch.tensor(alpha), persistent=False)
        self.weight = torch.nn.Parameter(torch.empty((in_features, out_features)))
        if init_mode == "uniform":
            torch.nn.init.uniform_(self.weight, -sqrt(3), sqrt(3))
        elif init_mode == "normal":
            torch.nn.init.normal_(self.weight, mean=0.0, std=1.0)
        else:
            raise ValueError(f"Unknown init_mode: {init_mode}")

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return torch.mm(input, self.weight * self.alpha)

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, out_features={self.out_features}, alpha={float(self.alpha):.6f}"


class ScalarMLPFunction(torch.nn.Module):
    def __init__(self, input_dim: int, output_dim: int, hidden_layers_depth: int = 0,
                 hidden_layers_width: Optional[int] = None, nonlinearity: Optional[str] = "silu", bias: bool = False,
                 forward_weight_init: bool = True, init_mode: str = "uniform"):
        super().__init__()
        assert nonlinearity in ("silu", None), "the hot path hard-codes SiLU (interaction_block.py:124)"
        assert forward_weight_init
        if hidden_layers_depth != 0:
            assert hidden_layers_depth > 0 and hidden_layers_width > 0
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

    def _fused_ok(self, x: torch.Tensor) -> bool:
        if self.training or not x.is_cuda or x.dtype != torch.float32 or self.num_layers != 2 or not self.is_nonlinear:
            return False
        ok = getattr(self, "_fused_supported", None)
        if ok is None:
            lib = _lib.load()
            ok = bool(lib.nqa_radial_mlp_supported(_lib.NQA_F32, self.dims[0], self.dims[1], self.dims[2]))
            self._fused_supported = ok
            self._alphas = (float(self.mlp[0].alpha), float(self.mlp[2].alpha))
        return ok

    def forward(self, x):
        # inference on the GPU: one fused MFMA kernel (hidden layer stays on chip); training keeps the
        # mm/SiLU formulation so that parameter gradients and double backward come from autograd
        if self._fused_ok(x):
            cache = getattr(self, "_weight_images", None)
            if cache is None:
                cache = self._weight_images = _WeightImages()
            # (the fused path only runs in eval mode: weights are constants there, as for o3.Linear)
            cache.validate(self.mlp[2].weight)
            return _RadialMLPFn.apply(x, self.mlp[0].weight.detach(), self.mlp[2].weight.detach(), self._alphas[0],
                                      self._alphas[1], radial_mlp_mode(), cache)
        return self.mlp(x)


class ScalarMLP(GraphModuleMixin, torch.nn.Module):
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

    def forward(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        data[self.out_field] = self.mlp_module(data[self.field])
        return data
