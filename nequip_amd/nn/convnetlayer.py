"""ConvNetLayer: InteractionBlock + Gate nonlinearity (mirror of ``nequip/nn/convnetlayer.py:26-170``)."""

from typing import Any, Callable, Dict, Optional

import os

import torch

from ..data import AtomicDataDict
from ..o3.irreps import Irreps
from ..o3 import _node_kernels
from ..o3.modules import Gate, NormActivation
from ..utils.tracing import traceable
from ._graph_mixin import GraphModuleMixin
from .interaction_block import InteractionBlock
from .utils import tp_path_exists

acts = {
    "abs": torch.abs,
    "tanh": torch.tanh,
    "silu": torch.nn.functional.silu,
}


class ConvNetLayer(GraphModuleMixin, torch.nn.Module):
    resnet: bool

    def __init__(self, irreps_in, feature_irreps_hidden, convolution=InteractionBlock,
                 convolution_kwargs: Optional[Dict[str, Any]] = None, resnet: bool = False,
                 nonlinearity_type: str = "gate",
                 nonlinearity_scalars: Dict[str, str] = {"e": "silu", "o": "tanh"},
                 nonlinearity_gates: Dict[str, str] = {"e": "silu", "o": "tanh"}):
        super().__init__()
        assert nonlinearity_type in ("gate", "norm")
        convolution_kwargs = {} if convolution_kwargs is None else dict(convolution_kwargs)
        self.feature_irreps_hidden = Irreps(feature_irreps_hidden)
        self.resnet = resnet
        self._init_irreps(irreps_in=irreps_in, required_irreps_in=[AtomicDataDict.NODE_FEATURES_KEY])

        sh = self.irreps_in[AtomicDataDict.EDGE_ATTRS_KEY]
        x_prev = self.irreps_in[AtomicDataDict.NODE_FEATURES_KEY]
        # hidden irreps the convolution can actually produce from (previous features) x (spherical harmonics)
        reachable = [(mul, ir) for mul, ir in self.feature_irreps_hidden if tp_path_exists(x_prev, sh, ir)]
        scalars = Irreps([(mul, ir) for mul, ir in reachable if ir.l == 0])
        vectors = Irreps([(mul, ir) for mul, ir in reachable if ir.l > 0])
        layer_out = (scalars + vectors).simplify()
        act_of = {"scalars": {1: nonlinearity_scalars["e"], -1: nonlinearity_scalars["o"]},
                  "gates": {1: nonlinearity_gates["e"], -1: nonlinearity_gates["o"]}}
        build = self._gate_nonlinearity if nonlinearity_type == "gate" else self._norm_nonlinearity
        self.equivariant_nonlin, conv_irreps_out = build(scalars, vectors, layer_out, x_prev, sh, act_of)
        self.resnet = bool(resnet and layer_out == x_prev)

        convolution_kwargs.pop("irreps_in", None)
        convolution_kwargs.pop("irreps_out", None)
        self.conv = convolution(irreps_in=self.irreps_in, irreps_out=conv_irreps_out, **convolution_kwargs)
        self.irreps_out.update(self.conv.irreps_out)
        self.irreps_out[AtomicDataDict.NODE_FEATURES_KEY] = self.equivariant_nonlin.irreps_out

    @staticmethod
    def _gate_nonlinearity(scalars, vectors, layer_out, x_prev, sh, act_of):
        """e3nn ``Gate`` (nequip/nn/convnetlayer.py:94-112): one extra scalar per gated irrep copy, of the parity the
        convolution can produce; the convolution then outputs scalars (+) gates (+) gated."""
        gate_ir = "0e" if tp_path_exists(x_prev, sh, "0e") else "0o"
        gates = Irreps([(mul, gate_ir) for mul, _ in vectors])
        nonlin = Gate(
            irreps_scalars=scalars, act_scalars=[acts[act_of["scalars"][ir.p]] for _, ir in scalars],
            irreps_gates=gates, act_gates=[acts[act_of["gates"][ir.p]] for _, ir in gates],
            irreps_gated=vectors,
        )
        return nonlin, nonlin.irreps_in.simplify()

    @staticmethod
    def _norm_nonlinearity(scalars, vectors, layer_out, x_prev, sh, act_of):
        """e3nn ``NormActivation`` (nequip/nn/convnetlayer.py:113-125): the norm of every irrep copy goes through the scalar
        activation of EVEN scalars (a norm is even); no gate scalars, the convolution outputs the hidden irreps."""
        nonlin = NormActivation(irreps_in=layer_out, scalar_nonlinearity=acts[act_of["scalars"][1]], normalize=True,
                                epsilon=1e-8, bias=False)
        return nonlin, layer_out

    # Set by the model builder when the NEXT module of the network is a ConvNetLayer around an InteractionBlock: in eval mode
    # on the GPU the gate is then not applied here but folded into its two consumers (linear_1 and the self-connection of the
    # next block: one launch, o3/_node_kernels.py::fused_node_stage), and its backward into the launch that produces their
    # input gradient.  The pre-gate rows travel as `data["_nqa_pregate"]`; NODE_FEATURES_KEY is not valid in between.
    defer_gate: bool = False

    def _gate_deferred(self, h: torch.Tensor) -> bool:
        if not self.defer_gate or self.resnet or not isinstance(self.equivariant_nonlin, Gate):
            return False
        meta = self.equivariant_nonlin._kernel_meta
        if meta is None or not h.is_cuda or h.dtype != torch.float32 or self.training:
            return False
        if traceable() and os.environ.get("NQA_TRACE_NO_NODE_FUSION", "") not in ("", "0"):
            return False
        return _node_kernels.fusion_enabled() and meta.fusable()

    def forward(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        old_x = data[AtomicDataDict.NODE_FEATURES_KEY]
        data = self.conv(data)
        if self._gate_deferred(data[AtomicDataDict.NODE_FEATURES_KEY]):
            data["_nqa_pregate"] = (data[AtomicDataDict.NODE_FEATURES_KEY], self.equivariant_nonlin._kernel_meta,
                                    self.equivariant_nonlin._op_key)
            return data
        data[AtomicDataDict.NODE_FEATURES_KEY] = self.equivariant_nonlin(data[AtomicDataDict.NODE_FEATURES_KEY])
        if self.resnet:
            data[AtomicDataDict.NODE_FEATURES_KEY] = old_x + data[AtomicDataDict.NODE_FEATURES_KEY]
        return data
