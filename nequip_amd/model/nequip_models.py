"""NequIP GNN model builders (mirror of ``nequip/model/nequip_models.py:116-399`` and the ``@model_builder``
wrapper ``nequip/model/utils.py:104-216``): same arguments, same module names/ordering
(``type_embed -> spharm -> edge_norm -> bessel_encode -> factor -> layer{i}_convnet -> per_atom_energy_readout
-> per_type_energy_scale_shift -> total_energy_sum``), wrapped in ``ForceStressOutput`` and ``GraphModel``.
The modules on the hot path are the HIP-backed ones of ``nequip_amd.nn``.
"""

from __future__ import annotations

import contextlib
import math
from typing import Dict, List, Optional, Sequence, Union

import torch

from ..data import AtomicDataDict
from ..nn import (
    ApplyFactor,
    AtomwiseReduce,
    ConvNetLayer,
    ForceStressOutput,
    GraphModel,
    PerTypeScaleShift,
    ScalarMLP,
    SequentialGraphNetwork,
)
from ..nn.embedding import (
    BesselEdgeLengthEncoding,
    EdgeLengthNormalizer,
    NodeTypeEmbed,
    PolynomialCutoff,
    SphericalHarmonicEdgeAttrs,
)
from ..o3.irreps import Irreps

_NEQUIP_GNN_PRESETS = {
    "S": {"num_layers": 2, "l_max": 1, "num_features": [128, 64]},
    "M": {"num_layers": 4, "l_max": 2, "num_features": [128, 64, 32]},
    "L": {"num_layers": 6, "l_max": 3, "num_features": [128, 64, 32, 32]},
    "XL": {"num_layers": 6, "l_max": 4, "num_features": [320, 96, 64, 32, 32]},
}
_NEQUIP_GNN_STANDARD_PRESET = {
    "parity": False,
    "type_embed_num_features": 32,
    "radial_mlp_depth": 1,
    "radial_mlp_width": 128,
}


@contextlib.contextmanager
def torch_default_dtype(dtype):
    orig = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        yield
    finally:
        torch.set_default_dtype(orig)


def _dtype_from_name(name) -> torch.dtype:
    if isinstance(name, torch.dtype):
        return name
    return {"float32": torch.float32, "float64": torch.float64}[name]


def _build(builder, seed: int, model_dtype, **kwargs) -> GraphModel:
    """``@model_builder`` semantics: seed under an isolated RNG, build under ``model_dtype`` as default dtype."""
    dtype = _dtype_from_name(model_dtype)
    cpu_state = torch.get_rng_state()
    try:
        torch.manual_seed(seed)
        with torch_default_dtype(dtype):
            model = builder(**kwargs)
    finally:
        torch.set_rng_state(cpu_state)
    return GraphModel(model, type_names=kwargs.get("type_names", ()), model_dtype=dtype, r_max=kwargs.get("r_max"))


def PresetNequIPGNNModel(preset: str, **kwargs) -> GraphModel:
    preset = preset.upper()
    assert preset in _NEQUIP_GNN_PRESETS
    model_kwargs = {**_NEQUIP_GNN_STANDARD_PRESET, **_NEQUIP_GNN_PRESETS[preset]}
    model_kwargs.update(kwargs)
    return NequIPGNNModel(**model_kwargs)


def NequIPGNNModel(
    num_layers: int = 4,
    l_max: int = 1,
    parity: bool = True,
    num_features: Union[int, List[int]] = 32,
    type_embed_num_features: Optional[int] = None,
    radial_mlp_depth: int = 1,
    radial_mlp_width: int = 128,
    **kwargs,
) -> GraphModel:
    assert num_layers > 0
    irreps_edge_sh = repr(Irreps.spherical_harmonics(lmax=l_max))
    if isinstance(num_features, int):
        num_features = [num_features] * (l_max + 1)
    assert len(num_features) == l_max + 1
    type_embed_num_features = type_embed_num_features if type_embed_num_features is not None else num_features[0]
    feature_irreps_hidden = repr(
        Irreps(
            [
                (num_features[l], (l, p))
                for l in range(l_max + 1)
                for p in ((1, -1) if parity else ((1,) if l % 2 == 0 else (-1,)))
            ]
        )
    )
    feature_irreps_hidden_list = [feature_irreps_hidden] * (num_layers - 1)
    feature_irreps_hidden_list += [repr(Irreps([(num_features[0], (0, 1))]))]
    return FullNequIPGNNModel(
        irreps_edge_sh=irreps_edge_sh,
        type_embed_num_features=type_embed_num_features,
        feature_irreps_hidden=feature_irreps_hidden_list,
        radial_mlp_depth=[radial_mlp_depth] * num_layers,
        radial_mlp_width=[radial_mlp_width] * num_layers,
        **kwargs,
    )


def FullNequIPGNNModel(seed: int = 0, model_dtype="float32", **kwargs) -> GraphModel:
    return _build(_full_nequip_energy_model, seed, model_dtype, **kwargs)


def _full_nequip_energy_model(
    r_max: float,
    type_names: Sequence[str],
    radial_mlp_depth: Sequence[int],
    radial_mlp_width: Sequence[int],
    feature_irreps_hidden: Sequence[Union[str, Irreps]],
    irreps_edge_sh: Union[int, str, Irreps],
    type_embed_num_features: int,
    readout_mlp_hidden_layers_depth: int = 0,
    readout_mlp_hidden_layers_width: Optional[int] = None,
    readout_mlp_nonlinearity: Optional[str] = "silu",
    num_bessels: int = 8,
    bessel_trainable: bool = False,
    per_edge_type_cutoff: Optional[Dict[str, Union[float, Dict[str, float]]]] = None,
    polynomial_cutoff_p: int = 6,
    avg_num_neighbors: Optional[Union[float, Dict[str, float]]] = None,
    per_type_energy_scales: Optional[Union[float, Dict[str, float]]] = None,
    per_type_energy_shifts: Optional[Union[float, Dict[str, float]]] = None,
    do_derivatives: bool = True,
    convnet_sc: bool = True,
    convnet_resnet: bool = False,
    convnet_nonlinearity_type: str = "gate",
    convnet_nonlinearity_scalars: Dict[str, str] = {"e": "silu", "o": "tanh"},
    convnet_nonlinearity_gates: Dict[str, str] = {"e": "silu", "o": "tanh"},
):
    if not all(name.isalnum() for name in type_names):
        raise AssertionError("type names must be alphanumeric")
    depths, widths, hidden = list(radial_mlp_depth), list(radial_mlp_width), list(feature_irreps_hidden)
    if not (len(depths) == len(widths) == len(hidden)):
        raise AssertionError("radial_mlp_depth, radial_mlp_width and feature_irreps_hidden list one entry per layer")
    if any(l != 0 for l in Irreps(hidden[-1]).ls):
        raise AssertionError("the last layer keeps scalars only (the readout acts on them)")

    chain = _Chain()
    # ---- geometry: type embedding, spherical harmonics, normalised lengths, Bessel x cutoff (x 2 pi / r_max^2, folded) ----
    chain.add("type_embed", lambda prev: NodeTypeEmbed(type_names=type_names, num_features=type_embed_num_features))
    chain.add("spharm", lambda prev: SphericalHarmonicEdgeAttrs(irreps_edge_sh=irreps_edge_sh, irreps_in=prev))
    edge_norm = chain.add("edge_norm", lambda prev: EdgeLengthNormalizer(
        r_max=r_max, type_names=type_names, per_edge_type_cutoff=per_edge_type_cutoff, irreps_in=prev))
    bessel = chain.add("bessel_encode", lambda prev: BesselEdgeLengthEncoding(
        num_bessels=num_bessels, trainable=bessel_trainable, cutoff=PolynomialCutoff(polynomial_cutoff_p),
        edge_invariant_field=AtomicDataDict.EDGE_EMBEDDING_KEY, irreps_in=prev))
    chain.add("factor", lambda prev: ApplyFactor(in_field=AtomicDataDict.EDGE_EMBEDDING_KEY,
                                                 factor=(2 * math.pi) / (r_max * r_max), irreps_in=prev, fold_into=bessel))

    # ---- message passing ----
    convnets = []
    for k, (irreps_k, depth_k, width_k) in enumerate(zip(hidden, depths, widths)):
        first = k == 0
        conv_kwargs = dict(radial_mlp_depth=depth_k, radial_mlp_width=width_k, use_sc=convnet_sc and not first,
                           is_first_layer=first, avg_num_neighbors=avg_num_neighbors, type_names=type_names)
        convnets.append(chain.add(f"layer{k}_convnet", lambda prev: ConvNetLayer(
            irreps_in=prev, feature_irreps_hidden=irreps_k, convolution_kwargs=conv_kwargs,
            resnet=convnet_resnet and not first, nonlinearity_type=convnet_nonlinearity_type,
            nonlinearity_scalars=convnet_nonlinearity_scalars, nonlinearity_gates=convnet_nonlinearity_gates)))

    # ---- energy head: readout -> per-type scale / shift (float64) -> per-frame sum ----
    readout_width = Irreps(hidden[-1]).dim if readout_mlp_hidden_layers_width is None else readout_mlp_hidden_layers_width
    energy = AtomicDataDict.PER_ATOM_ENERGY_KEY
    readout = chain.add("per_atom_energy_readout", lambda prev: ScalarMLP(
        output_dim=1, hidden_layers_depth=readout_mlp_hidden_layers_depth, hidden_layers_width=readout_width,
        nonlinearity=readout_mlp_nonlinearity, bias=False, forward_weight_init=True,
        field=AtomicDataDict.NODE_FEATURES_KEY, out_field=energy, irreps_in=prev))
    scale_shift = chain.add("per_type_energy_scale_shift", lambda prev: PerTypeScaleShift(
        type_names=type_names, field=energy, out_field=energy, scales=per_type_energy_scales,
        shifts=per_type_energy_shifts, irreps_in=prev))
    chain.add("total_energy_sum", lambda prev: AtomwiseReduce(irreps_in=prev, reduce="sum", field=energy,
                                                             out_field=AtomicDataDict.TOTAL_ENERGY_KEY))

    _plan_fusions(convnets, edge_norm, readout, scale_shift, readout_mlp_hidden_layers_depth)
    _plan_embedding_fusion(chain.modules["spharm"], edge_norm, bessel)
    return ForceStressOutput(SequentialGraphNetwork(chain.modules), do_derivatives)


class _Chain:
    """The module sequence under construction: ``add(name, factory)`` builds the next module from the irreps the previous
    one leaves behind (names and order are the reference's: they are the state-dict keys)."""

    def __init__(self):
        self.modules = {}
        self._irreps = None

    def add(self, name: str, factory):
        module = factory(self._irreps)
        self.modules[name] = module
        self._irreps = module.irreps_out
        return module


def _plan_embedding_fusion(spharm, edge_norm, bessel) -> None:
    """Spherical harmonics and radial basis in ONE launch per direction (`nn/embedding/_edge.py`): only for a plain `r_max`
    (per-edge-type cutoffs reach the radial kernel through `edge_norm`, which runs between the two modules).  A plain list
    entry, no registration: module names, parameters and state-dict keys are untouched."""
    if not edge_norm._per_edge_type and spharm._output_dtype == bessel._output_dtype:
        spharm.__dict__["_fuse_radial"] = [bessel, edge_norm]


def _plan_fusions(convnets, edge_norm, readout, scale_shift, readout_depth: int) -> None:
    """What the eval-mode GPU path may fuse across module boundaries (none of it changes a module's parameters or keys):

    * a Gate between two convolution layers is consumed by the next layer's ``linear_1`` / self-connection only, so it may be
      folded into them (``ConvNetLayer.defer_gate`` -> ``o3/_node_kernels.py::fused_node_stage``);
    * the last layer's Gate, a depth-0 readout and the scale / shift run as one launch per direction (``nn/_energy_head.py``);
      the readout finds the scale / shift module through a plain list entry -- no second registration, no new keys;
    * asymmetric per-edge-type cutoffs (cutoff(A <- B) != cutoff(B <- A)): the two directed edges of a pair no longer share
      their radial weights, the reverse-edge pairing is switched off."""
    for layer in convnets[:-1]:
        layer.defer_gate = True
    if readout_depth == 0:
        readout.__dict__["_scale_shift"] = [scale_shift]
        convnets[-1].defer_gate = True
    if not edge_norm.symmetric:
        for layer in convnets:
            layer.conv.paired_radial_ok = False
