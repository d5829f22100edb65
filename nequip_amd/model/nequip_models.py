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
    assert all(tn.isalnum() for tn in type_names)
    assert len(radial_mlp_depth) == len(radial_mlp_width) == len(feature_irreps_hidden)
    num_layers = len(radial_mlp_depth)
    assert all(l == 0 for l in Irreps(feature_irreps_hidden[-1]).ls)

    type_embed = NodeTypeEmbed(type_names=type_names, num_features=type_embed_num_features)
    spharm = SphericalHarmonicEdgeAttrs(irreps_edge_sh=irreps_edge_sh, irreps_in=type_embed.irreps_out)
    edge_norm = EdgeLengthNormalizer(r_max=r_max, type_names=type_names, per_edge_type_cutoff=per_edge_type_cutoff,
                                     irreps_in=spharm.irreps_out)
    bessel_encode = BesselEdgeLengthEncoding(
        num_bessels=num_bessels,
        trainable=bessel_trainable,
        cutoff=PolynomialCutoff(polynomial_cutoff_p),
        edge_invariant_field=AtomicDataDict.EDGE_EMBEDDING_KEY,
        irreps_in=edge_norm.irreps_out,
    )
    factor = ApplyFactor(
        in_field=AtomicDataDict.EDGE_EMBEDDING_KEY,
        factor=(2 * math.pi) / (r_max * r_max),
        irreps_in=bessel_encode.irreps_out,
        fold_into=bessel_encode,
    )
    modules = {
        "type_embed": type_embed,
        "spharm": spharm,
        "edge_norm": edge_norm,
        "bessel_encode": bessel_encode,
        "factor": factor,
    }
    prev_irreps_out = factor.irreps_out

    for layer_i in range(num_layers):
        current_convnet = ConvNetLayer(
            irreps_in=prev_irreps_out,
            feature_irreps_hidden=feature_irreps_hidden[layer_i],
            convolution_kwargs={
                "radial_mlp_depth": radial_mlp_depth[layer_i],
                "radial_mlp_width": radial_mlp_width[layer_i],
                "use_sc": (layer_i != 0) and convnet_sc,
                "is_first_layer": layer_i == 0,
                "avg_num_neighbors": avg_num_neighbors,
                "type_names": type_names,
            },
            resnet=(layer_i != 0) and convnet_resnet,
            nonlinearity_type=convnet_nonlinearity_type,
            nonlinearity_scalars=convnet_nonlinearity_scalars,
            nonlinearity_gates=convnet_nonlinearity_gates,
        )
        prev_irreps_out = current_convnet.irreps_out
        modules[f"layer{layer_i}_convnet"] = current_convnet
        if not edge_norm.symmetric:
            # cutoff(A <- B) != cutoff(B <- A): the two directed edges of a pair no longer share their radial weights
            current_convnet.conv.paired_radial_ok = False
        if layer_i > 0:
            # the previous layer's Gate is consumed by this layer's linear_1 / self-connection only: it may be folded into
            # them (eval mode, GPU; nn/convnetlayer.py::defer_gate)
            modules[f"layer{layer_i - 1}_convnet"].defer_gate = True

    if readout_mlp_hidden_layers_width is None:
        readout_mlp_hidden_layers_width = Irreps(feature_irreps_hidden[-1]).dim
    per_atom_energy_readout = ScalarMLP(
        output_dim=1,
        hidden_layers_depth=readout_mlp_hidden_layers_depth,
        hidden_layers_width=readout_mlp_hidden_layers_width,
        nonlinearity=readout_mlp_nonlinearity,
        bias=False,
        forward_weight_init=True,
        field=AtomicDataDict.NODE_FEATURES_KEY,
        out_field=AtomicDataDict.PER_ATOM_ENERGY_KEY,
        irreps_in=prev_irreps_out,
    )
    per_type_energy_scale_shift = PerTypeScaleShift(
        type_names=type_names,
        field=AtomicDataDict.PER_ATOM_ENERGY_KEY,
        out_field=AtomicDataDict.PER_ATOM_ENERGY_KEY,
        scales=per_type_energy_scales,
        shifts=per_type_energy_shifts,
        irreps_in=per_atom_energy_readout.irreps_out,
    )
    modules["per_atom_energy_readout"] = per_atom_energy_readout
    modules["per_type_energy_scale_shift"] = per_type_energy_scale_shift
    # eval mode on the GPU: the last layer's Gate, the readout and the scale / shift run as one launch per direction
    # (nn/_energy_head.py).  The link is a plain list entry: no second registration of the module, no new state-dict keys.
    if readout_mlp_hidden_layers_depth == 0:
        per_atom_energy_readout.__dict__["_scale_shift"] = [per_type_energy_scale_shift]
        modules[f"layer{num_layers - 1}_convnet"].defer_gate = True
    # nequip/model/energy_modules.py: total energy = sum of per-atom energies per frame
    modules["total_energy_sum"] = AtomwiseReduce(
        irreps_in=per_type_energy_scale_shift.irreps_out,
        reduce="sum",
        field=AtomicDataDict.PER_ATOM_ENERGY_KEY,
        out_field=AtomicDataDict.TOTAL_ENERGY_KEY,
    )
    energy_model = SequentialGraphNetwork(modules)
    return ForceStressOutput(energy_model, do_derivatives)
