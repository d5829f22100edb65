"""``AvgNumNeighborsNorm``: node features times ``1 / sqrt(avg_num_neighbors)`` -- one global value or one per atom type.

Interface of ``nequip/nn/norm.py:7-68`` (constructor arguments, the non-persistent ``norm_const`` buffer, the
``feature_norm_factor`` entry cached in the data dict for the later layers).  On the GPU path with a single global value
``InteractionBlock`` does not call ``forward`` at all: it reads ``norm_scalar`` and folds the factor into the ``linear_1``
launch (``nqa_node_linear``'s ``scale`` argument), so no separate pass over the ``[N, D]`` features exists there.
"""

import math
from typing import Dict, List, Sequence, Union

import torch

from ..data import AtomicDataDict

_FEATURES = AtomicDataDict.NODE_FEATURES_KEY
_CACHE_KEY = AtomicDataDict.FEATURE_NORM_FACTOR_KEY


def _per_type_values(spec, type_names: Sequence[str]) -> List[float]:
    """A number applies to every type; a dict must name exactly the model's types."""
    if spec is None:
        raise AssertionError("avg_num_neighbors must be specified")
    if isinstance(spec, dict):
        if set(spec) != set(type_names):
            raise AssertionError(f"avg_num_neighbors keys {sorted(spec)} do not match the type names {sorted(type_names)}")
        return [float(spec[name]) for name in type_names]
    if isinstance(spec, (int, float)):
        return [float(spec)]
    raise RuntimeError("Unrecognized format for `avg_num_neighbors`, only floats or dicts allowed.")


class AvgNumNeighborsNorm(torch.nn.Module):
    in_field = out_field = _FEATURES
    norm_key = _CACHE_KEY

    def __init__(self, type_names: Sequence[str], avg_num_neighbors: Union[float, Dict[str, float]]) -> None:
        super().__init__()
        inv_sqrt = [1.0 / math.sqrt(v) for v in _per_type_values(avg_num_neighbors, type_names)]
        # column vector: row t = factor of atom type t (a single row when the value is global)
        self.register_buffer("norm_const", torch.tensor(inv_sqrt).unsqueeze(1), persistent=False)
        self.norm_shortcut = len(inv_sqrt) == 1
        # python float for callers that fold the type-independent factor into their own kernel launch
        self.norm_scalar = inv_sqrt[0] if self.norm_shortcut else None

    def _factor(self, data: AtomicDataDict.Type, rows: int) -> torch.Tensor:
        cached = data.get(self.norm_key)
        if cached is not None and cached.size(0) == rows:
            return cached
        if self.norm_shortcut:
            factor = self.norm_const.expand(rows, 1)
        else:
            types = data[AtomicDataDict.ATOM_TYPE_KEY].view(-1)[:rows]
            factor = self.norm_const.index_select(0, types)
        data[self.norm_key] = factor  # [rows, 1], reused by the following layers
        return factor

    def forward(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        x = data[self.in_field]
        data[self.out_field] = x * self._factor(data, x.size(0))
        return data
