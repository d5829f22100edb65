"""Per-atom readout helpers (mirror of ``nequip/nn/atomwise.py:62-284``): ``AtomwiseReduce`` (per-frame energy sum)
and ``PerTypeScaleShift`` (float64 scale/shift).  O(N) elementwise work, outside the hot kernels."""

from typing import Dict, List, Optional, Union

import torch

from ..data import AtomicDataDict
from ._graph_mixin import GraphModuleMixin
from .utils import scatter

_GLOBAL_DTYPE = torch.float64


class AtomwiseReduce(GraphModuleMixin, torch.nn.Module):
    def __init__(self, field: str, out_field: Optional[str] = None, reduce="sum", irreps_in={}):
        super().__init__()
        assert reduce == "sum"
        self.reduce = reduce
        self.field = field
        self.out_field = f"{reduce}_{field}" if out_field is None else out_field
        self._init_irreps(
            irreps_in=irreps_in,
            irreps_out=({self.out_field: irreps_in[self.field]} if self.field in irreps_in else {}),
        )

    def forward(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        field = data[self.field]
        if AtomicDataDict.BATCH_KEY in data:
            result = scatter(field, data[AtomicDataDict.BATCH_KEY], dim=0, dim_size=AtomicDataDict.num_frames(data))
        else:
            result = field.sum(dim=0, keepdim=True)
        data[self.out_field] = result
        return data


class PerTypeScaleShift(GraphModuleMixin, torch.nn.Module):
    def __init__(self, type_names: List[str], field: str, out_field: Optional[str] = None,
                 scales: Optional[Union[float, Dict[str, float]]] = None,
                 shifts: Optional[Union[float, Dict[str, float]]] = None, irreps_in={}):
        super().__init__()
        self.type_names = type_names
        self.num_types = len(type_names)
        self.field = field
        self.out_field = field if out_field is None else out_field
        self._init_irreps(irreps_in=irreps_in, my_irreps_in={self.field: "0e"},
                          irreps_out={self.out_field: irreps_in[self.field]})
        self.out_dtype = _GLOBAL_DTYPE

        def prep(v):
            if v is None:
                return None
            if isinstance(v, (float, int)):
                v = [v]
            elif isinstance(v, dict):
                assert set(self.type_names) == set(v.keys())
                v = [v[name] for name in self.type_names]
            else:
                raise ValueError("per-type scales/shifts must be a float or a dict keyed by type name")
            return torch.as_tensor(v, dtype=self.out_dtype).reshape(-1, 1)

        scales, shifts = prep(scales), prep(shifts)
        self.has_scales = scales is not None
        self.has_shifts = shifts is not None
        self.register_buffer("scales", scales if self.has_scales else torch.Tensor())
        self.register_buffer("shifts", shifts if self.has_shifts else torch.Tensor())
        self.scales_shortcut = self.scales.numel() == 1
        self.shifts_shortcut = self.shifts.numel() == 1

    def forward(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        scaled_by = data.pop("_nqa_energy_scaled", None)
        if scaled_by is not None and scaled_by != id(self):
            raise RuntimeError("the fused energy head applied the scales / shifts of a PerTypeScaleShift module that is no "
                               "longer the one in the model (module replaced after the model was built): rebuild the model "
                               "or set NQA_NO_ENERGY_HEAD=1")
        if scaled_by is not None:
            # the fused energy head (nn/_energy_head.py) has produced `field` in float64 with this module's scales and
            # shifts applied
            if self.out_field != self.field:
                data[self.out_field] = data[self.field]
            return data
        if not (self.has_scales or self.has_shifts):
            data[self.out_field] = data[self.field].to(self.out_dtype)
            return data
        in_field = data[self.field]
        types = data[AtomicDataDict.ATOM_TYPE_KEY].view(-1)[: in_field.size(0)]
        scales = shifts = None
        if self.has_scales:
            scales = self.scales if self.scales_shortcut else torch.nn.functional.embedding(types, self.scales)
        if self.has_shifts:
            shifts = self.shifts if self.shifts_shortcut else torch.nn.functional.embedding(types, self.shifts)
        in_field = in_field.to(self.out_dtype)
        if self.has_scales and self.has_shifts:
            in_field = torch.addcmul(shifts, scales, in_field)
        else:
            if self.has_scales:
                in_field = scales * in_field
            if self.has_shifts:
                in_field = shifts + in_field
        data[self.out_field] = in_field
        return data
