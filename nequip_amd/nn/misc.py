"""``ApplyFactor`` (mirror of ``nequip/nn/misc.py:29-48``).  When the field it scales is produced by the fused
edge-embedding kernel the multiplication is folded into that kernel (``fold_into``) and this module is a no-op."""

from typing import Optional

import torch

from ..data import AtomicDataDict
from ._graph_mixin import GraphModuleMixin


class ApplyFactor(GraphModuleMixin, torch.nn.Module):
    def __init__(self, in_field: str, factor: float, out_field: Optional[str] = None, irreps_in={}, fold_into=None):
        super().__init__()
        self.in_field = in_field
        self.out_field = in_field if out_field is None else out_field
        self.factor = factor
        self._folded = False
        if fold_into is not None and self.out_field == self.in_field:
            # same rounding as the reference: factor * (bessel * cutoff), all in model dtype
            fold_into.factor = float(fold_into.factor) * float(factor)
            self._folded = True
        self._init_irreps(irreps_in=irreps_in)
        self.irreps_out[self.out_field] = self.irreps_in[self.in_field]

    def forward(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        if not self._folded:
            data[self.out_field] = self.factor * data[self.in_field]
        return data
