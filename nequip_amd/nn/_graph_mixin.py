"""``GraphModuleMixin`` / ``SequentialGraphNetwork``: the dict-in/dict-out module contract
(mirror of ``nequip/nn/_graph_mixin.py:21-95,146-149,235-238``, reduced to irreps bookkeeping)."""

from typing import Dict, Optional, Sequence

import torch

from ..data import AtomicDataDict
from ..o3.irreps import Irreps


def _fix(irreps: Optional[Dict]) -> Dict:
    out = {}
    for k, v in (irreps or {}).items():
        out[k] = None if v is None else Irreps(str(v) if not isinstance(v, (str, Irreps, list, tuple)) else v)
    return out


class GraphModuleMixin:
    def _init_irreps(self, irreps_in=None, my_irreps_in=None, required_irreps_in: Sequence[str] = (), irreps_out=None):
        irreps_in = _fix(irreps_in)
        my_irreps_in = _fix(my_irreps_in)
        irreps_out = _fix(irreps_out)
        for k in required_irreps_in:
            if k not in irreps_in:
                raise ValueError(f"{type(self).__name__} requires field '{k}' in irreps_in")
        for k, v in my_irreps_in.items():
            if k in irreps_in and irreps_in[k] != v:
                raise ValueError(f"field '{k}': irreps {irreps_in[k]} incompatible with required {v}")
        self.irreps_in = irreps_in
        new_out = dict(irreps_in)
        new_out.update(irreps_out)
        self.irreps_out = new_out


class SequentialGraphNetwork(GraphModuleMixin, torch.nn.Sequential):
    def __init__(self, modules: Dict[str, torch.nn.Module]):
        names = list(modules.keys())
        mods = list(modules.values())
        for (n1, m1), (n2, m2) in zip(zip(names, mods), zip(names[1:], mods[1:])):
            for k, v in m2.irreps_in.items():
                if k in m1.irreps_out and v is not None and m1.irreps_out[k] is not None:
                    assert m1.irreps_out[k] == v, f"{n1}.irreps_out[{k}]={m1.irreps_out[k]} != {n2}.irreps_in={v}"
        from collections import OrderedDict

        super().__init__(OrderedDict(modules))
        self._init_irreps(irreps_in=mods[0].irreps_in, irreps_out=mods[-1].irreps_out)

    def forward(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        for module in self:
            data = module(data)
        return data
