"""``GraphModel``: top-level wrapper (mirror of ``nequip/nn/graph_model.py:37-155``): copies the input dict,
keeps only the model's input fields and exposes the custom-ops metadata of accelerated submodules."""

import os
from typing import List

import torch

from ..data import AtomicDataDict
from ._graph_mixin import GraphModuleMixin


class GraphModel(GraphModuleMixin, torch.nn.Module):
    is_compile_graph_model: bool = False

    def __init__(self, model: GraphModuleMixin, type_names: List[str] = (), model_dtype=torch.float32,
                 r_max: float = None) -> None:
        super().__init__()
        self.model = model
        self.type_names = list(type_names)
        self.model_dtype = model_dtype
        self.r_max = r_max
        self.model_input_fields = [
            AtomicDataDict.POSITIONS_KEY, AtomicDataDict.EDGE_INDEX_KEY, AtomicDataDict.ATOM_TYPE_KEY,
            AtomicDataDict.CELL_KEY, AtomicDataDict.EDGE_CELL_SHIFT_KEY, AtomicDataDict.BATCH_KEY,
            AtomicDataDict.NUM_NODES_KEY, AtomicDataDict.EDGE_VECTORS_KEY,
            # local / ghost bookkeeping of a domain-decomposed caller (nequip/nn/graph_model.py:70-75)
            AtomicDataDict.LMP_MLIAP_DATA_KEY, AtomicDataDict.NUM_LOCAL_GHOST_NODES_KEY,
        ]  # fmt: skip
        self._init_irreps(irreps_in=self.model.irreps_in, irreps_out=self.model.irreps_out)

    @property
    def nequip_custom_ops_libs(self):
        libs = []
        for m in self.modules():
            for lib in getattr(m, "_nequip_custom_ops_libs", ()):
                if lib not in libs:
                    libs.append(lib)
        return tuple(libs)

    @property
    def metadata(self):
        """String-valued model metadata for compiled artefacts (nequip/nn/graph_model.py:20-36,100-146): model dtype,
        type names, cutoff, and the libraries whose import registers the custom ops the model calls."""
        out = {
            "model_dtype": {torch.float32: "float32", torch.float64: "float64"}.get(self.model_dtype, str(self.model_dtype)),
            "type_names": " ".join(self.type_names),
            "num_types": str(len(self.type_names)),
        }
        if self.r_max is not None:
            out["r_max"] = str(self.r_max)
        libs = self.nequip_custom_ops_libs
        if libs:
            out["nequip_custom_ops_libs"] = " ".join(sorted(libs))
        return out

    def forward(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        new_data: AtomicDataDict.Type = {k: v for k, v in data.items() if k in self.model_input_fields}
        # callers that are known to refill their index buffers in place (LAMMPS ML-IAP hands in views of its own arrays)
        # never get a topology cached across calls; NQA_TOPOLOGY_CACHE=0 extends that to everybody
        from ..utils.tracing import traceable
        from ._topology import topology_cache

        if traceable():  # (a tracer follows the model only; topologies are built inside the ops at run time)
            return self.model(new_data)
        trust = (AtomicDataDict.LMP_MLIAP_DATA_KEY not in new_data
                 and os.environ.get("NQA_TOPOLOGY_CACHE", "1") not in ("0",))
        with topology_cache.scope(trust_identity=trust):
            self._start_pairing(new_data)
            return self.model(new_data)

    def _start_pairing(self, data) -> None:
        """A new edge list (MD: every step): start its reverse-edge pairing before anything else is launched, so that the
        verdict is on the host by the time the first convolution asks for it (``EdgeTopology.start_pairing``).  Only for the
        evaluations that pair at all: float32 models differentiated w.r.t. positions."""
        K = AtomicDataDict
        ei, pos = data.get(K.EDGE_INDEX_KEY), data.get(K.POSITIONS_KEY)
        if (ei is None or pos is None or not ei.is_cuda or K.EDGE_VECTORS_KEY in data or self.model_dtype != torch.float32
                or os.environ.get("NQA_NO_EARLY_PAIRING", "") not in ("", "0")):
            return
        from ._topology import topology_cache

        topology_cache.get(ei[0], ei[1], pos.shape[0]).start_pairing(data.get(K.EDGE_CELL_SHIFT_KEY))
