"""Node type embedding (mirror of ``nequip/nn/embedding/node.py:39-175`` without categorical graph fields)."""

from typing import List, Optional

import torch

from ...data import AtomicDataDict
from ...o3.irreps import Irreps
from ...utils.wgrad import differentiable_parameters
from .._graph_mixin import GraphModuleMixin


def _tracing() -> bool:
    from ...utils.tracing import traceable

    return traceable()


class NodeTypeEmbed(GraphModuleMixin, torch.nn.Module):
    def __init__(self, type_names: List[str], num_features: int, set_features: bool = True, irreps_in=None):
        super().__init__()
        self.num_types = len(type_names)
        self.set_features = set_features
        self.embed_module = torch.nn.Embedding(num_embeddings=self.num_types, embedding_dim=num_features)
        irreps_out = {AtomicDataDict.NODE_ATTRS_KEY: Irreps([(num_features, (0, 1))])}
        if set_features:
            irreps_out[AtomicDataDict.NODE_FEATURES_KEY] = irreps_out[AtomicDataDict.NODE_ATTRS_KEY]
        self._init_irreps(irreps_in=irreps_in, irreps_out=irreps_out)

    def forward(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        atom_types = data[AtomicDataDict.ATOM_TYPE_KEY].view(-1)
        # Out-of-range types: `F.embedding` (eval path; the reference's only path) raises / faults on them, the one-hot
        # product of the training path below would silently give a zero row.  Host tensors are checked here (free);
        # device tensors are not (the check would synchronise every step) -- validate type maps where the data is built.
        if not atom_types.is_cuda and atom_types.numel() > 0 and not _tracing():
            lo, hi = int(atom_types.min()), int(atom_types.max())
            if lo < 0 or hi >= self.num_types:
                raise IndexError(f"atom type index out of range: [{lo}, {hi}] for {self.num_types} types")
        # eval mode: parameters are constants (same convention as o3.Linear) -- keeps autograd from carrying the
        # position-independent embedding through every backward kernel when only forces are requested
        w = self.embed_module.weight
        table = w if differentiable_parameters(self.training, w) else w.detach()
        if table.requires_grad:
            # training: one-hot product instead of a row gather -- same values (each row is 1.0 x one table row plus exact
            # zeros); its backward is a [T, N] x [N, F] product instead of embedding_dense_backward's sort + segmented
            # reduction (127 -> ~15 us per step at 8192 atoms / 5 types), and it is differentiable again as is
            # (not F.one_hot: its range check of the indices synchronises with the device)
            kinds = torch.arange(self.num_types, device=atom_types.device).view(1, -1)
            onehot = (atom_types.view(-1, 1) == kinds).to(table.dtype)
            embedding = onehot @ table
        else:
            embedding = torch.nn.functional.embedding(atom_types, table)
        data[AtomicDataDict.NODE_ATTRS_KEY] = embedding
        # node_attrs == table[types]: lets the self-connection contract its weights per type first
        data["_nqa_node_attrs_table"] = table
        if self.set_features:
            data[AtomicDataDict.NODE_FEATURES_KEY] = embedding
        return data
