"""``ForceStressOutput``: forces / virial / stress by autograd of the total energy
(mirror of ``nequip/nn/grad_output.py:107-298``; symmetric-displacement trick for the virial,
``create_graph=self.training`` so that force-matching training can differentiate again)."""

import torch

from ..data import AtomicDataDict
from ._graph_mixin import GraphModuleMixin


class ForceStressOutput(GraphModuleMixin, torch.nn.Module):
    def __init__(self, func: GraphModuleMixin, do_derivatives: bool = True):
        super().__init__()
        self.func = func
        self.do_derivatives = do_derivatives
        self._init_irreps(irreps_in=self.func.irreps_in.copy(), irreps_out=self.func.irreps_out.copy())
        self.irreps_out[AtomicDataDict.FORCE_KEY] = "1o"
        self.irreps_out[AtomicDataDict.STRESS_KEY] = "1o"
        self.irreps_out[AtomicDataDict.VIRIAL_KEY] = "1o"

    def forward(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        if not self.do_derivatives:
            return self.func(data)
        K = AtomicDataDict
        if K.EDGE_VECTORS_KEY in data:
            # LAMMPS ML-IAP branch (grad_output.py:276-296): differentiate w.r.t. the edge vectors
            edge_vectors = data[K.EDGE_VECTORS_KEY]
            edge_vectors.requires_grad_(True)
            data = self.func(data)
            edge_forces = torch.autograd.grad([data[K.TOTAL_ENERGY_KEY].sum()], [edge_vectors])[0]
            data[K.EDGE_FORCE_KEY] = edge_forces
            return data

        if K.BATCH_KEY in data:
            batch = data[K.BATCH_KEY]
            num_batch = K.num_frames(data)
        else:
            batch = None
            num_batch = 1
        pos = data[K.POSITIONS_KEY]
        has_cell = K.CELL_KEY in data
        if has_cell:
            orig_cell = data[K.CELL_KEY]
            cell = orig_cell.view(-1, 3, 3).expand(num_batch, 3, 3)
            data[K.CELL_KEY] = cell
        shape = (num_batch, 3, 3) if num_batch > 1 else (3, 3)
        displacement = torch.zeros(shape, dtype=pos.dtype, device=pos.device)
        displacement.requires_grad_(True)
        data["_displacement"] = displacement
        symmetric_displacement = 0.5 * (displacement + displacement.transpose(-1, -2))
        did_pos_req_grad = pos.requires_grad
        pos.requires_grad_(True)
        if num_batch > 1:
            data[K.POSITIONS_KEY] = pos + torch.bmm(
                pos.unsqueeze(-2), torch.index_select(symmetric_displacement, 0, batch)
            ).squeeze(-2)
        else:
            data[K.POSITIONS_KEY] = pos + torch.sum(pos.view(-1, 3, 1) * symmetric_displacement, 1)
        if has_cell:
            if num_batch > 1:
                data[K.CELL_KEY] = cell + torch.bmm(cell, symmetric_displacement)
            else:
                data[K.CELL_KEY] = (
                    cell.view(3, 3) + torch.sum(cell.view(3, 3, 1) * symmetric_displacement, 1)
                ).view(1, 3, 3)

        data = self.func(data)

        grads = torch.autograd.grad(
            [data[K.TOTAL_ENERGY_KEY].sum()], [pos, data["_displacement"]], create_graph=self.training
        )
        data[K.FORCE_KEY] = torch.neg(grads[0])
        virial = grads[1].view(num_batch, 3, 3)
        if has_cell:
            # |a . (b x c)|, written out (no LAPACK call: keeps the step capturable in a hipGraph)
            volume = torch.sum(cell[:, 0] * torch.linalg.cross(cell[:, 1], cell[:, 2], dim=-1), dim=-1).abs().unsqueeze(-1)
            data[K.STRESS_KEY] = virial / volume.view(num_batch, 1, 1)
            data[K.CELL_KEY] = orig_cell
        data[K.VIRIAL_KEY] = torch.neg(virial)
        del data["_displacement"]
        data[K.POSITIONS_KEY] = pos
        if not did_pos_req_grad:
            pos.requires_grad_(False)
        return data
