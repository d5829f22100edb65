"""Ghost-atom exchange for domain-decomposed callers (LAMMPS ML-IAP): mirror of the reference's
``nequip/nn/_ghost_exchange_base.py:9-60`` and ``nequip/nn/_ghost_exchange_lmp_mliap.py:11-64``.

A LAMMPS rank owns ``nlocal`` atoms and sees ``ntotal - nlocal`` ghost copies of atoms owned elsewhere (periodic images or
other ranks' atoms).  Every convolution after the first works on the local rows only (``interaction_block.py:166-168``),
so before its tensor product the ghosts' features have to be fetched from their owners; the reverse exchange sums the
ghosts' gradients back into the owners' rows.  The exchange itself is LAMMPS's (``lmp_data.forward_exchange`` /
``reverse_exchange``, device pointers under Kokkos); these modules only prepare the ``[ntotal, D]`` buffer and put the two
calls on the autograd tape.  The default module is a no-op; ``enable_LAMMPSMLIAPGhostExchange`` swaps the real one in,
exactly like the reference's private modifier of the same name.
"""

import torch

from ..data import AtomicDataDict
from ._graph_mixin import GraphModuleMixin
from .model_modifier_utils import model_modifier, replace_submodules


class GhostExchangeModule(GraphModuleMixin, torch.nn.Module):
    """Interface: ``forward(data, ghost_included)`` returns ``data`` with ``data[field]`` holding ``ntotal`` rows whose
    ghost rows carry their owners' values."""

    def __init__(self, field: str = AtomicDataDict.NODE_FEATURES_KEY, irreps_in={}):
        super().__init__()
        self.field = field
        self._init_irreps(irreps_in=irreps_in, my_irreps_in={field: irreps_in[field]},
                          irreps_out={field: irreps_in[field]})

    def forward(self, data: AtomicDataDict.Type, ghost_included: bool) -> AtomicDataDict.Type:
        raise NotImplementedError


class NoOpGhostExchangeModule(GhostExchangeModule):
    """Single-domain evaluation: every neighbour is a local atom, nothing to fetch."""

    def forward(self, data: AtomicDataDict.Type, ghost_included: bool) -> AtomicDataDict.Type:
        return data

    @model_modifier(persistent=True, private=True)
    @classmethod
    def enable_LAMMPSMLIAPGhostExchange(cls, model):
        """Enable LAMMPS ML-IAP ghost exchange for inference in LAMMPS ML-IAP."""
        return replace_submodules(
            model, cls, lambda old: LAMMPSMLIAPGhostExchangeModule(field=old.field, irreps_in=old.irreps_in)
        )


class _LAMMPSExchangeFn(torch.autograd.Function):
    """owners -> ghosts on the way forward, ghosts' gradients -> owners on the way back."""

    @staticmethod
    def forward(ctx, features, lmp_data):
        flat = features.reshape(features.size(0), -1).contiguous()
        out = torch.empty_like(flat)
        lmp_data.forward_exchange(flat, out, out.size(-1))
        ctx.lmp_data, ctx.shape = lmp_data, features.shape
        return out.view(features.shape)

    @staticmethod
    def backward(ctx, grad):
        flat = grad.reshape(grad.size(0), -1).contiguous()
        out = torch.empty_like(flat)
        ctx.lmp_data.reverse_exchange(flat, out, out.size(-1))
        return out.view(ctx.shape), None


class LAMMPSMLIAPGhostExchangeModule(GhostExchangeModule):
    def forward(self, data: AtomicDataDict.Type, ghost_included: bool = False) -> AtomicDataDict.Type:
        if AtomicDataDict.LMP_MLIAP_DATA_KEY not in data:
            raise RuntimeError("LAMMPSMLIAPGhostExchangeModule needs the LAMMPS ML-IAP data object in the input dict")
        lmp_data = data[AtomicDataDict.LMP_MLIAP_DATA_KEY]
        feats = data[self.field]
        local = feats[: lmp_data.nlocal] if ghost_included else feats
        n_ghost = lmp_data.ntotal - lmp_data.nlocal
        # local rows followed by empty ghost rows: the exchange fills the ghosts (zeros, not `empty`: the reverse
        # exchange of the zero rows' gradients must not see garbage)
        padded = torch.cat((local, local.new_zeros((n_ghost,) + tuple(local.shape[1:]))), dim=0)
        data[self.field] = _LAMMPSExchangeFn.apply(padded, lmp_data)
        return data
