"""LAMMPS ML-IAP (unified interface) wrapper for the HIP-backed models.

The object LAMMPS' ``pair_style mliap unified`` drives: it reads ``element_types`` / ``rcutfac`` when the pair style is
set up and calls ``compute_forces(data)`` every step with the rank's local + ghost atoms and its pair list.  Mirrors
``nequip/integrations/lammps_mliap/lmp_mliap_wrapper.py:29-263`` (attributes, lazy initialisation on the first call,
input keys, what is written back and in which sign convention) on top of what this repository already has for that
caller: the edge-vector branch of ``ForceStressOutput`` (``nn/grad_output.py``, reference ``grad_output.py:276-296``), the
local / ghost bookkeeping of ``InteractionBlock`` and the ghost-exchange modules (``nn/_ghost_exchange.py``), and the
per-evaluation scope of the topology cache for callers that refill index buffers in place (``nn/graph_model.py``).

Differences that follow from the design here, not omissions: the model is held as a module (``torch.save(wrapper, path)``
is the file LAMMPS loads, as ``create_lmp_mliap_file.py`` produces for the reference) instead of as package bytes --
checkpoint / package files are outside this repository's scope --, nothing is ``torch.compile``d (the kernels are the
hand-written HIP ones either way), and a CPU LAMMPS build is refused: there is no CPU implementation of the hot path.
"""

from __future__ import annotations

from typing import Optional

import torch

from ..data import AtomicDataDict

try:  # the real base class when LAMMPS' Python package is present
    from lammps.mliap.mliap_unified_abc import MLIAPUnified as _Base
except Exception:  # pragma: no cover  (build / test containers have no LAMMPS)

    class _Base:  # the attribute surface LAMMPS reads (mliap_unified_abc.py)
        def __init__(self, interface=None, element_types=None, ndescriptors=None, nparams=None, rcutfac=None):
            self.interface = interface
            self.element_types = element_types
            self.ndescriptors = ndescriptors
            self.nparams = nparams
            self.rcutfac = rcutfac


class NequIPLAMMPSMLIAPWrapper(_Base):
    """``model``: a ``NequIPGNNModel`` (energy + ``ForceStressOutput``), any device; it is moved to the GPU on the first
    ``compute_forces`` call.  ``device``: override of the device choice (default: ``cuda`` for a Kokkos data object, as the
    reference decides, ``lmp_mliap_wrapper.py:137-140``)."""

    def __init__(self, model: torch.nn.Module, device: Optional[str] = None, sync_inputs: bool = True):
        super().__init__()
        r_max = getattr(model, "r_max", None)
        if r_max is None:
            md = getattr(model, "metadata", None) or {}
            if "r_max" not in md:
                raise ValueError("NequIPLAMMPSMLIAPWrapper needs the model's cutoff: the GraphModel was built without "
                                 "`r_max` (pass it to the model builder so that it lands in `model.metadata`)")
            r_max = float(md["r_max"])
        self.rcutfac = 0.5 * float(r_max)  # LAMMPS multiplies by 2 (`lmp_mliap_wrapper.py:72`)
        self.element_types = list(model.type_names)
        self.nparams = 1
        self.ndescriptors = 1
        self._model = model.eval()
        self._device = device
        self._ready = False
        # the LAMMPS arrays are written by Kokkos kernels on another stream than PyTorch's: without a device
        # synchronisation before they are read, stale pair indices reach the kernels (`lmp_mliap_wrapper.py:181-189`)
        self.sync_inputs = sync_inputs

    # ---- lazy set-up on the first call (the reference's _initialize_model) ----
    def _initialize(self, lmp_data) -> None:
        from ..nn import NoOpGhostExchangeModule

        device = self._device
        if device is None:
            device = "cuda" if "kokkos" in type(lmp_data).__module__.lower() else "cpu"
        if not str(device).startswith("cuda"):
            raise RuntimeError(
                "nequip_amd runs on the GPU only (hand-written HIP kernels, no CPU implementation): LAMMPS has to be built "
                "with Kokkos / HIP so that ML-IAP hands over device arrays")
        model = self._model
        has_exchange = any(hasattr(m, "enable_LAMMPSMLIAPGhostExchange") for m in model.modules())
        if has_exchange:  # fetch the ghosts' features from their owners before every layer but the first
            model = NoOpGhostExchangeModule.enable_LAMMPSMLIAPGhostExchange(model)
        self._model = model.to(device)
        self._device = device
        self._ready = True

    # ---- what LAMMPS calls ----
    def compute_forces(self, lmp_data) -> None:
        if not self._ready:
            self._initialize(lmp_data)
        if lmp_data.nlocal == 0 or lmp_data.npairs <= 1:
            return
        if self.sync_inputs:
            torch.cuda.synchronize()
        K = AtomicDataDict
        dev = self._device
        nlocal, ntotal = int(lmp_data.nlocal), int(lmp_data.ntotal)
        edge_vectors = torch.as_tensor(lmp_data.rij, dtype=torch.float64).to(dev)
        data = {
            K.EDGE_VECTORS_KEY: edge_vectors,
            K.EDGE_INDEX_KEY: torch.vstack([torch.as_tensor(lmp_data.pair_i, dtype=torch.int64).to(dev),
                                            torch.as_tensor(lmp_data.pair_j, dtype=torch.int64).to(dev)]),
            K.ATOM_TYPE_KEY: torch.as_tensor(lmp_data.elems, dtype=torch.int64).to(dev),
            K.LMP_MLIAP_DATA_KEY: lmp_data,
            K.NUM_LOCAL_GHOST_NODES_KEY: torch.tensor([nlocal, ntotal - nlocal], dtype=torch.int64, device=dev),
        }
        # ForceStressOutput takes its edge-vector branch: EDGE_FORCE_KEY = dE / d(edge vector), LAMMPS' sign convention
        out = self._model(data)
        e_atoms = out[K.PER_ATOM_ENERGY_KEY].detach().view(-1)
        if e_atoms.size(0) != nlocal:  # a model that keeps ghost rows to the end: the energy of this rank is its local atoms'
            e_atoms = torch.narrow(e_atoms, 0, 0, nlocal)
            e_total = e_atoms.sum()
        else:
            e_total = out[K.TOTAL_ENERGY_KEY].detach().sum()
        torch.as_tensor(lmp_data.eatoms).copy_(e_atoms)
        lmp_data.energy = e_total
        lmp_data.update_pair_forces_gpu(out[K.EDGE_FORCE_KEY].detach())

    def compute_descriptors(self, lmp_data) -> None:  # (not a descriptor-based potential)
        pass

    def compute_gradients(self, lmp_data) -> None:
        pass


def create_lmp_mliap_file(model: torch.nn.Module, output_path: str, **kwargs) -> str:
    """The file ``pair_style mliap unified <file>`` loads (``create_lmp_mliap_file.py:60-88``: ``torch.save`` of the
    wrapper object)."""
    if not str(output_path).endswith(".nequip.lmp.pt"):
        raise ValueError("the LAMMPS ML-IAP file must be named *.nequip.lmp.pt (the reference's convention)")
    # a CPU COPY is pickled: the caller's (possibly GPU-resident, possibly still in use) model is left untouched, and the
    # copy carries no device-side derived state (`Module.to` does not move cached weight images)
    import copy

    torch.save(NequIPLAMMPSMLIAPWrapper(copy.deepcopy(model).to("cpu"), **kwargs), output_path)
    return str(output_path)
