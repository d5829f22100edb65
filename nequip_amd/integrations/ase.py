"""ASE calculator on top of the MI355X path (mirror of ``nequip/integrations/ase.py:13-160``).

``NequIPCalculator(model, device, r_max, chemical_symbols)`` follows the reference's calculator: same constructor
meaning (an eval-mode model, a device, unit conversion factors), the same ``calculate`` results (``energy``,
``free_energy``, ``energies``, ``forces``, ``stress`` in ASE's Voigt order and units) and the same
``atoms_to_data`` / ``call_model`` / ``save_extra_outputs`` hooks.  What differs is the data path before the model:
where the reference converts the ``Atoms`` on the host and (with its default neighbour-list backends) builds the graph on
the CPU before the H2D copy, this calculator ships positions / cell / numbers once and builds the neighbour list on the
GPU (``nequip_amd.data.compute_neighborlist_``, ``csrc/neighbor_list.hip``); the list comes out grouped by centre atom,
so the tensor-product kernels take its row pointer as their CSR without sorting.

ASE itself is optional: with ``ase`` installed the class derives from ``ase.calculators.calculator.Calculator`` and can
be attached to ``Atoms`` as usual; without it (the build container has no ``ase``) a minimal stand-in base class keeps
``calculate(atoms)`` / ``get_potential_energy`` / ``get_forces`` / ``get_stress`` working for any object with the
``Atoms`` accessors used here (``get_positions``, ``get_cell``, ``get_pbc``, ``get_chemical_symbols``).
"""

from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from ..data import AtomicDataDict
from ..data._nl import compute_neighborlist_

try:  # pragma: no cover - ase is not installed in the build container
    from ase.calculators.calculator import Calculator, all_changes

    HAVE_ASE = True
except Exception:  # noqa: BLE001
    HAVE_ASE = False
    all_changes = ["positions", "numbers", "cell", "pbc", "initial_charges", "initial_magmoms"]

    class Calculator:  # minimal stand-in with the part of ASE's interface the class below relies on
        implemented_properties: List[str] = []

        def __init__(self, **kwargs):
            self.atoms = None
            self.results: Dict[str, np.ndarray] = {}

        def calculate(self, atoms=None, properties=("energy",), system_changes=all_changes):
            if atoms is not None:
                self.atoms = atoms

        def get_property(self, name: str, atoms=None):
            if name not in self.implemented_properties:
                raise NotImplementedError(f"{name} property not implemented")
            self.calculate(atoms, [name], all_changes)
            if name not in self.results:
                raise NotImplementedError(f"{name} not present in this calculation")
            return self.results[name]

        def get_potential_energy(self, atoms=None):
            return self.get_property("energy", atoms)

        def get_forces(self, atoms=None):
            return self.get_property("forces", atoms)

        def get_stress(self, atoms=None):
            return self.get_property("stress", atoms)


def full_3x3_to_voigt_6_stress(stress: np.ndarray) -> np.ndarray:
    """ASE's Voigt order (xx, yy, zz, yz, xz, xy), off-diagonals symmetrised (``ase.stress``)."""
    s = np.asarray(stress).reshape(3, 3)
    return np.array([s[0, 0], s[1, 1], s[2, 2], 0.5 * (s[1, 2] + s[2, 1]), 0.5 * (s[0, 2] + s[2, 0]),
                     0.5 * (s[0, 1] + s[1, 0])])


class NequIPCalculator(Calculator):
    """Energy / forces / stress of an ``Atoms`` object through the HIP kernels (one GPU, one frame per call)."""

    implemented_properties = ["energy", "energies", "forces", "stress", "free_energy"]

    def __init__(
        self,
        model: torch.nn.Module,
        device: Union[str, torch.device],
        r_max: float,
        chemical_symbols: Optional[Union[Sequence[str], Dict[str, str]]] = None,
        energy_units_to_eV: float = 1.0,
        length_units_to_A: float = 1.0,
        transforms: Sequence[Callable] = (),
        graphed_md: bool = False,
        graphed_md_headroom: float = 1.02,
        **kwargs,
    ):
        """``graphed_md=True``: successive calls on the same atoms (same species, same periodicity -- a molecular-dynamics
        run) replay positions -> neighbour list -> model as one hipGraph (``integrations/graphed_step.py``) instead of
        launching ~110 kernels per step from Python; any change of the atoms' identity captures anew, a changed cell is
        written in place.  Needs a cell, an eager ``nequip_amd`` model (not a compiled package) and no ``transforms``."""
        Calculator.__init__(self, **kwargs)
        self.results = {}
        assert not model.training, "make sure to call .eval() on model before building NequIPCalculator"
        self.device = torch.device(device) if isinstance(device, str) else device
        if self.device.type != "cuda":
            raise RuntimeError("nequip_amd's calculator runs on the GPU only (HIP kernels; there is no CPU path)")
        self.model = model.to(self.device)
        self.r_max = float(r_max)
        self.energy_units_to_eV = energy_units_to_eV
        self.length_units_to_A = length_units_to_A
        self.transforms = list(transforms)
        self.graphed_md = bool(graphed_md)
        self.graphed_md_headroom = float(graphed_md_headroom)
        self._graphed = None  # (identity of the atoms, GraphedStep, cell as last written)
        if self.graphed_md and self.transforms:
            raise ValueError("graphed_md replays a fixed pipeline: it cannot run user transforms")
        if self.graphed_md and type(model).__name__ == "DictInputOutputWrapper":
            raise ValueError("graphed_md needs the eager nequip_amd model: a compiled package's graph is fixed at export time")
        # chemical symbol -> atom type index (`ChemicalSpeciesToAtomTypeMapper`, nequip/data/transforms): a list means
        # "type_names are chemical symbols in this order", a dict maps symbol -> type name
        type_names = list(getattr(model, "type_names", []) or [])
        if chemical_symbols is None:
            chemical_symbols = type_names
        if isinstance(chemical_symbols, dict):
            self._type_of_symbol = {sym: type_names.index(name) for sym, name in chemical_symbols.items()}
        else:
            self._type_of_symbol = {sym: i for i, sym in enumerate(chemical_symbols)}
        if not self._type_of_symbol:
            raise ValueError("no chemical species mapping: pass chemical_symbols or a model with type_names")

    @classmethod
    def from_compiled_model(cls, compile_path: str, device: Union[str, torch.device] = "cuda",
                            chemical_symbols: Optional[Union[Sequence[str], Dict[str, str]]] = None, **kwargs):
        """Calculator around an AOTInductor package made by ``nequip_amd.utils.aot.aot_export_model`` (the reference's
        recommended route: ``NequIPCalculator.from_compiled_model`` on a ``nequip-compile --target ase`` artefact,
        nequip/integrations/ase.py:16-19).  Cutoff and type names come from the package metadata."""
        from ..utils.aot import load_aotinductor_model

        model, metadata = load_aotinductor_model(str(compile_path), device=device)
        model = model.eval()
        model.type_names = metadata["type_names"].split()
        return cls(model, device=device, r_max=float(metadata["r_max"]), chemical_symbols=chemical_symbols, **kwargs)

    # ---- data ----------------------------------------------------------------------------------------------------
    def atoms_to_data(self, atoms) -> AtomicDataDict.Type:
        """``from_ase`` + species mapping + neighbour list (``nequip/integrations/ase.py:142-150``), on the device."""
        K = AtomicDataDict
        symbols = atoms.get_chemical_symbols()
        try:
            types = np.fromiter((self._type_of_symbol[s] for s in symbols), dtype=np.int64, count=len(symbols))
        except KeyError as e:
            raise ValueError(f"chemical species {e.args[0]!r} is not among the model's types "
                             f"{sorted(self._type_of_symbol)}") from None
        pbc = np.asarray(atoms.get_pbc(), dtype=bool).reshape(3)
        data = {
            K.POSITIONS_KEY: torch.as_tensor(np.asarray(atoms.get_positions(), dtype=np.float64)).to(self.device),
            K.ATOM_TYPE_KEY: torch.as_tensor(types).to(self.device),
        }
        if pbc.any():
            cell = np.asarray(atoms.get_cell(), dtype=np.float64).reshape(3, 3)
            data[K.CELL_KEY] = torch.as_tensor(cell).view(1, 3, 3).to(self.device)
            data[K.PBC_KEY] = torch.as_tensor(pbc).view(1, 3).to(self.device)
        for t in self.transforms:
            data = t(data)
        if K.EDGE_INDEX_KEY not in data:
            compute_neighborlist_(data, self.r_max)
        return data

    def call_model(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        return self.model(data)

    def _graphed_outputs(self, atoms) -> Optional[AtomicDataDict.Type]:
        """The replayed step for ``atoms`` (``None``: not applicable -- no periodic direction, i.e. no cell to pad with)."""
        from .graphed_step import GraphedStep

        K = AtomicDataDict
        pbc = tuple(bool(b) for b in np.asarray(atoms.get_pbc(), dtype=bool).reshape(3))
        if not any(pbc):
            return None
        symbols = tuple(atoms.get_chemical_symbols())
        cell = np.asarray(atoms.get_cell(), dtype=np.float64).reshape(3, 3)
        ident = (symbols, pbc)
        if self._graphed is None or self._graphed[0] != ident:
            try:
                types = np.fromiter((self._type_of_symbol[s] for s in symbols), dtype=np.int64, count=len(symbols))
            except KeyError as e:
                raise ValueError(f"chemical species {e.args[0]!r} is not among the model's types "
                                 f"{sorted(self._type_of_symbol)}") from None
            step = GraphedStep(
                self.model, torch.as_tensor(types).to(self.device), torch.as_tensor(cell).to(self.device), pbc, self.r_max,
                headroom=self.graphed_md_headroom,
                outputs=(K.TOTAL_ENERGY_KEY, K.PER_ATOM_ENERGY_KEY, K.FORCE_KEY, K.STRESS_KEY, K.VIRIAL_KEY))
            self._graphed = (ident, step, cell.copy())
        _, step, last_cell = self._graphed
        if not np.array_equal(cell, last_cell):
            step.set_cell(torch.as_tensor(cell).to(self.device))
            self._graphed = (ident, step, cell.copy())
        pos = torch.as_tensor(np.asarray(atoms.get_positions(), dtype=np.float64)).to(self.device)
        return step(pos)

    def save_extra_outputs(self, out: AtomicDataDict.Type) -> None:
        """Hook for subclasses (as in the reference)."""

    # ---- ASE interface ---------------------------------------------------------------------------------------------
    def calculate(self, atoms=None, properties=("energy",), system_changes=all_changes):
        Calculator.calculate(self, atoms)
        atoms = atoms if atoms is not None else self.atoms
        K = AtomicDataDict
        out = self._graphed_outputs(atoms) if self.graphed_md else None
        if out is None:
            out = self.call_model(self.atoms_to_data(atoms))
        self.results = {}
        e2ev, l2a = self.energy_units_to_eV, self.length_units_to_A
        if K.TOTAL_ENERGY_KEY in out:
            self.results["energy"] = e2ev * out[K.TOTAL_ENERGY_KEY].detach().cpu().numpy().reshape(tuple())
            self.results["free_energy"] = self.results["energy"]
        if K.PER_ATOM_ENERGY_KEY in out:
            self.results["energies"] = e2ev * out[K.PER_ATOM_ENERGY_KEY].detach().squeeze(-1).cpu().numpy()
        if K.FORCE_KEY in out:
            self.results["forces"] = (e2ev / l2a) * out[K.FORCE_KEY].detach().cpu().numpy()
        if K.STRESS_KEY in out and out[K.STRESS_KEY] is not None and out[K.STRESS_KEY].numel() == 9:
            stress = out[K.STRESS_KEY].detach().cpu().numpy().reshape(3, 3) * (e2ev / l2a**3)
            self.results["stress"] = full_3x3_to_voigt_6_stress(stress)
        self.save_extra_outputs(out)
