"""One molecular-dynamics force evaluation -- positions -> neighbour list -> reverse-edge pairing -> model -- as ONE hipGraph.

The reference's drivers rebuild the neighbour list on the host at every step and launch the model's kernels one by one
(``nequip/integrations/ase.py:125-160``: ``from_ase`` -> ``compute_neighborlist_`` (``nequip/data/_nl.py:63-165``, a CPU
library) -> ``model(data)``).  Here the list is built on the device into a fixed number of edge slots
(``nequip_amd.data._nl.PaddedNeighborList``: the unused slots hold edges beyond the cutoff, which carry no interaction), the
pairing verdict is left on the device, and therefore every launch of the step has static shapes and no host read-back: the
step is captured once and replayed with new positions.  After a replay the host reads two flags together with the results:

* the list did not fit its capacity  -> the step is captured again with more slots and repeated;
* the list did not pair up (cannot happen for lists built here, which are symmetric by construction; kept as a guard)
  -> this step is evaluated through the ordinary eager path.

There is no CPU path: the model, the cell and the atom types live on the GPU.
"""

from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple, Union

import torch

from ..data import AtomicDataDict
from ..data._nl import PaddedNeighborList, compute_neighborlist_, compute_neighborlist_padded_
from ..nn._topology import topology_cache


class GraphedStep:
    """``step(pos) -> {total_energy, forces, ...}`` for a fixed set of atoms in a (possibly changing) cell.

    ``model``: an eval-mode ``GraphModel`` on the GPU (float32); ``atom_types`` int64 ``[N]``; ``cell`` ``[3, 3]`` (rows =
    lattice vectors); ``pbc`` three flags; ``r_max`` the model's cutoff.  ``headroom``: slots per edge of the first list
    (the capacity grows by the same factor whenever a list does not fit).  ``outputs``: the fields of the model's output to
    keep (static tensors, overwritten by the next step: copy what must survive).  Positions are float64 by default, as the
    reference's data is (``nequip/data/_key_registry`` / ASE): the model casts where it computes."""

    def __init__(self, model: torch.nn.Module, atom_types: torch.Tensor, cell: torch.Tensor,
                 pbc: Union[bool, Sequence[bool], torch.Tensor], r_max: float, headroom: float = 1.02,
                 outputs: Sequence[str] = (AtomicDataDict.TOTAL_ENERGY_KEY, AtomicDataDict.FORCE_KEY),
                 pos_dtype: torch.dtype = torch.float64, edge_capacity: Optional[int] = None):
        if not atom_types.is_cuda:
            raise RuntimeError("GraphedStep runs on the GPU only (HIP kernels; there is no CPU path)")
        assert not model.training, "call .eval() on the model before building a GraphedStep"
        if headroom < 1.0:
            raise ValueError("headroom must be >= 1")
        self.model = model
        self.device = atom_types.device
        self.num_atoms = int(atom_types.numel())
        self.r_max = float(r_max)
        self.headroom = float(headroom)
        self.outputs = tuple(outputs)
        self._types = atom_types.detach().view(-1).to(torch.int64).contiguous()
        self._cell = cell.detach().reshape(1, 3, 3).to(device=self.device).clone()
        if isinstance(pbc, bool):
            pbc = (pbc,) * 3
        elif isinstance(pbc, torch.Tensor):
            pbc = pbc.detach().cpu().view(-1).tolist()
        self._pbc = tuple(bool(b) for b in pbc)
        self._pbc_t = torch.tensor([list(self._pbc)], device=self.device)
        self._pos = torch.zeros(self.num_atoms, 3, dtype=pos_dtype, device=self.device)
        self._capacity = edge_capacity
        self._side = torch.cuda.Stream(self.device)
        # (the evaluations that pair at all: float32 models, see GraphModel._start_pairing)
        self._pairs = getattr(model, "model_dtype", torch.float32) == torch.float32
        self._nl: Optional[PaddedNeighborList] = None
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._out: Dict[str, torch.Tensor] = {}
        self._flags: Optional[torch.Tensor] = None
        self._topo = None
        self.num_captures = 0
        self.num_eager_fallbacks = 0
        self.last_num_edges = 0

    # ---- pieces ----------------------------------------------------------------------------------------------------
    def _data(self) -> AtomicDataDict.Type:
        K = AtomicDataDict
        return {K.POSITIONS_KEY: self._pos, K.ATOM_TYPE_KEY: self._types, K.CELL_KEY: self._cell, K.PBC_KEY: self._pbc_t}

    def _evaluate_padded(self) -> Tuple[Dict[str, torch.Tensor], torch.Tensor]:
        """The step's launches on the current stream (eager or under capture): ``(outputs, flags)`` with
        ``flags = [list did not fit, E, list paired up]`` on the device."""
        K = AtomicDataDict
        data = compute_neighborlist_padded_(self._data(), self._nl)
        ei = data[K.EDGE_INDEX_KEY]
        topo = topology_cache.get(ei[0], ei[1], self.num_atoms)
        topo.defer_pairing_verdict = True
        if self._pairs:
            # the lists of the backward pass are built next to the forward pass, not in front of their first consumer
            topo.prefetch_backward_lists(data.get(K.EDGE_CELL_SHIFT_KEY), self._side)
        out = self.model(data)
        torch.cuda.current_stream(self.device).wait_stream(self._side)
        ok = topo.pairing_ok
        if ok is None:  # (a model that does not pair -- float64, NQA_NO_PAIRED: nothing to verify)
            ok = torch.ones(1, dtype=torch.int32, device=self.device)
        flags = torch.cat([self._nl.status_tensor, ok])
        self._topo = topo
        # nothing of the autograd graph must outlive the step (or be alive during capture)
        return {k: out[k].detach() for k in self.outputs if k in out and out[k] is not None}, flags

    def _evaluate_eager(self) -> Dict[str, torch.Tensor]:
        data = compute_neighborlist_(self._data(), self.r_max)
        out = self.model(data)
        return {k: out[k].detach() for k in self.outputs if k in out and out[k] is not None}

    def _first_capacity(self) -> int:
        K = AtomicDataDict
        data = compute_neighborlist_(self._data(), self.r_max)
        return int(data[K.EDGE_INDEX_KEY].shape[1])

    def _capture(self) -> None:
        """(Re)build the neighbour-list buffers for the current capacity and capture the step."""
        if self._topo is not None:
            topology_cache.forget(self._topo)
        self._graph = None
        self._out, self._flags, self._topo = {}, None, None
        self._nl = PaddedNeighborList(self.num_atoms, self.r_max, self._cell[0], self._pbc, self._capacity,
                                      shift_dtype=self._pos.dtype)
        self._capacity = self._nl.edge_capacity
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):  # warm-up off the default stream: allocator pools, lazily built tables
            for _ in range(2):
                self._evaluate_padded()
                topology_cache.forget(self._topo)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._out, self._flags = self._evaluate_padded()
        # the captured topology lives in the graph's memory pool and belongs to this object, not to the cross-call cache
        topology_cache.forget(self._topo)
        self._graph = graph
        self.num_captures += 1

    # ---- interface -------------------------------------------------------------------------------------------------
    def set_cell(self, cell: torch.Tensor) -> None:
        """New lattice vectors (variable-cell dynamics); takes effect at the next step, no new capture."""
        self._cell.copy_(cell.detach().reshape(1, 3, 3))
        if self._nl is not None:
            self._nl.set_cell(self._cell[0])

    @property
    def edge_capacity(self) -> Optional[int]:
        return self._capacity

    def __call__(self, pos: torch.Tensor) -> Dict[str, torch.Tensor]:
        self._pos.copy_(pos.detach().reshape(self.num_atoms, 3))
        if self._graph is None:
            if self._capacity is None:
                self._capacity = int(self._first_capacity() * self.headroom) + 2
            self._capture()
        for _ in range(4):
            self._graph.replay()
            bad, num_edges, paired = self._flags.cpu().tolist()  # (the step's one synchronisation, with its results)
            self.last_num_edges = int(num_edges)
            if bad:
                # E > capacity (or an odd number of free slots): more slots, new capture, same positions
                self._capacity = max(int(num_edges * self.headroom) + 2, self._capacity + 2)
                self._capture()
                continue
            if paired:
                return self._out
            break
        # a list that does not pair up (or that no capacity fits: an odd edge count, i.e. not symmetric): the ordinary path
        self.num_eager_fallbacks += 1
        return self._evaluate_eager()
