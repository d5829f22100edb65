"""Neighbour lists on the device (mirror of ``nequip/data/_nl.py:63-381`` for one backend, ``"nequip_amd"``).

``compute_neighborlist_(data, r_max)`` keeps the reference's contract (``_nl.py:364-381``): it adds ``edge_index``
(int64 ``[2, E]``, row 0 = convolution centre, row 1 = neighbour) and -- iff the data has a cell -- ``edge_cell_shift``
(``[E, 3]``, dtype of the positions) to ``data`` in place; batched input gives batched output, unbatched gives
unbatched, everything stays on the device of the positions.  Where the reference moves the positions to the host and
calls matscipy / ASE / vesin, this backend runs a cell-list search on the GPU (``nqa_neighbor_list_count/fill``,
``nequip_amd/csrc/neighbor_list.hip``) with the same pair semantics (``|r| < r_max``, no self pair with zero shift,
mixed periodicity, cells thinner than the cutoff).  Edges come out grouped by centre atom, so the dst-CSR of the
tensor-product kernels needs no sort for them.
"""

import ctypes
from typing import Dict, Final, Optional, Tuple, Union

import torch

from .. import _lib
from . import AtomicDataDict

NEIGHBORLIST_BACKEND_NEQUIP_AMD: Final[str] = "nequip_amd"
DEFAULT_NEIGHBORLIST_BACKEND: Final[str] = NEIGHBORLIST_BACKEND_NEQUIP_AMD


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()


def _complete_cell(cell64: torch.Tensor, pbc: Tuple[bool, bool, bool]) -> torch.Tensor:
    """Cells with zero lattice vectors along non-periodic directions (ASE slabs / wires / molecules with pbc = (T, T, F)
    and c = 0): complete them with unit vectors orthogonal to the span of the others, as ``ase.geometry.complete_cell``
    does and the reference's backends accept -- the kernel inverts the cell.  A zero (or linearly dependent) vector
    along a *periodic* direction is an error.

    The check reads the nine numbers on the host: one small copy per call, next to the one synchronisation the
    neighbour list needs anyway (the data-dependent edge count).  Deliberately not memoised on the tensor's storage:
    a recycled allocation would look like the same cell."""
    return _complete_cell_host(cell64, pbc)


def _complete_cell_host(cell64: torch.Tensor, pbc: Tuple[bool, bool, bool]) -> torch.Tensor:
    import numpy as np

    c = cell64.detach().cpu().numpy().copy()
    norms = np.linalg.norm(c, axis=1)
    missing = [i for i in range(3) if norms[i] < 1e-12]
    if not missing:
        if abs(np.linalg.det(c)) < 1e-12 * max(1.0, norms.prod()):
            raise ValueError("cell vectors are linearly dependent")
        return cell64
    for i in missing:
        if pbc[i]:
            raise ValueError(f"lattice vector {i} is zero but the direction is periodic")
    # orthonormal basis of the span of the present vectors (Gram-Schmidt), then unit vectors orthogonal to it
    basis = []
    for i in range(3):
        if i in missing:
            continue
        v = c[i].copy()
        for b in basis:
            v -= np.dot(v, b) * b
        n = np.linalg.norm(v)
        if n < 1e-12 * norms[i]:
            raise ValueError("cell vectors are linearly dependent")
        basis.append(v / n)
    for i in missing:
        if len(basis) == 2:
            v = np.cross(basis[0], basis[1])
        else:
            # the coordinate axis with the largest component orthogonal to the basis so far
            best = None
            for axis in np.eye(3):
                w = axis.copy()
                for b in basis:
                    w -= np.dot(w, b) * b
                n = np.linalg.norm(w)
                if best is None or n > best[0]:
                    best = (n, w)
            v = best[1]
        v = v / np.linalg.norm(v)
        c[i] = v
        basis.append(v)
    if abs(np.linalg.det(c)) < 1e-12:
        raise ValueError("cell vectors are linearly dependent")
    return torch.as_tensor(c, dtype=torch.float64, device=cell64.device).contiguous()


def _compute_neighborlist_single_frame(
    pos: torch.Tensor,
    r_max: float,
    cell: Optional[torch.Tensor] = None,
    pbc: Union[bool, Tuple[bool, bool, bool], torch.Tensor] = False,
    return_rowptr: bool = False,
):
    """``(edge_index [2, E] int64, edge_cell_shift [E, 3])`` of one frame (``nequip/data/_nl.py:63-165``)."""
    if not pos.is_cuda:
        raise RuntimeError("the `nequip_amd` neighbour list runs on the GPU: positions must be a CUDA/HIP tensor")
    if isinstance(pbc, bool):
        pbc = (pbc,) * 3
    elif isinstance(pbc, torch.Tensor):
        pbc = tuple(bool(b) for b in pbc.detach().cpu().view(-1).tolist())
    if cell is None and any(pbc):
        raise ValueError("Periodic boundary conditions requested but no cell was provided.")
    lib = _lib.load()
    device = pos.device
    out_dtype = pos.dtype
    N = pos.shape[0]
    pos64 = pos.detach().to(torch.float64).contiguous()
    cell64 = cell.detach().to(torch.float64).reshape(3, 3).contiguous() if cell is not None else None
    if cell64 is not None:
        cell64 = _complete_cell(cell64, pbc)
    pbc_dev = torch.tensor([int(b) for b in pbc], dtype=torch.int32, device=device)
    ws_bytes = lib.nqa_neighbor_list_workspace_bytes(N)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=device)
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    with torch.cuda.device(device):
        rc = lib.nqa_neighbor_list_count(_ptr(pos64), _ptr(cell64), _ptr(pbc_dev), float(r_max), N, _ptr(ws), ws_bytes,
                                         _ptr(rowptr), stream)
        _lib.check(rc, "nqa_neighbor_list_count")
        E = int(rowptr[N].item())  # the one synchronisation: the edge count is data dependent
        if E < 0:
            raise RuntimeError("neighbour list has more than 2^31 - 1 edges")
        edge_index = torch.empty((2, E), dtype=torch.int64, device=device)
        shifts = torch.empty((E, 3), dtype=torch.float64, device=device)
        rc = lib.nqa_neighbor_list_fill(_ptr(ws), _ptr(rowptr), N, E, _ptr(edge_index), _ptr(shifts), stream)
        _lib.check(rc, "nqa_neighbor_list_fill")
    if return_rowptr:
        return edge_index, shifts.to(out_dtype), rowptr
    return edge_index, shifts.to(out_dtype)


class PaddedNeighborList:
    """Neighbour list of ONE frame with a fixed number of edge slots and no host read-back (``nqa_neighbor_list_count`` +
    ``nqa_neighbor_list_fill_padded``): every launch of ``build`` goes to the current stream and every shape is static, so
    ``positions -> neighbour list -> pairing -> model`` can be captured in one hipGraph and replayed (molecular dynamics;
    the reference builds the list on the host at every step, ``nequip/integrations/ase.py:125-160`` ->
    ``nequip/data/_nl.py:63-165``).

    The ``edge_capacity - E`` unused slots hold padding edges: self-image pairs ``(i <- i, +-S)`` longer than ``r_max``, dealt
    out evenly over the atoms.  They lie outside the polynomial cutoff, so their radial embedding, their (bias-free) radial-MLP
    weights and all derivatives vanish: energies, forces and virials are those of the unpadded list.  Needs a cell.

    ``status()`` (a synchronising read, for AFTER the step) reports ``(fits, E)``: when the list did not fit, the output of that
    ``build`` holds padding only and the caller repeats the step with a larger capacity."""

    def __init__(self, num_atoms: int, r_max: float, cell: torch.Tensor,
                 pbc: Union[bool, Tuple[bool, bool, bool], torch.Tensor], edge_capacity: int,
                 shift_dtype: torch.dtype = torch.float32):
        if cell is None:
            raise ValueError("a capacity-padded neighbour list needs a cell (its padding edges are lattice images)")
        if not cell.is_cuda:
            raise RuntimeError("the `nequip_amd` neighbour list runs on the GPU: the cell must be a CUDA/HIP tensor")
        if isinstance(pbc, bool):
            pbc = (pbc,) * 3
        elif isinstance(pbc, torch.Tensor):
            pbc = tuple(bool(b) for b in pbc.detach().cpu().view(-1).tolist())
        self.num_atoms = int(num_atoms)
        if self.num_atoms < 1:
            raise ValueError("a capacity-padded neighbour list needs at least one atom")
        self.r_max = float(r_max)
        self.edge_capacity = int(edge_capacity) + (int(edge_capacity) & 1)  # (pairs of edges)
        if not 0 <= self.edge_capacity < 2**31 - 1:
            raise ValueError("edge capacity outside the int32 index range of the kernels")
        self.shift_dtype = shift_dtype
        self.device = cell.device
        self.pbc = tuple(pbc)
        lib = _lib.load()
        self.cell64 = torch.empty(3, 3, dtype=torch.float64, device=self.device)
        self.set_cell(cell)
        self._pbc_dev = torch.tensor([int(b) for b in pbc], dtype=torch.int32, device=self.device)
        self._ws_bytes = lib.nqa_neighbor_list_workspace_bytes(self.num_atoms)
        self._ws = torch.empty(max(self._ws_bytes, 1), dtype=torch.uint8, device=self.device)
        self._rowptr = torch.empty(self.num_atoms + 1, dtype=torch.int32, device=self.device)
        self._status = torch.zeros(2, dtype=torch.int32, device=self.device)
        self._edge_ids = torch.arange(max(self.edge_capacity, 1), dtype=torch.int32, device=self.device)
        self.last_csr = None

    def set_cell(self, cell: torch.Tensor) -> None:
        """New lattice vectors (variable-cell dynamics): written into the buffer the captured launches read.  Reads the cell on
        the host once (completion of missing lattice vectors, as ``_compute_neighborlist_single_frame`` does)."""
        c = _complete_cell(cell.detach().to(torch.float64).reshape(3, 3).contiguous(), self.pbc)
        self.cell64.copy_(c)

    def build(self, pos: torch.Tensor):
        """``(edge_index [2, capacity] int64, edge_cell_shift [capacity, 3], rowptr [N + 1] int32)`` for ``pos``; also leaves
        the fit flag in ``status_tensor`` (device).  No synchronisation."""
        if not pos.is_cuda or pos.shape != (self.num_atoms, 3):
            raise RuntimeError(f"positions must be a CUDA/HIP tensor of shape ({self.num_atoms}, 3)")
        lib = _lib.load()
        dev = self.device
        pos64 = pos.detach().to(torch.float64).contiguous()
        cap = self.edge_capacity
        edge_index = torch.empty((2, cap), dtype=torch.int64, device=dev)
        shifts = torch.empty((cap, 3), dtype=torch.float64, device=dev)
        rowptr = torch.empty(self.num_atoms + 1, dtype=torch.int32, device=dev)
        src32 = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            rc = lib.nqa_neighbor_list_count(_ptr(pos64), _ptr(self.cell64), _ptr(self._pbc_dev), self.r_max,
                                             self.num_atoms, _ptr(self._ws), self._ws_bytes, _ptr(self._rowptr), stream)
            _lib.check(rc, "nqa_neighbor_list_count")
            rc = lib.nqa_neighbor_list_fill_padded(_ptr(self._ws), _ptr(self._rowptr), self.num_atoms, cap, _ptr(rowptr),
                                                   _ptr(edge_index), _ptr(shifts), _ptr(src32), _ptr(self._status), stream)
            _lib.check(rc, "nqa_neighbor_list_fill_padded")
        self.last_csr = (rowptr, self._edge_ids, src32)  # the dst-CSR of the list just built (edge ids = 0 .. capacity - 1)
        return edge_index, shifts.to(self.shift_dtype), rowptr

    @property
    def status_tensor(self) -> torch.Tensor:
        """int32[2] on the device: ``[did not fit, E]`` of the last ``build``."""
        return self._status

    def status(self) -> Tuple[bool, int]:
        """``(fits, E)`` of the last ``build`` (synchronises)."""
        bad, e = self._status.cpu().tolist()
        return bad == 0, int(e)


def compute_neighborlist_padded_(data: AtomicDataDict.Type, nl: PaddedNeighborList) -> AtomicDataDict.Type:
    """``compute_neighborlist_`` for one unbatched frame through a ``PaddedNeighborList``: adds ``edge_index`` and
    ``edge_cell_shift`` (both ``nl.edge_capacity`` long) in place and hands the row pointer to the topology cache."""
    K = AtomicDataDict
    if K.BATCH_KEY in data and K.num_frames(data) != 1:
        raise ValueError("a capacity-padded neighbour list holds one frame")
    edge_index, shifts, rowptr = nl.build(data[K.POSITIONS_KEY])
    data[K.EDGE_INDEX_KEY] = edge_index
    data[K.EDGE_CELL_SHIFT_KEY] = shifts
    from ..nn._topology import topology_cache

    topology_cache.hint_sorted(edge_index, rowptr, csr=nl.last_csr)
    return data


def _frame_from_batched(data: AtomicDataDict.Type, idx: int, node_offsets) -> AtomicDataDict.Type:
    K = AtomicDataDict
    lo, hi = int(node_offsets[idx]), int(node_offsets[idx + 1])
    out = {K.POSITIONS_KEY: data[K.POSITIONS_KEY][lo:hi]}
    if K.ATOM_TYPE_KEY in data:
        out[K.ATOM_TYPE_KEY] = data[K.ATOM_TYPE_KEY].view(-1)[lo:hi]
    if K.CELL_KEY in data:
        out[K.CELL_KEY] = data[K.CELL_KEY].view(-1, 3, 3)[idx]
    if K.PBC_KEY in data:
        out[K.PBC_KEY] = data[K.PBC_KEY].view(-1, 3)[idx]
    return out


def compute_neighborlist_(data: AtomicDataDict.Type, r_max: float,
                          backend: str = DEFAULT_NEIGHBORLIST_BACKEND) -> AtomicDataDict.Type:
    """Add a neighbour list to ``data`` in place (contract of ``nequip/data/_nl.py:364-381``)."""
    if backend not in NEIGHBORLIST_BACKEND_OPTIONS:
        supported = ", ".join(f"`{b}`" for b in NEIGHBORLIST_BACKEND_OPTIONS)
        raise ValueError(f"Unknown neighborlist backend = `{backend}`. Supported backends: {supported}")
    K = AtomicDataDict
    batched = K.BATCH_KEY in data
    nframes = K.num_frames(data)
    if batched:
        counts = data[K.NUM_NODES_KEY].view(-1).cpu().tolist() if K.NUM_NODES_KEY in data else torch.bincount(
            data[K.BATCH_KEY], minlength=nframes).cpu().tolist()
    else:
        counts = [data[K.POSITIONS_KEY].shape[0]]
    offsets = [0]
    for c in counts:
        offsets.append(offsets[-1] + int(c))
    has_cell = data.get(K.CELL_KEY, None) is not None
    eidx, shifts, rowptrs = [], [], []
    for f in range(nframes):
        frame = _frame_from_batched(data, f, offsets)
        cell = frame.get(K.CELL_KEY, None)
        pbc = frame.get(K.PBC_KEY, None)
        if pbc is None:
            pbc = False
        ei, sh, rp = _compute_neighborlist_single_frame(frame[K.POSITIONS_KEY], r_max, cell=cell, pbc=pbc,
                                                        return_rowptr=True)
        eidx.append(ei + offsets[f])
        rowptrs.append(rp)
        shifts.append(sh)
    data[K.EDGE_INDEX_KEY] = torch.cat(eidx, dim=1) if len(eidx) > 1 else eidx[0]
    # the list is grouped by centre atom: give the tensor-product kernels its row pointer (saves the dst sort)
    from ..nn._topology import topology_cache

    if len(rowptrs) == 1:
        rowptr = rowptrs[0]
    else:
        parts, eoff = [], 0
        for f, rp in enumerate(rowptrs):
            parts.append(rp[:-1] + eoff)
            eoff += int(eidx[f].shape[1])
        parts.append(torch.tensor([eoff], dtype=torch.int32, device=rowptrs[0].device))
        rowptr = torch.cat(parts).to(torch.int32)
    topology_cache.hint_sorted(data[K.EDGE_INDEX_KEY], rowptr)
    if has_cell:
        data[K.EDGE_CELL_SHIFT_KEY] = torch.cat(shifts, dim=0) if len(shifts) > 1 else shifts[0]
    return data


NEIGHBORLIST_BACKEND_OPTIONS: Final[Dict[str, object]] = {NEIGHBORLIST_BACKEND_NEQUIP_AMD: compute_neighborlist_}
