"""Seeded synthetic periodic boxes and a host-side neighbour list for benchmarks and parity tests.

The reference builds neighbour lists on the CPU with matscipy / ASE / vesin (``nequip/data/_nl.py:63-165``),
none of which is available here; this is a small numpy/scipy stand-in that follows the same conventions
(``edge_index[0]`` = centre atom that receives the message, ``edge_index[1]`` = neighbour,
``edge_vec = pos[edge_index[1]] - pos[edge_index[0]] + edge_cell_shift @ cell``, ``nequip/data/_nl.py:74-90``,
``nequip/nn/utils.py:88-114``) and emits edges sorted by centre atom.  It is input preparation, not part of
the timed hot path ("neighbor list prebuilt and resident on device").

Box recipes are the BASELINE.json configs as specified in SURVEY.md 8(d).
"""

from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from ..data import AtomicDataDict


def neighbor_list(pos: np.ndarray, r_max: float, cell: Optional[np.ndarray] = None, pbc: bool = True):
    """Full (both directions) neighbour list.  Returns ``edge_index [2,E]`` int64 and ``edge_cell_shift [E,3]``."""
    from scipy.spatial import cKDTree

    pos = np.asarray(pos, dtype=np.float64)
    N = pos.shape[0]
    if cell is None or not pbc:
        tree = cKDTree(pos)
        coo = tree.sparse_distance_matrix(tree, r_max, output_type="coo_matrix")
        i, j = coo.row, coo.col
        keep = i != j
        i, j = i[keep], j[keep]
        order = np.lexsort((j, i))
        return np.stack([i[order], j[order]]).astype(np.int64), np.zeros((len(i), 3), dtype=np.float64)

    cell = np.asarray(cell, dtype=np.float64).reshape(3, 3)
    inv = np.linalg.inv(cell)
    # perpendicular heights of the cell -> how many images are needed along each lattice vector
    vol = abs(np.linalg.det(cell))
    heights = np.array(
        [
            vol / np.linalg.norm(np.cross(cell[1], cell[2])),
            vol / np.linalg.norm(np.cross(cell[2], cell[0])),
            vol / np.linalg.norm(np.cross(cell[0], cell[1])),
        ]
    )
    nimg = np.ceil(r_max / heights).astype(int)
    frac = pos @ inv
    shifts = np.array(
        [
            (a, b, c)
            for a in range(-nimg[0], nimg[0] + 1)
            for b in range(-nimg[1], nimg[1] + 1)
            for c in range(-nimg[2], nimg[2] + 1)
        ],
        dtype=np.float64,
    )
    img_pos, img_idx, img_shift = [], [], []
    margin = r_max / heights  # fractional skin that can reach into the home cell
    fmin, fmax = frac.min(0) - margin, frac.max(0) + margin
    for s in shifts:
        f = frac + s
        keep = np.all((f >= fmin) & (f <= fmax), axis=1)
        if not keep.any():
            continue
        idx = np.nonzero(keep)[0]
        img_pos.append(pos[idx] + s @ cell)
        img_idx.append(idx)
        img_shift.append(np.broadcast_to(s, (len(idx), 3)))
    img_pos = np.concatenate(img_pos)
    img_idx = np.concatenate(img_idx)
    img_shift = np.concatenate(img_shift)
    tree_c = cKDTree(pos)
    tree_i = cKDTree(img_pos)
    coo = tree_c.sparse_distance_matrix(tree_i, r_max, output_type="coo_matrix")
    i, k = coo.row, coo.col
    j = img_idx[k]
    S = img_shift[k]
    keep = ~((i == j) & np.all(S == 0, axis=1))
    i, j, S = i[keep], j[keep], S[keep]
    order = np.lexsort((S[:, 2], S[:, 1], S[:, 0], j, i))
    return np.stack([i[order], j[order]]).astype(np.int64), S[order]


def morton_order(pos: np.ndarray, cell_size: float = 2.25) -> np.ndarray:
    """Permutation that sorts atoms along a Z-order (Morton) curve of `cell_size` boxes.  Spatially close atoms get
    close indices, so the node rows gathered by neighbouring atoms stay resident in an XCD's 4 MiB L2 (the model is
    permutation equivariant; this is pure input ordering, like the cell binning of any MD neighbour list)."""
    pos = np.asarray(pos, dtype=np.float64)
    q = np.floor((pos - pos.min(0)) / cell_size).astype(np.uint64)
    code = np.zeros(len(pos), dtype=np.uint64)
    for bit in range(16):
        for ax in range(3):
            code |= ((q[:, ax] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + ax)
    return np.argsort(code, kind="stable")


def make_data(pos, types, r_max: float, cell=None, pbc: bool = True, spatial_sort: bool = False) -> AtomicDataDict.Type:
    if spatial_sort:
        perm = morton_order(pos)
        pos, types = np.asarray(pos)[perm], np.asarray(types)[perm]
    edge_index, shifts = neighbor_list(pos, r_max, cell, pbc)
    data = {
        AtomicDataDict.POSITIONS_KEY: torch.as_tensor(pos, dtype=torch.float64),
        AtomicDataDict.ATOM_TYPE_KEY: torch.as_tensor(types, dtype=torch.long),
        AtomicDataDict.EDGE_INDEX_KEY: torch.as_tensor(edge_index, dtype=torch.long),
    }
    if cell is not None and pbc:
        data[AtomicDataDict.CELL_KEY] = torch.as_tensor(cell, dtype=torch.float64).view(1, 3, 3)
        data[AtomicDataDict.EDGE_CELL_SHIFT_KEY] = torch.as_tensor(shifts, dtype=torch.float64)
    return data


# ---- BASELINE configs ------------------------------------------------------------


def silicon_box(reps: int = 5, a: float = 5.431, rattle: float = 0.05, seed: int = 0):
    """cfg-2: diamond-cubic Si, reps^3 conventional cells (8 atoms each); 5 -> 1000 atoms."""
    rng = np.random.default_rng(seed)
    basis = np.array(
        [[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0], [0.25, 0.25, 0.25], [0.25, 0.75, 0.75],
         [0.75, 0.25, 0.75], [0.75, 0.75, 0.25]]
    )  # fmt: skip
    cells = np.array([(i, j, k) for i in range(reps) for j in range(reps) for k in range(reps)], dtype=np.float64)
    pos = ((cells[:, None, :] + basis[None, :, :]) * a).reshape(-1, 3)
    pos = pos + rng.normal(0.0, rattle, size=pos.shape)
    cell = np.eye(3) * a * reps
    return pos, np.zeros(len(pos), dtype=np.int64), cell, ["Si"]


def water_box(n_side: int = 15, seed: int = 0):
    """cfg-3: n_side^3 H2O on a jittered cubic grid, random orientations, 0.0334 molecules/A^3; 15 -> 10125 atoms."""
    rng = np.random.default_rng(seed)
    nmol = n_side**3
    L = (nmol / 0.0334) ** (1.0 / 3.0)
    spacing = L / n_side
    grid = np.array([(i, j, k) for i in range(n_side) for j in range(n_side) for k in range(n_side)], dtype=np.float64)
    centers = (grid + 0.5) * spacing + rng.uniform(-0.1, 0.1, size=(nmol, 3)) * spacing
    r_oh, ang = 0.9572, np.deg2rad(104.52)
    h1 = np.array([r_oh * np.sin(ang / 2), r_oh * np.cos(ang / 2), 0.0])
    h2 = np.array([-r_oh * np.sin(ang / 2), r_oh * np.cos(ang / 2), 0.0])
    # random rotations (QR of Gaussian matrices)
    A = rng.normal(size=(nmol, 3, 3))
    Q, R = np.linalg.qr(A)
    Q = Q * np.sign(np.diagonal(R, axis1=1, axis2=2))[:, None, :]
    pos = np.empty((nmol, 3, 3))
    pos[:, 0] = centers
    pos[:, 1] = centers + h1 @ np.transpose(Q, (0, 2, 1))
    pos[:, 2] = centers + h2 @ np.transpose(Q, (0, 2, 1))
    types = np.tile(np.array([1, 0, 0]), nmol)  # type_names = ["H", "O"]
    return pos.reshape(-1, 3), types, np.eye(3) * L, ["H", "O"]


def random_frame(n_atoms: int = 256, n_species: int = 5, density: float = 0.05, min_dist: float = 1.6, seed: int = 0):
    """cfg-4: random positions with a minimum distance in a cubic box at `density` atoms/A^3."""
    rng = np.random.default_rng(seed)
    L = (n_atoms / density) ** (1.0 / 3.0)
    pos = np.empty((0, 3))
    while len(pos) < n_atoms:
        cand = rng.uniform(0, L, size=(4 * n_atoms, 3))
        for c in cand:
            if len(pos) == 0:
                pos = c[None]
                continue
            d = pos - c
            d -= L * np.round(d / L)
            if (np.einsum("ij,ij->i", d, d) > min_dist**2).all():
                pos = np.vstack([pos, c])
                if len(pos) == n_atoms:
                    break
    types = rng.integers(0, n_species, size=n_atoms)
    return pos, types, np.eye(3) * L, [f"X{i}" for i in range(n_species)]


def copper_box(reps: Tuple[int, int, int] = (25, 25, 40), a: float = 3.615, rattle: float = 0.05, seed: int = 0):
    """cfg-5: fcc Cu, reps conventional cells (4 atoms each); (25,25,40) -> 100000 atoms."""
    rng = np.random.default_rng(seed)
    basis = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
    cells = np.array(
        [(i, j, k) for i in range(reps[0]) for j in range(reps[1]) for k in range(reps[2])], dtype=np.float64
    )
    pos = ((cells[:, None, :] + basis[None, :, :]) * a).reshape(-1, 3)
    pos = pos + rng.normal(0.0, rattle, size=pos.shape)
    cell = np.diag(np.array(reps, dtype=np.float64) * a)
    return pos, np.zeros(len(pos), dtype=np.int64), cell, ["Cu"]


def aspirin_like(seed: int = 0, rattle: float = 0.05):
    """cfg-1: a fixed 21-atom C9H8O4-like geometry (types C,H,O), rattled; non-periodic."""
    rng = np.random.default_rng(seed)
    # planar ring + substituents on a 1.4 A scale; any fixed conformer is acceptable
    ring = np.array([[1.40 * np.cos(t), 1.40 * np.sin(t), 0.0] for t in np.arange(6) * np.pi / 3])
    extra_c = np.array([[2.9, 0.0, 0.0], [-2.1, 2.3, 0.3], [-3.4, 2.9, -0.2]])
    oxy = np.array([[3.6, 1.0, 0.2], [3.5, -1.1, -0.2], [-1.4, 2.6, 1.3], [-0.9, -2.5, 0.1]])
    hyd = np.array(
        [[2.4 * np.cos(t), 2.4 * np.sin(t), 0.1] for t in (np.pi / 3, 2 * np.pi / 3 + 2.2, 4 * np.pi / 3, 5 * np.pi / 3)]
        + [[4.5, 0.9, 0.3], [-3.3, 3.9, -0.6], [-4.1, 2.8, 0.7], [-3.9, 2.3, -0.9]]
    )
    pos = np.concatenate([ring, extra_c, oxy, hyd]) + rng.normal(0, rattle, size=(21, 3))
    types = np.array([0] * 9 + [2] * 4 + [1] * 8)  # type_names = ["C", "H", "O"]
    return pos, types, None, ["C", "H", "O"]
