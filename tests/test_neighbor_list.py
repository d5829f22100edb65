"""Device neighbour list (nqa_neighbor_list_count/fill) against

* the reference's own known-answer test (tests/unit/data/test_neighborlist.py:46-100: two-atom silicon cell, r_max 2.5:
  edge set {(0,1) x4, (1,0) x4}) and its no-neighbour cases (:102-114);
* a brute-force enumeration over lattice images (float64) on random triclinic / thin / partially periodic boxes with
  atoms outside the cell: identical (i, j, S) sets, and |pos_j - pos_i + S @ cell| < r_max for every edge;
* the contracts of compute_neighborlist_ (nequip/data/_nl.py:364-381): batched in -> batched out, shift only with a
  cell, device preserved, edges grouped by centre atom.
"""

import itertools

import numpy as np
import pytest
import torch


def _brute_force(pos, cell, pbc, r_max):
    pos = np.asarray(pos, dtype=np.float64)
    n = len(pos)
    if cell is None:
        cell = np.eye(3)
        pbc = (False,) * 3
    cell = np.asarray(cell, dtype=np.float64)
    inv = np.linalg.inv(cell)
    heights = 1.0 / np.linalg.norm(inv, axis=0)
    frac = pos @ inv
    spread = np.ceil(frac.max(0) - frac.min(0)).astype(int) + 1 if n else np.zeros(3, int)
    rng = [range(-(int(np.ceil(r_max / heights[d])) + spread[d]), int(np.ceil(r_max / heights[d])) + spread[d] + 1)
           if pbc[d] else range(0, 1) for d in range(3)]
    out = set()
    for S in itertools.product(*rng):
        d = pos[None, :, :] + (np.array(S, dtype=np.float64) @ cell)[None, None, :] - pos[:, None, :]
        r2 = (d * d).sum(-1)
        ii, jj = np.nonzero(r2 < r_max * r_max)
        for i, j in zip(ii, jj):
            if i == j and S == (0, 0, 0):
                continue
            out.add((int(i), int(j)) + tuple(int(s) for s in S))
    return out


def _as_set(edge_index, shift):
    ei = edge_index.cpu().numpy()
    sh = shift.cpu().numpy()
    assert np.array_equal(sh, np.round(sh))
    return {(int(a), int(b), int(s[0]), int(s[1]), int(s[2])) for a, b, s in zip(ei[0], ei[1], sh)}


@pytest.mark.gpu
def test_silicon_known_answer(device):
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.data._nl import compute_neighborlist_

    lattice = torch.tensor([[3.34939851, 0, 1.93377613], [1.11646617, 3.1578432, 1.93377613], [0, 0, 3.86755226]])
    coords = torch.tensor([[0, 0, 0], [1.11646617, 0.7894608, 1.93377613]])
    data = {K.POSITIONS_KEY: coords.to(device), K.CELL_KEY: lattice.view(1, 3, 3).to(device),
            K.PBC_KEY: torch.tensor([[True, True, True]], device=device)}
    data = compute_neighborlist_(data, r_max=2.5)
    ei = data[K.EDGE_INDEX_KEY].cpu().numpy()
    assert sorted(zip(ei[0], ei[1])) == [(0, 1)] * 4 + [(1, 0)] * 4  # tests/unit/data/test_neighborlist.py:88-100
    assert data[K.EDGE_CELL_SHIFT_KEY].shape == (8, 3) and data[K.EDGE_CELL_SHIFT_KEY].dtype == coords.dtype
    assert data[K.EDGE_INDEX_KEY].device.type == "cuda"


@pytest.mark.gpu
def test_no_neighbors(device):
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.data._nl import compute_neighborlist_

    pbc = torch.tensor([[True, True, True]], device=device)
    # isolated atom in a 20 A box (reference test_no_neighbors)
    d = compute_neighborlist_({K.POSITIONS_KEY: torch.zeros(1, 3, dtype=torch.float64, device=device),
                               K.CELL_KEY: (20 * torch.eye(3, dtype=torch.float64, device=device)).view(1, 3, 3),
                               K.PBC_KEY: pbc}, r_max=2.5)
    assert d[K.EDGE_INDEX_KEY].numel() == 0 and d[K.EDGE_CELL_SHIFT_KEY].numel() == 0
    # fcc Cu, a = 3.6: nearest neighbour 2.546 > 2.5
    a = 3.6
    pos = torch.tensor([[0, 0, 0], [0, a / 2, a / 2], [a / 2, 0, a / 2], [a / 2, a / 2, 0]], dtype=torch.float64)
    d = compute_neighborlist_({K.POSITIONS_KEY: pos.to(device), K.PBC_KEY: pbc,
                               K.CELL_KEY: (a * torch.eye(3, dtype=torch.float64, device=device)).view(1, 3, 3)}, r_max=2.5)
    assert d[K.EDGE_INDEX_KEY].numel() == 0
    # ... and with a cutoff just beyond it: 12 neighbours each
    d = compute_neighborlist_({K.POSITIONS_KEY: pos.to(device), K.PBC_KEY: pbc,
                               K.CELL_KEY: (a * torch.eye(3, dtype=torch.float64, device=device)).view(1, 3, 3)}, r_max=2.6)
    assert d[K.EDGE_INDEX_KEY].shape[1] == 4 * 12
    # no cell, no pbc: no shift key
    d = compute_neighborlist_({K.POSITIONS_KEY: pos.to(device)}, r_max=2.6)
    assert K.EDGE_CELL_SHIFT_KEY not in d and d[K.EDGE_INDEX_KEY].shape[1] == 12
    with pytest.raises(ValueError):
        compute_neighborlist_({K.POSITIONS_KEY: pos.to(device), K.PBC_KEY: pbc}, r_max=2.6)


CASES = [
    # (n_atoms, cell, pbc, r_max, spread): spread > 1 puts atoms outside the home cell
    (60, np.diag([9.0, 8.0, 10.0]), (True, True, True), 3.0, 1.0),
    (40, np.array([[7.0, 0.3, -0.4], [1.9, 6.5, 0.2], [-1.1, 2.4, 8.0]]), (True, True, True), 3.5, 2.5),
    (25, np.array([[2.2, 0.0, 0.0], [0.4, 2.5, 0.0], [0.3, -0.2, 2.8]]), (True, True, True), 5.0, 1.0),  # thin cell: many images
    (50, np.diag([6.0, 7.0, 30.0]), (True, True, False), 3.0, 1.0),  # slab
    (50, np.diag([6.0, 40.0, 40.0]), (True, False, False), 3.2, 1.0),  # wire
    (70, None, (False, False, False), 2.5, 1.0),  # molecule / cluster
    (3, np.diag([50.0, 50.0, 50.0]), (True, True, True), 4.0, 1.0),  # sparse box (grid coarsening)
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(CASES)))
def test_matches_brute_force(device, case):
    from nequip_amd.data._nl import _compute_neighborlist_single_frame

    n, cell, pbc, r_max, spread = CASES[case]
    rng = np.random.default_rng(100 + case)
    if cell is None:
        pos = rng.uniform(-4.0, 4.0, size=(n, 3))
    else:
        pos = (rng.uniform(-(spread - 1.0), spread, size=(n, 3))) @ cell
    ref = _brute_force(pos, cell, pbc, r_max)
    ei, sh = _compute_neighborlist_single_frame(
        torch.tensor(pos, device=device), r_max, cell=None if cell is None else torch.tensor(cell, device=device), pbc=pbc)
    got = _as_set(ei, sh)
    assert len(got) == ei.shape[1], "duplicate edges"
    assert got == ref
    assert bool((ei[0][1:] >= ei[0][:-1]).all()), "edges must be grouped by centre atom in ascending order"
    if cell is not None and ei.shape[1]:
        p = torch.tensor(pos, device=device)
        vec = p[ei[1]] - p[ei[0]] + sh @ torch.tensor(cell, device=device)
        assert float(vec.norm(dim=1).max()) < r_max


@pytest.mark.gpu
def test_batched_contract_and_model_consistency(device):
    """Batched input -> batched output with node offsets; the list drives the model to the same energy as the host-side
    scipy list used elsewhere in the tests (edge order differs, energies must not)."""
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.data._nl import compute_neighborlist_
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(3, seed=3)
    ref = syn.make_data(pos, types, 4.5, cell)
    frames = []
    for shift in (0.0, 0.37):
        frames.append({K.POSITIONS_KEY: torch.tensor(pos + shift, dtype=torch.float64), K.ATOM_TYPE_KEY: ref[K.ATOM_TYPE_KEY],
                       K.CELL_KEY: torch.tensor(cell, dtype=torch.float64).view(1, 3, 3), K.EDGE_INDEX_KEY: torch.zeros(2, 0, dtype=torch.long),
                       K.PBC_KEY: torch.tensor([[True, True, True]])})
    batched = K.batched_from_list(frames)
    batched[K.PBC_KEY] = torch.tensor([[True, True, True]] * 2)
    del batched[K.EDGE_INDEX_KEY]
    batched = K.to_device(batched, device)
    batched = compute_neighborlist_(batched, 4.5)
    n = len(pos)
    ei = batched[K.EDGE_INDEX_KEY]
    assert ei.shape[1] == 2 * ref[K.EDGE_INDEX_KEY].shape[1]
    half = ei.shape[1] // 2
    assert int(ei[:, :half].max()) < n and int(ei[:, half:].min()) >= n
    model = NequIPGNNModel(seed=0, model_dtype="float32", r_max=4.5, type_names=names, num_layers=2, l_max=1,
                           parity=False, num_features=8, radial_mlp_width=64, radial_mlp_depth=1,
                           avg_num_neighbors=20.0).to(device).eval()
    out_b = model(dict(batched))
    out_r = model(K.to_device(dict(ref), device))
    torch.testing.assert_close(out_b[K.TOTAL_ENERGY_KEY][0], out_r[K.TOTAL_ENERGY_KEY][0], atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(out_b[K.TOTAL_ENERGY_KEY][1], out_r[K.TOTAL_ENERGY_KEY][0], atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(out_b[K.FORCE_KEY][:n], out_r[K.FORCE_KEY], atol=5e-5, rtol=5e-5)
