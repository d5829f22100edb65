"""``AtomicDataDict``: the dict-of-tensors data model (mirror of ``nequip/data/AtomicDataDict.py:34-305``,
restricted to what the hot path touches: key names, frame/node counts, batching by concatenation)."""

from typing import Dict, List

import torch

from ._keys import *  # noqa: F401,F403
from . import _keys

Type = Dict[str, torch.Tensor]


def num_nodes(data: Type) -> int:
    # edge-vector based callers (LAMMPS ML-IAP) pass no positions (nequip/data/AtomicDataDict.py:252-262)
    if _keys.POSITIONS_KEY in data:
        return data[_keys.POSITIONS_KEY].size(0)
    if _keys.ATOM_TYPE_KEY in data:
        return data[_keys.ATOM_TYPE_KEY].size(0)
    raise RuntimeError(
        f"No basic input node variable found, expecting either {_keys.POSITIONS_KEY} or {_keys.ATOM_TYPE_KEY}"
    )


def num_edges(data: Type) -> int:
    return data[_keys.EDGE_INDEX_KEY].size(1)


def num_frames(data: Type) -> int:
    if _keys.NUM_NODES_KEY in data:
        return data[_keys.NUM_NODES_KEY].size(0)
    if _keys.BATCH_KEY not in data:
        return 1
    return int(data[_keys.BATCH_KEY].max()) + 1


def batched_from_list(frames: List[Type]) -> Type:
    """Concatenate frames with node-index offsets (``nequip/data/AtomicDataDict.py:71-140``)."""
    out: Type = {}
    offset = 0
    pos, types, eidx, shift, batch, cells, nn = [], [], [], [], [], [], []
    for f, d in enumerate(frames):
        n = d[_keys.POSITIONS_KEY].size(0)
        pos.append(d[_keys.POSITIONS_KEY])
        types.append(d[_keys.ATOM_TYPE_KEY].view(-1))
        eidx.append(d[_keys.EDGE_INDEX_KEY] + offset)
        if _keys.EDGE_CELL_SHIFT_KEY in d:
            shift.append(d[_keys.EDGE_CELL_SHIFT_KEY])
        if _keys.CELL_KEY in d:
            cells.append(d[_keys.CELL_KEY].view(1, 3, 3))
        batch.append(torch.full((n,), f, dtype=torch.long, device=d[_keys.POSITIONS_KEY].device))
        nn.append(n)
        offset += n
    out[_keys.POSITIONS_KEY] = torch.cat(pos)
    out[_keys.ATOM_TYPE_KEY] = torch.cat(types)
    out[_keys.EDGE_INDEX_KEY] = torch.cat(eidx, dim=1)
    out[_keys.BATCH_KEY] = torch.cat(batch)
    out[_keys.NUM_NODES_KEY] = torch.tensor(nn, dtype=torch.long, device=out[_keys.POSITIONS_KEY].device)
    if shift:
        out[_keys.EDGE_CELL_SHIFT_KEY] = torch.cat(shift)
    if cells:
        out[_keys.CELL_KEY] = torch.cat(cells)
    return out


def to_device(data: Type, device) -> Type:
    return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in data.items()}
