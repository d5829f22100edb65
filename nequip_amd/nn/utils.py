"""Helpers shared by the nn modules (mirror of ``nequip/nn/utils.py``)."""

from typing import Optional

import torch

from ..data import AtomicDataDict
from ..o3.irreps import Irrep, Irreps


def scatter(src: torch.Tensor, index: torch.Tensor, dim: int = 0, dim_size: Optional[int] = None) -> torch.Tensor:
    """Sum-scatter along dim 0 (``nequip/nn/utils.py:24-53``); used for the per-frame energy sum only --
    the edge->node reduction of the hot path is fused into the HIP TensorProductScatter kernels."""
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    out_dtype = src.dtype if src.dtype in (torch.float32, torch.float64) else torch.float32
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=out_dtype, device=src.device)
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    return out.scatter_add_(0, idx, src.to(out_dtype))


def tp_path_exists(irreps_in1, irreps_in2, ir_out) -> bool:
    """``nequip/nn/utils.py:56-65``"""
    irreps_in1 = Irreps(irreps_in1).simplify()
    irreps_in2 = Irreps(irreps_in2).simplify()
    ir_out = Irrep(ir_out)
    for _, ir1 in irreps_in1:
        for _, ir2 in irreps_in2:
            if ir_out in list(ir1 * ir2):
                return True
    return False


def with_edge_vectors_(data: AtomicDataDict.Type, with_lengths: bool = True) -> AtomicDataDict.Type:
    """Edge displacement vectors ``pos[edge_index[1]] - pos[edge_index[0]] (+ shift @ cell)``, differentiable
    w.r.t. positions and cell (``nequip/nn/utils.py:68-118``).  A [E,3] float64 tensor: index plumbing done
    with ATen; everything per-edge and heavy downstream of it is in the HIP kernels."""
    K = AtomicDataDict
    if K.EDGE_VECTORS_KEY in data:
        if with_lengths and K.EDGE_LENGTH_KEY not in data:
            data[K.EDGE_LENGTH_KEY] = data[K.EDGE_VECTORS_KEY].square().sum(1, keepdim=True).sqrt()
        return data
    pos = data[K.POSITIONS_KEY]
    edge_index = data[K.EDGE_INDEX_KEY]
    edge_vec = torch.index_select(pos, 0, edge_index[1]) - torch.index_select(pos, 0, edge_index[0])
    if K.CELL_KEY in data:
        cell = data[K.CELL_KEY]
        edge_cell_shift = data[K.EDGE_CELL_SHIFT_KEY]
        if K.BATCH_KEY in data:
            edge_batch = torch.index_select(data[K.BATCH_KEY], 0, edge_index[0])
            edge_vec = torch.baddbmm(
                edge_vec.view(-1, 1, 3),
                edge_cell_shift.view(-1, 1, 3),
                torch.index_select(cell.view(-1, 3, 3), 0, edge_batch),
            ).view(-1, 3)
        else:
            edge_vec = edge_vec + torch.sum(edge_cell_shift.view(-1, 3, 1) * cell.view(3, 3), 1)
    data[K.EDGE_VECTORS_KEY] = edge_vec
    if with_lengths:
        data[K.EDGE_LENGTH_KEY] = edge_vec.square().sum(1, keepdim=True).sqrt()
    return data
