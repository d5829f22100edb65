"""Dispatcher-op form of the edge-vector kernels: ``torch.ops.nequip_amd.edge_vectors / edge_vectors_adj``.

``with_edge_vectors_`` (``nequip/nn/utils.py:68-118``) -- ``vec = pos[src] - pos[dst] (+ shift @ cell[frame])`` -- and its
adjoint (per-atom ordered sums over both CSRs instead of float64 ``index_add_`` atomics) in a form a tracer keeps
(``utils/tracing.py``).  The map is linear in (pos, cell), so the two ops are each other's derivative: the family is closed
under differentiation to any order.  CUDA only.
"""

from __future__ import annotations

from typing import Optional

import torch

_NS = "nequip_amd"
_lib_def = torch.library.Library(_NS, "FRAGMENT")
_lib_def.define("edge_vectors(Tensor pos, Tensor? cell, Tensor edge_index, Tensor? shift, Tensor? batch) -> Tensor")
_lib_def.define("edge_vectors_adj(Tensor g_vec, Tensor edge_index, Tensor? shift, Tensor? batch, SymInt num_nodes, "
                "SymInt num_frames, bool need_cell) -> (Tensor, Tensor)")


def _fwd_cuda(pos, cell, edge_index, shift, batch):
    from .utils import _EdgeVectorsFn

    return _EdgeVectorsFn.apply(pos.detach(), None if cell is None else cell.detach(), edge_index, shift, batch)


def _adj_cuda(g_vec, edge_index, shift, batch, num_nodes, num_frames, need_cell):
    from .utils import _EdgeVectorsAdjFn

    cell_shape = (int(num_frames), 3, 3) if need_cell else None
    g_pos, g_cell = _EdgeVectorsAdjFn.apply(g_vec.detach(), edge_index, shift, batch, int(num_nodes), cell_shape)
    return g_pos, (g_cell if g_cell is not None else g_vec.new_empty(0))


_lib_def.impl("edge_vectors", _fwd_cuda, "CUDA")
_lib_def.impl("edge_vectors_adj", _adj_cuda, "CUDA")


@torch.library.register_fake(f"{_NS}::edge_vectors")
def _fwd_fake(pos, cell, edge_index, shift, batch):
    return pos.new_empty((edge_index.shape[1], 3), dtype=torch.float64)


@torch.library.register_fake(f"{_NS}::edge_vectors_adj")
def _adj_fake(g_vec, edge_index, shift, batch, num_nodes, num_frames, need_cell):
    g_pos = g_vec.new_empty((num_nodes, 3), dtype=torch.float64)
    g_cell = g_vec.new_empty((num_frames, 3, 3), dtype=torch.float64) if need_cell else g_vec.new_empty(0)
    return g_pos, g_cell


def _fwd_setup(ctx, inputs, output):
    pos, cell, edge_index, shift, batch = inputs
    ctx.edge_index, ctx.shift, ctx.batch = edge_index, shift, batch
    ctx.num_nodes = pos.shape[0]
    ctx.num_frames = None if cell is None else cell.reshape(-1, 3, 3).shape[0]
    ctx.pos_dtype = pos.dtype
    ctx.cell_meta = None if cell is None else (cell.dtype, tuple(cell.shape))


def _fwd_backward(ctx, g_vec):
    need_cell = ctx.cell_meta is not None and ctx.needs_input_grad[1]
    g_pos, g_cell = torch.ops.nequip_amd.edge_vectors_adj(g_vec, ctx.edge_index, ctx.shift, ctx.batch, ctx.num_nodes,
                                                          ctx.num_frames if ctx.num_frames is not None else 1, need_cell)
    g_pos = g_pos.to(ctx.pos_dtype) if ctx.needs_input_grad[0] else None
    g_cell = g_cell.view(ctx.cell_meta[1]).to(ctx.cell_meta[0]) if need_cell else None
    return g_pos, g_cell, None, None, None


torch.library.register_autograd(f"{_NS}::edge_vectors", _fwd_backward, setup_context=_fwd_setup)


def _adj_setup(ctx, inputs, output):
    g_vec, edge_index, shift, batch, num_nodes, num_frames, need_cell = inputs
    ctx.edge_index, ctx.shift, ctx.batch, ctx.need_cell = edge_index, shift, batch, need_cell
    ctx.set_materialize_grads(False)


def _adj_backward(ctx, c_pos, c_cell):
    # adjoint of the adjoint = the (linear) forward map applied to the cotangents
    if c_pos is None:
        raise RuntimeError("double backward through edge vectors needs a position cotangent")
    cell = c_cell if (ctx.need_cell and c_cell is not None) else None
    vec = torch.ops.nequip_amd.edge_vectors(c_pos, cell, ctx.edge_index, ctx.shift if cell is not None else None,
                                            ctx.batch)
    return vec, None, None, None, None, None, None


torch.library.register_autograd(f"{_NS}::edge_vectors_adj", _adj_backward, setup_context=_adj_setup)


def edge_vectors(pos, cell: Optional[torch.Tensor], edge_index, shift: Optional[torch.Tensor],
                 batch: Optional[torch.Tensor]) -> torch.Tensor:
    return torch.ops.nequip_amd.edge_vectors(pos, cell, edge_index, shift, batch)
