"""Helpers shared by the nn modules (mirror of ``nequip/nn/utils.py``)."""

from typing import Optional

import torch

from ..data import AtomicDataDict
from ..o3.irreps import Irrep, Irreps
from ..utils.tracing import traceable


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


_FRAME_SCAN_MAX = 1 << 24


def _frame_kernel_ok(like: torch.Tensor, batch: torch.Tensor, num_frames: int) -> bool:
    """float64 GPU rows of <= 16 values per atom, and few enough frames for one scan of `batch` per frame."""
    width = 1
    for d in like.shape[1:]:
        width *= d
    return (like.is_cuda and not traceable() and like.dtype == torch.float64 and batch.dtype == torch.int64 and 1 <= width <= 16
            and num_frames * batch.shape[0] <= _FRAME_SCAN_MAX)


class _FrameSumFn(torch.autograd.Function):
    """``out[f] = sum of rows[n] over the atoms of frame f`` (``nqa_frame_sum``); linear, its adjoint is ``_FrameRowsFn``."""

    @staticmethod
    def forward(ctx, rows, batch, num_frames: int):
        from .. import _lib
        from ._topology import _ptr, current_stream_ptr

        rows_c = rows.contiguous()
        width = rows_c.numel() // max(rows_c.shape[0], 1)
        out = torch.empty((num_frames,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
        with torch.cuda.device(rows.device):
            rc = _lib.load().nqa_frame_sum(_ptr(rows_c), _ptr(batch), rows_c.shape[0], width, num_frames, _ptr(out),
                                           current_stream_ptr(rows.device))
        _lib.check(rc, "nqa_frame_sum")
        ctx.batch = batch
        return out

    @staticmethod
    def backward(ctx, c):
        return _FrameRowsFn.apply(c, ctx.batch), None, None


class _FrameRowsFn(torch.autograd.Function):
    """``per_frame[batch]``; the backward is the ordered per-frame sum instead of float64 ``index_add_`` atomics."""

    @staticmethod
    def forward(ctx, per_frame, batch):
        ctx.batch, ctx.num_frames = batch, per_frame.shape[0]
        return torch.index_select(per_frame, 0, batch)

    @staticmethod
    def backward(ctx, g):
        return _FrameSumFn.apply(g, ctx.batch, ctx.num_frames), None


def frame_sum(rows: torch.Tensor, batch: torch.Tensor, num_frames: int) -> torch.Tensor:
    """Per-frame sum of per-atom rows ``[N, ...]`` (float64 on the GPU: ``nqa_frame_sum``; otherwise ``index_add_``)."""
    if _frame_kernel_ok(rows, batch, num_frames):
        return _FrameSumFn.apply(rows, batch.contiguous(), num_frames)
    return torch.zeros((num_frames,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device).index_add_(0, batch, rows)


def frame_rows(per_frame: torch.Tensor, batch: torch.Tensor) -> torch.Tensor:
    """``per_frame[batch]`` with a deterministic, atomics-free backward where the kernel applies."""
    if _frame_kernel_ok(per_frame, batch, per_frame.shape[0]):
        return _FrameRowsFn.apply(per_frame, batch.contiguous())
    return torch.index_select(per_frame, 0, batch)


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


def _as_f64(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.detach().to(torch.float64).contiguous()


class _EdgeVectorsFn(torch.autograd.Function):
    """(pos, cell) -> edge_vec on the GPU (``nqa_edge_vectors_fwd``); linear, so its adjoint ``_EdgeVectorsAdjFn``
    and the adjoint's adjoint (this forward again) close the family for double backward."""

    @staticmethod
    def forward(ctx, pos, cell, edge_index, shift, batch):
        from .. import _lib
        from ._topology import _ptr, current_stream_ptr

        lib = _lib.load()
        # the kernel reads float64 / int64 through raw pointers: normalise here (the reference's with_edge_vectors_
        # accepts any floating dtype, nequip/nn/utils.py:88-114; float32 positions or an integer shift tensor must not be
        # reinterpreted).  autograd casts the float64 gradient back to the input dtype.
        if edge_index.dtype != torch.int64:
            raise TypeError(f"edge_index must be int64, got {edge_index.dtype}")
        pos_c = _as_f64(pos)
        E = edge_index.shape[1]
        dst, src = edge_index[0].contiguous(), edge_index[1].contiguous()
        vec = torch.empty((E, 3), dtype=torch.float64, device=pos.device)
        cell_c = _as_f64(cell)
        shift = _as_f64(shift)
        if batch is not None and batch.dtype != torch.int64:
            batch = batch.to(torch.int64)
        with torch.cuda.device(pos.device):
            rc = lib.nqa_edge_vectors_fwd(_ptr(pos_c), _ptr(dst), _ptr(src), _ptr(shift), _ptr(cell_c), _ptr(batch),
                                          E, _ptr(vec), current_stream_ptr(pos.device))
        _lib.check(rc, "nqa_edge_vectors_fwd")
        ctx.edge_index, ctx.shift, ctx.batch = edge_index, shift, batch
        ctx.num_nodes = pos.shape[0]
        ctx.cell_shape = None if cell is None else tuple(cell.shape)
        return vec

    @staticmethod
    def backward(ctx, g_vec):
        need_cell = ctx.cell_shape is not None and ctx.needs_input_grad[1]
        g_pos, g_cell = _EdgeVectorsAdjFn.apply(
            g_vec, ctx.edge_index, ctx.shift, ctx.batch, ctx.num_nodes, ctx.cell_shape if need_cell else None
        )
        return g_pos, g_cell, None, None, None


class _EdgeVectorsAdjFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g_vec, edge_index, shift, batch, num_nodes, cell_shape):
        from .. import _lib
        from ._topology import _ptr, current_stream_ptr, topology_cache

        lib = _lib.load()
        g = _as_f64(g_vec)
        shift = _as_f64(shift)
        topo = topology_cache.get(edge_index[0], edge_index[1], num_nodes)
        rp_d, eid_d, _ = topo.by_dst
        rp_s, eid_s, _ = topo.by_src
        g_pos = torch.empty((num_nodes, 3), dtype=torch.float64, device=g.device)
        part = torch.empty((num_nodes, 9), dtype=torch.float64, device=g.device) if cell_shape is not None else None
        with torch.cuda.device(g.device):
            rc = lib.nqa_edge_vectors_bwd(_ptr(g), _ptr(shift), _ptr(rp_d), _ptr(eid_d), _ptr(rp_s), _ptr(eid_s),
                                          num_nodes, 1.0, _ptr(g_pos), _ptr(part), current_stream_ptr(g.device))
        _lib.check(rc, "nqa_edge_vectors_bwd")
        g_cell = None
        if cell_shape is not None:
            nframes = 1
            for d in cell_shape[:-2]:
                nframes *= d
            if batch is None or nframes == 1:
                g_cell = part.sum(0).view(cell_shape)
            else:
                # (ordered per-frame tree sum; a dense one-hot product was measured too: rocBLAS runs the float64
                # [frames, N] x [N, 9] shape in 235 us against 38 us for the index_add_ atomics)
                with torch.no_grad():
                    g_cell = frame_sum(part, batch, nframes).view(cell_shape)
        ctx.edge_index, ctx.shift, ctx.batch = edge_index, shift, batch
        ctx.has_cell = cell_shape is not None
        return g_pos, g_cell

    @staticmethod
    def backward(ctx, c_pos, c_cell):
        # adjoint of the adjoint = the (linear) forward map applied to the cotangents
        if c_pos is None:
            raise RuntimeError("double backward through edge vectors needs a position cotangent")
        cell = c_cell if (ctx.has_cell and c_cell is not None) else None
        vec = _EdgeVectorsFn.apply(c_pos, cell, ctx.edge_index, ctx.shift if cell is not None else None, ctx.batch)
        return vec, None, None, None, None, None


def with_edge_vectors_(data: AtomicDataDict.Type, with_lengths: bool = True) -> AtomicDataDict.Type:
    """Edge displacement vectors ``pos[edge_index[1]] - pos[edge_index[0]] (+ shift @ cell)``, differentiable
    w.r.t. positions and cell (``nequip/nn/utils.py:68-118``).  On the GPU this is the HIP kernel pair
    ``nqa_edge_vectors_fwd/bwd`` (atomics-free adjoint); CPU tensors take the reference's ATen formulation
    (host-side data preparation and tests only -- the modules downstream of it are GPU-only)."""
    K = AtomicDataDict
    if K.EDGE_VECTORS_KEY in data:
        if with_lengths and K.EDGE_LENGTH_KEY not in data:
            data[K.EDGE_LENGTH_KEY] = data[K.EDGE_VECTORS_KEY].square().sum(1, keepdim=True).sqrt()
        return data
    pos = data[K.POSITIONS_KEY]
    edge_index = data[K.EDGE_INDEX_KEY]
    if pos.is_cuda:
        cell = data.get(K.CELL_KEY)
        shift = data[K.EDGE_CELL_SHIFT_KEY].contiguous() if cell is not None else None
        batch = data[K.BATCH_KEY].contiguous() if (cell is not None and K.BATCH_KEY in data) else None
        if traceable():  # the same kernels as dispatcher ops (nn/_edge_vector_ops.py)
            from ._edge_vector_ops import edge_vectors as _edge_vectors_op

            edge_vec = _edge_vectors_op(pos, cell, edge_index, shift, batch)
        else:
            edge_vec = _EdgeVectorsFn.apply(pos, cell, edge_index, shift, batch)
        data[K.EDGE_VECTORS_KEY] = edge_vec
        if with_lengths:
            data[K.EDGE_LENGTH_KEY] = edge_vec.square().sum(1, keepdim=True).sqrt()
        return data
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
