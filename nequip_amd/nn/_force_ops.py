"""Dispatcher-op form of the inference force / virial tail: ``torch.ops.nequip_amd.force_virial``.

``ForceStressOutput`` (``nequip/nn/grad_output.py:107-298``) in eval mode differentiates the energy w.r.t. the edge vectors
and maps that gradient ``g`` to forces, virial and stress with two kernels (``ForceStressOutput._forward_inference``:
``nqa_edge_vectors_bwd`` -- the atomics-free adjoint of the edge-vector map, which also returns the per-atom
``sum_e edge_vec_e (x) g_e`` -- and ``nqa_virial_finalize``).  This op is that tail in a form a tracer keeps, so a compiled
graph ends the way the eager evaluation does instead of with the reference's strain bookkeeping in float64 ATen kernels:

``force_virial(g_vec [E, 3], edge_vec [E, 3], edge_index [2, E], batch?, cell?, num_nodes, num_frames)
  -> (forces [N, 3], virial [F, 3, 3], stress [F, 3, 3] or [0])``   (float64; first order, no autograd formula).
"""

from __future__ import annotations

import torch

_NS = "nequip_amd"
_lib_def = torch.library.Library(_NS, "FRAGMENT")
_lib_def.define("force_virial(Tensor g_vec, Tensor edge_vec, Tensor edge_index, Tensor? batch, Tensor? cell, SymInt num_nodes, "
                "SymInt num_frames) -> (Tensor, Tensor, Tensor)")


def _launch(g_vec, edge_vec, edge_index, batch, cell, num_nodes: int, num_frames: int):
    from .. import _lib
    from ._topology import _ptr, current_stream_ptr, topology_cache

    lib = _lib.load()
    dev = g_vec.device
    g = g_vec.detach().to(torch.float64).contiguous()
    ev = edge_vec.detach().to(torch.float64).contiguous()
    topo = topology_cache.get(edge_index[0], edge_index[1], num_nodes)
    rp_d, eid_d, _ = topo.by_dst
    rp_s, eid_s, _ = topo.by_src
    forces = torch.empty((num_nodes, 3), dtype=torch.float64, device=dev)
    part = torch.empty((num_nodes, 9), dtype=torch.float64, device=dev)
    virial = torch.empty((num_frames, 3, 3), dtype=torch.float64, device=dev)
    has_cell = cell is not None
    stress = torch.empty((num_frames, 3, 3), dtype=torch.float64, device=dev) if has_cell else None
    cell_c = (cell.detach().reshape(-1, 3, 3).expand(num_frames, 3, 3).to(torch.float64).contiguous()
              if has_cell else None)
    stream = current_stream_ptr(dev)
    with torch.cuda.device(dev):
        # part[n] = sum_{e: centre(e) = n} edge_vec_e (x) g_e;  sign -1: forces = -dE/dpos directly
        rc = lib.nqa_edge_vectors_bwd(_ptr(g), _ptr(ev), _ptr(rp_d), _ptr(eid_d), _ptr(rp_s), _ptr(eid_s), num_nodes, -1.0,
                                      _ptr(forces), _ptr(part), stream)
        _lib.check(rc, "nqa_edge_vectors_bwd")
        # virial = -sym(sum_n part[n]) per frame, stress = sym / volume
        rc = lib.nqa_virial_finalize(_ptr(part), _ptr(batch.contiguous() if batch is not None else None), _ptr(cell_c),
                                     num_nodes, num_frames, _ptr(virial), _ptr(stress), stream)
        _lib.check(rc, "nqa_virial_finalize")
    return forces, virial, stress


def _cuda(g_vec, edge_vec, edge_index, batch, cell, num_nodes, num_frames):
    forces, virial, stress = _launch(g_vec, edge_vec, edge_index, batch, cell, int(num_nodes), int(num_frames))
    return forces, virial, (stress if stress is not None else g_vec.new_empty(0, dtype=torch.float64))


_lib_def.impl("force_virial", _cuda, "CUDA")


@torch.library.register_fake(f"{_NS}::force_virial")
def _fake(g_vec, edge_vec, edge_index, batch, cell, num_nodes, num_frames):
    f64 = torch.float64
    return (g_vec.new_empty((num_nodes, 3), dtype=f64), g_vec.new_empty((num_frames, 3, 3), dtype=f64),
            g_vec.new_empty((num_frames, 3, 3), dtype=f64) if cell is not None else g_vec.new_empty(0, dtype=f64))


def force_virial(g_vec, edge_vec, edge_index, batch, cell, num_nodes, num_frames):
    return torch.ops.nequip_amd.force_virial(g_vec, edge_vec, edge_index, batch, cell, num_nodes, num_frames)
