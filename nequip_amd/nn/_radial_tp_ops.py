"""Dispatcher-op form of one convolution's edge side (inference): ``torch.ops.nequip_amd.radial_tp_fwd / radial_tp_bwd``.

``InteractionBlock.forward`` runs ``edge_mlp`` on the edge embedding and hands the ``[E, W]`` result to ``tp_scatter``
(``nequip/nn/interaction_block.py:190-199``).  The eager path of this package evaluates the MLP once per reverse-edge PAIR
and runs the pair-centric tensor-product backward (``nn/_paired_radial.py``) -- a decision that depends on the edge list
(does every edge have exactly one reverse partner?) and is read on the host, which a traced graph cannot do.  These two
ops move the decision behind the dispatcher: the graph sees

* ``radial_tp_fwd(emb [E, nb], x, edge_attr, w0, w1, alpha0, alpha1, edge_dst, edge_src, edge_shift?, plan)
  -> (out [N, D_mid], w_rows [(E // 2) * W])``
* ``radial_tp_bwd(grad_out, emb, x, edge_attr, w_rows, w0, w1, alpha0, alpha1, edge_dst, edge_src, edge_shift?, plan,
  need_emb, need_x, need_y) -> (g_emb [E, nb], g_x, g_edge_attr)``

with shapes that are functions of ``E`` and ``N`` alone (``w_rows`` is flat: a ``[E // 2, W]`` result would make an exporter
guard on ``E // 2 != 1``, its 0 / 1 size specialisation, and reject two-edge graphs), and the implementation pairs the list when it pairs up (the
topology, the pairing and its owner lists come from the cache of ``nn/_topology.py``, shared by the ops of one evaluation
and across evaluations of an unchanged list): MLP on the pairs' representative rows, ``w_rows`` = those ``E / 2`` weight
rows (saved for the backward), pair-centric backward, MLP backward on the summed gradient rows, and ``g_emb`` with the
pair's gradient on its representative edge and zero on the reverse one -- what the eager ``pair_rows`` gather hands back.
A list that does not pair up takes the per-edge kernels inside the same ops (``w_rows`` is then unused and the backward
re-evaluates the MLP).  The weights are constants of these ops and the family is first order: a graph that differentiates
the forces again (training) keeps ``radial_mlp_*`` + ``tp_scatter_*``.
"""

from __future__ import annotations

import os
from typing import Optional

import torch

from . import _tp_scatter_ops as _tpo

_NS = "nequip_amd"
_lib_def = torch.library.Library(_NS, "FRAGMENT")
_lib_def.define("radial_tp_fwd(Tensor emb, Tensor x, Tensor edge_attr, Tensor w0, Tensor w1, float alpha0, float alpha1, "
                "Tensor edge_dst, Tensor edge_src, Tensor? edge_shift, str plan) -> (Tensor, Tensor)")
_lib_def.define("radial_tp_bwd(Tensor grad_out, Tensor emb, Tensor x, Tensor edge_attr, Tensor w_rows, Tensor w0, Tensor w1, "
                "float alpha0, float alpha1, Tensor edge_dst, Tensor edge_src, Tensor? edge_shift, str plan, bool need_emb, "
                "bool need_x, bool need_y) -> (Tensor, Tensor, Tensor)")


def _context(x, edge_dst, edge_src, edge_shift, plan):
    """(kernels, topology, pairing or None) of one call."""
    k = _tpo._kernels(plan, x.device)
    topo = _tpo._topology(edge_dst, edge_src, x.size(0))
    pairing = None
    if k.has_spec(torch.float32) and os.environ.get("NQA_NO_PAIRED", "") in ("", "0"):
        pairing = topo.pairing(edge_shift)
    return k, topo, pairing


def _images(w1, alpha1):
    from ._mlp_ops import _cache_for

    return _cache_for(w1, alpha1)


def _fwd_cuda(emb, x, edge_attr, w0, w1, alpha0, alpha1, edge_dst, edge_src, edge_shift, plan):
    from . import mlp as m
    from . import _paired_radial as pr

    emb, x, y = emb.contiguous(), x.contiguous(), edge_attr.contiguous()
    k, topo, pairing = _context(x, edge_dst, edge_src, edge_shift, plan)
    mode, cache = m.radial_mlp_mode(), _images(w1, alpha1)
    if pairing is not None:
        with torch.no_grad():
            emb_half = pr.pair_rows(emb, pairing)
        w_rows = m._launch_fwd(emb_half, w0, w1, alpha0, alpha1, mode, cache)
        return k.fwd(x, y, w_rows, topo, pairing), w_rows.view(-1)
    w = m._launch_fwd(emb, w0, w1, alpha0, alpha1, mode, cache)
    return k.fwd(x, y, w, topo), emb.new_empty(((emb.shape[0] // 2) * w1.shape[1],))


def _bwd_cuda(grad_out, emb, x, edge_attr, w_rows, w0, w1, alpha0, alpha1, edge_dst, edge_src, edge_shift, plan,
              need_emb, need_x, need_y):
    from . import mlp as m
    from . import _paired_radial as pr

    g, emb, x, y = grad_out.contiguous(), emb.contiguous(), x.contiguous(), edge_attr.contiguous()
    k, topo, pairing = _context(x, edge_dst, edge_src, edge_shift, plan)
    mode, cache = m.radial_mlp_mode(), _images(w1, alpha1)
    empty = x.new_empty(0)
    gx = gy = G = None
    if pairing is None:
        w = m._launch_fwd(emb, w0, w1, alpha0, alpha1, mode, cache)
        gx_, gy_, gw_ = _tpo._bwd_cuda(g, x, y, w, edge_dst, edge_src, plan, need_x, need_y, need_emb)
        g_emb = m._launch_bwd(emb, w0, w1, alpha0, alpha1, gw_, mode, cache) if need_emb else empty
        return g_emb, gx_, gy_
    # the same choices as _PairedRadialTPFn.backward (nn/_paired_radial.py), on the launching stream
    P = pairing.num_pairs
    w_rows = w_rows.contiguous().view(P, w1.shape[1])
    fused, folded = None, False
    if need_emb and need_x and need_y and k.prefer_fused_bwd and os.environ.get("NQA_NO_FUSED_BWD", "") in ("", "0"):
        if pr._pair_backward_pays(g):
            fused = k.bwd_pairs(x, y, w_rows, g, topo, pairing)
            folded = fused is not None
        if fused is None and k.fused_rows_ok:
            fused = k.bwd_fused(x, y, w_rows, g, topo, pairing=pairing)
    if fused is not None:
        gx, G, gy = fused
    else:
        if need_x:
            gx = k.bwd_x(y, w_rows, g, topo, pairing)
        pairs = None
        if need_emb and need_y and pr._pair_backward_pays(g):
            pairs = k.bwd_pairs(x, y, w_rows, g, topo, pairing, need_gx=False)
        if pairs is not None:
            _, G, gy = pairs
            folded = True
        elif need_emb or need_y:
            G, gy = k.bwd_edge(x, y, w_rows, g, topo, need_gw=need_emb, need_gy=need_y, pairing=pairing)
    g_emb = empty
    if need_emb:
        with torch.no_grad():
            emb_half = pr.pair_rows(emb, pairing)
            if folded:
                g_half = m._launch_bwd(emb_half, w0, w1, alpha0, alpha1, G, mode, cache)
            else:
                g_half = m._launch_bwd_paired(emb_half, w0, w1, alpha0, alpha1, G[:P], G[P:], mode, cache)
            g_emb = pr._PairExpandFn.apply(g_half, pairing, emb.shape[0])
    return g_emb, (gx if gx is not None else empty), (gy if gy is not None else empty)


_lib_def.impl("radial_tp_fwd", _fwd_cuda, "CUDA")
_lib_def.impl("radial_tp_bwd", _bwd_cuda, "CUDA")


@torch.library.register_fake(f"{_NS}::radial_tp_fwd")
def _fwd_fake(emb, x, edge_attr, w0, w1, alpha0, alpha1, edge_dst, edge_src, edge_shift, plan):
    d1, d2, do, wn = _tpo.plan_dims(plan)
    torch._check(x.dim() == 2 and x.shape[1] == d1, lambda: f"x must be [N, {d1}]")
    torch._check(edge_attr.dim() == 2 and edge_attr.shape[1] == d2, lambda: f"edge_attr must be [E, {d2}]")
    torch._check(w1.dim() == 2 and w1.shape[1] == wn, lambda: f"w1 must be [H, {wn}]")
    torch._check(emb.dim() == 2 and w0.dim() == 2 and emb.shape[1] == w0.shape[0], lambda: "emb [E, nb], w0 [nb, H]")
    return x.new_empty((x.shape[0], do)), x.new_empty(((emb.shape[0] // 2) * wn,))


@torch.library.register_fake(f"{_NS}::radial_tp_bwd")
def _bwd_fake(grad_out, emb, x, edge_attr, w_rows, w0, w1, alpha0, alpha1, edge_dst, edge_src, edge_shift, plan, need_emb,
              need_x, need_y):
    return (torch.empty_like(emb) if need_emb else x.new_empty(0), torch.empty_like(x) if need_x else x.new_empty(0),
            torch.empty_like(edge_attr) if need_y else x.new_empty(0))


def _fwd_setup(ctx, inputs, output):
    emb, x, y, w0, w1, alpha0, alpha1, dst, src, shift, plan = inputs
    ctx.save_for_backward(emb, x, y, output[1], w0, w1, dst, src, shift)
    ctx.alphas, ctx.plan = (alpha0, alpha1), plan
    ctx.set_materialize_grads(False)


def _fwd_backward(ctx, g, _g_rows):
    if g is None:
        return (None,) * 11
    emb, x, y, w_rows, w0, w1, dst, src, shift = ctx.saved_tensors
    ne, nx, ny = ctx.needs_input_grad[:3]
    g_emb, gx, gy = torch.ops.nequip_amd.radial_tp_bwd(g, emb, x, y, w_rows, w0, w1, *ctx.alphas, dst, src, shift,
                                                       ctx.plan, ne, nx, ny)
    return (g_emb if ne else None, gx if nx else None, gy if ny else None) + (None,) * 8


torch.library.register_autograd(f"{_NS}::radial_tp_fwd", _fwd_backward, setup_context=_fwd_setup)


def usable(edge_mlp, tp_scatter, x: torch.Tensor, emb: torch.Tensor) -> bool:
    """float32 GPU tensors, constant weights, a depth-1 radial MLP in the fused kernels' range (checked on shapes only:
    runs on fake tensors while a tracer follows the model)."""
    from . import mlp as _mlp
    from .. import _lib
    from ..utils.wgrad import differentiable_parameters

    if not x.is_cuda or x.dtype != torch.float32 or emb.dtype != torch.float32:
        return False
    if os.environ.get("NQA_NO_RADIAL_TP_OP", "") not in ("", "0"):
        return False
    if tp_scatter.model_dtype != torch.float32 or not edge_mlp._fused_ok(emb, tracing_ok=True):
        return False
    if _mlp.radial_mlp_mode() != _lib.NQA_MLP_BF16X6:
        return False
    return not differentiable_parameters(edge_mlp.training, edge_mlp.mlp[0].weight, edge_mlp.mlp[2].weight)


def radial_tp(edge_mlp, tp_scatter, emb, x, edge_attr, edge_dst, edge_src, edge_shift: Optional[torch.Tensor]):
    out, _ = torch.ops.nequip_amd.radial_tp_fwd(
        emb, x, edge_attr, edge_mlp.mlp[0].weight.detach(), edge_mlp.mlp[2].weight.detach(),
        float(edge_mlp._alphas[0]), float(edge_mlp._alphas[1]), edge_dst, edge_src, edge_shift, tp_scatter._plan_key)
    return out
