"""Radial MLP + tensor-product scatter with *paired* radial weights (inference).

``InteractionBlock.forward`` evaluates ``edge_mlp`` on every directed edge and hands the ``[E, W]`` result to
``tp_scatter`` (``nequip/nn/interaction_block.py:190-199``).  The MLP's input is a function of the edge length only
(``nequip/nn/embedding/_edge.py:136-150``) and a neighbour list holds every interaction in both directions with bitwise
identical lengths, so half of those rows -- half of the only dense GEMM on the edge side -- are duplicates.  When the list
pairs up (``EdgeTopology.pairing``, ``csrc/edge_pairs.hip``) this Function evaluates the MLP once per pair and lets the
tensor-product kernels read the shared row through the pair index (``nqa_tp_scatter_*_paired``); in the backward the two
directed edges of a pair write their weight-gradient halves to rows ``p`` and ``p + P`` and the MLP backward adds the two
streams while loading them (``nqa_radial_mlp_bwd_paired``).  Same numbers as the per-edge evaluation up to the order of
one fp32 addition in the backward.  The fused Function below is first order (eval mode); in training the same pairing
feeds the twice-differentiable per-module Functions (``paired_radial_tp``).
"""

from __future__ import annotations

import os

import torch

from .. import _lib
from . import mlp as _mlp
from ._tp_scatter_base import _Kernels
from ._topology import EdgePairing, EdgeTopology


_side_streams = {}


def side_stream(device: torch.device, slot: int = 0) -> torch.cuda.Stream:
    """Process-wide side HIP streams per device (slot 0: radial-MLP backward queue, slot 1: self-connection branch)."""
    key = (device.type, device.index, slot)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


class RadialBackwardQueue:
    """The radial-MLP backward launches of one model evaluation, on a side HIP stream.

    ``g_emb = radial_mlp_bwd(grad_w)`` of layer L is needed by nobody until the embedding's own backward at the very end
    of the step, it is bound by the matrix cores (split-bf16 MFMA) and by the read of ``grad_w``, while what the main
    stream runs next -- the gate / linear backward of the layer and the tensor-product backward of layer L-1 -- is bound
    by the vector ALU.  So the launches go to a side stream (forked after the tensor-product backward that produced
    ``grad_w``), accumulate the layers' ``g_emb`` there, and the Function of the first layer -- the last one autograd runs
    -- joins the stream and hands the sum to autograd (the other layers return no embedding gradient).  Measured on the
    cfg-3 middle layer: radial backward || fused tensor-product backward 1.09-1.13 ms instead of 1.28 ms back to back
    (scripts/r2_overlap.py).  Captured by hipGraphs as a parallel branch.  ``NQA_NO_OVERLAP=1`` switches it off."""

    def __init__(self, device: torch.device):
        self.stream = side_stream(device, 0)
        self.layers = 0
        self.acc = None
        self.pending = None  # a layer's launch that has not gone out yet (see `submit`)

    @staticmethod
    def lagged() -> bool:
        """Launch a layer's radial backward when the NEXT layer's backward begins, not at the fork itself (its place in the
        stream order -- behind the event recorded at the fork -- is the same).  In a captured hipGraph the first node created
        behind a fork continues on the parent's queue and later ones move to another queue: created at the fork, the side
        work stayed and the main chain hopped queues at every fork (~10 us of idle time each in the cfg-3 timeline; the
        training step's ~25 forks lost more than the overlap gained until its side launches trailed, utils/wgrad.py).
        ``NQA_RADIAL_LAG=0``: launch at the fork."""
        return os.environ.get("NQA_RADIAL_LAG", "1") not in ("", "0")

    def flush(self) -> None:
        item, self.pending = self.pending, None
        if item is not None:
            item()

    def submit(self, launch, ready: "torch.cuda.Event", reads, last: bool) -> None:
        """``launch()`` -> this layer's ``g_emb`` part, to run on the side stream behind ``ready``; ``reads``: tensors of the
        main stream it reads.  ``last``: nothing follows on the main chain (launch now)."""
        def item():
            self.stream.wait_event(ready)
            with torch.cuda.stream(self.stream):
                part = launch()
                self.acc = part if self.acc is None else self.acc.add_(part)
            for t in reads:  # read on the side stream: the allocator must not recycle them earlier
                t.record_stream(self.stream)

        self.flush()
        if last or not self.lagged():
            item()
        else:
            self.pending = item

    @staticmethod
    def enabled() -> bool:
        return os.environ.get("NQA_NO_OVERLAP", "") in ("", "0")


def _pair_backward_pays(g: torch.Tensor) -> bool:
    """The pair-centric backward halves the streamed bytes (weights, their gradient, the grad_x rows) and pays with a
    second gathered node row per pair -- grad_out[other], ``dim_out`` floats -- that has to come out of the cache
    hierarchy.  Measured a win both with ``grad_out`` resident in the 256 MB infinity cache (cfg-3, 90 MB: fused backward
    + radial backward 1.36 -> 1.10 ms) and far beyond it (81 000-atom water box, 725 MB: 9.9 -> 8.5 ms; the neighbours of
    spatially ordered atoms are re-used out of the L2), so there is no size limit by default.  ``NQA_PAIR_BWD_MAX_MB``
    sets one, ``NQA_NO_PAIR_BWD=1`` switches the kernel off."""
    if os.environ.get("NQA_NO_PAIR_BWD", "") not in ("", "0"):
        return False
    limit_mb = os.environ.get("NQA_PAIR_BWD_MAX_MB", "")
    return limit_mb == "" or g.numel() * g.element_size() <= float(limit_mb) * (1 << 20)


class _PairedRadialTPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb_half, x, y, w0, w1, alpha0: float, alpha1: float, mode: int, cache, k: _Kernels,
                topo: EdgeTopology, pairing: EdgePairing, queue=None):
        emb_half, x, y = emb_half.contiguous(), x.contiguous(), y.contiguous()
        w_half = _mlp._launch_fwd(emb_half, w0, w1, alpha0, alpha1, mode, cache)
        out = k.fwd(x, y, w_half, topo, pairing)
        ctx.save_for_backward(emb_half, x, y, w_half, w0, w1)
        ctx.args = (alpha0, alpha1, mode, cache, k, topo, pairing)
        ctx.queue = queue
        if queue is not None:
            queue.layers += 1
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        emb_half, x, y, w_half, w0, w1 = ctx.saved_tensors
        alpha0, alpha1, mode, cache, k, topo, pairing = ctx.args
        need_emb, need_x, need_y = ctx.needs_input_grad[:3]
        if ctx.queue is not None:
            # the previous layer's radial backward goes out now: the main chain's kernels since its fork are queued
            ctx.queue.flush()
        g = g.contiguous()
        P = pairing.num_pairs
        gx = gy = G = None
        fused = None
        folded = False  # G = [P, W] already summed over the two directed edges of every pair (else the two halves [2P, W])
        if need_emb and need_x and need_y and k.prefer_fused_bwd and os.environ.get("NQA_NO_FUSED_BWD", "") in ("", "0"):
            if _pair_backward_pays(g):
                fused = k.bwd_pairs(x, y, w_half, g, topo, pairing)
                folded = fused is not None
            if fused is None and k.fused_rows_ok:
                fused = k.bwd_fused(x, y, w_half, g, topo, pairing=pairing)
        if fused is not None:
            gx, G, gy = fused
        else:
            if need_x:
                gx = k.bwd_x(y, w_half, g, topo, pairing)
            pairs = k.bwd_pairs(x, y, w_half, g, topo, pairing, need_gx=False) if (need_emb and need_y and _pair_backward_pays(g)) else None
            if pairs is not None:
                _, G, gy = pairs
                folded = True
            else:
                G, gy = k.bwd_edge(x, y, w_half, g, topo, need_gw=need_emb, need_gy=need_y, pairing=pairing)
        g_emb = None
        q = ctx.queue
        if q is not None:
            q.layers -= 1
        if need_emb and q is not None:
            cur = torch.cuda.current_stream(g.device)
            try:
                ready = torch.cuda.Event()
                ready.record(cur)  # grad_w is complete
                last = q.layers <= 0

                def launch():
                    if folded:
                        # the first layer's backward is the last launch of the queue: the main chain is waiting for it, so
                        # it has the device to itself (the other layers' launches run next to the main chain)
                        return _mlp._launch_bwd(emb_half, w0, w1, alpha0, alpha1, G, mode, cache, device_idle=last)
                    return _mlp._launch_bwd_paired(emb_half, w0, w1, alpha0, alpha1, G[:P], G[P:], mode, cache)

                q.submit(launch, ready, (G, emb_half, w0, w1), last)
            except BaseException:
                # never leave an un-joined fork behind (it would invalidate a hipGraph capture) nor a stale partial sum
                cur.wait_stream(q.stream)
                q.acc, q.layers, q.pending = None, 0, None
                raise
            if q.layers <= 0:  # first layer of the model = last backward of the evaluation: join
                assert q.acc is not None and q.layers == 0, "radial backward queue out of step with the layers"
                cur.wait_stream(q.stream)
                g_emb, q.acc = q.acc, None
                g_emb.record_stream(cur)
        elif need_emb and folded:
            g_emb = _mlp._launch_bwd(emb_half, w0, w1, alpha0, alpha1, G, mode, cache)
        elif need_emb:
            g_emb = _mlp._launch_bwd_paired(emb_half, w0, w1, alpha0, alpha1, G[:P], G[P:], mode, cache)
        return (g_emb, gx, gy) + (None,) * 10


class _PairRowsFn(torch.autograd.Function):
    """``emb[pairing.rep_edge]`` (``nqa_pair_gather``); backward: the per-pair gradient goes to the representative edge,
    zero to its reverse (``nqa_pair_expand``).  Differentiable again (the two maps are adjoint, both linear)."""

    @staticmethod
    def forward(ctx, emb, pairing: EdgePairing):
        from ._topology import _ptr, current_stream_ptr

        emb = emb.contiguous()
        assert emb.dim() == 2 and emb.element_size() == 4
        out = torch.empty((pairing.num_pairs, emb.shape[1]), dtype=emb.dtype, device=emb.device)
        with torch.cuda.device(emb.device):
            rc = _lib.load().nqa_pair_gather(_ptr(emb), _ptr(pairing.rep_edge), pairing.num_pairs, emb.shape[1],
                                             _ptr(out), current_stream_ptr(emb.device))
        _lib.check(rc, "nqa_pair_gather")
        ctx.pairing, ctx.num_edges = pairing, emb.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        return _PairExpandFn.apply(g, ctx.pairing, ctx.num_edges), None


class _PairExpandFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, pairing: EdgePairing, num_edges: int):
        from ._topology import _ptr, current_stream_ptr

        g = g.contiguous()
        out = torch.empty((num_edges, g.shape[1]), dtype=g.dtype, device=g.device)
        with torch.cuda.device(g.device):
            rc = _lib.load().nqa_pair_expand(_ptr(g), _ptr(pairing.rows), num_edges, pairing.num_pairs, g.shape[1],
                                             _ptr(out), current_stream_ptr(g.device))
        _lib.check(rc, "nqa_pair_expand")
        ctx.pairing = pairing
        return out

    @staticmethod
    def backward(ctx, c):
        return _PairRowsFn.apply(c, ctx.pairing), None, None


def pair_rows(emb: torch.Tensor, pairing: EdgePairing) -> torch.Tensor:
    return _PairRowsFn.apply(emb, pairing)


def available(edge_mlp, tp_scatter, x: torch.Tensor, emb: torch.Tensor) -> bool:
    """float32 GPU evaluation with the fused split-bf16 MLP and structure-specialised TP kernels."""
    if not x.is_cuda or x.dtype != torch.float32 or emb.dtype != torch.float32:
        return False
    from ..utils.tracing import traceable

    if traceable():
        return False  # pairing is decided from the data on the host
    if os.environ.get("NQA_NO_PAIRED", "") not in ("", "0"):
        return False
    if not edge_mlp._fused_ok(emb) or _mlp.radial_mlp_mode() != _lib.NQA_MLP_BF16X6:
        return False
    if tp_scatter.use_dispatcher_ops or tp_scatter.model_dtype != torch.float32:
        return False
    return tp_scatter._get_kernels().has_spec(torch.float32)


def paired_radial_tp(edge_mlp, tp_scatter, emb, x, edge_attr, topo: EdgeTopology, pairing: EdgePairing,
                     emb_half=None, queue=None):
    """``tp_scatter(x, edge_attr, edge_mlp(emb), ...)`` with the MLP evaluated on the pairs' representative edges
    (``emb_half = emb[pairing.rep_edge]``, shared by the layers of one evaluation when the caller passes it)."""
    cache = getattr(edge_mlp, "_weight_images", None)
    if cache is None:
        cache = edge_mlp._weight_images = _mlp._WeightImages()
    cache.validate(edge_mlp.mlp[2].weight)
    if emb_half is None:
        emb_half = pair_rows(emb, pairing)
    from ..utils.wgrad import differentiable_parameters

    if differentiable_parameters(edge_mlp.training, edge_mlp.mlp[0].weight, edge_mlp.mlp[2].weight):
        # training: the per-module twice-differentiable Functions, on P rows instead of E (the weight gradient of the
        # pair is the sum of its two halves, folded inside the tensor-product backward)
        w_half = edge_mlp(emb_half)
        return tp_scatter(x, edge_attr, w_half, topo._dst, topo._src, topology=topo, pairing=pairing)
    return _PairedRadialTPFn.apply(
        emb_half, x, edge_attr, edge_mlp.mlp[0].weight.detach(), edge_mlp.mlp[2].weight.detach(),
        edge_mlp._alphas[0], edge_mlp._alphas[1], _mlp.radial_mlp_mode(), cache, tp_scatter._get_kernels(), topo, pairing,
        queue,
    )
