"""Edge embedding modules backed by the fused HIP kernel ``nqa_edge_embed_fwd/bwd``.

Mirrors ``nequip/nn/embedding/_edge.py``: ``EdgeLengthNormalizer`` (:19-80), ``BesselEdgeLengthEncoding``
(:84-150) and ``SphericalHarmonicEdgeAttrs`` (:154-198) with the same constructor arguments, fields and
float64-in / model-dtype-out behaviour.  The real spherical harmonics (e3nn ``SphericalHarmonics(...,
normalize=True, normalization="component")``), the Bessel basis ``sinc(n x) n``, the polynomial cutoff and
the ``2 pi / r_max^2`` factor (``nequip/model/nequip_models.py:318-322``) are evaluated on the GPU in one
pass over the edges; their vector-Jacobian product feeds the force backward.
"""

from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Union

import torch

from ... import _lib
from ...utils import ktimer
from ...data import AtomicDataDict
from ...o3.irreps import Irreps
from .._graph_mixin import GraphModuleMixin
from .._topology import _ptr, current_stream_ptr
from ..utils import with_edge_vectors_
from ...utils.tracing import traceable


def _embed(edge_vec, bessel_weights, cfg):
    """The fused kernel as an autograd Function (eager) or as dispatcher ops (while tracing)."""
    if traceable():
        from ._edge_ops import edge_embed

        return edge_embed(edge_vec, bessel_weights, cfg)
    return _EdgeEmbedFn.apply(edge_vec, bessel_weights, cfg)

_GLOBAL_DTYPE = torch.float64  # nequip/utils/global_dtype.py:5


def _dt(dtype):
    return _lib.NQA_F32 if dtype == torch.float32 else _lib.NQA_F64


class _EdgeEmbedFn(torch.autograd.Function):
    """edge_vec [E,3] f64 -> (sh [E,S] | None, emb [E,nb] | None) in model dtype."""

    @staticmethod
    def forward(ctx, edge_vec, bessel_weights, cfg):
        if not edge_vec.is_cuda:
            raise RuntimeError("nequip_amd edge embedding runs on the GPU only (HIP kernel); no CPU fallback exists")
        assert edge_vec.dtype == torch.float64, "edge vectors must be float64 (nequip _GLOBAL_DTYPE)"
        lib = _lib.load()
        vec = edge_vec.contiguous()
        E = vec.shape[0]
        out_dtype = cfg["dtype"]
        lmax = cfg["lmax"]
        sh = torch.empty((E, (lmax + 1) ** 2), dtype=out_dtype, device=vec.device) if cfg["want_sh"] else None
        emb = torch.empty((E, cfg["nb"]), dtype=out_dtype, device=vec.device) if cfg["want_emb"] else None
        nbytes = E * (24 + out_dtype.itemsize * (((lmax + 1) ** 2 if cfg["want_sh"] else 0) + (cfg["nb"] if cfg["want_emb"] else 0)))
        with torch.cuda.device(vec.device), ktimer.region("edge_embed_fwd", nbytes):
            rc = lib.nqa_edge_embed_fwd(
                _dt(out_dtype), max(lmax, 0), _ptr(vec), E, cfg["rmax_recip"], _ptr(cfg.get("rmax_edge")), cfg["nb"],
                _ptr(bessel_weights), cfg["p"], cfg["factor"], _ptr(sh), _ptr(emb), ctypes.c_void_p(),
                current_stream_ptr(vec.device),
            )  # fmt: skip
        _lib.check(rc, "nqa_edge_embed_fwd")
        ctx.save_for_backward(vec, bessel_weights)
        ctx.cfg = cfg
        outs = tuple(t for t in (sh, emb) if t is not None)
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def backward(ctx, *grads):
        vec, bw = ctx.saved_tensors
        cfg = ctx.cfg
        grads = list(grads)
        g_sh = grads.pop(0) if cfg["want_sh"] else None
        g_emb = grads.pop(0) if cfg["want_emb"] else None
        return _EdgeEmbedBwdFn.apply(vec, bw, g_sh, g_emb, cfg), None, None


class _EdgeEmbedBwdFn(torch.autograd.Function):
    """(edge_vec, g_sh, g_emb) -> g_edge_vec = J^T g; differentiable once more through ``nqa_edge_embed_bwd_bwd``."""

    @staticmethod
    def forward(ctx, vec, bw, g_sh, g_emb, cfg):
        lib = _lib.load()
        g_sh = g_sh.contiguous() if g_sh is not None else None
        g_emb = g_emb.contiguous() if g_emb is not None else None
        E = vec.shape[0]
        g_vec = torch.empty((E, 3), dtype=torch.float64, device=vec.device)
        nbytes = E * (48 + cfg["dtype"].itemsize * (((cfg["lmax"] + 1) ** 2 if g_sh is not None else 0) + (cfg["nb"] if g_emb is not None else 0)))
        with torch.cuda.device(vec.device), ktimer.region("edge_embed_bwd", nbytes):
            rc = lib.nqa_edge_embed_bwd(
                _dt(cfg["dtype"]), max(cfg["lmax"], 0), _ptr(vec), E, cfg["rmax_recip"], _ptr(cfg.get("rmax_edge")),
                cfg["nb"], _ptr(bw), cfg["p"], cfg["factor"], _ptr(g_sh), _ptr(g_emb), _ptr(g_vec),
                current_stream_ptr(vec.device),
            )  # fmt: skip
        _lib.check(rc, "nqa_edge_embed_bwd")
        ctx.save_for_backward(vec, bw, g_sh, g_emb)
        ctx.cfg = cfg
        return g_vec

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, c):
        vec, bw, g_sh, g_emb = ctx.saved_tensors
        need_vec, _, need_gsh, need_gemb = ctx.needs_input_grad[:4]
        g_vec2, gg_sh, gg_emb = _edge_embed_second_order(vec, bw, g_sh, g_emb, c, ctx.cfg, need_vec, need_gsh, need_gemb)
        return g_vec2, None, gg_sh, gg_emb, None


def _edge_embed_second_order(vec, bw, g_sh, g_emb, c, cfg, need_vec: bool, need_gsh: bool, need_gemb: bool):
    """``nqa_edge_embed_bwd_bwd``: gradients of ``g_vec = J(v)^T g`` w.r.t. (v, g_sh, g_emb) for the cotangent ``c``."""
    lib = _lib.load()
    c = c.contiguous()
    g_sh = g_sh.contiguous() if g_sh is not None else None
    g_emb = g_emb.contiguous() if g_emb is not None else None
    E = vec.shape[0]
    g_vec2 = torch.empty((E, 3), dtype=torch.float64, device=vec.device) if need_vec else None
    gg_sh = torch.empty_like(g_sh) if (g_sh is not None and need_gsh) else None
    gg_emb = torch.empty_like(g_emb) if (g_emb is not None and need_gemb) else None
    with torch.cuda.device(vec.device):
        rc = lib.nqa_edge_embed_bwd_bwd(
            _dt(cfg["dtype"]), max(cfg["lmax"], 0), _ptr(vec), E, cfg["rmax_recip"], _ptr(cfg.get("rmax_edge")),
            cfg["nb"], _ptr(bw), cfg["p"], cfg["factor"], _ptr(g_sh), _ptr(g_emb), _ptr(c), _ptr(gg_sh),
            _ptr(gg_emb), _ptr(g_vec2), current_stream_ptr(vec.device),
        )  # fmt: skip
    _lib.check(rc, "nqa_edge_embed_bwd_bwd")
    return g_vec2, gg_sh, gg_emb


def cutoff_partialdict_to_tensor(partial_dict: Dict, type_names: List[str], r_max: float) -> torch.Tensor:
    """``{"H": 2.0, "C": {"H": 4.0}}`` -> ``[num_types, num_types]`` float64 cutoffs, rows = centre type, missing entries =
    ``r_max`` (semantics of ``nequip/nn/embedding/utils.py:16-85``)."""
    rows = []
    for centre in type_names:
        entry = partial_dict.get(centre, None)
        if entry is None:
            rows.append([float(r_max)] * len(type_names))
        elif isinstance(entry, (int, float)):
            rows.append([float(entry)] * len(type_names))
        else:
            rows.append([float(entry.get(other, r_max)) for other in type_names])
    table = torch.as_tensor(rows, dtype=_GLOBAL_DTYPE).contiguous()
    assert torch.all(table > 0), "per-edge-type cutoffs must be positive"  # (nequip/nn/embedding/utils.py)
    return table


class _EdgeEmbedPairedFn(torch.autograd.Function):
    """``edge_vec [E, 3] -> (sh [E, S], emb [E, nb], emb_pairs [P, nb])`` in one launch for a list with a reverse-edge pairing
    (``nqa_edge_embed_fwd_paired``): the per-pair rows are what ``pair_rows(emb, pairing)`` would gather, and the backward
    takes their cotangent as it comes out of the radial MLP's backward (``nqa_edge_embed_bwd_paired``) -- no ``pair_gather`` /
    ``pair_expand`` launches.  First order (eval mode)."""

    @staticmethod
    def forward(ctx, edge_vec, bessel_weights, cfg, pairing):
        if not edge_vec.is_cuda:
            raise RuntimeError("nequip_amd edge embedding runs on the GPU only (HIP kernel); no CPU fallback exists")
        assert edge_vec.dtype == torch.float64, "edge vectors must be float64 (nequip _GLOBAL_DTYPE)"
        lib = _lib.load()
        vec = edge_vec.contiguous()
        E, P = vec.shape[0], pairing.num_pairs
        out_dtype, lmax, nb = cfg["dtype"], cfg["lmax"], cfg["nb"]
        sh = torch.empty((E, (lmax + 1) ** 2), dtype=out_dtype, device=vec.device)
        emb = torch.empty((E, nb), dtype=out_dtype, device=vec.device)
        emb_pairs = torch.empty((P, nb), dtype=out_dtype, device=vec.device)
        nbytes = E * (24 + out_dtype.itemsize * ((lmax + 1) ** 2 + nb)) + P * nb * out_dtype.itemsize
        with torch.cuda.device(vec.device), ktimer.region("edge_embed_fwd", nbytes):
            rc = lib.nqa_edge_embed_fwd_paired(
                _dt(out_dtype), lmax, _ptr(vec), E, cfg["rmax_recip"], nb, _ptr(bessel_weights), cfg["p"], cfg["factor"],
                _ptr(pairing.rows), P, _ptr(sh), _ptr(emb), _ptr(emb_pairs), current_stream_ptr(vec.device),
            )  # fmt: skip
        _lib.check(rc, "nqa_edge_embed_fwd_paired")
        ctx.save_for_backward(vec, bessel_weights)
        ctx.cfg, ctx.pairing = cfg, pairing
        return sh, emb, emb_pairs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_sh, g_emb, g_pairs):
        vec, bw = ctx.saved_tensors
        cfg, pairing = ctx.cfg, ctx.pairing
        lib = _lib.load()
        g_sh = g_sh.contiguous() if g_sh is not None else None
        g_emb = g_emb.contiguous() if g_emb is not None else None
        g_pairs = g_pairs.contiguous() if g_pairs is not None else None
        E = vec.shape[0]
        g_vec = torch.empty((E, 3), dtype=torch.float64, device=vec.device)
        nbytes = E * (48 + cfg["dtype"].itemsize * ((cfg["lmax"] + 1) ** 2)) + pairing.num_pairs * cfg["nb"] * cfg["dtype"].itemsize
        with torch.cuda.device(vec.device), ktimer.region("edge_embed_bwd", nbytes):
            rc = lib.nqa_edge_embed_bwd_paired(
                _dt(cfg["dtype"]), cfg["lmax"], _ptr(vec), E, cfg["rmax_recip"], cfg["nb"], _ptr(bw), cfg["p"],
                cfg["factor"], _ptr(pairing.rows), pairing.num_pairs, _ptr(g_sh), _ptr(g_emb), _ptr(g_pairs), _ptr(g_vec),
                current_stream_ptr(vec.device),
            )  # fmt: skip
        _lib.check(rc, "nqa_edge_embed_bwd_paired")
        return g_vec, None, None, None


class EdgeLengthNormalizer(GraphModuleMixin, torch.nn.Module):
    """Holds ``1/r_max`` -- one number, or one per (centre type, neighbour type) with ``per_edge_type_cutoff``
    (``nequip/nn/embedding/_edge.py:19-80``); the product ``r * (1/r_max)`` itself is formed inside the fused radial kernel,
    which takes the per-edge reciprocal cutoffs as an ``[E]`` float64 operand (``rmax_recip_edge`` of ``nqa_edge_embed_*``)."""

    def __init__(self, r_max: float, type_names: List[str], per_edge_type_cutoff: Optional[Dict] = None,
                 edge_type_field: str = AtomicDataDict.EDGE_TYPE_KEY,
                 norm_length_field: str = AtomicDataDict.NORM_LENGTH_KEY, irreps_in=None):
        super().__init__()
        self.r_max = float(r_max)
        self.num_types = len(type_names)
        self.edge_type_field = edge_type_field
        self.norm_length_field = norm_length_field
        self._per_edge_type = per_edge_type_cutoff is not None
        if self._per_edge_type:
            table = cutoff_partialdict_to_tensor(per_edge_type_cutoff, list(type_names), self.r_max)
            assert float(table.max()) <= self.r_max + 1e-12, "per-edge-type cutoffs cannot exceed r_max"
            rmax_recip = table.reciprocal().view(-1)  # row-major (centre type, neighbour type)
            self.symmetric = bool(torch.equal(table, table.t()))
        else:
            rmax_recip = torch.as_tensor(1.0 / self.r_max, dtype=_GLOBAL_DTYPE)
            self.symmetric = True
        self.register_buffer("_rmax_recip", rmax_recip)
        irreps_out = {self.norm_length_field: Irreps("1x0e")}
        if self._per_edge_type:
            irreps_out[self.edge_type_field] = None
        self._init_irreps(irreps_in=irreps_in, irreps_out=irreps_out)

    def forward(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        data = with_edge_vectors_(data, with_lengths=False)
        data["_nqa_rmax_recip"] = 1.0 / self.r_max
        if self._per_edge_type:
            if self.edge_type_field not in data:  # with_edge_type_ (nequip/nn/utils.py:121-133)
                data[self.edge_type_field] = torch.index_select(
                    data[AtomicDataDict.ATOM_TYPE_KEY].reshape(-1), 0,
                    data[AtomicDataDict.EDGE_INDEX_KEY].reshape(-1)).view(2, -1)
            et = data[self.edge_type_field]
            data["_nqa_rmax_recip_edge"] = torch.index_select(self._rmax_recip, 0, et[0] * self.num_types + et[1])
        return data


class BesselEdgeLengthEncoding(GraphModuleMixin, torch.nn.Module):
    """``edge_embedding = sinc(n r/r_max) n * cutoff(r/r_max)`` (``nequip/nn/embedding/_edge.py:136-150``),
    optionally pre-multiplied by a constant ``factor`` (the reference applies ``2 pi / r_max^2`` in a separate
    ``ApplyFactor`` module, ``nequip_models.py:318-322``; here that module sets ``self.factor`` and becomes a no-op)."""

    def __init__(self, cutoff: torch.nn.Module, num_bessels: int = 8, trainable: bool = False,
                 edge_invariant_field: str = AtomicDataDict.EDGE_EMBEDDING_KEY,
                 norm_length_field: str = AtomicDataDict.NORM_LENGTH_KEY, irreps_in=None):
        super().__init__()
        self.cutoff = cutoff
        self.num_bessels = num_bessels
        self.trainable = trainable
        self.edge_invariant_field = edge_invariant_field
        self.norm_length_field = norm_length_field
        bessel_weights = torch.linspace(1.0, num_bessels, num_bessels, dtype=_GLOBAL_DTYPE).unsqueeze(0)
        if trainable:  # (nequip/nn/embedding/_edge.py:117-120)
            self.bessel_weights = torch.nn.Parameter(bessel_weights)
        else:
            self.register_buffer("bessel_weights", bessel_weights)
        self.factor = 1.0
        self._init_irreps(
            irreps_in=irreps_in,
            irreps_out={self.edge_invariant_field: Irreps([(num_bessels, (0, 1))])},
        )
        self._output_dtype = torch.get_default_dtype()

    def _forward_differentiable_weights(self, data, vec, rmax_edge):
        """Trainable Bessel roots WHILE they are being trained: the reference's ATen formulation (_edge.py:136-150,
        cutoffs.py:17-27), which autograd differentiates w.r.t. the roots to any order (force-matching training).  With
        constant roots -- eval mode, or ``trainable=False`` -- the fused kernel evaluates the same expression."""
        r = torch.linalg.norm(vec, dim=-1, keepdim=True)
        x = r * (rmax_edge.view(-1, 1) if rmax_edge is not None else float(data["_nqa_rmax_recip"]))
        bessel = (torch.sinc(x * self.bessel_weights) * self.bessel_weights).to(self._output_dtype)
        p = float(self.cutoff.p)
        cut = 1.0 - ((p + 1.0) * (p + 2.0) / 2.0) * torch.pow(x, p) + p * (p + 2.0) * torch.pow(x, p + 1.0) \
            - (p * (p + 1.0) / 2) * torch.pow(x, p + 2.0)
        cut = (cut * (x < 1.0)).to(self._output_dtype)
        emb = bessel * cut
        return emb * self.factor if self.factor != 1.0 else emb

    def forward(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        data = with_edge_vectors_(data, with_lengths=False)
        vec = data[AtomicDataDict.EDGE_VECTORS_KEY]
        rmax_edge = data.get("_nqa_rmax_recip_edge")
        pre = data.pop("_nqa_fused_edge_embedding", None)
        if pre is not None and pre[1] is self and rmax_edge is None:  # evaluated with the spherical harmonics (see there)
            data[self.edge_invariant_field] = pre[0]
            return data
        if self.trainable and self.bessel_weights.requires_grad and torch.is_grad_enabled() and self.training:
            data[self.edge_invariant_field] = self._forward_differentiable_weights(data, vec, rmax_edge)
            return data
        if rmax_edge is not None and traceable():
            raise NotImplementedError("per-edge-type cutoffs are not part of the dispatcher-op (compile) form of the edge "
                                      "embedding: evaluate this model eagerly")
        cfg = dict(dtype=self._output_dtype, lmax=0, want_sh=False, want_emb=True, nb=self.num_bessels,
                   rmax_recip=float(data["_nqa_rmax_recip"]), p=float(self.cutoff.p), factor=float(self.factor))
        if rmax_edge is not None:
            cfg["rmax_edge"] = rmax_edge.contiguous()
        data[self.edge_invariant_field] = _embed(vec, self.bessel_weights.detach().view(-1), cfg)
        return data


class SphericalHarmonicEdgeAttrs(GraphModuleMixin, torch.nn.Module):
    """``edge_attrs = Y(edge_vectors)`` for ``l = 0..lmax`` (``nequip/nn/embedding/_edge.py:154-198``)."""

    def __init__(self, irreps_edge_sh: Union[int, str, Irreps], edge_sh_normalization: str = "component",
                 edge_sh_normalize: bool = True, irreps_in=None, out_field: str = AtomicDataDict.EDGE_ATTRS_KEY):
        super().__init__()
        self.out_field = out_field
        if isinstance(irreps_edge_sh, int):
            self.irreps_edge_sh = Irreps.spherical_harmonics(irreps_edge_sh)
        else:
            self.irreps_edge_sh = Irreps(str(irreps_edge_sh))
        lmax = self.irreps_edge_sh.lmax
        if self.irreps_edge_sh != Irreps.spherical_harmonics(lmax):
            raise NotImplementedError("edge spherical harmonics must be the full range 0..lmax with parity (-1)^l")
        if edge_sh_normalization != "component" or not edge_sh_normalize:
            raise NotImplementedError("only the nequip defaults (normalize=True, 'component') are implemented")
        self.lmax = lmax
        self._init_irreps(irreps_in=irreps_in, irreps_out={out_field: self.irreps_edge_sh})
        self._output_dtype = torch.get_default_dtype()
        self.register_buffer("_dummy_bw", torch.ones(1, dtype=_GLOBAL_DTYPE), persistent=False)

    @staticmethod
    def _known_pairing(data, vec):
        """The reverse-edge pairing of this evaluation's edge list IF it is known without waiting (a cached topology whose
        verdict has been read: a static list, or any list once the first convolution asked) -- float32 eval evaluations
        differentiated w.r.t. positions only, as `GraphModel._start_pairing`."""
        K = AtomicDataDict
        if (vec.dtype != torch.float64 or K.POSITIONS_KEY not in data
                or os.environ.get("NQA_NO_PAIRED_EMBED", "") not in ("", "0")
                or os.environ.get("NQA_NO_PAIRED", "") not in ("", "0")):
            return None
        ei = data.get(K.EDGE_INDEX_KEY)
        if ei is None or not ei.is_cuda:
            return None
        from .._topology import topology_cache

        topo = topology_cache.get(ei[0], ei[1], data[K.POSITIONS_KEY].shape[0])
        return topo.pairing_if_known(data.get(K.EDGE_CELL_SHIFT_KEY))

    def forward(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        data = with_edge_vectors_(data, with_lengths=False)
        vec = data[AtomicDataDict.EDGE_VECTORS_KEY]
        fused = self.__dict__.get("_fuse_radial")
        if fused is not None and not traceable() and os.environ.get("NQA_NO_EMBED_FUSION", "") in ("", "0"):
            # one launch for the spherical harmonics AND the radial basis of the Bessel module further down the chain (set by
            # the model builder, `_plan_fusions`: plain r_max, constant roots), one launch for their joint backward -- the
            # kernel evaluates both in one pass over the edges anyway; saves a launch each way and the add of the two edge-vector
            # gradients.  The Bessel module picks its rows up from `data` (and evaluates itself if they are not there).
            bessel = fused[0]
            if (not (bessel.trainable and bessel.bessel_weights.requires_grad and torch.is_grad_enabled() and bessel.training)
                    and bessel._output_dtype == self._output_dtype):
                cfg = dict(dtype=self._output_dtype, lmax=self.lmax, want_sh=True, want_emb=True, nb=bessel.num_bessels,
                           rmax_recip=1.0 / float(fused[1].r_max), p=float(bessel.cutoff.p), factor=float(bessel.factor))
                pairing = None
                if not self.training and self._output_dtype == torch.float32:  # (first order only: eval mode)
                    pairing = self._known_pairing(data, vec)
                if pairing is not None:
                    # the list pairs up (verdict already on the host): the per-pair radial rows come out of the same launch
                    sh, emb, emb_pairs = _EdgeEmbedPairedFn.apply(vec, bessel.bessel_weights.detach().view(-1), cfg, pairing)
                    data["_nqa_edge_embedding_pairs"] = emb_pairs
                else:
                    sh, emb = _embed(vec, bessel.bessel_weights.detach().view(-1), cfg)
                data[self.out_field] = sh
                data["_nqa_fused_edge_embedding"] = (emb, bessel)
                return data
        cfg = dict(dtype=self._output_dtype, lmax=self.lmax, want_sh=True, want_emb=False, nb=0, rmax_recip=1.0,
                   p=6.0, factor=1.0)
        data[self.out_field] = _embed(vec, self._dummy_bw, cfg)
        return data
