"""Dispatcher-op form of the edge embedding: ``torch.ops.nequip_amd.edge_embed_fwd / _bwd / _bwd_bwd``.

Same role as ``nn/_tp_scatter_ops.py`` for the tensor-product scatter: the opaque, traceable form of the fused
``SphericalHarmonicEdgeAttrs`` / ``BesselEdgeLengthEncoding`` kernel (``nequip/nn/embedding/_edge.py:136-198``), with fake
kernels for shape propagation and autograd formulas that stay inside the family (the VJP is ``edge_embed_bwd``, its own
derivative ``edge_embed_bwd_bwd``), so first and second derivatives trace.  An output / cotangent that is not wanted is an
empty tensor.  CUDA only: the CPU key is not registered.
"""

from __future__ import annotations

import torch

_NS = "nequip_amd"
_lib_def = torch.library.Library(_NS, "FRAGMENT")
_CFG = "int lmax, bool want_sh, bool want_emb, int nb, float rmax_recip, float p, float factor, bool f32"
_lib_def.define(f"edge_embed_fwd(Tensor edge_vec, Tensor bessel_weights, {_CFG}) -> (Tensor, Tensor)")
_lib_def.define(f"edge_embed_bwd(Tensor edge_vec, Tensor bessel_weights, Tensor g_sh, Tensor g_emb, {_CFG}) -> Tensor")
_lib_def.define(f"edge_embed_bwd_bwd(Tensor edge_vec, Tensor bessel_weights, Tensor g_sh, Tensor g_emb, Tensor c, {_CFG}, "
                "bool need_vec, bool need_gsh, bool need_gemb) -> (Tensor, Tensor, Tensor)")


def _cfg(lmax, want_sh, want_emb, nb, rmax_recip, p, factor, f32):
    return dict(dtype=torch.float32 if f32 else torch.float64, lmax=lmax, want_sh=want_sh, want_emb=want_emb, nb=nb,
                rmax_recip=rmax_recip, p=p, factor=factor)


def _opt(t):
    return t if t.numel() > 0 else None


# ---- device implementations (the autograd Functions' kernels, called below the autograd key) --------------------------
def _fwd_cuda(edge_vec, bessel_weights, lmax, want_sh, want_emb, nb, rmax_recip, p, factor, f32):
    from ._edge import _EdgeEmbedFn

    cfg = _cfg(lmax, want_sh, want_emb, nb, rmax_recip, p, factor, f32)
    out = _EdgeEmbedFn.apply(edge_vec.detach(), bessel_weights, cfg)
    outs = list(out) if isinstance(out, tuple) else [out]
    empty = edge_vec.new_empty(0, dtype=cfg["dtype"])
    sh = outs.pop(0) if want_sh else empty
    emb = outs.pop(0) if want_emb else empty
    return sh, emb


def _bwd_cuda(edge_vec, bessel_weights, g_sh, g_emb, lmax, want_sh, want_emb, nb, rmax_recip, p, factor, f32):
    from ._edge import _EdgeEmbedBwdFn

    cfg = _cfg(lmax, want_sh, want_emb, nb, rmax_recip, p, factor, f32)
    return _EdgeEmbedBwdFn.apply(edge_vec.detach().contiguous(), bessel_weights, _opt(g_sh.detach()), _opt(g_emb.detach()), cfg)


def _bwd_bwd_cuda(edge_vec, bessel_weights, g_sh, g_emb, c, lmax, want_sh, want_emb, nb, rmax_recip, p, factor, f32,
                  need_vec, need_gsh, need_gemb):
    from ._edge import _edge_embed_second_order

    cfg = _cfg(lmax, want_sh, want_emb, nb, rmax_recip, p, factor, f32)
    g_vec2, gg_sh, gg_emb = _edge_embed_second_order(edge_vec.contiguous(), bessel_weights, _opt(g_sh), _opt(g_emb), c,
                                                     cfg, need_vec, need_gsh, need_gemb)
    e64 = edge_vec.new_empty(0)
    em = edge_vec.new_empty(0, dtype=cfg["dtype"])
    return (g_vec2 if g_vec2 is not None else e64, gg_sh if gg_sh is not None else em,
            gg_emb if gg_emb is not None else em)


_lib_def.impl("edge_embed_fwd", _fwd_cuda, "CUDA")
_lib_def.impl("edge_embed_bwd", _bwd_cuda, "CUDA")
_lib_def.impl("edge_embed_bwd_bwd", _bwd_bwd_cuda, "CUDA")


# ---- fake kernels ---------------------------------------------------------------------------------------------------
@torch.library.register_fake(f"{_NS}::edge_embed_fwd")
def _fwd_fake(edge_vec, bessel_weights, lmax, want_sh, want_emb, nb, rmax_recip, p, factor, f32):
    torch._check(edge_vec.dim() == 2 and edge_vec.shape[1] == 3, lambda: "edge_vec must be [E, 3]")
    dt = torch.float32 if f32 else torch.float64
    E = edge_vec.shape[0]
    sh = edge_vec.new_empty((E, (lmax + 1) ** 2), dtype=dt) if want_sh else edge_vec.new_empty(0, dtype=dt)
    emb = edge_vec.new_empty((E, nb), dtype=dt) if want_emb else edge_vec.new_empty(0, dtype=dt)
    return sh, emb


@torch.library.register_fake(f"{_NS}::edge_embed_bwd")
def _bwd_fake(edge_vec, bessel_weights, g_sh, g_emb, lmax, want_sh, want_emb, nb, rmax_recip, p, factor, f32):
    return edge_vec.new_empty((edge_vec.shape[0], 3), dtype=torch.float64)


@torch.library.register_fake(f"{_NS}::edge_embed_bwd_bwd")
def _bwd_bwd_fake(edge_vec, bessel_weights, g_sh, g_emb, c, lmax, want_sh, want_emb, nb, rmax_recip, p, factor, f32,
                  need_vec, need_gsh, need_gemb):
    e = edge_vec.new_empty(0)
    return (torch.empty_like(c) if need_vec else e, torch.empty_like(g_sh) if (need_gsh and g_sh.numel() > 0) else g_sh.new_empty(0),
            torch.empty_like(g_emb) if (need_gemb and g_emb.numel() > 0) else g_emb.new_empty(0))


# ---- autograd: fwd -> bwd -> bwd_bwd ----------------------------------------------------------------------------------
def _fwd_setup(ctx, inputs, output):
    edge_vec, bw, *cfg = inputs
    ctx.save_for_backward(edge_vec, bw)
    ctx.cfg = tuple(cfg)
    ctx.set_materialize_grads(False)


def _fwd_backward(ctx, g_sh, g_emb):
    edge_vec, bw = ctx.saved_tensors
    lmax, want_sh, want_emb, nb, rr, p, factor, f32 = ctx.cfg
    dt = torch.float32 if f32 else torch.float64
    empty = edge_vec.new_empty(0, dtype=dt)
    g_sh = g_sh if (want_sh and g_sh is not None) else empty
    g_emb = g_emb if (want_emb and g_emb is not None) else empty
    g_vec = None
    if ctx.needs_input_grad[0] and (g_sh.numel() > 0 or g_emb.numel() > 0):
        g_vec = torch.ops.nequip_amd.edge_embed_bwd(edge_vec, bw, g_sh, g_emb, *ctx.cfg)
    return (g_vec, None) + (None,) * 8


torch.library.register_autograd(f"{_NS}::edge_embed_fwd", _fwd_backward, setup_context=_fwd_setup)


def _bwd_setup(ctx, inputs, output):
    edge_vec, bw, g_sh, g_emb, *cfg = inputs
    ctx.save_for_backward(edge_vec, bw, g_sh, g_emb)
    ctx.cfg = tuple(cfg)
    ctx.set_materialize_grads(False)


def _bwd_backward(ctx, c):
    if c is None:
        return (None,) * 12
    edge_vec, bw, g_sh, g_emb = ctx.saved_tensors
    need_vec, _, need_gsh, need_gemb = ctx.needs_input_grad[:4]
    g_vec2, gg_sh, gg_emb = torch.ops.nequip_amd.edge_embed_bwd_bwd(edge_vec, bw, g_sh, g_emb, c, *ctx.cfg, need_vec,
                                                                    need_gsh, need_gemb)
    return (g_vec2 if need_vec else None, None, gg_sh if (need_gsh and g_sh.numel() > 0) else None,
            gg_emb if (need_gemb and g_emb.numel() > 0) else None) + (None,) * 8


torch.library.register_autograd(f"{_NS}::edge_embed_bwd", _bwd_backward, setup_context=_bwd_setup)


def edge_embed(edge_vec, bessel_weights, cfg):
    """Dispatcher-op counterpart of ``_EdgeEmbedFn.apply(edge_vec, bessel_weights, cfg)`` (same return convention)."""
    sh, emb = torch.ops.nequip_amd.edge_embed_fwd(
        edge_vec, bessel_weights, int(cfg["lmax"]), bool(cfg["want_sh"]), bool(cfg["want_emb"]), int(cfg["nb"]),
        float(cfg["rmax_recip"]), float(cfg["p"]), float(cfg["factor"]), cfg["dtype"] == torch.float32)
    outs = tuple(t for t, want in ((sh, cfg["want_sh"]), (emb, cfg["want_emb"])) if want)
    return outs if len(outs) > 1 else outs[0]
