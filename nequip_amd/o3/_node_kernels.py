"""Host side of the fused node kernels (``nqa_node_linear`` / ``nqa_gate``, ``nequip_amd/csrc/node_ops.hip``).

Builds the chunk / instruction tables from irreps bookkeeping and wraps the launches in autograd Functions:

* ``node_linear(x, Wp, types, meta, addend)``: every per-irrep channel-mixing matrix of an e3nn ``o3.Linear`` (or of
  the type-pre-contracted self-connection) in ONE launch; the gradient w.r.t. ``x`` is the same kernel run on
  transposed tables; the gradient w.r.t. the packed weights (training only) is formed with a few small einsums.
* ``gate(x, meta)``: e3nn ``Gate`` forward / backward in one launch each.
"""

from __future__ import annotations

import contextlib
import ctypes
import os
import struct
import weakref
from typing import List, Optional, Sequence, Tuple

import torch

from .. import _lib
from ..utils import constcache as _constcache
from ..utils import ktimer
from ..utils import wgrad as _wgrad
from .irreps import Irreps


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _dt(dtype):
    if dtype == torch.float32:
        return _lib.NQA_F32
    if dtype == torch.float64:
        return _lib.NQA_F64
    raise RuntimeError(f"node kernels support float32/float64, got {dtype}")


class NodeLinearMeta:
    """Tables for ``out = sum_{(i -> o)} x_i @ W_(i,o)`` over irreps blocks, forward and transposed (backward)."""

    def __init__(self, irreps_in: Irreps, irreps_out: Irreps, instructions: Sequence[Tuple[int, int]]):
        self.irreps_in, self.irreps_out = Irreps(irreps_in), Irreps(irreps_out)
        self.instructions = [(int(i), int(o)) for i, o in instructions]
        self.din, self.dout = self.irreps_in.dim, self.irreps_out.dim
        in_off, out_off = self.irreps_in.offsets(), self.irreps_out.offsets()
        # forward weights: instruction order, each [mul_in, mul_out] row-major
        self.w_off, off = [], 0
        for i, o in self.instructions:
            self.w_off.append(off)
            off += self.irreps_in[i].mul * self.irreps_out[o].mul
        self.wstride = off
        # transposed weights: same instruction order, each [mul_out, mul_in]
        self.fwd = self._tables(self.irreps_out, out_off, in_off, self.irreps_in, True, 64)
        self.bwd = self._tables(self.irreps_in, in_off, out_off, self.irreps_out, False, 64)
        self._host = {}

    def _tables(self, side_out: Irreps, off_out, off_in, side_in: Irreps, by_out: bool, width: int):
        chunks: List[Tuple[int, ...]] = []
        instr: List[Tuple[int, ...]] = []
        for b, (mul_o, ir) in enumerate(side_out):
            if mul_o == 0:
                continue
            begin = len(instr)
            for k, (i, o) in enumerate(self.instructions):
                tgt, srcb = (o, i) if by_out else (i, o)
                if tgt == b:
                    instr.append((off_in[srcb], side_in[srcb].mul, self.w_off[k], 0))
            end = len(instr)
            for c0 in range(0, mul_o, width):
                chunks.append((off_out[b], ir.dim, mul_o, c0, begin, end, width, 0))
        return chunks, instr

    def host_tables(self, which: str):
        """(chunk bytes, n_chunks, instr bytes, n_instr): the tables are passed by host pointer (kernel arguments)."""
        if which not in self._host:
            chunks, instr = getattr(self, which)
            cb = b"".join(struct.pack("<8i", *c) for c in chunks)
            ib = b"".join(struct.pack("<4i", *i) for i in instr)
            self._host[which] = (ctypes.create_string_buffer(cb, max(len(cb), 1)), len(chunks),
                                 ctypes.create_string_buffer(ib, max(len(ib), 1)), len(instr))
        return self._host[which]

    def wgrad_table(self) -> "_wgrad.WgradTable":
        """Records of ``nqa_wgrad`` for the packed forward weights: A = input rows, B = output-gradient rows."""
        if getattr(self, "_wgrad_table", None) is None:
            in_off, out_off = self.irreps_in.offsets(), self.irreps_out.offsets()
            recs = []
            for (i, o), off in zip(self.instructions, self.w_off):
                mi, ir = self.irreps_in[i]
                recs.append((in_off[i], out_off[o], mi, self.irreps_out[o].mul, ir.dim, off))
            self._wgrad_table = _wgrad.WgradTable(recs, self.wstride)
        return self._wgrad_table

    def transpose_weights(self, wp: torch.Tensor) -> torch.Tensor:
        """[T, wstride] packed forward weights -> packed transposed weights with the same offsets (one gather with a
        precomputed permutation instead of a transpose + copy per instruction)."""
        key = str(wp.device)
        perm = self._perm.get(key) if hasattr(self, "_perm") else None
        if perm is None:
            idx = []
            for (i, o), off in zip(self.instructions, self.w_off):
                mi, mo = self.irreps_in[i].mul, self.irreps_out[o].mul
                idx.append((torch.arange(mi * mo).view(mi, mo).t().reshape(-1) + off))
            perm = (torch.cat(idx) if idx else torch.zeros(0, dtype=torch.long)).to(wp.device)
            if not hasattr(self, "_perm"):
                self._perm = {}
            self._perm[key] = perm
        return wp.index_select(1, perm)


def exact_fp32() -> bool:
    """``NQA_NODE_EXACT_FP32=1``: the channel-mixing products on the fp32 MFMA (bitwise an fma chain) instead of split-bf16
    operands on the bf16 MFMA (fp32-accurate: six partial products per fp32 product, as the radial MLP's default mode)."""
    return os.environ.get("NQA_NODE_EXACT_FP32", "") not in ("", "0")


def packed_weights(wp: torch.Tensor, meta: NodeLinearMeta, which: str) -> torch.Tensor:
    """``wp [T, wstride]`` split into 16-bit planes in MFMA-fragment order (``nqa_node_weights_pack``: two fp16 planes of
    power-of-two scaled K blocks by default, three bf16 planes with ``NQA_NODE_F16=0``).  Constant
    weights (eval mode: the modules keep ``wp`` alive across steps) are packed once per version of the tensor; a ``wp``
    that is part of an autograd graph (training: rebuilt from the parameter every step) is packed per call."""
    key = ("node_packed", id(meta), which, os.environ.get("NQA_NODE_F16", ""))  # (the layout depends on the split in use)
    # constants (eval mode: the module keeps `wp`; a compiled graph: a constant buffer) are packed once per version of the
    # tensor -- the cache is keyed on the storage (utils/constcache.py); a graph tensor of a training step lives for that
    # step only and is packed per call when NQA_NODE_TRAIN_PACKED=1 asks for it at all
    if wp.requires_grad or getattr(wp, "_nqa_volatile", False):
        return _pack(wp, meta, which)
    return _constcache.get(wp, key, lambda: _pack(wp, meta, which))


def _pack(wp: torch.Tensor, meta: NodeLinearMeta, which: str) -> torch.Tensor:
    lib = _lib.load()
    ct, nchunks, it, ninstr = meta.host_tables(which)
    T = wp.shape[0]
    nbytes = lib.nqa_node_weights_pack_bytes(ctypes.cast(ct, ctypes.c_void_p), nchunks, ctypes.cast(it, ctypes.c_void_p),
                                             ninstr, T)
    if nbytes < 0:
        raise RuntimeError("nqa_node_weights_pack_bytes: inconsistent tables")
    w = wp.detach().contiguous()
    out = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=wp.device)
    with torch.cuda.device(wp.device):
        rc = lib.nqa_node_weights_pack(_ptr(w), ctypes.cast(ct, ctypes.c_void_p), nchunks,
                                       ctypes.cast(it, ctypes.c_void_p), ninstr, T, w.shape[1], _ptr(out),
                                       _stream(wp.device))
    _lib.check(rc, "nqa_node_weights_pack")
    return out


def train_packed() -> bool:
    """Opt-in (NQA_NODE_TRAIN_PACKED=1): pack the weights of a training step too.  Measured on train256: the packed kernel
    is no faster than the exact one at 8192 atoms (0.95 vs 0.94 ms per step) and the 18 pack launches cost 0.25 ms."""
    return os.environ.get("NQA_NODE_TRAIN_PACKED", "") == "1"


def _launch_linear(x, wp, addend, types, meta: NodeLinearMeta, which: str, scale: float):
    lib = _lib.load()
    width = 64  # 64-channel chunks: float32 on the MFMA kernels, float64 on the VALU kernel
    ct, nchunks, it, ninstr = meta.host_tables(which)
    din, dout = (meta.din, meta.dout) if which == "fwd" else (meta.dout, meta.din)
    N = x.shape[0]
    out = torch.empty((N, dout), dtype=x.dtype, device=x.device)
    flops = 2.0 * N * sum(c[1] * min(64, c[2] - c[3]) * sum(meta_i[1] for meta_i in (meta.fwd if which == "fwd" else meta.bwd)[1][c[4]:c[5]]) for c in (meta.fwd if which == "fwd" else meta.bwd)[0])
    # (weights that are part of an autograd graph -- training -- change every step: the exact-fp32 kernel reads them as
    # they are; see train_packed())
    # a typed map (one weight set per atom type) walks the atoms grouped by type: a work unit then runs the stages of the
    # one or two types it holds instead of one masked pass per type (float32 MFMA kernels; any order gives the same result)
    order = None
    if (wp.shape[0] > 1 and types is not None and x.dtype == torch.float32
            and os.environ.get("NQA_NODE_TYPE_ORDER", "") != "0"):
        order = type_order(types)
    if (x.dtype == torch.float32 and not exact_fp32() and ninstr > 0
            and (train_packed() or not (wp.requires_grad or getattr(wp, "_nqa_volatile", False)))):
        wf = packed_weights(wp, meta, which)
        with torch.cuda.device(x.device), ktimer.region("node_linear", x.element_size() * N * (din + dout), flops):
            rc = lib.nqa_node_linear_packed_ordered(
                _ptr(x), _ptr(wf), _ptr(addend), _ptr(out), _ptr(types), _ptr(order),
                ctypes.cast(ct, ctypes.c_void_p), nchunks, ctypes.cast(it, ctypes.c_void_p), ninstr, wp.shape[0], din,
                dout, N, float(scale), _stream(x.device),
            )  # fmt: skip
        _lib.check(rc, "nqa_node_linear_packed_ordered")
        return out
    with torch.cuda.device(x.device), ktimer.region("node_linear", x.element_size() * N * (din + dout), flops):
        rc = lib.nqa_node_linear_ordered(
            _dt(x.dtype), _ptr(x), _ptr(wp), _ptr(addend), _ptr(out), _ptr(types), _ptr(order),
            ctypes.cast(ct, ctypes.c_void_p), nchunks, ctypes.cast(it, ctypes.c_void_p), ninstr,
            wp.shape[0], wp.shape[1], din, dout, N, float(scale), width, _stream(x.device),
        )  # fmt: skip
    _lib.check(rc, "nqa_node_linear_ordered")
    return out


class _NodeLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wp, addend, types, meta: NodeLinearMeta, scale: float):
        x = x.contiguous()
        wp = wp.contiguous()
        if addend is not None:
            addend = addend.contiguous()
        out = _launch_linear(x, wp, addend, types, meta, "fwd", scale)
        ctx.save_for_backward(x, wp, types)
        ctx.meta, ctx.scale, ctx.has_addend = meta, scale, addend is not None
        ctx.param_side = _wgrad.is_param_side(wp)  # the consumers of grad(wp) are nodes of the parameter-side stream
        ctx.param_link = getattr(wp, "_nqa_param", None)  # (parameter, scale vector, wp is the transposed copy)
        return out

    @staticmethod
    def backward(ctx, g):
        x, wp, types = ctx.saved_tensors
        meta: NodeLinearMeta = ctx.meta
        gx = gwp = gadd = None
        link = ctx.param_link
        if (ctx.needs_input_grad[1] and link is not None and wp.shape[0] == 1 and _wgrad.deferring()
                and _wgrad.supported(x, g)):
            # deferred parameter gradients (utils/wgrad.py): product, sum and scalings on the side stream, straight into the
            # parameter's bucket; autograd gets nothing for `wp` from this node
            param, scale_vec, transposed = link
            scale = ctx.scale

            def make():
                gw = _weight_grad(x, g, types, meta, 1)
                if transposed:  # this node ran on the transposed copy: back to the layout of the parameter
                    gw = meta.transpose_weights(gw)
                gw = gw.view(-1) * scale_vec
                return gw * scale if scale != 1.0 else gw

            _wgrad.defer(param, make, x, g)
        elif ctx.needs_input_grad[1] and _wgrad.param_grads_wanted():
            # first-order parameter gradient of a parameter-side weight tensor: on the parameter-side stream, next to the data
            # chain (no join here: its consumers are nodes of that stream, utils/wgrad.py)
            side = ctx.param_side and not torch.is_grad_enabled() and _wgrad.supported(x, g)
            with (_wgrad.parameter_side(x.device, x, g, types) if side else contextlib.nullcontext()):
                gwp = _weight_grad(x, g, types, meta, wp.shape[0])
                if ctx.scale != 1.0:
                    gwp = gwp * ctx.scale
        if ctx.needs_input_grad[0]:
            gx = _NodeLinearFn.apply(g, meta_transposed_weights(meta, wp), None, types, _transposed(meta), ctx.scale)
        if ctx.has_addend and ctx.needs_input_grad[2]:
            gadd = g
        return gx, gwp, gadd, None, None, None


_TRANSPOSED_CACHE = {}


def _transposed(meta: NodeLinearMeta) -> NodeLinearMeta:
    """The adjoint map as a NodeLinearMeta of its own (so that double backward closes on the same kernel)."""
    if getattr(meta, "_adjoint_of", None) is not None:
        return meta._adjoint_of
    key = id(meta)
    if key not in _TRANSPOSED_CACHE:
        mt = NodeLinearMeta(meta.irreps_out, meta.irreps_in, [(o, i) for i, o in meta.instructions])
        mt._adjoint_of = meta
        _TRANSPOSED_CACHE[key] = mt
    return _TRANSPOSED_CACHE[key]


def meta_transposed_weights(meta: NodeLinearMeta, wp: torch.Tensor) -> torch.Tensor:
    if wp.requires_grad:
        # one transposed copy per step: the force pass (create_graph) makes it, the loss backward finds it on the step's
        # weight tensor; the adjoint of the adjoint is the tensor it was made from (same meta, see _transposed)
        src = getattr(wp, "_nqa_adjoint_src", None)
        if src is not None and src[0] is meta:
            orig = src[1]()  # (weak: no reference cycle between the two copies)
            if orig is not None:
                return orig
        hit = getattr(wp, "_nqa_step_transposed", None)
        if hit is not None and hit[0] is meta and (hit[1].requires_grad or not torch.is_grad_enabled()):
            return hit[1]
        if _wgrad.is_param_side(wp):  # (the transposed copy of a parameter-side tensor is one too)
            with _wgrad.parameter_side(wp.device):
                wt = meta.transpose_weights(wp)
            _wgrad.publish(wp.device, wt)
        else:
            wt = meta.transpose_weights(wp)
        link = getattr(wp, "_nqa_param", None)
        if link is not None:
            wt._nqa_param = (link[0], link[1], not link[2])
        # inside a backward pass that builds no graph this copy does not require grad although it changes every step
        wt._nqa_volatile = True
        wt._nqa_adjoint_src = (_transposed(meta), weakref.ref(wp))
        wp._nqa_step_transposed = (meta, wt)
        return wt
    # constant weights (eval mode / a compiled graph's constant): one transposed copy per version of the tensor
    return _constcache.get(wp, ("node_transposed", id(meta)), lambda: meta.transpose_weights(wp).contiguous())


def _weight_grad(x, g, types, meta: NodeLinearMeta, T: int):
    """d/dWp of sum(out * g): per instruction  gW[t, u, w] = sum_{z: type z = t} sum_m x[z,u,m] g[z,w,m].

    First-order training backward (no graph requested), float32: one ``nqa_wgrad`` launch for all instructions and atom
    types; otherwise (float64, or a differentiable result is asked for) the einsum formulation below."""
    if not torch.is_grad_enabled() and _wgrad.supported(x, g):
        return _wgrad.wgrad(x, g.contiguous(), meta.wgrad_table(), types, T)
    Z = x.shape[0]
    in_off, out_off = meta.irreps_in.offsets(), meta.irreps_out.offsets()
    parts = []
    onehot = None
    if T > 1:
        onehot = torch.nn.functional.one_hot(types.view(-1), T).to(x.dtype)
    for i, o in meta.instructions:
        mi, ir = meta.irreps_in[i]
        mo = meta.irreps_out[o].mul
        d = ir.dim
        xb = x[:, in_off[i] : in_off[i] + mi * d].reshape(Z, mi, d)
        gb = g[:, out_off[o] : out_off[o] + mo * d].reshape(Z, mo, d)
        if T == 1:
            parts.append(torch.einsum("zum,zwm->uw", xb, gb).reshape(1, mi * mo))
        else:
            parts.append(torch.einsum("zt,zum,zwm->tuw", onehot, xb, gb).reshape(T, mi * mo))
    return torch.cat(parts, dim=1) if len(parts) > 1 else parts[0]


def node_linear(x, wp, types, meta: NodeLinearMeta, addend=None, scale: float = 1.0):
    return _NodeLinearFn.apply(x, wp, addend, types, meta, scale)


# ---- Gate --------------------------------------------------------------------------------------------------------
_ACT_IDS = {"identity": 0, "silu": 1, "tanh": 2}


class GateMeta:
    """Column tables of ``nqa_gate`` (see include/nequip_amd.h): one 32-byte record per output (forward) / input
    (backward) column."""

    def __init__(self, irreps_scalars: Irreps, act_scalars: Sequence[Tuple[str, float]], irreps_gates: Irreps,
                 act_gates: Sequence[Tuple[str, float]], irreps_gated: Irreps):
        self.ns, self.ng = irreps_scalars.dim, irreps_gates.dim
        self.din = self.ns + self.ng + irreps_gated.dim
        self.dout = self.ns + irreps_gated.dim
        col_act = []  # (act id, cst) of every scalar / gate column
        for (mul, _), (name, cst) in list(zip(irreps_scalars, act_scalars)) + list(zip(irreps_gates, act_gates)):
            col_act += [(_ACT_IDS[name], float(cst))] * mul
        fwd = [None] * self.dout
        bwd = [None] * self.din
        rec = struct.Struct("<iiiidii")
        for c in range(self.ns):
            act, cst = col_act[c]
            fwd[c] = rec.pack(c, -1, act, 0, cst, 0, 0)
            bwd[c] = rec.pack(0, act, c, 0, cst, 0, 0)
        in_off, out_off, goff = self.ns + self.ng, self.ns, 0
        for mul, ir in irreps_gated:
            d = ir.dim
            for u in range(mul):
                gcol = self.ns + goff + u
                act, cst = col_act[gcol]
                assert d <= 9, "nqa_gate unrolls the gated components up to 2 l + 1 = 9 (l <= 4)"
                bwd[gcol] = rec.pack(1, act, out_off + u * d, in_off + u * d, cst, d, 0)
                for m in range(d):
                    fwd[out_off + u * d + m] = rec.pack(in_off + u * d + m, gcol, act, 0, cst, 0, 0)
                    bwd[in_off + u * d + m] = rec.pack(2, act, out_off + u * d + m, 0, cst, 0, gcol)
            in_off += mul * d
            out_off += mul * d
            goff += mul
        # block form (nqa_gate_block, the fused node stage): (out_off, d, mul, val_off, gate_off, act, cst) per output block
        self.blocks = []
        off = 0
        for (mul, _), (name, cst) in zip(irreps_scalars, act_scalars):
            if mul > 0:
                self.blocks.append((off, 1, mul, off, -1, _ACT_IDS[name], float(cst)))
            off += mul
        b_in, b_out, b_gate = self.ns + self.ng, self.ns, self.ns
        for (mul, ir), (name, cst) in zip(irreps_gated, act_gates):
            if mul > 0:
                self.blocks.append((b_out, ir.dim, mul, b_in, b_gate, _ACT_IDS[name], float(cst)))
            b_in += mul * ir.dim
            b_out += mul * ir.dim
            b_gate += mul
        self._blocks_c = None
        zero = rec.pack(3, 0, 0, 0, 1.0, 0, 0)
        self._fwd = b"".join(r if r is not None else rec.pack(0, -1, 0, 0, 1.0, 0, 0) for r in fwd)
        self._bwd = b"".join(r if r is not None else zero for r in bwd)
        self._dev = {}

    def blocks_c(self):
        """The blocks as a ctypes array of ``nqa_gate_block`` (host memory, kept alive by this object)."""
        if self._blocks_c is None:
            arr = (_lib.GateBlock * max(len(self.blocks), 1))()
            for k, b in enumerate(self.blocks):
                arr[k] = _lib.GateBlock(*b)
            self._blocks_c = arr
        return self._blocks_c, len(self.blocks)

    def fusable(self) -> bool:
        """What ``nqa_node_fused`` asks of a gate: gated blocks with d >= 3, everything a multiple of 4."""
        for out_off, d, mul, val_off, gate_off, act, cst in self.blocks:
            if gate_off >= 0 and d < 3:
                return False
            if (out_off | mul | val_off | max(gate_off, 0)) & 3:
                return False
        return (self.din & 3) == 0 and (self.dout & 3) == 0

    def device_tables(self, device):
        key = str(device)
        if key not in self._dev:
            ft = torch.frombuffer(bytearray(self._fwd), dtype=torch.uint8).clone().to(device)
            bt = torch.frombuffer(bytearray(self._bwd), dtype=torch.uint8).clone().to(device)
            self._dev[key] = (ft, bt)
        return self._dev[key]


def _launch_gate(x, gout, meta: GateMeta, mode: int, cot=None):
    """mode 0: forward; 1: grad_in(x, gout); 2 / 3: gradient of <cot, grad_in> w.r.t. gout / x (second order)."""
    lib = _lib.load()
    ft, bt = meta.device_tables(x.device)
    N = x.shape[0]
    out = torch.empty((N, meta.din if mode in (1, 3) else meta.dout), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device), ktimer.region("gate", x.element_size() * N * (meta.din + meta.dout)):
        rc = lib.nqa_gate(_dt(x.dtype), mode, _ptr(x), _ptr(gout), _ptr(cot), _ptr(out),
                          _ptr(bt if mode in (1, 3) else ft), meta.din, meta.dout, N, _stream(x.device))
    _lib.check(rc, "nqa_gate")
    return out


class _GateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, meta: GateMeta):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.meta = meta
        return _launch_gate(x, None, meta, 0)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return _GateBwdFn.apply(x, g, ctx.meta), None


class _GateBwdFn(torch.autograd.Function):
    """(x, grad_out) -> grad_in; differentiable once more (force-matching training)."""

    @staticmethod
    def forward(ctx, x, g, meta: GateMeta):
        g = g.contiguous()
        ctx.save_for_backward(x, g)
        ctx.meta = meta
        return _launch_gate(x, g, meta, 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, c):
        x, g = ctx.saved_tensors
        c = c.contiguous()
        gx = _launch_gate(x, g, ctx.meta, 3, cot=c) if ctx.needs_input_grad[0] else None
        gg = _launch_gate(x, None, ctx.meta, 2, cot=c) if ctx.needs_input_grad[1] else None
        return gx, gg, None


def gate(x, meta: GateMeta):
    return _GateFn.apply(x, meta)


def apply_deferred_gate(pregate):
    """The gate a ``ConvNetLayer`` left to its consumer (``data["_nqa_pregate"] = (h, meta, op key)``), for a consumer whose
    fused form does not apply: the gate kernel, as a dispatcher op while a tracer follows the model."""
    from ..utils.tracing import traceable

    h, meta, key = pregate
    if traceable():
        from . import _node_ops

        return _node_ops.gate_op(h, key)
    return gate(h, meta)


# ---- fused node stage (nqa_node_fused): Gate folded into its consumers / into the producers of its gradient -------------
def fusion_enabled() -> bool:
    """The fused node stage needs the default fp16-split packing; ``NQA_NO_NODE_FUSION=1`` keeps the separate launches."""
    return (os.environ.get("NQA_NO_NODE_FUSION", "") in ("", "0") and os.environ.get("NQA_NODE_F16", "") != "0"
            and not exact_fp32())


class FusedPart:
    """One operand set + destination of a fused launch: ``out = scale * x @ W(meta.fwd tables)`` (+ addend)."""

    def __init__(self, x, wp, meta: NodeLinearMeta, scale: float = 1.0, in_gate: Optional[GateMeta] = None,
                 accumulate: bool = False, addend=None):
        self.x, self.wp, self.meta, self.scale = x, wp, meta, float(scale)
        self.in_gate, self.accumulate, self.addend = in_gate, accumulate, addend


def _fill_part(cp, part: FusedPart, out, keep: list):
    lib_tables = part.meta.host_tables("fwd")
    ct, nchunks, it, ninstr = lib_tables
    wf = packed_weights(part.wp, part.meta, "fwd")
    keep.append(wf)
    cp.x = part.x.data_ptr()
    cp.packed = wf.data_ptr()
    cp.chunk_table = ctypes.cast(ct, ctypes.c_void_p)
    cp.instr_table = ctypes.cast(it, ctypes.c_void_p)
    cp.n_chunks, cp.n_instr, cp.n_types = nchunks, ninstr, part.wp.shape[0]
    cp.dim_in = part.x.shape[1]
    cp.out = out.data_ptr() if out is not None else None
    cp.addend = part.addend.data_ptr() if part.addend is not None else None
    cp.dim_out = out.shape[1] if out is not None else 0
    cp.accumulate = 1 if part.accumulate else 0
    cp.scale = part.scale
    if part.in_gate is not None:
        arr, n = part.in_gate.blocks_c()
        cp.in_gate = ctypes.cast(arr, ctypes.c_void_p)
        cp.n_in_gate = n
    else:
        cp.in_gate = None
        cp.n_in_gate = 0


_TYPE_ORDER: dict = {}


def type_order(types: torch.Tensor) -> torch.Tensor:
    """A permutation of the atoms that groups them by type (int32, stable), for the ``atom_order`` operand of
    ``nqa_node_fused``.  ANY permutation gives the same results there (a unit skips the types it does not hold), so the cache
    -- keyed on the type tensor's memory and version, a few entries -- can at worst cost speed, never correctness."""
    key = (types.data_ptr(), types._version, types.numel(), str(types.device))
    hit = _TYPE_ORDER.get(key)
    if hit is None:
        if len(_TYPE_ORDER) >= 8:
            _TYPE_ORDER.clear()
        hit = _TYPE_ORDER[key] = torch.argsort(types.view(-1), stable=True).to(torch.int32).contiguous()
    return hit


def launch_fused(parts: Sequence[FusedPart], types, out_gate: Optional[GateMeta] = None, gate_h=None,
                 order: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """``nqa_node_fused``: returns the destination tensors (one per non-accumulating part).  With ``out_gate`` the single
    destination is the gradient w.r.t. the gate's input rows ``gate_h``.  ``order``: optional int32 ``[N]`` atom order."""
    lib = _lib.load()
    x0 = parts[0].x
    N, dev = x0.shape[0], x0.device
    keep: list = []
    cparts = (_lib.NodePart * len(parts))()
    outs = []
    nbytes = 0
    flops = 0.0
    for k, part in enumerate(parts):
        assert part.x.dtype == torch.float32 and part.x.is_contiguous() and part.x.shape[0] == N
        if part.accumulate:
            out = None
        else:
            dout = out_gate.din if out_gate is not None else part.meta.dout
            out = torch.empty((N, dout), dtype=torch.float32, device=dev)
            outs.append(out)
            nbytes += 4 * N * dout
        if k == 0 or part.x.data_ptr() != parts[0].x.data_ptr():
            nbytes += 4 * N * part.x.shape[1]
        chunks, instr = part.meta.fwd
        flops += 2.0 * N * sum(c[1] * min(64, c[2] - c[3]) * sum(i[1] for i in instr[c[4]:c[5]]) for c in chunks)
        _fill_part(cparts[k], part, out, keep)
    if out_gate is not None:
        arr, n = out_gate.blocks_c()
        og, nog, gh, gdim = ctypes.cast(arr, ctypes.c_void_p), n, gate_h.data_ptr(), out_gate.din
        nbytes += 4 * N * out_gate.din
    else:
        og, nog, gh, gdim = None, 0, None, 0
    tp = _ptr(types) if types is not None else ctypes.c_void_p()
    if order is not None:
        assert order.dtype == torch.int32 and order.numel() == N and order.is_contiguous() and order.device == dev
    with torch.cuda.device(dev), ktimer.region("node_fused", nbytes, flops):
        rc = lib.nqa_node_fused(ctypes.cast(cparts, ctypes.c_void_p), len(parts), tp, _ptr(order), N, og, nog, gh, gdim,
                                _stream(dev))
    _lib.check(rc, "nqa_node_fused")
    return outs


def _scaled(wp: torch.Tensor, scale: float) -> torch.Tensor:
    """``wp * scale`` as a constant riding on ``wp`` (eval mode: folded once per weight version)."""
    if scale == 1.0:
        return wp
    return _constcache.get(wp, ("node_scaled", float(scale)), lambda: (wp * scale).contiguous())


class _FusedNodeStageFn(torch.autograd.Function):
    """``(h, types) -> (x1, sc)`` with ``x = Gate(h)`` (or ``x = h`` without a gate), ``x1 = scale * linear_1(x)``,
    ``sc = self_connection(x, types)`` -- one launch; backward ``(g_x1, g_sc) -> g_h`` one launch (constant weights)."""

    @staticmethod
    def forward(ctx, h, types, gate_meta, wp1, meta1, scale1, wps, metas):
        h = h.contiguous()
        parts = [FusedPart(h, wp1, meta1, scale1, in_gate=gate_meta)]
        if wps is not None:
            parts.append(FusedPart(h, wps, metas, 1.0, in_gate=gate_meta))
        typed = wps is not None and wps.shape[0] > 1
        # typed self-connection: the units walk the atoms grouped by type, so a unit stages one type's weights, not all
        ctx.order = type_order(types) if (typed and os.environ.get("NQA_NODE_TYPE_ORDER", "") != "0") else None
        outs = launch_fused(parts, types if typed else None, order=ctx.order)
        ctx.save_for_backward(h, types, wp1, wps)
        ctx.gate_meta, ctx.meta1, ctx.metas, ctx.scale1 = gate_meta, meta1, metas, scale1
        return (outs[0], outs[1]) if wps is not None else (outs[0], None)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g1, gs):
        h, types, wp1, wps = ctx.saved_tensors
        gate_meta, meta1, metas = ctx.gate_meta, ctx.meta1, ctx.metas
        if not ctx.needs_input_grad[0]:
            return (None,) * 8
        N = h.shape[0]
        if g1 is None:
            g1 = torch.zeros((N, meta1.dout), dtype=h.dtype, device=h.device)
        parts = [FusedPart(g1.contiguous(), _scaled(meta_transposed_weights(meta1, wp1), ctx.scale1), _transposed(meta1))]
        if wps is not None and gs is not None:
            parts.append(FusedPart(gs.contiguous(), meta_transposed_weights(metas, wps), _transposed(metas), accumulate=True))
        typed = len(parts) > 1 and wps.shape[0] > 1
        order = ctx.order if typed else None
        if gate_meta is not None:
            (gh,) = launch_fused(parts, types if typed else None, out_gate=gate_meta, gate_h=h, order=order)
        else:
            (gh,) = launch_fused(parts, types if typed else None, order=order)
        return gh, None, None, None, None, None, None, None


def fused_node_stage(h, types, gate_meta: Optional[GateMeta], wp1, meta1: NodeLinearMeta, scale1: float, wps=None,
                     metas: Optional[NodeLinearMeta] = None):
    """``x1, sc = scale1 * linear_1(Gate(h)), sc(Gate(h), types)`` (``sc`` None without self-connection weights)."""
    return _FusedNodeStageFn.apply(h, types, gate_meta, wp1, meta1, scale1, wps, metas)
