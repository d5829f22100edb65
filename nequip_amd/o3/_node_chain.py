"""Host side of ``nqa_node_chain`` (``nequip_amd/csrc/node_chain.hip``): the node-side chain across one layer boundary,

    forward   h = linear_2(a) + sc_prev;  x' = Gate(h);  y = scale * linear_1'(x');  s = sc'(x', atom type)
    backward  g' = scale * linear_1'^T(g_y) + sc'^T(g_s);  g_h = Gate'(g', h);  g_a = linear_2^T(g_h)

in one launch each way (inference: the weights are constants).  ``NodeStage`` builds the chunk / instruction tables of
both directions from the modules' own metadata -- ``o3.Linear`` / ``FullyConnectedTensorProduct`` (``NodeLinearMeta``)
and ``Gate`` -- splitting every operand block at the gate's segment boundaries so that each instruction reads a
homogeneous *gate view* (scalar, gate or gated segment); ``node_stage`` is the autograd Function around the two launches.
Replaces the call sequence ``nequip/nn/interaction_block.py:201-204`` -> ``nequip/nn/convnetlayer.py:162-164`` ->
``nequip/nn/interaction_block.py:175-177`` of the reference.
"""

from __future__ import annotations

import ctypes
import struct
from typing import List, Optional, Tuple

import torch

from .. import _lib
from ..utils import ktimer
from ._node_kernels import _ACT_IDS, NodeLinearMeta, _ptr, _stream, meta_transposed_weights

K_PLAIN, K_FWD_SCALAR, K_FWD_GATED, K_BWD_SCALAR, K_BWD_GATED, K_BWD_GATE = range(6)

_CHUNK = struct.Struct("<8i")
_INSTR = struct.Struct("<8i2f6i")


class _Seg:
    """One homogeneous segment of the gate: `kind` in {"scalar", "gate", "gated"}."""

    def __init__(self, kind, mul, d, h_off, xp_off, act, cst, gate_h_off=-1, gated_h_off=-1, gated_xp_off=-1, gated_d=0):
        self.kind, self.mul, self.d, self.h_off, self.xp_off = kind, mul, d, h_off, xp_off
        self.act, self.cst = act, cst
        self.gate_h_off, self.gated_h_off, self.gated_xp_off, self.gated_d = gate_h_off, gated_h_off, gated_xp_off, gated_d


def gate_segments(gate) -> List[_Seg]:
    """Segments of a ``Gate`` in its input layout h = [scalars | gates | gated] and output layout x' = [scalars | gated]."""
    names = {torch.nn.functional.silu: "silu", torch.tanh: "tanh"}
    segs: List[_Seg] = []
    h_off = xp_off = 0
    for (mul, _), act, cst in zip(gate.irreps_scalars, gate.act_scalars, gate._cst_scalars):
        segs.append(_Seg("scalar", mul, 1, h_off, xp_off, _ACT_IDS[names[act]], float(cst)))
        h_off += mul
        xp_off += mul
    ns = gate.irreps_scalars.dim
    ng = gate.irreps_gates.dim
    gate_off = ns
    gated_h, gated_xp = ns + ng, ns
    gate_segs, gated_segs = [], []
    for (mul, ir), (gmul, _), act, cst in zip(gate.irreps_gated, gate.irreps_gates, gate.act_gates, gate._cst_gates):
        assert mul == gmul
        a = _ACT_IDS[names[act]]
        gate_segs.append(_Seg("gate", mul, 1, gate_off, -1, a, float(cst), gated_h_off=gated_h, gated_xp_off=gated_xp,
                              gated_d=ir.dim))
        gated_segs.append(_Seg("gated", mul, ir.dim, gated_h, gated_xp, a, float(cst), gate_h_off=gate_off))
        gate_off += mul
        gated_h += mul * ir.dim
        gated_xp += mul * ir.dim
    return segs + gate_segs + gated_segs


def _split(block_off: int, mul: int, d: int, segs: List[_Seg], layout: str):
    """Intersect the operand block [block_off, block_off + mul d) (channels of dimension d) with the gate segments in
    the given layout ("xp" = gate output, "h" = gate input): yields (u0, u1, segment, channel offset inside it)."""
    out = []
    for sg in segs:
        off = sg.xp_off if layout == "xp" else sg.h_off
        if off < 0:
            continue
        lo, hi = max(block_off, off), min(block_off + mul * d, off + sg.mul * sg.d)
        if lo >= hi:
            continue
        assert sg.d == d and (lo - block_off) % d == 0 and (hi - block_off) % d == 0 and (lo - off) % d == 0, \
            "gate segments and operand blocks must share channel boundaries"
        out.append(((lo - block_off) // d, (hi - block_off) // d, sg, (lo - off) // d))
    covered = sum(u1 - u0 for u0, u1, _, _ in out)
    assert covered == mul, "operand block not covered by the gate's segments"
    return out


class NodeStage:
    """Tables and launches of one fused layer boundary.  ``lin2`` / ``gate`` belong to layer L, ``lin1`` / ``sc`` (optional)
    to layer L + 1; ``scale`` is layer L + 1's 1/sqrt(avg_num_neighbors)."""

    def __init__(self, lin2, gate, lin1, sc, scale: float):
        self.lin2, self.gate, self.lin1, self.sc, self.scale = lin2, gate, lin1, sc, float(scale)
        self.segs = gate_segments(gate)
        self.dim_a, self.dim_h = lin2._meta.din, lin2._meta.dout
        self.dim_x = lin1._meta.din
        self.dim_y = lin1._meta.dout
        self.dim_s = sc._meta.dout if sc is not None else 0
        assert self.dim_h == gate._kernel_meta.din and self.dim_x == gate._kernel_meta.dout
        if sc is not None:
            assert sc._meta.din == self.dim_x
        self.dmax = max(ir.dim for _, ir in list(lin2.irreps_in) + list(lin2.irreps_out) + list(lin1.irreps_out))
        self._tables = {}
        self._build()

    # ------------------------------------------------------------------ table construction
    def _build(self):
        # ---- forward.  sources: 0 = a (plain), 1 = h (gate-forward views);  destinations: 0 = h, 1 = y, 2 = s;
        #      weight sets: 0 = linear_2, 1 = linear_1', 2 = sc' (typed)
        chunks, instr = [], []

        def add_chunk(dst, c, ibeg, iend, flags):
            o_off, d, mul_o, c0 = c[0], c[1], c[2], c[3]
            chunks.append(_CHUNK.pack(dst, o_off, d, mul_o, c0, ibeg, iend, flags))

        def emit(src, x_off, mul_in, wset, w_off, kind=K_PLAIN, act=0, store=-1, scale=1.0, cst=1.0, aux=(0, 0, 0), ln=0):
            instr.append(_INSTR.pack(src, x_off, mul_in, wset, w_off, kind, act, store, scale, cst, aux[0], aux[1], aux[2],
                                     ln, 0, 0))

        def meta_chunks(meta: NodeLinearMeta, which: str):
            return getattr(meta, which)

        # phase 0: h = linear_2(a) (+ addend)
        ch2, in2 = meta_chunks(self.lin2._meta, "fwd")
        for c in ch2:
            ibeg = len(instr)
            for (x_off, mul_in, w_off, _) in in2[c[4]:c[5]]:
                emit(0, x_off, mul_in, 0, w_off)
            add_chunk(0, c, ibeg, len(instr), 1)
        p0 = len(chunks)

        # phase 1: y = scale * linear_1'(Gate(h)), s = sc'(Gate(h))
        def fwd_view_instrs(x_off, mul_in, w_off, mul_out, d, wset):
            for u0, u1, sg, cu in _split(x_off, mul_in, d, self.segs, "xp"):
                w = w_off + u0 * mul_out
                if sg.kind == "scalar":
                    emit(1, sg.h_off + cu, u1 - u0, wset, w, K_FWD_SCALAR if sg.act else K_PLAIN, sg.act, cst=sg.cst)
                else:
                    emit(1, sg.h_off + cu * d, u1 - u0, wset, w, K_FWD_GATED, sg.act, cst=sg.cst,
                         aux=(sg.gate_h_off + cu, 0, 0))

        for dst, mod, wset in ((1, self.lin1, 1), (2, self.sc, 2)):
            if mod is None:
                continue
            chs, ins = meta_chunks(mod._meta, "fwd")
            for c in chs:
                ibeg = len(instr)
                for (x_off, mul_in, w_off, _) in ins[c[4]:c[5]]:
                    fwd_view_instrs(x_off, mul_in, w_off, c[2], c[1], wset)
                add_chunk(dst, c, ibeg, len(instr), 0)
        self._fwd = (b"".join(chunks), b"".join(instr), (0, p0, len(chunks)))

        # ---- backward.  sources: 0 = g_y, 1 = g_s, 2 = (g', h) gate-backward views storing g_h;
        #      destinations: 0 = g' (scratch), 1 = g_a;  weight sets: 0 = linear_1'^T, 1 = sc'^T (typed), 2 = linear_2^T
        chunks, instr = [], []
        ch1, in1 = meta_chunks(self.lin1._meta, "bwd")
        if self.sc is not None:
            chs, ins = meta_chunks(self.sc._meta, "bwd")
            assert [c[:4] for c in chs] == [c[:4] for c in ch1], "linear_1 and the self-connection read the same blocks"
        for k, c in enumerate(ch1):
            ibeg = len(instr)
            for (x_off, mul_in, w_off, _) in in1[c[4]:c[5]]:
                emit(0, x_off, mul_in, 0, w_off, scale=self.scale)
            if self.sc is not None:
                cs = chs[k]
                for (x_off, mul_in, w_off, _) in ins[cs[4]:cs[5]]:
                    emit(1, x_off, mul_in, 1, w_off)
            add_chunk(0, c, ibeg, len(instr), 0)
        p0 = len(chunks)
        ch2, in2 = meta_chunks(self.lin2._meta, "bwd")
        stored = set()
        for c in ch2:
            ibeg = len(instr)
            mul_out, d = c[2], c[1]
            for (x_off, mul_in, w_off, _) in in2[c[4]:c[5]]:
                for u0, u1, sg, cu in _split(x_off, mul_in, d, self.segs, "h"):
                    w = w_off + u0 * mul_out
                    hcol = sg.h_off + cu * d
                    key = (hcol, u1 - u0)
                    store = hcol if key not in stored and c[3] == 0 else -1  # first chunk that stages the segment
                    if store >= 0:
                        stored.add(key)
                    if sg.kind == "scalar":
                        emit(2, hcol, u1 - u0, 2, w, K_BWD_SCALAR, sg.act, store, cst=sg.cst, aux=(0, sg.xp_off + cu, 0))
                    elif sg.kind == "gate":
                        emit(2, hcol, u1 - u0, 2, w, K_BWD_GATE, sg.act, store, cst=sg.cst,
                             aux=(0, sg.gated_xp_off + cu * sg.gated_d, sg.gated_h_off + cu * sg.gated_d), ln=sg.gated_d)
                    else:
                        emit(2, hcol, u1 - u0, 2, w, K_BWD_GATED, sg.act, store, cst=sg.cst,
                             aux=(sg.gate_h_off + cu, sg.xp_off + cu * d, 0))
            add_chunk(1, c, ibeg, len(instr), 0)
        self._bwd = (b"".join(chunks), b"".join(instr), (0, p0, len(chunks)))
        covered = sum(n for _, n in stored)
        self._gh_complete = covered == sum(sg.mul for sg in self.segs)

    def _device_tables(self, which: str, device):
        key = (which, str(device))
        if key not in self._tables:
            cb, ib, ph = getattr(self, "_" + which)
            ct = torch.frombuffer(bytearray(cb), dtype=torch.uint8).clone().to(device)
            it = torch.frombuffer(bytearray(ib), dtype=torch.uint8).clone().to(device)
            self._tables[key] = (ct, it, ph)
        return self._tables[key]

    # ------------------------------------------------------------------ weights (constants, eval mode)
    def _weights(self, device, dtype, table):
        w2 = self.lin2.eval_weights(device, dtype)
        w1 = self.lin1.eval_weights(device, dtype)
        ws = self.sc.eval_weights_typed(table, dtype) if self.sc is not None else None
        return w2, w1, ws

    # ------------------------------------------------------------------ launches
    def _launch(self, which, srcs, dsts, wsets, types, N, device, nbytes, flops):
        lib = _lib.load()
        ct, it, ph = self._device_tables(which, device)
        desc = _lib.ChainDesc()
        for i, s in enumerate(srcs):
            rows, gin, store = s
            desc.src[i].rows = rows.data_ptr() if rows is not None else None
            desc.src[i].gate_input = gin.data_ptr() if gin is not None else None
            desc.src[i].store = store.data_ptr() if store is not None else None
            desc.src[i].dim = rows.shape[1] if rows is not None else 0
            desc.src[i].gate_input_dim = gin.shape[1] if gin is not None else 0
            desc.src[i].store_dim = store.shape[1] if store is not None else 0
        for i, d in enumerate(dsts):
            rows, addend, scale = d
            desc.dst[i].rows = rows.data_ptr() if rows is not None else None
            desc.dst[i].addend = addend.data_ptr() if addend is not None else None
            desc.dst[i].scale = float(scale)
            desc.dst[i].dim = rows.shape[1] if rows is not None else 0
        for i, w in enumerate(wsets):
            if w is not None:
                desc.weights[i].data = w.data_ptr()
                desc.weights[i].stride = w.shape[1]
                desc.weights[i].n_types = w.shape[0]
        desc.chunks, desc.instr = ct.data_ptr(), it.data_ptr()
        desc.atom_types = types.data_ptr() if types is not None else None
        desc.num_nodes = N
        for k, v in enumerate(ph):
            desc.phase_begin[k] = v
        desc.n_phases = len(ph) - 1
        desc.max_irrep_dim = self.dmax
        with torch.cuda.device(device), ktimer.region("node_chain", nbytes, flops):
            rc = lib.nqa_node_chain(ctypes.byref(desc), _stream(device))
        _lib.check(rc, "nqa_node_chain")

    def _flops(self, N):
        def f(meta):
            chs, ins = meta.fwd
            return 2.0 * N * sum(c[1] * min(64, c[2] - c[3]) * sum(i[1] for i in ins[c[4]:c[5]]) for c in chs)
        T = 1
        return f(self.lin2._meta) + f(self.lin1._meta) + (f(self.sc._meta) if self.sc is not None else 0.0) * T

    def forward(self, a, addend, types, table):
        """-> (h, y, s)"""
        N, dev = a.shape[0], a.device
        w2, w1, ws = self._weights(dev, a.dtype, table)
        h = torch.empty((N, self.dim_h), dtype=a.dtype, device=dev)
        y = torch.empty((N, self.dim_y), dtype=a.dtype, device=dev)
        s = torch.empty((N, self.dim_s), dtype=a.dtype, device=dev) if self.sc is not None else None
        nbytes = 4.0 * N * (self.dim_a + self.dim_h + self.dim_y + self.dim_s + (self.dim_h if addend is not None else 0))
        self._launch("fwd", [(a, None, None), (h, None, None)], [(h, addend, 1.0), (y, None, self.scale), (s, None, 1.0)],
                     [w2, w1, ws], types if self.sc is not None else None, N, dev, nbytes, self._flops(N))
        return h, y, s

    def backward(self, g_y, g_s, h, types, table, want_gh: bool):
        """-> (g_a, g_h or None)"""
        N, dev = h.shape[0], h.device
        w2, w1, ws = self._weights(dev, h.dtype, table)
        w1t = meta_transposed_weights(self.lin1._meta, w1)
        w2t = meta_transposed_weights(self.lin2._meta, w2)
        wst = meta_transposed_weights(self.sc._meta, ws) if self.sc is not None else None
        gp = torch.empty((N, self.dim_x), dtype=h.dtype, device=dev)
        ga = torch.empty((N, self.dim_a), dtype=h.dtype, device=dev)
        gh = torch.empty((N, self.dim_h), dtype=h.dtype, device=dev) if want_gh else None
        if want_gh and not self._gh_complete:
            gh.zero_()
        nbytes = 4.0 * N * (self.dim_y + self.dim_s + self.dim_h + self.dim_a + (self.dim_h if want_gh else 0))
        self._launch("bwd_nostore" if not want_gh else "bwd",
                     [(g_y, None, None), (g_s, None, None), (gp, h, gh)], [(gp, None, 1.0), (ga, None, 1.0)],
                     [w1t, wst, w2t], types if self.sc is not None else None, N, dev, nbytes, self._flops(N))
        return ga, gh

    @property
    def _bwd_nostore(self):
        """The backward tables without the g_h stores (first layer boundary of a model whose first layer has no sc)."""
        if "_bwd_ns" not in self.__dict__:
            cb, ib, ph = self._bwd
            recs = []
            for k in range(0, len(ib), _INSTR.size):
                f = list(_INSTR.unpack(ib[k:k + _INSTR.size]))
                f[7] = -1
                recs.append(_INSTR.pack(*f))
            self.__dict__["_bwd_ns"] = (cb, b"".join(recs), ph)
        return self.__dict__["_bwd_ns"]


class _NodeStageFn(torch.autograd.Function):
    """(a, addend) -> (y, s); h is kept for the backward.  Weights are constants (eval mode)."""

    @staticmethod
    def forward(ctx, a, addend, types, table, stage: NodeStage):
        a = a.contiguous()
        addend = addend.contiguous() if addend is not None else None
        h, y, s = stage.forward(a, addend, types, table)
        ctx.save_for_backward(h, types, table)
        ctx.stage, ctx.has_addend = stage, addend is not None
        if s is None:
            s = y.new_empty(0)
            ctx.mark_non_differentiable(s)
        return y, s

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_y, g_s):
        h, types, table = ctx.saved_tensors
        stage: NodeStage = ctx.stage
        g_y = g_y.contiguous() if g_y is not None else h.new_zeros((h.shape[0], stage.dim_y))
        if stage.sc is not None:
            g_s = g_s.contiguous() if g_s is not None else h.new_zeros((h.shape[0], stage.dim_s))
        else:
            g_s = None
        want_gh = ctx.has_addend and ctx.needs_input_grad[1]
        ga, gh = stage.backward(g_y, g_s, h, types, table, want_gh)
        return ga, gh, None, None, None


def node_stage(stage: NodeStage, a, addend, types, table) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    y, s = _NodeStageFn.apply(a, addend, types, table, stage)
    return y, (s if stage.sc is not None else None)


def stage_supported(lin2, gate, lin1, sc) -> bool:
    """float32 GPU evaluation of these modules through nqa_node_chain: kernel-backed gate (silu / tanh), node_attrs as one
    scalar block (typed self-connection), matching layouts."""
    if getattr(gate, "_kernel_meta", None) is None or lin2._meta is None or lin1._meta is None:
        return False
    if sc is not None and sc._meta is None:
        return False
    if lin2._meta.dout != gate._kernel_meta.din or lin1._meta.din != gate._kernel_meta.dout:
        return False
    if sc is not None and sc._meta.din != lin1._meta.din:
        return False
    try:
        _split  # noqa: B018
        segs = gate_segments(gate)
        for meta, which, layout in ((lin1._meta, "fwd", "xp"), (lin2._meta, "bwd", "h")):
            chs, ins = getattr(meta, which)
            for c in chs:
                for (x_off, mul_in, _, _) in ins[c[4]:c[5]]:
                    _split(x_off, mul_in, c[1], segs, layout)
    except AssertionError:
        return False
    return True
