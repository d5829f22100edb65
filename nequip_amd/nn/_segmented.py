"""Non-uniform multiplicities (the reference's S / M / L / XL presets, ``nequip/model/nequip_models.py:30-51``:
``num_features = [128, 64]``, ``[128, 64, 32]``, ...) on the structure-specialised kernels.

A 'uvu' path couples channel ``u`` of its input irrep with channel ``u`` of its output slot and with weight column
``u`` of the path -- nothing mixes channels.  So a convolution whose input irreps have multiplicities
``m_0 >= m_1 >= ...`` splits into independent convolutions over channel ranges in which the set of live irreps is
constant (``[0, 32)``: all of them, ``[32, 64)``: those with ``mul > 32``, ...), each with ONE multiplicity -- exactly
the structures ``csrc/gen_spec.py`` generates kernels for (the truncated-input ones are prebuilt next to the BASELINE
shapes).  Per segment: gather the segment's columns of the node rows (node-sized copies), evaluate the radial MLP on the
segment's columns of its last layer (the hidden layer is recomputed per segment: 8 -> 128, cheap), run the uniform
tensor-product scatter (paired radial weights, pair-centric backward, training kernels: all as for uniform models) and
put the output columns back.  Autograd composes the pieces; parameters stay where the reference has them.
"""

from __future__ import annotations

from typing import List, Optional

import torch

from ..o3.irreps import Irreps
from ..utils.wgrad import differentiable_parameters
from .mlp import ScalarMLPFunction


class _SegmentMLP(ScalarMLPFunction):
    """``edge_mlp`` restricted to a set of output columns: shares the parent's parameters (first layer as is, last layer
    through a column gather that autograd differentiates in training and that is cached per parameter version in eval)."""

    def __init__(self, parent: ScalarMLPFunction, cols: torch.Tensor):
        depth = parent.num_layers - 1
        super().__init__(input_dim=parent.dims[0], output_dim=int(cols.numel()), hidden_layers_depth=depth,
                         hidden_layers_width=parent.dims[1] if depth > 0 else None, nonlinearity="silu", bias=False)
        self._parent = [parent]  # (a list: not a registered submodule)
        self._cols = cols
        self._last = len(self.mlp) - 1
        for layer in self.mlp:  # own no parameters
            if hasattr(layer, "weight"):
                del layer._parameters["weight"]
                layer.weight = None
        self._cached = None

    def sync(self) -> "_SegmentMLP":
        parent = self._parent[0]
        self.train(parent.training)
        for i, layer in enumerate(self.mlp):
            if not hasattr(layer, "weight"):
                continue
            pw = parent.mlp[i].weight
            if i != self._last:
                layer.weight = pw
                continue
            cols = self._cols if self._cols.device == pw.device else self._cols.to(pw.device)
            self._cols = cols
            if differentiable_parameters(parent.training, pw):
                layer.weight = pw.index_select(1, cols)
            else:
                key = (id(pw), pw._version, pw.data_ptr(), pw.device)
                if self._cached is None or self._cached[0] != key:
                    self._cached = (key, pw.detach().index_select(1, cols).contiguous())
                layer.weight = self._cached[1]
        return self

    def release(self) -> None:
        """Drop the derived last-layer weight once the segment's forward has been recorded.  In training it is a non-leaf
        ``index_select`` of the parent's parameter: left on the module it would keep the previous step's autograd graph
        alive and make ``copy.deepcopy(model)`` raise ("Only Tensors created explicitly by the user ... support the deepcopy
        protocol") -- best-model snapshots, lazily created EMA copies.  ``sync()`` derives it again on the next forward;
        autograd keeps what the backward needs."""
        layer = self.mlp[self._last]
        w = getattr(layer, "weight", None)
        if w is not None and (w.grad_fn is not None or w.requires_grad):
            layer.weight = None

    def __getstate__(self):
        # deepcopy / pickle: never carry derived tensors (the gathered columns, the eval-mode cache)
        for layer in self.mlp:
            w = getattr(layer, "weight", None)
            if w is not None and not isinstance(w, torch.nn.Parameter):
                layer.weight = None
        state = dict(self.__dict__)
        state["_cached"] = None
        return state


class Segment:
    def __init__(self, c0: int, c1: int, x_cols, w_cols, out_cols, tp):
        self.c0, self.c1 = c0, c1
        self.x_cols, self.w_cols, self.out_cols = x_cols, w_cols, out_cols
        self.tp = tp
        self.mlp: Optional[_SegmentMLP] = None

    def to(self, device):
        if self.x_cols.device != device:
            self.x_cols, self.w_cols, self.out_cols = (t.to(device) for t in (self.x_cols, self.w_cols, self.out_cols))
        return self


def channel_segments(feature_irreps_in, irreps_edge_attr, irreps_mid, instructions, tp_factory) -> Optional[List[Segment]]:
    """The uniform pieces of a 'uvu' convolution, or None when it is uniform already (or has no instructions).
    ``tp_factory(irreps_in, irreps_edge_attr, irreps_mid, instructions)`` builds the tensor-product module of a piece."""
    f_in, e_at, mid = Irreps(str(feature_irreps_in)), Irreps(str(irreps_edge_attr)), Irreps(str(irreps_mid))
    ins = [(int(t[0]), int(t[1]), int(t[2])) for t in instructions]
    muls = sorted({m.mul for m in f_in if m.mul > 0})
    if len(muls) <= 1 or not ins:
        return None
    x_off, o_off = f_in.offsets(), mid.offsets()
    w_off, off = [], 0
    for a, b, _ in ins:
        w_off.append(off)
        off += f_in[a].mul * e_at[b].mul
    segs = []
    bounds = [0] + muls
    for c0, c1 in zip(bounds, bounds[1:]):
        live = [b for b, m in enumerate(f_in) if m.mul > c0]
        keep = [k for k, (a, _, _) in enumerate(ins) if a in live]
        if not keep:
            continue
        slots = sorted({ins[k][2] for k in keep})
        width = c1 - c0
        sub_in = Irreps([(width, f_in[b].ir) for b in live])
        sub_mid = Irreps([(width, mid[s].ir) for s in slots])
        sub_ins = [(live.index(ins[k][0]), ins[k][1], slots.index(ins[k][2]), "uvu", True) for k in keep]
        ar = torch.arange(c0, c1)

        def block_cols(base, d):
            return (base + (ar.view(-1, 1) * d + torch.arange(d).view(1, -1))).reshape(-1)

        x_cols = torch.cat([block_cols(x_off[b], f_in[b].ir.dim) for b in live])
        out_cols = torch.cat([block_cols(o_off[s], mid[s].ir.dim) for s in slots])
        w_cols = torch.cat([w_off[k] + ar for k in keep])
        segs.append(Segment(c0, c1, x_cols, w_cols, out_cols, tp_factory(sub_in, e_at, sub_mid, sub_ins)))
    return segs


def output_permutation(segs: List[Segment], dim_out: int) -> torch.Tensor:
    """``cat([out_s for s in segs], 1).index_select(1, perm)`` is the output in the reference's column order."""
    cat = torch.cat([s.out_cols for s in segs])
    assert cat.numel() == dim_out and torch.equal(torch.sort(cat).values, torch.arange(dim_out))
    return torch.argsort(cat)
