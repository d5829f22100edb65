"""Oracle for TensorProductScatter (the hot op) in the reference's own op order.

Reference: nequip/nn/_tp_scatter_base.py:35-38
    edge_features = self.tp(x[edge_src], edge_attr, edge_weight)
    x = scatter(edge_features, edge_dst, dim=0, dim_size=x.size(0))
with self.tp = e3nn TensorProduct(feature_irreps_in, irreps_edge_attr, irreps_mid, instructions,
shared_weights=False, internal_weights=False) (:24-31) and scatter = zeros().scatter_add_ over the
expanded index (nequip/nn/utils.py:24-53).  e3nn 'uvu' semantics per SURVEY.md A.2:
    out[z, slot, u, k] = c_p * w[z, off_p + u] * sum_ij C_ijk x1[z, i1, u, i] x2[z, i2, 0, j]
    c_p = sqrt((2 l_out + 1) / #instructions into the same slot)   ("component" / "element")
weights consumed in instruction-list order, outputs concatenated in irreps_out order.
"""

import math

import torch

from . import irreps as ir
from .wigner import wigner_3j


def build_instructions(feature_irreps_in, irreps_edge_attr, irreps_out_filter):
    """Instruction list exactly as InteractionBlock builds it (nequip/nn/interaction_block.py:89-109),
    also used verbatim by tests/unit/nn/test_tp_scatter_kernel.py:78-97.
    Returns (irreps_mid_sorted, instructions)."""
    f_in = ir.parse(feature_irreps_in)
    e_at = ir.parse(irreps_edge_attr)
    f_out = ir.parse(irreps_out_filter)
    mid, instructions = [], []
    for i, (mul, l1, p1) in enumerate(f_in):
        for j, (_, l2, p2) in enumerate(e_at):
            for (l3, p3) in ir.product((l1, p1), (l2, p2)):
                if ir.contains(f_out, (l3, p3)):
                    k = len(mid)
                    mid.append((mul, l3, p3))
                    instructions.append((i, j, k, "uvu", True))
    mid_sorted, p, _ = ir.sort(mid)
    instructions = [(i1, i2, p[io], mode, train) for i1, i2, io, mode, train in instructions]
    return mid_sorted, instructions


def weight_numel(feature_irreps_in, irreps_edge_attr, instructions):
    f_in, e_at = ir.parse(feature_irreps_in), ir.parse(irreps_edge_attr)
    return sum(f_in[i1][0] * e_at[i2][0] for i1, i2, *_ in instructions)


def tensor_product_uvu(x1, x2, w, irreps_in1, irreps_in2, irreps_out, instructions):
    """e3nn TensorProduct.forward for 'uvu' instructions with per-sample weights.
    x1 [Z, dim1], x2 [Z, dim2], w [Z, weight_numel] -> [Z, dim_out]."""
    in1, in2, out = ir.parse(irreps_in1), ir.parse(irreps_in2), ir.parse(irreps_out)
    s1, s2 = ir.slices(in1), ir.slices(in2)
    Z = x1.shape[0]
    n_into = [0] * len(out)
    for i1, i2, io, *_ in instructions:
        n_into[io] += in2[i2][0]
    outs = [None] * len(out)
    woff = 0
    for ins in instructions:
        i1, i2, io = ins[0], ins[1], ins[2]
        pw = ins[5] if len(ins) > 5 else 1.0
        mul1, l1, p1 = in1[i1]
        mul2, l2, p2 = in2[i2]
        mulo, l3, p3 = out[io]
        assert mul1 == mulo and p1 * p2 == p3 and abs(l1 - l2) <= l3 <= l1 + l2
        a = x1[:, s1[i1]].reshape(Z, mul1, 2 * l1 + 1)
        b = x2[:, s2[i2]].reshape(Z, mul2, 2 * l2 + 1)
        wp = w[:, woff : woff + mul1 * mul2].reshape(Z, mul1, mul2)
        woff += mul1 * mul2
        C = wigner_3j(l1, l2, l3).to(x1.dtype)
        # xx = einsum("zui,zvj->zuvij"); result = einsum("zuv,ijk,zuvij->zuk")
        xx = torch.einsum("zui,zvj->zuvij", a, b)
        res = torch.einsum("zuv,ijk,zuvij->zuk", wp, C, xx)
        alpha = (2 * l3 + 1) / n_into[io] * pw
        res = math.sqrt(alpha) * res
        res = res.reshape(Z, mulo * (2 * l3 + 1))
        outs[io] = res if outs[io] is None else outs[io] + res
    assert woff == w.shape[1]
    cols = []
    for io, (mulo, l3, _) in enumerate(out):
        if outs[io] is None:
            cols.append(x1.new_zeros(Z, mulo * (2 * l3 + 1)))
        else:
            cols.append(outs[io])
    return torch.cat(cols, dim=-1) if cols else x1.new_zeros(Z, 0)


def scatter(src, index, dim_size):
    """nequip/nn/utils.py:24-53 with dim=0, reduce='sum'."""
    idx = index.view(-1, 1).expand_as(src)
    out = torch.zeros((dim_size, src.shape[1]), dtype=src.dtype, device=src.device)
    return out.scatter_add_(0, idx, src)


def tp_scatter(x, edge_attr, edge_weight, edge_dst, edge_src, feature_irreps_in, irreps_edge_attr, irreps_mid,
               instructions, edge_chunk=None):
    """TensorProductScatter.forward, nequip/nn/_tp_scatter_base.py:35-38.

    ``edge_chunk`` (test infrastructure for the full-size parity tests): evaluate the same two lines over consecutive
    ranges of ``edge_chunk`` edges -- the per-edge arithmetic is untouched (same einsum chain per edge), the ranges are
    added to the node rows in edge order like the sequential ``scatter_add_`` -- with every range under
    ``torch.utils.checkpoint`` so that autograd keeps the range's operands instead of the ``[E, mul, d1, d2]``
    temporaries (the reference formulation of the 10k-atom / l_max = 3 boxes does not fit a host otherwise, SURVEY
    App. B note iv)."""
    E = edge_dst.shape[0]
    if edge_chunk is None or E <= edge_chunk:
        edge_features = tensor_product_uvu(
            x[edge_src], edge_attr, edge_weight, feature_irreps_in, irreps_edge_attr, irreps_mid, instructions
        )
        return scatter(edge_features, edge_dst, x.size(0))
    from torch.utils.checkpoint import checkpoint

    def piece(x_, y_, w_, dst_, src_):
        f = tensor_product_uvu(x_[src_], y_, w_, feature_irreps_in, irreps_edge_attr, irreps_mid, instructions)
        return scatter(f, dst_, x_.size(0))

    out = None
    for a in range(0, E, edge_chunk):
        b = min(E, a + edge_chunk)
        part = checkpoint(piece, x, edge_attr[a:b], edge_weight[a:b], edge_dst[a:b], edge_src[a:b],
                          use_reentrant=False)
        out = part if out is None else out + part
    return out
