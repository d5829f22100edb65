"""Oracle restatements of the remaining modules on the path (all plain torch, reference op order)."""

import math

import torch

from . import irreps as ir
from .sh import spherical_harmonics
from .wigner import wigner_3j


# --------------------------------------------------------------------------------------------------
# edge geometry: nequip/nn/utils.py:68-118 (with_edge_vectors_)
# --------------------------------------------------------------------------------------------------
def edge_vectors(pos, edge_index, cell=None, edge_cell_shift=None, batch=None):
    vec = torch.index_select(pos, 0, edge_index[1]) - torch.index_select(pos, 0, edge_index[0])
    if cell is not None:
        if batch is not None:
            eb = torch.index_select(batch, 0, edge_index[0])
            vec = torch.baddbmm(
                vec.view(-1, 1, 3), edge_cell_shift.view(-1, 1, 3), torch.index_select(cell.view(-1, 3, 3), 0, eb)
            ).view(-1, 3)
        else:
            vec = vec + torch.sum(edge_cell_shift.view(-1, 3, 1) * cell.view(3, 3), 1)
    return vec


# --------------------------------------------------------------------------------------------------
# radial basis: nequip/nn/embedding/_edge.py:65-80,136-150; cutoffs.py:17-27; nequip_models.py:318-322
# --------------------------------------------------------------------------------------------------
def polynomial_cutoff(x, p=6.0):
    p = float(p)
    out = 1.0
    out = out - (((p + 1.0) * (p + 2.0) / 2.0) * torch.pow(x, p))
    out = out + (p * (p + 2.0) * torch.pow(x, p + 1.0))
    out = out - ((p * (p + 1.0) / 2) * torch.pow(x, p + 2.0))
    return out * (x < 1.0)


def bessel_embedding(edge_vec, r_max, num_bessels=8, p=6.0, model_dtype=torch.float32, bessel_weights=None,
                     per_edge_type_cutoff=None, edge_types=None):
    """EdgeLengthNormalizer -> BesselEdgeLengthEncoding (x PolynomialCutoff) -> ApplyFactor(2 pi / r_max^2).
    ``per_edge_type_cutoff`` [T, T] (rows = centre type) with ``edge_types`` [2, E]: _edge.py:71-78."""
    r = edge_vec.square().sum(1, keepdim=True).sqrt()  # utils.py:117
    if per_edge_type_cutoff is not None:
        T = per_edge_type_cutoff.shape[0]
        recip = per_edge_type_cutoff.reciprocal().view(-1)  # _edge.py:52
        x = r * torch.index_select(recip, 0, edge_types[0] * T + edge_types[1]).unsqueeze(-1)  # :74-79
    else:
        x = r * torch.as_tensor(1.0 / r_max, dtype=edge_vec.dtype)  # _edge.py:79
    if bessel_weights is None:
        bessel_weights = torch.linspace(1.0, num_bessels, num_bessels, dtype=edge_vec.dtype).unsqueeze(0)
    bessel = (torch.sinc(x * bessel_weights) * bessel_weights).to(model_dtype)  # _edge.py:140-142
    cutoff = polynomial_cutoff(x, p).to(model_dtype)  # :145
    emb = bessel * cutoff  # :149
    factor = (2 * math.pi) / (r_max * r_max)  # nequip_models.py:320
    return factor * emb, cutoff


def sh_edge_attrs(edge_vec, lmax, model_dtype=torch.float32):
    """SphericalHarmonicEdgeAttrs.forward, nequip/nn/embedding/_edge.py:193-198."""
    return spherical_harmonics(edge_vec, lmax, normalize=True).to(model_dtype)


# --------------------------------------------------------------------------------------------------
# ScalarMLPFunction: nequip/nn/mlp.py:81-196,262-268
# --------------------------------------------------------------------------------------------------
def scalar_mlp(x, weights, nonlinearity="silu"):
    """weights: list of [h_in, h_out]; alpha = gain/sqrt(h_in), gain = 1 for layer 0 (or no nonlinearity) else sqrt(2)."""
    n = len(weights)
    for layer, W in enumerate(weights):
        h_in = W.shape[0]
        gain = 1.0 if (nonlinearity is None or layer == 0) else math.sqrt(2)
        alpha = torch.tensor(gain / math.sqrt(h_in), dtype=W.dtype)  # model-dtype buffer, mlp.py:259
        x = torch.mm(x, W * alpha)
        if layer != n - 1 and nonlinearity is not None:
            x = torch.nn.functional.silu(x)
    return x


# --------------------------------------------------------------------------------------------------
# e3nn o3.Linear, used at nequip/nn/interaction_block.py:82-87,129-138
# --------------------------------------------------------------------------------------------------
def linear_instructions(irreps_in, irreps_out):
    i_in, i_out = ir.parse(irreps_in), ir.parse(irreps_out)
    return [
        (a, b) for a, (_, l1, p1) in enumerate(i_in) for b, (_, l2, p2) in enumerate(i_out) if (l1, p1) == (l2, p2)
    ]


def linear_weight_numel(irreps_in, irreps_out):
    i_in, i_out = ir.parse(irreps_in), ir.parse(irreps_out)
    return sum(i_in[a][0] * i_out[b][0] for a, b in linear_instructions(irreps_in, irreps_out))


def o3_linear(x, weight, irreps_in, irreps_out):
    """out[z, i_out, w, m] = sum_{i_in} 1/sqrt(fan_in(i_out)) sum_u x[z, i_in, u, m] W[u, w];
    flat weight in instruction order (i_in-major, then i_out), each [mul_in, mul_out]."""
    i_in, i_out = ir.parse(irreps_in), ir.parse(irreps_out)
    s_in = ir.slices(i_in)
    instr = linear_instructions(i_in, i_out)
    fan_in = [0] * len(i_out)
    for a, b in instr:
        fan_in[b] += i_in[a][0]
    Z = x.shape[0]
    outs = [None] * len(i_out)
    off = 0
    for a, b in instr:
        mul_in, l, _ = i_in[a]
        mul_out = i_out[b][0]
        W = weight[off : off + mul_in * mul_out].reshape(mul_in, mul_out)
        off += mul_in * mul_out
        xa = x[:, s_in[a]].reshape(Z, mul_in, 2 * l + 1)
        r = torch.einsum("uw,zui->zwi", W, xa) * (1.0 / math.sqrt(fan_in[b]))
        r = r.reshape(Z, mul_out * (2 * l + 1))
        outs[b] = r if outs[b] is None else outs[b] + r
    assert off == weight.numel()
    cols = []
    for b, (mul, l, _) in enumerate(i_out):
        cols.append(outs[b] if outs[b] is not None else x.new_zeros(Z, mul * (2 * l + 1)))
    return torch.cat(cols, dim=-1)


# --------------------------------------------------------------------------------------------------
# e3nn FullyConnectedTensorProduct, the self-connection at interaction_block.py:142-146,175
# --------------------------------------------------------------------------------------------------
def fctp_instructions(irreps_in1, irreps_in2, irreps_out):
    a1, a2, ao = ir.parse(irreps_in1), ir.parse(irreps_in2), ir.parse(irreps_out)
    return [
        (i1, i2, io)
        for i1, (_, l1, p1) in enumerate(a1)
        for i2, (_, l2, p2) in enumerate(a2)
        for io, (_, l3, p3) in enumerate(ao)
        if (l3, p3) in ir.product((l1, p1), (l2, p2))
    ]


def fctp_weight_numel(irreps_in1, irreps_in2, irreps_out):
    a1, a2, ao = ir.parse(irreps_in1), ir.parse(irreps_in2), ir.parse(irreps_out)
    return sum(a1[i1][0] * a2[i2][0] * ao[io][0] for i1, i2, io in fctp_instructions(a1, a2, ao))


def fully_connected_tp(x1, x2, weight, irreps_in1, irreps_in2, irreps_out):
    """'uvw' paths, shared internal weights; alpha = (2 l_o + 1) / sum_{paths->i_o} mul1*mul2."""
    a1, a2, ao = ir.parse(irreps_in1), ir.parse(irreps_in2), ir.parse(irreps_out)
    s1, s2 = ir.slices(a1), ir.slices(a2)
    instr = fctp_instructions(a1, a2, ao)
    fan = [0] * len(ao)
    for i1, i2, io in instr:
        fan[io] += a1[i1][0] * a2[i2][0]
    Z = x1.shape[0]
    outs = [None] * len(ao)
    off = 0
    for i1, i2, io in instr:
        mul1, l1, _ = a1[i1]
        mul2, l2, _ = a2[i2]
        mulo, l3, _ = ao[io]
        W = weight[off : off + mul1 * mul2 * mulo].reshape(mul1, mul2, mulo)
        off += mul1 * mul2 * mulo
        xa = x1[:, s1[i1]].reshape(Z, mul1, 2 * l1 + 1)
        xb = x2[:, s2[i2]].reshape(Z, mul2, 2 * l2 + 1)
        C = wigner_3j(l1, l2, l3).to(x1.dtype)
        xx = torch.einsum("zui,zvj->zuvij", xa, xb)
        r = torch.einsum("uvw,ijk,zuvij->zwk", W, C, xx) * math.sqrt((2 * l3 + 1) / fan[io])
        r = r.reshape(Z, mulo * (2 * l3 + 1))
        outs[io] = r if outs[io] is None else outs[io] + r
    assert off == weight.numel()
    cols = []
    for io, (mul, l, _) in enumerate(ao):
        cols.append(outs[io] if outs[io] is not None else x1.new_zeros(Z, mul * (2 * l + 1)))
    return torch.cat(cols, dim=-1)


# --------------------------------------------------------------------------------------------------
# e3nn Gate + normalize2mom, nequip/nn/convnetlayer.py:104-112,162-164
# --------------------------------------------------------------------------------------------------
_ACTS = {"silu": torch.nn.functional.silu, "tanh": torch.tanh, "abs": torch.abs}
_CST_CACHE = {}


def normalize2mom_const(name):
    if name not in _CST_CACHE:
        gen = torch.Generator(device="cpu").manual_seed(0)
        z = torch.randn(1_000_000, generator=gen, dtype=torch.float64)
        cst = _ACTS[name](z).pow(2).mean().pow(-0.5).item()
        _CST_CACHE[name] = 1.0 if abs(cst - 1) < 1e-4 else cst
    return _CST_CACHE[name]


def gate(x, irreps_scalars, act_scalars, irreps_gates, act_gates, irreps_gated):
    """input layout scalars (+) gates (+) gated; output act(scalars) (+) act(gates)[u] * gated[u, :]."""
    sc, ga, gd = ir.parse(irreps_scalars), ir.parse(irreps_gates), ir.parse(irreps_gated)
    ns, ng = ir.dim(sc), ir.dim(ga)
    scalars, gates, gated = x[:, :ns], x[:, ns : ns + ng], x[:, ns + ng :]

    def activation(t, irreps, acts):
        cols, off = [], 0
        for (mul, l, p), name in zip(irreps, acts):
            assert l == 0
            seg = t[:, off : off + mul]
            off += mul
            cols.append(_ACTS[name](seg) * normalize2mom_const(name))
        return torch.cat(cols, dim=-1) if cols else t[:, :0]

    scalars = activation(scalars, sc, act_scalars)
    if ng == 0:
        return scalars
    gates = activation(gates, ga, act_gates)
    cols, goff, xoff = [], 0, 0
    Z = x.shape[0]
    for mul, l, _ in gd:
        g = gates[:, goff : goff + mul]
        goff += mul
        blk = gated[:, xoff : xoff + mul * (2 * l + 1)].reshape(Z, mul, 2 * l + 1)
        xoff += mul * (2 * l + 1)
        cols.append((blk * g.unsqueeze(-1)).reshape(Z, mul * (2 * l + 1)))
    return torch.cat([scalars] + cols, dim=-1)


# --------------------------------------------------------------------------------------------------
# e3nn NormActivation (used at nequip/nn/convnetlayer.py:116-125 for nonlinearity_type="norm") [RECALLED, e3nn 0.5/0.6]:
#   norms = o3.Norm(irreps, squared=True)(x)           # per irrep copy: sum_m x_m^2
#   norms[norms < eps^2] = eps^2; norms = norms.sqrt()  # eps = 1e-8
#   scalings = act(norms) / norms                       # normalize=True, bias=False; act is the RAW function (no
#   out = ElementwiseTensorProduct(scalings, x)         #   normalize2mom here); 0e x l -> l with unit coefficient
# --------------------------------------------------------------------------------------------------
def norm_activation(x, irreps, act_name="silu", epsilon=1e-8):
    Z = x.shape[0]
    cols, off = [], 0
    for mul, l, _ in ir.parse(irreps):
        d = 2 * l + 1
        blk = x[:, off : off + mul * d].reshape(Z, mul, d)
        off += mul * d
        norms = (blk * blk).sum(-1)
        norms = torch.where(norms < epsilon * epsilon, torch.full_like(norms, epsilon * epsilon), norms).sqrt()
        scalings = _ACTS[act_name](norms) / norms
        cols.append((blk * scalings.unsqueeze(-1)).reshape(Z, mul * d))
    return torch.cat(cols, dim=-1)
