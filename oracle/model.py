"""Oracle NequIP GNN energy + forces model: the reference call stack of SURVEY.md 3.1 with plain torch ops.

Architecture replay: nequip/model/nequip_models.py:116-399 (NequIPGNNModel -> FullNequIPGNNModel),
nequip/nn/convnetlayer.py:34-170, nequip/nn/interaction_block.py:24-207, nequip/nn/grad_output.py:107-298.
Weights are taken from a flat ``{name: tensor}`` dict that uses the reference's parameter names
(``layer{i}_convnet.conv.linear_1.weight`` ... see nequip/model/param_groups.py:62-71), so the very same
state dict drives this oracle and the HIP-backed product model in the parity tests.
"""

import math

import torch

from . import irreps as ir
from . import nn as onn
from . import tp as otp


def tp_path_exists(irreps_in1, irreps_in2, ir_out):
    """nequip/nn/utils.py:56-65"""
    for _, l1, p1 in ir.simplify(ir.parse(irreps_in1)):
        for _, l2, p2 in ir.simplify(ir.parse(irreps_in2)):
            if tuple(ir_out) in ir.product((l1, p1), (l2, p2)):
                return True
    return False


class LayerSpec:
    pass


def build_specs(cfg):
    """Replay the irreps derivation of every ConvNetLayer / InteractionBlock."""
    l_max = cfg["l_max"]
    parity = cfg.get("parity", True)
    nf = cfg["num_features"]
    nf = [nf] * (l_max + 1) if isinstance(nf, int) else list(nf)
    assert len(nf) == l_max + 1
    num_layers = cfg["num_layers"]
    n_embed = cfg.get("type_embed_num_features") or nf[0]
    hidden = [
        (nf[l], l, p) for l in range(l_max + 1) for p in ((1, -1) if parity else ((1,) if l % 2 == 0 else (-1,)))
    ]  # nequip_models.py:180-190
    hidden_list = [hidden] * (num_layers - 1) + [[(nf[0], 0, 1)]]
    edge_sh = ir.spherical_harmonics(l_max)
    node_attrs = [(n_embed, 0, 1)]
    prev = [(n_embed, 0, 1)]
    acts_s = {1: "silu", -1: "tanh"}
    specs = []
    for li in range(num_layers):
        S = LayerSpec()
        S.feature_irreps_in = prev
        scalars = [(m, l, p) for m, l, p in hidden_list[li] if l == 0 and tp_path_exists(prev, edge_sh, (l, p))]
        gated = [(m, l, p) for m, l, p in hidden_list[li] if l > 0 and tp_path_exists(prev, edge_sh, (l, p))]
        gate_ir = (0, 1) if tp_path_exists(prev, edge_sh, (0, 1)) else (0, -1)
        gates = [(m, gate_ir[0], gate_ir[1]) for m, _, _ in gated]
        S.irreps_scalars, S.irreps_gates, S.irreps_gated = scalars, gates, gated
        S.act_scalars = [acts_s[p] for _, _, p in scalars]
        S.act_gates = [acts_s[p] for _, _, p in gates]
        S.nonlin = cfg.get("convnet_nonlinearity_type", "gate")
        if S.nonlin == "gate":
            S.conv_irreps_out = ir.simplify(ir.simplify(scalars + gates + gated))
        else:  # "norm" (nequip/nn/convnetlayer.py:113-125): no gate scalars, NormActivation on the simplified output
            S.conv_irreps_out = ir.simplify(scalars + gated)
        S.irreps_mid, S.instructions = otp.build_instructions(prev, edge_sh, S.conv_irreps_out)
        S.use_sc = (li != 0) and cfg.get("convnet_sc", True)
        S.radial_depth = cfg.get("radial_mlp_depth", 1)
        S.radial_width = cfg.get("radial_mlp_width", 128)
        S.weight_numel = otp.weight_numel(prev, edge_sh, S.instructions)
        S.node_attrs = node_attrs
        S.edge_sh = edge_sh
        # Gate output: activated scalars (+) gated (tanh keeps 0o odd, silu keeps 0e)
        prev = scalars + gated if S.nonlin == "gate" else S.conv_irreps_out
        S.irreps_out = prev
        specs.append(S)
    return specs


def energy_model(data, cfg, weights, specs=None):
    """SequentialGraphNetwork.forward of the energy model; returns (total_energy [F,1] float64, per-atom energy)."""
    specs = specs or build_specs(cfg)
    dt = {"float32": torch.float32, "float64": torch.float64}[cfg.get("model_dtype", "float32")]
    r_max = float(cfg["r_max"])
    pos = data["pos"]
    edge_index = data["edge_index"]
    types = data["atom_types"].view(-1)
    batch = data.get("batch")
    # type_embed (nequip/nn/embedding/node.py:146-175)
    node_attrs = torch.nn.functional.embedding(types, weights["type_embed.embed_module.weight"])
    x = node_attrs
    # spharm / edge_norm / bessel_encode / factor
    if "edge_vectors" in data:
        vec = data["edge_vectors"]
    else:
        vec = onn.edge_vectors(pos, edge_index, data.get("cell"), data.get("edge_cell_shift"), batch)
    edge_attrs = onn.sh_edge_attrs(vec, cfg["l_max"], dt)
    pt = cfg.get("per_edge_type_cutoff_table")  # [T, T] float64, rows = centre type (EdgeLengthNormalizer, _edge.py:27-52)
    bw = weights.get("bessel_encode.bessel_weights")  # trained roots (bessel_trainable), else the buffer 1..num_bessels
    edge_emb, _ = onn.bessel_embedding(
        vec, r_max, cfg.get("num_bessels", 8), cfg.get("polynomial_cutoff_p", 6), dt,
        bessel_weights=None if bw is None else bw.to(vec.dtype).view(1, -1),
        per_edge_type_cutoff=None if pt is None else torch.as_tensor(pt, dtype=vec.dtype),
        edge_types=None if pt is None else torch.stack([types[edge_index[0]], types[edge_index[1]]]),
    )
    norm = torch.tensor(1.0 / math.sqrt(cfg["avg_num_neighbors"]), dtype=dt)  # nequip/nn/norm.py:39
    acts_norm = cfg.get("convnet_nonlinearity_scalars", {"e": "silu"})["e"]
    for li, S in enumerate(specs):
        pre = f"layer{li}_convnet.conv."
        # InteractionBlock.forward, nequip/nn/interaction_block.py:158-207
        if S.use_sc:
            sc = onn.fully_connected_tp(x, node_attrs, weights[pre + "sc.weight"], S.feature_irreps_in, S.node_attrs,
                                        S.conv_irreps_out)
        x = onn.o3_linear(x, weights[pre + "linear_1.weight"], S.feature_irreps_in, S.feature_irreps_in)
        x = norm * x
        mlp_w = [weights[pre + f"edge_mlp.mlp.{2 * k}.weight"] for k in range(S.radial_depth + 1)]
        w = onn.scalar_mlp(edge_emb, mlp_w, "silu")
        x = otp.tp_scatter(x, edge_attrs, w, edge_index[0], edge_index[1], S.feature_irreps_in, S.edge_sh,
                           S.irreps_mid, S.instructions, edge_chunk=cfg.get("oracle_edge_chunk"))
        x = onn.o3_linear(x, weights[pre + "linear_2.weight"], ir.simplify(S.irreps_mid), S.conv_irreps_out)
        if S.use_sc:
            x = x + sc
        # ConvNetLayer: Gate, nequip/nn/convnetlayer.py:162-164
        if S.nonlin == "gate":
            x = onn.gate(x, S.irreps_scalars, S.act_scalars, S.irreps_gates, S.act_gates, S.irreps_gated)
        else:
            x = onn.norm_activation(x, S.conv_irreps_out, acts_norm)
    # readout: ScalarMLP(output_dim=1, depth 0) -> PerTypeScaleShift (float64) -> AtomwiseReduce
    e_atom = onn.scalar_mlp(x, [weights["per_atom_energy_readout.mlp_module.mlp.0.weight"]], "silu")
    e_atom = e_atom.to(torch.float64)
    scales = weights.get("per_type_energy_scale_shift.scales")
    shifts = weights.get("per_type_energy_scale_shift.shifts")
    if scales is not None and scales.numel() > 0:
        sc_ = scales if scales.numel() == 1 else torch.nn.functional.embedding(types, scales.view(-1, 1))
        e_atom = sc_ * e_atom
    if shifts is not None and shifts.numel() > 0:
        sh_ = shifts if shifts.numel() == 1 else torch.nn.functional.embedding(types, shifts.view(-1, 1))
        e_atom = sh_ + e_atom
    if batch is not None:
        nframes = int(batch.max()) + 1 if "num_frames" not in data else int(data["num_frames"])
        total = otp.scatter(e_atom, batch, nframes)
    else:
        total = e_atom.sum(dim=0, keepdim=True)
    return total, e_atom


def energy_forces(data, cfg, weights, specs=None, with_virial=False, create_graph=False):
    """ForceStressOutput.forward (nequip/nn/grad_output.py:107-298): forces = -dE/dpos by autograd
    (``create_graph=True`` = training mode, grad_output.py:220)."""
    data = dict(data)
    pos = data["pos"].detach().clone().requires_grad_(True)
    data["pos"] = pos
    batch = data.get("batch")
    if with_virial:
        nb = 1 if batch is None else int(batch.max()) + 1
        disp = torch.zeros((3, 3) if nb == 1 else (nb, 3, 3), dtype=pos.dtype, requires_grad=True)
        sym = 0.5 * (disp + disp.transpose(-1, -2))
        if nb == 1:
            data["pos"] = pos + torch.sum(pos.view(-1, 3, 1) * sym, 1)
            if data.get("cell") is not None:
                cell = data["cell"].view(3, 3)
                data["cell"] = (cell + torch.sum(cell.view(3, 3, 1) * sym, 1)).view(1, 3, 3)
        else:
            data["pos"] = pos + torch.bmm(pos.unsqueeze(-2), torch.index_select(sym, 0, batch)).squeeze(-2)
            if data.get("cell") is not None:
                cell = data["cell"].view(-1, 3, 3)
                data["cell"] = cell + torch.bmm(cell, sym)
    total, e_atom = energy_model(data, cfg, weights, specs)
    wrt = [pos] + ([disp] if with_virial else [])
    grads = torch.autograd.grad([total.sum()], wrt, create_graph=create_graph)
    if create_graph:
        out = {"total_energy": total, "atomic_energy": e_atom, "forces": -grads[0]}
    else:
        out = {"total_energy": total.detach(), "atomic_energy": e_atom.detach(), "forces": -grads[0]}
    if with_virial:
        out["virial"] = -grads[1].view(-1, 3, 3)
    return out
