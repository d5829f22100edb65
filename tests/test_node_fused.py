"""The node side of a layer boundary as one launch per direction (``nqa_node_fused``, csrc/node_fused.h; SURVEY.md 8(f)-2:
``nequip/nn/convnetlayer.py:156-170`` Gate + ``nequip/nn/interaction_block.py:175-181`` linear_1 / self-connection).

CPU: the merged kernel tables (``nqa_node_fused_plan``) of the forward (gate folded into both consumers, two destinations)
and of the backward (two operand sets accumulated into one tile, gate backward as epilogue) for the BASELINE cfg-3 layer
shapes; the block form of the gate against its column tables.
GPU: the fused launches against (a) the separate launches (``nqa_gate`` + ``nqa_node_linear_packed``), (b) the ATen
formulation of e3nn's Gate / Linear / FullyConnectedTensorProduct in float64 -- forward values and the gradient w.r.t. the
pre-gate rows; the model takes the fused path in eval mode and gives the separate-launch energies / forces.
"""
import ctypes
import os
import struct

import pytest
import torch

from nequip_amd import _lib
from nequip_amd.o3 import _node_kernels as nk
from nequip_amd.o3.irreps import Irreps
from nequip_amd.o3.modules import FullyConnectedTensorProduct, Gate, Linear

SILU = torch.nn.functional.silu


def _layer(hidden: str, n_attr: int = 8):
    """(gate, linear_1, sc) of a middle layer whose hidden irreps are ``hidden`` (uniform nequip construction)."""
    hid = Irreps(hidden)
    scalars = Irreps([(m, ir) for m, ir in hid if ir.l == 0])
    gated = Irreps([(m, ir) for m, ir in hid if ir.l > 0])
    gates = Irreps([(m, (0, 1)) for m, _ in gated])
    acts = {1: SILU, -1: torch.tanh}
    gate = Gate(scalars, [acts[ir.p] for _, ir in scalars], gates, [SILU for _ in gates], gated)
    x_ir = gate.irreps_out
    lin1 = Linear(x_ir, x_ir)
    conv_out = gate.irreps_in  # (the next layer's convolution output = its own gate's input: same shape here)
    sc = FullyConnectedTensorProduct(x_ir, Irreps([(n_attr, (0, 1))]), conv_out)
    return gate, lin1, sc


def _part(meta, n_types, dim_in, dim_out, accumulate=False, in_gate=None, scale=1.0):
    ct, nchunks, it, ninstr = meta.host_tables("fwd")
    p = _lib.NodePart()
    p.chunk_table = ctypes.cast(ct, ctypes.c_void_p)
    p.instr_table = ctypes.cast(it, ctypes.c_void_p)
    p.n_chunks, p.n_instr, p.n_types, p.dim_in, p.dim_out = nchunks, ninstr, n_types, dim_in, dim_out
    p.accumulate = 1 if accumulate else 0
    p.scale = scale
    if in_gate is not None:
        arr, n = in_gate.blocks_c()
        p.in_gate = ctypes.cast(arr, ctypes.c_void_p)
        p.n_in_gate = n
    return p


def _plan(parts, out_gate=None):
    lib = _lib.load()
    arr = (_lib.NodePart * len(parts))(*parts)
    cb = (ctypes.c_int32 * (12 * 64))()
    ib = (ctypes.c_int32 * (8 * 64))()
    og, nog = (None, 0)
    if out_gate is not None:
        a, nog = out_gate.blocks_c()
        og = ctypes.cast(a, ctypes.c_void_p)
    rc = lib.nqa_node_fused_plan(ctypes.cast(arr, ctypes.c_void_p), len(parts), og, nog, cb, len(cb), ib, len(ib))
    if rc < 0:
        raise RuntimeError(lib.nqa_last_error().decode())
    nc, ni = rc >> 16, rc & 0xFFFF
    raw_c, raw_i = bytes(cb), bytes(ib)
    chunks = [struct.unpack("<11if", raw_c[48 * k: 48 * k + 48]) for k in range(nc)]
    instr = [struct.unpack("<6ifi", raw_i[32 * k: 32 * k + 32]) for k in range(ni)]
    return chunks, instr


def test_gate_blocks_agree_with_the_column_tables():
    for hidden in ("64x0e+64x1o+64x2e", "32x0e+32x0o+32x1e+32x1o", "128x0e+128x1o+128x2e+128x3o"):
        gate, _, _ = _layer(hidden)
        meta = gate._kernel_meta
        assert meta.fusable()
        rec = struct.Struct("<iiiidii")
        cols = [rec.unpack(meta._fwd[32 * c: 32 * c + 32]) for c in range(meta.dout)]
        seen = [False] * meta.dout
        for out_off, d, mul, val_off, gate_off, act, cst in meta.blocks:
            for u in range(mul):
                for m in range(d):
                    c = out_off + u * d + m
                    src, gcol, a, _, k, _, _ = cols[c]
                    assert src == val_off + u * d + m and a == act and abs(k - cst) < 1e-12
                    assert gcol == (gate_off + u if gate_off >= 0 else -1)
                    seen[c] = True
        assert all(seen)
    odd = Gate(Irreps("6x0e"), [SILU], Irreps("6x0e"), [SILU], Irreps("6x1o"))  # multiplicities that are not multiples of 4
    assert not odd._kernel_meta.fusable()


def test_forward_plan_cfg3_gate_folded_into_both_consumers():
    gate, lin1, sc = _layer("64x0e+64x1o+64x2e")
    gm = gate._kernel_meta
    assert (gm.din, gm.dout) == (704, 576)
    parts = [_part(lin1._meta, 1, 704, 576, in_gate=gm, scale=0.16), _part(sc._meta, 2, 704, 704, in_gate=gm)]
    chunks, instr = _plan(parts)
    assert len(chunks) == 3 + 5 and len(instr) == 6
    # destinations: linear_1's three blocks -> 0, the self-connection's 192x0e (three chunks) + 1o + 2e -> 1
    assert sorted(c[6] for c in chunks) == [0, 0, 0, 1, 1, 1, 1, 1]
    assert all(c[7] == 0 for c in chunks)  # plain epilogues
    by_set = {0: [], 1: []}
    for x_off, mul_in, frag_off, exp_off, gate_off, act, cst, st in instr:
        by_set[st].append((x_off, mul_in, gate_off, act))
        assert abs(cst - gm.blocks[0][6]) < 1e-6
    for st in (0, 1):  # 0e: activated in place; 1o / 2e: values at 192 / 384 of h, gate scalars at 64 / 128
        assert by_set[st] == [(0, 64, -1, 1), (192, 64, 64, 1), (384, 64, 128, 1)]
    # every chunk's instruction range holds exactly the one instruction of its own operand set and irrep
    for o_off, d, mul_out, c0, ib, ie, dst, *_ in chunks:
        assert ie - ib == 1 and instr[ib][7] == dst
        assert {1: 0, 3: 192, 5: 384}[d] == instr[ib][0]


def test_backward_plan_cfg3_two_sets_one_tile_and_gate_epilogue():
    gate, lin1, sc = _layer("64x0e+64x1o+64x2e")
    gm = gate._kernel_meta
    t1, ts = nk._transposed(lin1._meta), nk._transposed(sc._meta)
    parts = [_part(t1, 1, 576, 704), _part(ts, 2, 704, 704, accumulate=True)]
    chunks, instr = _plan(parts, out_gate=gm)
    assert len(chunks) == 3 and len(instr) == 6
    chunks.sort(key=lambda c: c[1])
    eps = [(c[1], c[7], c[8], c[9]) for c in chunks]
    assert eps == [(1, 1, 0, -1), (3, 2, 192, 64), (5, 2, 384, 128)]
    for o_off, d, mul_out, c0, ib, ie, dst, ep, ev, eg, act, cst in chunks:
        assert ie - ib == 2 and dst == 0 and act == 1
        assert [instr[q][7] for q in range(ib, ie)] == [0, 1]  # linear_1^T's rows first, then the self-connection's
        assert all(instr[q][4] == -2 for q in range(ib, ie))  # gradients are read as they are
    # an output block that is no block of the gate / parts that do not line up are refused
    with pytest.raises(RuntimeError, match="same output chunks"):
        short = nk.NodeLinearMeta(Irreps("64x0e+64x1o"), sc._meta.irreps_out, [(0, 0), (1, 1)])  # two output blocks, not three
        _plan([_part(t1, 1, 576, 704), _part(nk._transposed(short), 2, 704, 704, accumulate=True)], out_gate=gm)
    with pytest.raises(RuntimeError, match="one scale"):
        _plan([_part(t1, 1, 576, 704, scale=0.5), _part(ts, 2, 704, 704, accumulate=True)], out_gate=gm)


# ---- GPU ------------------------------------------------------------------------------------------------------------
def _reference_f64(h, types, table, gate, lin1, sc, scale1):
    """e3nn semantics in float64 with ATen ops (the modules' CPU formulation)."""
    import copy

    g64, l64, s64 = (copy.deepcopy(m).double() for m in (gate, lin1, sc))
    x = g64(h)
    return scale1 * l64(x), s64.forward_typed(x, types, table)


def _reference_oracle_f64(h, types, table, gate, lin1, sc, scale1):
    """The same stage through the ORACLE's restatements of e3nn's Gate / Linear / FullyConnectedTensorProduct
    (oracle/nn.py: gate, o3_linear, fully_connected_tp), float64 -- independent of the product's own
    ATen formulation in o3/modules.py."""
    from oracle import nn as onn

    names = {SILU: "silu", torch.tanh: "tanh"}
    x = onn.gate(h, str(gate.irreps_scalars), [names[a] for a in gate.act_scalars], str(gate.irreps_gates),
                 [names[a] for a in gate.act_gates], str(gate.irreps_gated))
    y1 = onn.o3_linear(x, lin1.weight.detach().double().cpu(), str(lin1.irreps_in), str(lin1.irreps_out)) * scale1
    ys = onn.fully_connected_tp(x, table[types], sc.weight.detach().double().cpu(), str(sc.irreps_in1), str(sc.irreps_in2),
                                str(sc.irreps_out))
    return y1, ys


CASES = [
    ("64x0e+64x1o+64x2e", 2, 1003),                      # cfg-3 middle layer; N not a multiple of 32 / 10 / 6
    ("32x0e+32x0o+32x1e+32x1o", 4, 105),                  # tutorial shape (parity=True: tanh on odd scalars), 4 species
    ("128x0e+128x1o+128x2e+128x3o", 1, 257),              # cfg-5 (two 64-channel chunks per block, d = 7)
    ("64x0e+64x1o+64x2e+64x3o+64x4e", 5, 70),             # d = 9, five species
]


@pytest.mark.gpu
@pytest.mark.parametrize("stage_loop", ["default", "one_wait_per_stage"])
@pytest.mark.parametrize("hidden,n_types,n_atoms", CASES)
def test_fused_stage_matches_separate_launches_and_float64(device, hidden, n_types, n_atoms, stage_loop, monkeypatch):
    # `one_wait_per_stage`: the opt-in stage loop of node_fused_kernel<true> / node_linear_pipe_kernel (NQA_NODE_PIPE=1, read
    # at every launch; measured no faster) has to give the default's results
    monkeypatch.setenv("NQA_NODE_PIPE", "1" if stage_loop == "one_wait_per_stage" else "0")
    torch.manual_seed(1)
    gate, lin1, sc = _layer(hidden)
    gate, lin1, sc = gate.eval(), lin1.eval(), sc.eval()
    gm = gate._kernel_meta
    table = torch.randn(n_types, 8, dtype=torch.float64)
    types = torch.randint(0, n_types, (n_atoms,))
    h = torch.randn(n_atoms, gm.din, dtype=torch.float64)
    h[::7] *= 30.0  # a few rows far out in the activations' tails
    h[5] = 0.0
    scale1 = 0.1617
    ref1, refs = _reference_f64(h, types, table, gate, lin1, sc, scale1)
    c1 = torch.randn_like(ref1)
    cs = torch.randn_like(refs)
    hr = h.clone().requires_grad_(True)
    o1, os_ = _reference_f64(hr, types, table, gate, lin1, sc, scale1)
    (g_ref,) = torch.autograd.grad((o1 * c1).sum() + (os_ * cs).sum(), hr)

    gate, lin1, sc = gate.to(device), lin1.to(device), sc.to(device)
    hd = h.float().to(device).requires_grad_(True)
    td, tabd = types.to(device), table.float().to(device)
    wp1 = lin1.eval_weights(device, torch.float32)
    wps = sc.eval_weights_typed(tabd, torch.float32)
    x1, s = nk.fused_node_stage(hd, td, gm, wp1, lin1._meta, scale1, wps, sc._meta)
    (g,) = torch.autograd.grad((x1 * c1.float().to(device)).sum() + (s * cs.float().to(device)).sum(), hd)

    def close(a, b, what):
        scale = float(b.abs().max())
        err = float((a.double().cpu() - b).abs().max())
        assert err <= 2e-6 * max(1.0, scale), f"{what}: {err:.3e} (scale {scale:.3e})"

    close(x1, ref1, "linear_1(Gate(h)) / sqrt(avg)")
    close(s, refs, "sc(Gate(h))")
    close(g, g_ref, "gradient w.r.t. the pre-gate rows")
    # ... and against the oracle's restatement of the three e3nn modules (not the product's own ATen formulation)
    hq = h.clone().requires_grad_(True)
    q1, qs = _reference_oracle_f64(hq, types, table, gate.cpu(), lin1.cpu(), sc.cpu(), scale1)
    (g_orc,) = torch.autograd.grad((q1 * c1).sum() + (qs * cs).sum(), hq)
    close(x1, q1.detach(), "linear_1(Gate(h)) / sqrt(avg) vs oracle.nn")
    close(s, qs.detach(), "sc(Gate(h)) vs oracle.nn")
    close(g, g_orc, "gradient w.r.t. the pre-gate rows vs oracle.nn")
    gate, lin1, sc = gate.to(device), lin1.to(device), sc.to(device)

    # the separate launches (nqa_gate, nqa_node_linear_packed x 2; backward x 3 + an add)
    h2 = h.float().to(device).requires_grad_(True)
    x = nk.gate(h2, gm)
    y1 = nk.node_linear(x, wp1, None, lin1._meta, scale=scale1)
    y2 = nk.node_linear(x, wps, td.contiguous(), sc._meta)
    (g2,) = torch.autograd.grad((y1 * c1.float().to(device)).sum() + (y2 * cs.float().to(device)).sum(), h2)
    for a, b, what in ((x1, y1, "x1"), (s, y2, "sc"), (g, g2, "grad")):
        err = float((a - b).abs().max())
        assert err <= 2e-6 * max(1.0, float(b.abs().max())), (what, err)


@pytest.mark.gpu
def test_fused_stage_without_self_connection_and_without_gate(device):
    """One part only (use_sc = False), and parts without an input gate (plain two-destination launch)."""
    torch.manual_seed(2)
    gate, lin1, sc = _layer("64x0e+64x1o+64x2e")
    gm = gate._kernel_meta
    n = 333
    h = torch.randn(n, gm.din, device=device, requires_grad=True)
    lin1, sc = lin1.to(device).eval(), sc.to(device).eval()
    wp1 = lin1.eval_weights(device, torch.float32)
    x1, s = nk.fused_node_stage(h, None, gm, wp1, lin1._meta, 0.3)
    assert s is None
    c = torch.randn_like(x1)
    (g,) = torch.autograd.grad((x1 * c).sum(), h)
    h2 = h.detach().clone().requires_grad_(True)
    y = nk.node_linear(nk.gate(h2, gm), wp1, None, lin1._meta, scale=0.3)
    (g2,) = torch.autograd.grad((y * c).sum(), h2)
    assert float((x1 - y).abs().max()) <= 2e-6 * max(1.0, float(y.abs().max()))
    assert float((g - g2).abs().max()) <= 2e-6 * max(1.0, float(g2.abs().max()))
    # no gate: x = Gate-output-shaped rows as they are
    x = torch.randn(n, gm.dout, device=device, requires_grad=True)
    types = torch.randint(0, 3, (n,), device=device)
    table = torch.randn(3, 8, device=device)
    wps = sc.eval_weights_typed(table, torch.float32)
    a1, a2 = nk.fused_node_stage(x, types, None, wp1, lin1._meta, 0.3, wps, sc._meta)
    c1, c2 = torch.randn_like(a1), torch.randn_like(a2)
    (g,) = torch.autograd.grad((a1 * c1).sum() + (a2 * c2).sum(), x)
    x2 = x.detach().clone().requires_grad_(True)
    b1 = nk.node_linear(x2, wp1, None, lin1._meta, scale=0.3)
    b2 = nk.node_linear(x2, wps, types, sc._meta)
    (g2,) = torch.autograd.grad((b1 * c1).sum() + (b2 * c2).sum(), x2)
    for a, b in ((a1, b1), (a2, b2), (g, g2)):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))


@pytest.mark.gpu
def test_model_takes_the_fused_path_in_eval_mode(device, monkeypatch):
    """Eval mode on the GPU: every gate between two convolution layers is deferred and consumed by ``nqa_node_fused`` (two
    launches per layer boundary and evaluation); ``NQA_NO_NODE_FUSION=1`` gives the separate launches and the same energies /
    forces to rounding."""
    from nequip_amd.data import AtomicDataDict
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=3, seed=2)
    data = AtomicDataDict.to_device(syn.make_data(pos, types, 4.5, cell), device)
    model = NequIPGNNModel(seed=0, model_dtype="float32", r_max=4.5, type_names=names, num_layers=3, l_max=2, parity=False,
                           num_features=64, radial_mlp_depth=1, radial_mlp_width=128, avg_num_neighbors=38.0).to(device).eval()
    calls = []
    real = nk.launch_fused
    monkeypatch.setattr(nk, "launch_fused", lambda *a, **k: (calls.append(k.get("out_gate") is not None), real(*a, **k))[1])
    out = model(dict(data))
    assert calls == [False, False, True, True], calls  # two boundaries: forward launches, then their backward launches
    monkeypatch.setenv("NQA_NO_NODE_FUSION", "1")
    calls.clear()
    ref = model(dict(data))
    assert calls == []
    fs = max(1.0, float(ref["forces"].abs().max()))
    torch.testing.assert_close(out["total_energy"], ref["total_energy"], atol=2e-5 * len(pos), rtol=2e-6)
    torch.testing.assert_close(out["forces"], ref["forces"], atol=3e-6 * fs, rtol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("n_types,per_type", [(1, False), (3, True), (2, False)])
def test_energy_head_matches_the_module_chain(device, n_types, per_type):
    """``nqa_energy_head`` against Gate(scalars) -> ScalarMLP(depth 0) -> PerTypeScaleShift as the modules compute them
    (float32 readout, float64 scale / shift), values and the gradient w.r.t. the pre-gate scalars."""
    from nequip_amd.nn._energy_head import energy_head

    torch.manual_seed(3)
    n, d = 1001, 64
    h = (torch.randn(n, d) * 2.0).to(device).requires_grad_(True)
    w = (torch.rand(d, 1) * 2 * 3 ** 0.5 - 3 ** 0.5).to(device)
    alpha = 1.0 / d ** 0.5
    types = torch.randint(0, n_types, (n,), device=device)
    scales = (torch.rand(n_types if per_type else 1, dtype=torch.float64) + 0.5).to(device)
    shifts = torch.randn(n_types if per_type else 1, dtype=torch.float64).to(device)
    cst = 1.6791767923989418
    e = energy_head(h, (w.view(-1) * alpha).contiguous(), scales, shifts, types, 1, cst)
    assert e.dtype == torch.float64 and e.shape == (n, 1)
    (g,) = torch.autograd.grad(e.sum(), h)
    h2 = h.detach().clone().requires_grad_(True)
    x = cst * SILU(h2)
    e32 = torch.mm(x, w * alpha)
    sc = scales[types].view(-1, 1) if per_type else scales.view(1, 1)
    sh = shifts[types].view(-1, 1) if per_type else shifts.view(1, 1)
    ref = torch.addcmul(sh, sc, e32.to(torch.float64))
    (g2,) = torch.autograd.grad(ref.sum(), h2)
    torch.testing.assert_close(e, ref, atol=2e-6 * float(ref.abs().max()), rtol=0)
    torch.testing.assert_close(g, g2, atol=2e-6 * float(g2.abs().max()), rtol=0)
    # a weighted sum as the loss (a non-trivial incoming gradient), scales only
    c = torch.randn(n, 1, dtype=torch.float64, device=device)
    e = energy_head(h, (w.view(-1) * alpha).contiguous(), scales, None, types, 1, cst)
    (g,) = torch.autograd.grad((e * c).sum(), h)
    ref = sc * torch.mm(cst * SILU(h2), w * alpha).to(torch.float64)
    (g2,) = torch.autograd.grad((ref * c).sum(), h2)
    torch.testing.assert_close(g, g2, atol=2e-6 * float(g2.abs().max()), rtol=0)


@pytest.mark.gpu
def test_model_runs_the_energy_head_and_matches_the_module_chain(device, monkeypatch):
    from nequip_amd.data import AtomicDataDict
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.nn import _energy_head
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=3, seed=4)
    data = AtomicDataDict.to_device(syn.make_data(pos, types, 4.5, cell), device)
    model = NequIPGNNModel(seed=0, model_dtype="float32", r_max=4.5, type_names=names, num_layers=2, l_max=1, parity=False,
                           num_features=32, radial_mlp_depth=1, radial_mlp_width=64, avg_num_neighbors=38.0,
                           per_type_energy_scales={"H": 1.3, "O": 0.7}, per_type_energy_shifts={"H": -1.0, "O": 2.0}
                           ).to(device).eval()
    calls = []
    real = _energy_head._launch
    monkeypatch.setattr(_energy_head, "_launch", lambda b, *a: (calls.append(b), real(b, *a))[1])
    out = model(dict(data))
    assert calls == [0, 1]
    monkeypatch.setenv("NQA_NO_ENERGY_HEAD", "1")
    calls.clear()
    ref = model(dict(data))
    assert calls == []
    torch.testing.assert_close(out["atomic_energy"], ref["atomic_energy"], atol=1e-5, rtol=1e-6)
    torch.testing.assert_close(out["total_energy"], ref["total_energy"], atol=1e-5 * len(pos), rtol=1e-6)
    torch.testing.assert_close(out["forces"], ref["forces"], atol=3e-6 * max(1.0, float(ref["forces"].abs().max())), rtol=0)


@pytest.mark.gpu
def test_atom_order_is_free_and_type_grouping_skips_stages(device, monkeypatch):
    """``atom_order`` of ``nqa_node_fused``: any permutation gives the results of the natural order (bitwise: the same
    arithmetic per atom); the default groups the atoms by type, which only changes which typed stages a unit runs."""
    torch.manual_seed(5)
    gate, lin1, sc = _layer("64x0e+64x1o+64x2e")
    gate, lin1, sc = gate.to(device).eval(), lin1.to(device).eval(), sc.to(device).eval()
    gm = gate._kernel_meta
    n, n_types = 997, 5
    h = torch.randn(n, gm.din, device=device)
    types = torch.randint(0, n_types, (n,), device=device)
    table = torch.randn(n_types, 8, device=device)
    wp1 = lin1.eval_weights(device, torch.float32)
    wps = sc.eval_weights_typed(table, torch.float32)

    def run(order):
        parts = [nk.FusedPart(h, wp1, lin1._meta, 0.2, in_gate=gm), nk.FusedPart(h, wps, sc._meta, 1.0, in_gate=gm)]
        return nk.launch_fused(parts, types, order=order)

    ref = run(None)
    for order in (nk.type_order(types), torch.randperm(n, device=device).to(torch.int32),
                  torch.arange(n - 1, -1, -1, device=device, dtype=torch.int32)):
        assert sorted(order.tolist()) == list(range(n))
        out = run(order.contiguous())
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
    srt = types[nk.type_order(types).long()]
    assert bool((srt[1:] >= srt[:-1]).all())
    # backward with the output gate, both orders
    g1, gs = torch.randn(n, gm.dout, device=device), torch.randn(n, gm.din, device=device)
    t1, ts = nk._transposed(lin1._meta), nk._transposed(sc._meta)

    def run_bwd(order):
        parts = [nk.FusedPart(g1, nk.meta_transposed_weights(lin1._meta, wp1), t1),
                 nk.FusedPart(gs, nk.meta_transposed_weights(sc._meta, wps), ts, accumulate=True)]
        return nk.launch_fused(parts, types, out_gate=gm, gate_h=h, order=order)[0]

    assert torch.equal(run_bwd(nk.type_order(types)), run_bwd(None))
