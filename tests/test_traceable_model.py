"""Whole-model traceability (the reference compiles a model by symbolic tracing, nequip/nn/compile.py:176-191):
inside `traceable_forms()` every module picks a form `make_fx` can follow -- dispatcher ops for the tensor-product scatter
and the edge embedding, ATen formulations for the rest -- so energy AND forces (autograd inside the model) trace into one
graph.  Traced here without a GPU through fake tensors; on the GPU the traced graph must reproduce the eager model."""
import pytest
import torch

from nequip_amd.data import AtomicDataDict
from nequip_amd.model import NequIPGNNModel
from nequip_amd.utils import synthetic as syn
from nequip_amd.utils.tracing import traceable, traceable_forms

FIELDS = ("pos", "cell", "edge_index", "edge_cell_shift", "atom_types")


def _model_and_data(device, parity=False, l_max=2):
    pos, types, cell, names = syn.water_box(n_side=2, seed=5)
    data = syn.make_data(pos, types, 4.0, cell)
    model = NequIPGNNModel(seed=3, model_dtype="float32", r_max=4.0, type_names=names, num_layers=2, l_max=l_max,
                           parity=parity, num_features=8, radial_mlp_depth=1, radial_mlp_width=16,
                           avg_num_neighbors=float(data["edge_index"].shape[1] / data["pos"].shape[0]),
                           per_type_energy_scales=1.0, per_type_energy_shifts=0.0).to(device).eval()
    data = AtomicDataDict.to_device(data, device)
    return model, {k: data[k] for k in FIELDS}


def _fn(model):
    def f(inputs):
        out = model(dict(inputs))
        return out["total_energy"], out["forces"], out["virial"]

    return f


def _functional(model):
    """(params, buffers, inputs) -> outputs: the weights are graph inputs, as in the reference's compile path
    (nequip/nn/compile.py:150-191)."""
    def f(params, buffers, inputs):
        out = torch.func.functional_call(model, (params, buffers), (dict(inputs),))
        return out["total_energy"], out["forces"], out["virial"]

    return f, dict(model.named_parameters()), dict(model.named_buffers())


def test_flag_scopes():
    assert not traceable()
    with traceable_forms():
        assert traceable()
        with traceable_forms(False):
            assert traceable()
    assert not traceable()


def test_whole_model_traces_with_fake_tensors_on_cpu():
    from torch.fx.experimental.proxy_tensor import make_fx

    model, inputs = _model_and_data(torch.device("cpu"))
    f, params, buffers = _functional(model)
    with traceable_forms():
        gm = make_fx(f, tracing_mode="fake")(params, buffers, inputs)
    targets = {str(n.target) for n in gm.graph.nodes if n.op == "call_function"}
    ours = {t for t in targets if t.startswith("nequip_amd.")}
    assert {"nequip_amd.tp_scatter_fwd.default", "nequip_amd.tp_scatter_bwd.default",
            "nequip_amd.edge_embed_fwd.default", "nequip_amd.edge_embed_bwd.default"} <= ours, ours
    assert all(t.startswith(("aten.", "nequip_amd.", "<built-in", "_operator", "prims.")) for t in targets), targets
    # outputs: energy [1, 1], forces [N, 3], virial [1, 3, 3]
    outs = [n for n in gm.graph.nodes if n.op == "output"][0].args[0]
    assert len(outs) == 3


def test_whole_model_traces_symbolically_on_cpu():
    """`tracing_mode="symbolic"` (what nequip/nn/compile.py:176-191 uses): the edge count stays a symbol."""
    from torch.fx.experimental.proxy_tensor import make_fx

    model, inputs = _model_and_data(torch.device("cpu"))
    f, params, buffers = _functional(model)
    with traceable_forms():
        gm = make_fx(f, tracing_mode="symbolic")(params, buffers, inputs)
    shift = [n for n in gm.graph.nodes if n.op == "placeholder"][-2].meta["val"]
    assert not isinstance(shift.shape[0], int), "the number of edges must be symbolic"


def test_fold_constants_leaves_no_weight_only_computation_in_the_graph():
    """`fold_constants` (deployment-time freezing, utils/tracing.py): afterwards no call node depends on weight inputs /
    folded buffers alone, the graph got smaller, and the folded buffers are what the graph computed from the weights
    (checked on one of them: the path-normalised o3.Linear weights)."""
    from nequip_amd.utils.tracing import fold_constants, trace_model

    model, inputs = _model_and_data(torch.device("cpu"))
    gm, params, buffers = trace_model(model, inputs, tracing_mode="symbolic")
    before = sum(1 for n in gm.graph.nodes if n.op == "call_function")
    n_weights = len(params) + len(buffers)
    made = fold_constants(gm, params, buffers)
    after = sum(1 for n in gm.graph.nodes if n.op == "call_function")
    assert made > 0 and after < before, (made, before, after)
    const = {n for i, n in enumerate(n for n in gm.graph.nodes if n.op == "placeholder") if i < n_weights}
    const |= {n for n in gm.graph.nodes if n.op == "get_attr"}
    for n in gm.graph.nodes:
        if n.op == "call_function":
            ins = n.all_input_nodes
            assert not ins or any(i not in const for i in ins), f"{n.format_node()} depends on constants alone"
    lin = [m for m in model.modules() if type(m).__name__ == "Linear" and getattr(m, "weight_numel", 0) > 0][0]
    want = lin.weight.detach() * lin._scale_vec
    folded = [b for name, b in gm.named_buffers() if name.startswith("_folded_") and b.numel() == want.numel()]
    assert any(torch.equal(b.reshape(-1), want) for b in folded)


def test_constant_cache_is_keyed_on_storage_identity_and_version():
    """utils/constcache.py: views / new wrapper objects of one storage hit, a clone or an in-place write misses."""
    from nequip_amd.utils import constcache

    constcache.clear()
    calls = []
    w = torch.arange(12.0).view(3, 4)

    def build():
        calls.append(1)
        return len(calls)

    assert constcache.get(w, "t", build) == 1
    assert constcache.get(w.view(3, 4), "t", build) == 1          # another tensor object over the same memory
    assert constcache.get(w.detach(), "t", build) == 1
    assert constcache.get(w, "other tag", build) == 2
    assert constcache.get(w.clone(), "t", build) == 3             # same values, another storage
    assert constcache.get(w[1:], "t", build) == 4                 # same storage, another data pointer / shape
    w.add_(1.0)                                                   # PyTorch sees the write: version counter
    assert constcache.get(w, "t", build) == 5
    constcache.clear()
    assert constcache.get(w, "t", build) == 6


def test_energy_graph_of_the_benchmark_shape_on_fake_gpu_tensors():
    """No GPU needed: fake CUDA tensors take the GPU branches of every module, so the FORWARD graph of a cfg-3-shaped model
    (64 features, radial MLP 8-128-W) is checked here -- the edge side of every convolution is ONE op (`radial_tp_fwd`: the
    pairing decision lives behind the dispatcher), every layer boundary ONE (`node_stage_fwd`: Gate + linear_1 +
    self-connection), the readout ONE (`energy_head_fwd`), and no separate gate / radial-MLP / tensor-product op is left.
    (The backward half needs the autograd engine, which needs the device: GPU test below.)"""
    from collections import Counter

    from torch._subclasses.fake_tensor import FakeTensorMode
    from torch.fx.experimental.proxy_tensor import make_fx

    pos, types, cell, names = syn.water_box(n_side=2, seed=5)
    data = syn.make_data(pos, types, 4.0, cell)
    model = NequIPGNNModel(seed=3, model_dtype="float32", r_max=4.0, type_names=names, num_layers=3, l_max=2, parity=False,
                           num_features=64, radial_mlp_depth=1, radial_mlp_width=128, avg_num_neighbors=20.0,
                           per_type_energy_scales=1.0, per_type_energy_shifts=0.0).eval()
    energy = model.model.func  # (the energy model inside ForceStressOutput)
    mode = FakeTensorMode()

    def fake(t):
        with mode:
            return torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device="cuda")

    params = {k: fake(v) for k, v in energy.named_parameters()}
    buffers = {k: fake(v) for k, v in energy.named_buffers()}
    inputs = {k: fake(data[k]) for k in FIELDS}

    def f(params, buffers, inputs):
        return torch.func.functional_call(energy, (params, buffers), (dict(inputs),))["total_energy"]

    with mode, traceable_forms():
        gm = make_fx(f, tracing_mode="real")(params, buffers, inputs)
    ours = Counter(str(n.target).split(".")[1] for n in gm.graph.nodes
                   if n.op == "call_function" and str(n.target).startswith("nequip_amd."))
    assert ours["radial_tp_fwd"] == 3 and ours["node_stage_fwd"] == 2 and ours["energy_head_fwd"] == 1, ours
    assert not ({"gate", "radial_mlp_fwd", "tp_scatter_fwd"} & set(ours)), ours
    out = [n for n in gm.graph.nodes if n.op == "output"][0].args[0]
    out = out[0] if isinstance(out, (list, tuple)) else out
    assert tuple(out.meta["val"].shape) == (1, 1) and out.meta["val"].dtype == torch.float64


@pytest.mark.gpu
@pytest.mark.parametrize("parity,l_max", [(False, 2), (True, 1)])
def test_traced_graph_reproduces_the_eager_model(device, parity, l_max):
    from torch.fx.experimental.proxy_tensor import make_fx

    model, inputs = _model_and_data(device, parity, l_max)
    e0, f0, v0 = _fn(model)(inputs)  # eager: fused kernels
    with traceable_forms():
        e1, f1, v1 = _fn(model)(inputs)  # the traceable forms themselves, eagerly
        f, params, buffers = _functional(model)
        gm = make_fx(f, tracing_mode="real")(params, buffers, inputs)
    e2, f2, v2 = gm(params, buffers, inputs)  # the traced graph, outside the context
    for a, b in ((e1, e0), (f1, f0), (v1, v0), (e2, e0), (f2, f0), (v2, v0)):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=2e-5 * max(1.0, float(b.abs().max())))
    # other positions through the same graph (no constants of the traced example baked in)
    inputs2 = dict(inputs)
    inputs2["pos"] = inputs["pos"] + 0.01 * torch.randn_like(inputs["pos"])
    e3, f3, _ = gm(params, buffers, inputs2)
    e4, f4, _ = _fn(model)(inputs2)
    torch.testing.assert_close(e3, e4, rtol=1e-5, atol=2e-5 * max(1.0, float(e4.abs().max())))
    torch.testing.assert_close(f3, f4, rtol=1e-5, atol=2e-5 * max(1.0, float(f4.abs().max())))


@pytest.mark.gpu
def test_trace_model_helper_and_edge_embed_ops(device):
    """`trace_model` (dict outputs) and the edge-embedding ops against the autograd-Function form, second order included."""
    from nequip_amd.nn.embedding._edge import _EdgeEmbedFn
    from nequip_amd.nn.embedding._edge_ops import edge_embed
    from nequip_amd.utils.tracing import trace_model

    model, inputs = _model_and_data(device)
    gm, params, buffers = trace_model(model, inputs, tracing_mode="real")
    out = gm(params, buffers, inputs)
    ref = model(dict(inputs))
    torch.testing.assert_close(out["forces"], ref["forces"], rtol=1e-5, atol=2e-5 * float(ref["forces"].abs().max()))

    g = torch.Generator().manual_seed(2)
    vec = (torch.randn(300, 3, generator=g, dtype=torch.float64) * 2.0).to(device)
    bw = torch.linspace(1.0, 8, 8, dtype=torch.float64, device=device)
    for cfg in (dict(dtype=torch.float32, lmax=2, want_sh=True, want_emb=False, nb=0, rmax_recip=1.0, p=6.0, factor=1.0),
                dict(dtype=torch.float32, lmax=0, want_sh=False, want_emb=True, nb=8, rmax_recip=1 / 4.5, p=6.0, factor=0.31),
                dict(dtype=torch.float64, lmax=3, want_sh=True, want_emb=True, nb=8, rmax_recip=1 / 4.5, p=6.0, factor=0.31)):
        res = []
        for form in (_EdgeEmbedFn.apply, edge_embed):
            v = vec.clone().requires_grad_(True)
            out = form(v, bw, cfg)
            outs = list(out) if isinstance(out, tuple) else [out]
            cot = [torch.randn(o.shape, generator=torch.Generator().manual_seed(7 + i), dtype=o.dtype).to(device).requires_grad_(True)
                   for i, o in enumerate(outs)]
            (gv,) = torch.autograd.grad(outs, [v], cot, create_graph=True)
            c = torch.randn(gv.shape, generator=torch.Generator().manual_seed(11), dtype=gv.dtype).to(device)
            second = torch.autograd.grad([gv], [v] + cot, [c])
            res.append([o.detach() for o in outs] + [gv.detach()] + [s.detach() for s in second])
        for a, b in zip(*res):
            torch.testing.assert_close(a, b, rtol=0, atol=0)  # the same kernels


@pytest.mark.gpu
def test_radial_mlp_ops_match_the_function_form(device):
    """`torch.ops.nequip_amd.radial_mlp_fwd/_bwd/_bwd_bwd` against the module's eager (training-capable) Functions: value,
    gradient w.r.t. the embedding, and the second-order terms w.r.t. the embedding and the incoming gradient."""
    from nequip_amd.nn._mlp_ops import radial_mlp
    from nequip_amd.nn.mlp import ScalarMLPFunction

    torch.manual_seed(0)
    mlp = ScalarMLPFunction(8, 192, 1, 128).to(device)
    emb = (torch.randn(1000, 8, device=device) * 0.5)
    g = torch.randn(1000, 192, device=device)
    c = torch.randn(1000, 8, device=device)
    w0, w1 = mlp.mlp[0].weight.detach(), mlp.mlp[2].weight.detach()
    a0, a1 = float(mlp.mlp[0].alpha), float(mlp.mlp[2].alpha)

    def run(fwd):
        e = emb.clone().requires_grad_(True)
        gg = g.clone().requires_grad_(True)
        out = fwd(e)
        (ge,) = torch.autograd.grad(out, e, gg, create_graph=True)
        s_e, s_g = torch.autograd.grad(ge, [e, gg], c)
        return out.detach(), ge.detach(), s_e, s_g

    mlp.train()  # the twice-differentiable Function pair
    ref = run(lambda e: mlp(e))
    got = run(lambda e: radial_mlp(e, w0, w1, a0, a1))
    for r, o, what in zip(ref, got, ("value", "grad emb", "second order emb", "second order g")):
        torch.testing.assert_close(o, r, rtol=2e-5, atol=2e-5 * max(1.0, float(r.abs().max())), msg=lambda m: f"{what}: {m}")


@pytest.mark.gpu
def test_traced_benchmark_shaped_model_keeps_the_fused_kernels(device):
    """cfg-3-shaped model (64 features, radial MLP 8-128-W) on a small box: traced on the GPU, every fused kernel family
    appears as a dispatcher op in the graph, and the graph reproduces the eager model (energy, forces, virial)."""
    from nequip_amd.utils.tracing import trace_model

    pos, types, cell, names = syn.water_box(n_side=3, seed=2)
    data = syn.make_data(pos, types, 4.5, cell)
    model = NequIPGNNModel(seed=1, model_dtype="float32", r_max=4.5, type_names=names, num_layers=3, l_max=2,
                           parity=False, num_features=64, radial_mlp_depth=1, radial_mlp_width=128,
                           avg_num_neighbors=float(data["edge_index"].shape[1] / data["pos"].shape[0]),
                           per_type_energy_scales=1.0, per_type_energy_shifts=0.0).to(device).eval()
    data = AtomicDataDict.to_device(data, device)
    inputs = {k: data[k] for k in FIELDS}
    ref = model(dict(inputs))
    gm, params, buffers = trace_model(model, inputs, tracing_mode="real")
    ours = {str(n.target) for n in gm.graph.nodes if n.op == "call_function" and str(n.target).startswith("nequip_amd.")}
    for op in ("radial_tp_fwd", "radial_tp_bwd", "edge_embed_fwd", "edge_embed_bwd", "node_linear", "node_stage_fwd",
               "node_stage_bwd", "energy_head_fwd", "energy_head_bwd", "edge_vectors", "force_virial"):
        assert f"nequip_amd.{op}.default" in ours, (op, ours)
    assert not any(f"nequip_amd.{op}.default" in ours for op in ("gate", "radial_mlp_fwd", "tp_scatter_fwd")), ours
    out = gm(params, buffers, inputs)
    for k in ("total_energy", "forces", "virial", "stress"):
        torch.testing.assert_close(out[k], ref[k], rtol=1e-5, atol=3e-5 * max(1.0, float(ref[k].abs().max())))


@pytest.mark.gpu
def test_folded_graph_reproduces_the_eager_model_and_keeps_its_weight_images(device):
    """`trace_model(fold=True)`: the weight-only part of the graph is evaluated once; what is left reproduces the eager model
    on the box it was traced on and on another one (dynamic sizes), and the dispatcher ops find their packed / transposed /
    split weight images again at the second call (utils/constcache.py: no new entries)."""
    from nequip_amd.utils import constcache
    from nequip_amd.utils.tracing import trace_model

    pos, types, cell, names = syn.water_box(n_side=3, seed=2)
    data = syn.make_data(pos, types, 4.5, cell)
    model = NequIPGNNModel(seed=1, model_dtype="float32", r_max=4.5, type_names=names, num_layers=3, l_max=2,
                           parity=False, num_features=64, radial_mlp_depth=1, radial_mlp_width=128,
                           avg_num_neighbors=float(data["edge_index"].shape[1] / data["pos"].shape[0]),
                           per_type_energy_scales=1.0, per_type_energy_shifts=0.0).to(device).eval()
    data = AtomicDataDict.to_device(data, device)
    inputs = {k: data[k] for k in FIELDS}
    gm, params, buffers = trace_model(model, inputs, tracing_mode="symbolic", fold=True)
    folded = [n for n, _ in gm.named_buffers() if n.startswith("_folded_")]
    assert len(folded) >= 8, folded
    aten_math = [str(n.target) for n in gm.graph.nodes if n.op == "call_function"
                 and str(n.target) in ("aten.bmm.default", "aten.mm.default", "aten.embedding.default")]
    assert aten_math == ["aten.embedding.default"], aten_math  # (the type embedding depends on the input, nothing else is left)
    pos2, types2, cell2, _ = syn.water_box(n_side=4, seed=7)
    d2 = AtomicDataDict.to_device(syn.make_data(pos2, types2, 4.5, cell2), device)
    for k, inp in enumerate((inputs, {f: d2[f] for f in FIELDS}, inputs)):
        ref = model(dict(inp))
        if k == 1:
            held = len(constcache._entries)
        out = gm(params, buffers, inp)
        for f in ("total_energy", "forces", "virial", "stress"):
            torch.testing.assert_close(out[f], ref[f], rtol=1e-5, atol=3e-5 * max(1.0, float(ref[f].abs().max())))
    assert len(constcache._entries) == held, "the ops rebuilt a weight image for a constant they had seen"


@pytest.mark.gpu
def test_traced_graph_on_two_frames_and_on_a_list_that_does_not_pair_up(device):
    """The fused inference ops away from their comfortable case, against the eager model AND the CPU oracle:
    (a) a batch of two periodic frames (per-frame cells; `force_virial` sums virials per frame),
    (b) an edge list that does not pair up -- one directed edge removed, so `radial_tp_*` take the per-edge kernels inside the
        same ops (the list is still what the model is asked to evaluate: the oracle gets the same list)."""
    from nequip_amd.utils.tracing import trace_model
    from oracle import model as omodel

    cfg = dict(r_max=4.0, num_layers=2, l_max=2, parity=False, num_features=16, radial_mlp_depth=1, radial_mlp_width=64,
               num_bessels=8, polynomial_cutoff_p=6, avg_num_neighbors=18.0, model_dtype="float32")
    frames = []
    for seed in (11, 12):
        pos, types, cell, names = syn.water_box(n_side=2, seed=seed)
        frames.append(syn.make_data(pos, types, 4.0, cell))
    batch_cpu = AtomicDataDict.batched_from_list(frames)
    model = NequIPGNNModel(seed=2, type_names=names, per_type_energy_scales=1.0, per_type_energy_shifts=0.0,
                           **{k: v for k, v in cfg.items()}).to(device).eval()
    weights = {k.replace("model.func.", ""): v.detach().cpu() for k, v in model.state_dict().items()}
    fields = FIELDS + ("batch", "num_atoms")
    single = dict(frames[0])
    keep = torch.ones(single["edge_index"].shape[1], dtype=torch.bool)
    keep[5] = False
    single["edge_index"] = single["edge_index"][:, keep].contiguous()
    single["edge_cell_shift"] = single["edge_cell_shift"][keep].contiguous()
    for label, cpu, keys in (("two frames", batch_cpu, fields), ("unpaired list", single, FIELDS)):
        data = AtomicDataDict.to_device(cpu, device)
        inputs = {k: data[k] for k in keys if k in data}
        ref = model(dict(inputs))
        gm, params, buffers = trace_model(model, inputs, tracing_mode="symbolic", fold=True)
        ours = {str(n.target) for n in gm.graph.nodes if n.op == "call_function" and str(n.target).startswith("nequip_amd.")}
        assert {"nequip_amd.radial_tp_fwd.default", "nequip_amd.radial_tp_bwd.default",
                "nequip_amd.force_virial.default"} <= ours, (label, ours)
        out = gm(params, buffers, inputs)
        orc = omodel.energy_forces(cpu, cfg, weights, with_virial=True)
        n = cpu["pos"].shape[0]
        fs = max(1.0, float(orc["forces"].abs().max()))
        for k in ("total_energy", "forces", "virial"):
            torch.testing.assert_close(out[k], ref[k], rtol=1e-5, atol=3e-5 * max(1.0, float(ref[k].abs().max())),
                                       msg=lambda m: f"{label}: {k} vs eager: {m}")
        df = float((orc["forces"] - out["forces"].cpu()).abs().max())
        assert df < 1e-4, f"{label}: forces differ from the oracle by {df:.3e} eV/A"
        torch.testing.assert_close(out["total_energy"].cpu(), orc["total_energy"].view_as(out["total_energy"].cpu()),
                                   atol=5e-5 * n, rtol=5e-5)
        torch.testing.assert_close(out["virial"].cpu().view(-1, 3, 3), orc["virial"].view(-1, 3, 3), atol=5e-5 * n * fs,
                                   rtol=5e-4)


@pytest.mark.gpu
def test_traced_graph_with_the_round_3_op_set(device, monkeypatch):
    """The separate-op forms stay selectable (and are what a graph that differentiates twice, or a deep radial MLP, gets):
    per-edge radial MLP + tensor-product ops, gate ops, the reference's strain formulation of the virial."""
    from nequip_amd.utils.tracing import trace_model

    monkeypatch.setenv("NQA_NO_RADIAL_TP_OP", "1")
    monkeypatch.setenv("NQA_TRACE_NO_NODE_FUSION", "1")
    monkeypatch.setenv("NQA_TRACE_REFERENCE_TAIL", "1")
    pos, types, cell, names = syn.water_box(n_side=3, seed=2)
    data = syn.make_data(pos, types, 4.5, cell)
    model = NequIPGNNModel(seed=1, model_dtype="float32", r_max=4.5, type_names=names, num_layers=3, l_max=2,
                           parity=False, num_features=64, radial_mlp_depth=1, radial_mlp_width=128,
                           avg_num_neighbors=float(data["edge_index"].shape[1] / data["pos"].shape[0]),
                           per_type_energy_scales=1.0, per_type_energy_shifts=0.0).to(device).eval()
    data = AtomicDataDict.to_device(data, device)
    inputs = {k: data[k] for k in FIELDS}
    ref = model(dict(inputs))
    gm, params, buffers = trace_model(model, inputs, tracing_mode="real")
    ours = {str(n.target) for n in gm.graph.nodes if n.op == "call_function" and str(n.target).startswith("nequip_amd.")}
    for op in ("tp_scatter_fwd", "tp_scatter_bwd", "edge_embed_fwd", "edge_embed_bwd", "radial_mlp_fwd", "radial_mlp_bwd",
               "node_linear", "gate", "gate_bwd", "edge_vectors", "edge_vectors_adj"):
        assert f"nequip_amd.{op}.default" in ours, (op, ours)
    out = gm(params, buffers, inputs)
    for k in ("total_energy", "forces", "virial"):
        torch.testing.assert_close(out[k], ref[k], rtol=1e-5, atol=3e-5 * max(1.0, float(ref[k].abs().max())))


def test_interaction_block_compiles_under_dynamo_on_cpu():
    """`torch.compile(dynamic=True)` (the reference's train-time compile path, nequip/nn/compile.py:176-191) must be able to
    trace InteractionBlock.forward -- including the self-connection branch, which runs inside a side-stream context in
    eager mode and must not enter `torch.cuda.stream(None)` while Dynamo traces.  The backend below never executes the graph
    (the kernels have no CPU form): it returns shape-faithful zeros from the fake-tensor metadata."""
    from nequip_amd.nn.interaction_block import InteractionBlock

    model, inputs = _model_and_data(torch.device("cpu"))
    block = [m for m in model.modules() if isinstance(m, InteractionBlock) and m.sc is not None][0]
    n, e = inputs["pos"].shape[0], inputs["edge_index"].shape[1]
    irr = block.irreps_in
    g = torch.Generator().manual_seed(0)
    data = {
        "node_features": torch.randn(n, irr["node_features"].dim, generator=g),
        "node_attrs": torch.randn(n, irr["node_attrs"].dim, generator=g),
        "edge_attrs": torch.randn(e, irr["edge_attrs"].dim, generator=g),
        "edge_embedding": torch.randn(e, irr["edge_embedding"].dim, generator=g),
        "edge_index": inputs["edge_index"], "atom_types": inputs["atom_types"], "pos": inputs["pos"],
    }
    graphs = []

    def backend(gm, example_inputs):
        graphs.append(gm)
        outs = [nd for nd in gm.graph.nodes if nd.op == "output"][0].args[0]

        def run(*args):
            return tuple(torch.zeros([int(s) for s in o.meta["example_value"].shape], dtype=o.meta["example_value"].dtype)
                         for o in outs)

        return run

    def f(d):
        return block(dict(d))["node_features"]

    out = torch.compile(f, backend=backend, dynamic=True, fullgraph=True)(data)
    assert out.shape == (n, block.irreps_out["node_features"].dim)
    targets = {str(nd.target) for gm in graphs for nd in gm.graph.nodes if nd.op == "call_function"}
    assert any("tp_scatter_fwd" in t for t in targets), targets
