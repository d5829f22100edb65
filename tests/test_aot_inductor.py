"""AOTInductor packaging of the HIP-backed model (the reference: `nequip-compile --mode aotinductor`,
nequip/scripts/compile.py:248-344, loaded by nequip/model/inference_models/aotinductor.py:57-125).

The whole energy + forces + virial evaluation is traced (make_fx, symbolic edge / atom counts), exported and compiled into
a `.nequip.pt2` package whose kernels are the `torch.ops.nequip_amd.*` dispatcher ops; the package carries the
`nequip_custom_ops_libs` entry, is loaded through the same steps as the reference's loader, and must reproduce the eager
fused model on the box it was compiled for AND on a box of another size (dynamic shapes)."""
import zipfile

import pytest
import torch


def test_custom_ops_entry_roundtrip(tmp_path):
    """The zip entry written next to the compiled artefact is what the loader imports (CPU, no compiler involved)."""
    from nequip_amd.utils import aot

    p = tmp_path / "m.nequip.pt2"
    with zipfile.ZipFile(p, "w") as zf:
        zf.writestr("data/placeholder", "x")
    aot.embed_custom_ops_libs(str(p), ["nequip_amd", "json"])
    with zipfile.ZipFile(p) as zf:
        assert zf.read("nequip_custom_ops_libs.txt").decode().split() == ["json", "nequip_amd"]
    aot.import_custom_ops_libs(str(p))  # imports nequip_amd: the ops exist afterwards
    assert hasattr(torch.ops.nequip_amd, "tp_scatter_fwd") and hasattr(torch.ops.nequip_amd, "node_linear")


def test_modifier_declares_aotinductor():
    from nequip_amd.integrations import nequip_extension as ext

    class Stub:
        pass

    ext.register(Stub)
    mod = getattr(Stub, ext.MODIFIER_NAME)
    fn = getattr(mod, "__func__", mod)
    modes = getattr(fn, "_nequip_model_modifier_supported_compile_modes")  # (nequip/nn/model_modifier_utils.py:30-36)
    assert "aotinductor" in list(modes)


def test_folded_graph_exports_with_dynamic_sizes_on_cpu():
    """The first half of `aot_export_model` without a GPU and without the compiler: trace (symbolic sizes) -> fold the
    weight-only part -> wrap -> `torch.export` with dynamic atom / edge counts.  The folded constants travel as buffers of the
    exported program, and the exported module's signature is the positional ASE input list."""
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.utils import aot
    from nequip_amd.utils import synthetic as syn
    from nequip_amd.utils.tracing import trace_model

    pos, types, cell, names = syn.water_box(n_side=2, seed=5)
    data = syn.make_data(pos, types, 4.0, cell)
    model = NequIPGNNModel(seed=3, model_dtype="float32", r_max=4.0, type_names=names, num_layers=2, l_max=1, parity=False,
                           num_features=8, radial_mlp_depth=1, radial_mlp_width=16, avg_num_neighbors=20.0,
                           per_type_energy_scales=1.0, per_type_energy_shifts=0.0).eval()
    fields = aot.ASE_INPUTS
    inputs = {k: data[k] for k in fields}
    gm, params, buffers = trace_model(model, inputs, tracing_mode="symbolic", fold=True)
    for nd in list(gm.graph.nodes):
        if nd.op == "get_attr" and len(nd.users) == 0:
            gm.graph.erase_node(nd)
    gm.graph.eliminate_dead_code()
    gm.recompile()
    wrapped = aot._ListIO(gm, params, buffers, fields, ["total_energy", "forces", "virial"])
    bm = {"graph": torch.export.Dim.STATIC, "node": torch.export.Dim("num_nodes", min=2, max=1 << 26),
          "edge": torch.export.Dim("num_edges", min=2, max=1 << 30)}
    ep = torch.export.export(wrapped, tuple(inputs[k] for k in fields),
                             dynamic_shapes=(tuple(aot._field_dims(k, bm) for k in fields),), strict=False)
    folded = [k for k in ep.state_dict if "_folded_" in k]
    assert len(folded) >= 10, folded
    placeholders = [n for n in ep.graph.nodes if n.op == "placeholder"]
    user_inputs = [s for s in ep.graph_signature.user_inputs]
    assert len(user_inputs) == len(fields) and len(placeholders) > len(fields)
    # the edge count is a symbol of the exported graph, not the example's number
    ei = [n for n in placeholders if n.name == user_inputs[fields.index("edge_index")]][0]
    assert not isinstance(ei.meta["val"].shape[1], int)


@pytest.mark.gpu
def test_aotinductor_package_reproduces_eager_model(device, tmp_path):
    from nequip_amd.data import AtomicDataDict
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.utils import aot
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=3, seed=5)
    data = syn.make_data(pos, types, 4.5, cell)
    model = NequIPGNNModel(seed=3, model_dtype="float32", r_max=4.5, type_names=names, num_layers=3, l_max=2,
                           parity=False, num_features=64, radial_mlp_depth=1, radial_mlp_width=128,
                           avg_num_neighbors=float(data["edge_index"].shape[1] / data["pos"].shape[0]),
                           per_type_energy_scales=1.0, per_type_energy_shifts=0.0).to(device).eval()
    model_avg = float(data["edge_index"].shape[1] / data["pos"].shape[0])
    data = AtomicDataDict.to_device(data, device)
    assert model.metadata["nequip_custom_ops_libs"] == "nequip_amd" and model.metadata["r_max"] == "4.5"
    ref = model(dict(data))
    path = aot.aot_export_model(model, data, str(tmp_path / "water.nequip.pt2"))
    with zipfile.ZipFile(path) as zf:
        assert zf.read("nequip_custom_ops_libs.txt").decode().split() == ["nequip_amd"]
    compiled, md = aot.load_aotinductor_model(path, device="cuda")
    assert md["nequip_aoti_inputs"].split() == aot.ASE_INPUTS and md["type_names"] == " ".join(names)
    out = compiled(dict(data))
    n = data["pos"].shape[0]
    fscale = max(1.0, float(ref["forces"].abs().max()))
    torch.testing.assert_close(out["total_energy"], ref["total_energy"], atol=2e-5 * n, rtol=2e-5)
    torch.testing.assert_close(out["forces"], ref["forces"], atol=2e-5 * fscale, rtol=2e-5)
    torch.testing.assert_close(out["virial"], ref["virial"], atol=2e-5 * n * fscale, rtol=2e-4)
    # dynamic shapes: another box (different atom and edge counts) through the same package
    pos, types, cell, _ = syn.water_box(n_side=4, seed=9)
    d2 = AtomicDataDict.to_device(syn.make_data(pos, types, 4.5, cell), device)
    ref2, out2 = model(dict(d2)), compiled(dict(d2))
    fscale = max(1.0, float(ref2["forces"].abs().max()))
    torch.testing.assert_close(out2["total_energy"], ref2["total_energy"], atol=2e-5 * len(pos), rtol=2e-5)
    torch.testing.assert_close(out2["forces"], ref2["forces"], atol=2e-5 * fscale, rtol=2e-5)
    # parity proper: the compiled package against the CPU oracle (not only against the eager HIP model), on the box it
    # was NOT compiled for -- bars of tests/test_baseline_size_parity.py
    from oracle import model as omodel

    cfg = dict(r_max=4.5, num_layers=3, l_max=2, parity=False, num_features=64, radial_mlp_depth=1,
               radial_mlp_width=128, num_bessels=8, polynomial_cutoff_p=6, avg_num_neighbors=model_avg,
               model_dtype="float32")
    weights = {k.replace("model.func.", ""): v.detach().cpu() for k, v in model.state_dict().items()}
    d2_cpu = syn.make_data(pos, types, 4.5, cell)
    orc = omodel.energy_forces(d2_cpu, cfg, weights, with_virial=True)
    df = float((orc["forces"] - out2["forces"].cpu()).abs().max())
    assert df < 1e-4, f"compiled package: forces differ from the oracle by {df:.3e} eV/A"
    fs = max(1.0, float(orc["forces"].abs().max()))
    torch.testing.assert_close(out2["total_energy"].cpu(), orc["total_energy"], atol=5e-5 * len(pos), rtol=5e-5)
    torch.testing.assert_close(out2["forces"].cpu(), orc["forces"], atol=5e-5 * fs, rtol=5e-5)
    torch.testing.assert_close(out2["virial"].cpu().view(-1, 3, 3), orc["virial"], atol=5e-5 * len(pos) * fs, rtol=5e-4)
