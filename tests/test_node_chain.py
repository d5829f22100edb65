"""nqa_node_chain (one launch per layer boundary and direction) against the module-by-module evaluation it replaces:
linear_2 (+ sc) -> Gate -> {linear_1 * 1/sqrt(avg), sc} forward, and the reverse chain backward.  The per-module kernels
are themselves compared with the oracle in test_node_kernels.py / test_model_parity.py; here the chain must reproduce
them (same fp32 MFMA arithmetic, different tiling: tolerance 2e-6 of the row scale) for the BASELINE shapes and for
ragged ones (odd multiplicities, parity irreps, atoms not a multiple of the group size, one atom type only)."""
import pytest
import torch


def _layers(device, **kw):
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.nn import ConvNetLayer

    cfg = dict(seed=1, model_dtype="float32", r_max=4.5, type_names=["H", "O"], num_layers=3, l_max=2, parity=False,
               num_features=64, radial_mlp_depth=1, radial_mlp_width=128, avg_num_neighbors=38.0)
    cfg.update(kw)
    model = NequIPGNNModel(**cfg).to(device).eval()
    return model, [m for m in model.modules() if isinstance(m, ConvNetLayer)]


CASES = [
    dict(),                                         # cfg-3 model
    dict(l_max=1, num_features=32, parity=True),    # parity irreps, 32 channels (half-filled 64-channel chunks)
    dict(l_max=3, num_features=128, num_layers=3),  # cfg-5 model: d = 7 blocks, two chunks per block
    dict(l_max=2, num_features=24, type_names=["Si"]),  # odd multiplicity, one atom type
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("n_atoms", [16, 37])
def test_chain_matches_module_sequence(device, case, n_atoms):
    from nequip_amd.o3._node_chain import NodeStage, node_stage, stage_supported

    model, layers = _layers(device, **CASES[case])
    ntypes = len(CASES[case].get("type_names", ["H", "O"]))
    g = torch.Generator().manual_seed(5 + case)
    types = torch.randint(0, ntypes, (n_atoms,), generator=g).to(device)
    embed = [m for m in model.modules() if type(m).__name__ == "NodeTypeEmbed"][0]
    table = embed.embed_module.weight.detach()
    for L, (cur, nxt) in enumerate(zip(layers, layers[1:])):
        lin2, gate, lin1, sc = cur.conv.linear_2, cur.equivariant_nonlin, nxt.conv.linear_1, nxt.conv.sc
        assert stage_supported(lin2, gate, lin1, sc)
        scale = float(nxt.conv.avg_num_neighbors_norm.norm_scalar)
        stage = NodeStage(lin2, gate, lin1, sc, scale)
        a = torch.randn(n_atoms, stage.dim_a, generator=g).to(device).requires_grad_(True)
        addend = torch.randn(n_atoms, stage.dim_h, generator=g).to(device).requires_grad_(True) if cur.conv.sc is not None else None
        gy = torch.randn(n_atoms, stage.dim_y, generator=g).to(device)
        gs = torch.randn(n_atoms, stage.dim_s, generator=g).to(device) if sc is not None else None

        # reference: the modules, one launch each
        h = lin2(a, addend=addend)
        xp = gate(h)
        y_ref = lin1(xp, scale=scale)
        s_ref = sc.forward_typed(xp, types, table) if sc is not None else None
        outs, gouts = [y_ref], [gy]
        if sc is not None:
            outs.append(s_ref)
            gouts.append(gs)
        inputs = [a] + ([addend] if addend is not None else [])
        grads_ref = torch.autograd.grad(outs, inputs, gouts)

        a2 = a.detach().clone().requires_grad_(True)
        add2 = addend.detach().clone().requires_grad_(True) if addend is not None else None
        y, s = node_stage(stage, a2, add2, types, table)
        ysc = max(1.0, float(y_ref.abs().max()))
        torch.testing.assert_close(y, y_ref, atol=2e-6 * ysc, rtol=1e-5, msg=lambda m: f"layer {L} y: {m}")
        outs2 = [y]
        if sc is not None:
            torch.testing.assert_close(s, s_ref, atol=2e-6 * max(1.0, float(s_ref.abs().max())), rtol=1e-5,
                                       msg=lambda m: f"layer {L} s: {m}")
            outs2.append(s)
        inputs2 = [a2] + ([add2] if add2 is not None else [])
        grads = torch.autograd.grad(outs2, inputs2, gouts)
        for name, gr, gref in zip(("grad_a", "grad_addend"), grads, grads_ref):
            sc_ = max(1.0, float(gref.abs().max()))
            torch.testing.assert_close(gr, gref, atol=3e-6 * sc_, rtol=1e-5, msg=lambda m: f"layer {L} {name}: {m}")


@pytest.mark.gpu
def test_model_with_fused_chain_matches_module_path(device, monkeypatch):
    """NQA_CHAIN=1 routes the two inner layer boundaries of the model through nqa_node_chain (pending-stage protocol of
    InteractionBlock / ConvNetLayer): same energies, forces and stress as the per-module launches."""
    from nequip_amd.data import AtomicDataDict
    from nequip_amd.o3 import _node_chain
    from nequip_amd.utils import synthetic as syn

    model, _ = _layers(device)
    pos, types, cell, names = syn.water_box(n_side=3, seed=9)
    data = AtomicDataDict.to_device(syn.make_data(pos, types, 4.5, cell), device)
    monkeypatch.delenv("NQA_CHAIN", raising=False)
    ref = model(dict(data))
    calls = []
    orig = _node_chain.NodeStage.forward
    monkeypatch.setattr(_node_chain.NodeStage, "forward", lambda self, *a, **k: (calls.append(1), orig(self, *a, **k))[1])
    monkeypatch.setenv("NQA_CHAIN", "1")
    out = model(dict(data))
    assert len(calls) == 2, "both inner layer boundaries must take the fused launch"
    torch.testing.assert_close(out["total_energy"].detach(), ref["total_energy"].detach(), rtol=1e-6, atol=1e-5)
    f = ref["forces"].detach()
    torch.testing.assert_close(out["forces"].detach(), f, rtol=0, atol=3e-6 * max(1.0, float(f.abs().max())))
    torch.testing.assert_close(out["stress"].detach(), ref["stress"].detach(), rtol=0,
                               atol=3e-6 * max(1e-3, float(ref["stress"].abs().max())))
