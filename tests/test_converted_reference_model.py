"""The model of tests/golden/converted_reference_model.pt -- built by the REFERENCE's `NequIPGNNModel` builder, converted by
the REFERENCE's `modify` with `enable_NequipAMD_full` (tests/golden/make_converted_model_fixture.py; the build container has
the reference, the GPU box does not) -- evaluated on the HIP path and compared with the oracle on the same parameters.
Bars as tests/test_model_parity.py: energy / forces 5e-5 abs / rel, forces <= 1e-4 eV/A."""

import os

import pytest
import torch

from oracle import model as omodel

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "converted_reference_model.pt")


def _load():
    blob = torch.load(FIXTURE, map_location="cpu", weights_only=False)
    return blob["model"], blob["hyper"]


def test_fixture_holds_the_modifier_output():
    model, hyper = _load()
    chain = model.model.func
    assert [n for n, _ in chain.named_children()][:5] == ["type_embed", "spharm", "edge_norm", "bessel_encode", "factor"]
    assert all(type(m).__module__.startswith("nequip_amd.") for m in chain.children())
    assert chain.layer0_convnet.defer_gate and chain.layer2_convnet.defer_gate and chain.factor._folded
    assert "_scale_shift" in chain.per_atom_energy_readout.__dict__
    assert hyper["num_layers"] == 3 and hyper["l_max"] == 2


@pytest.mark.gpu
def test_converted_reference_model_matches_the_oracle(device):
    from nequip_amd.data import AtomicDataDict
    from nequip_amd.utils import synthetic as syn

    model, hyper = _load()
    model = model.to(device).eval()
    pos, types, cell, names = syn.water_box(n_side=4, seed=5)
    assert names == hyper["type_names"]
    data = syn.make_data(pos, types, hyper["r_max"], cell)
    out = model(AtomicDataDict.to_device(data, device))
    cfg = dict(r_max=hyper["r_max"], num_layers=hyper["num_layers"], l_max=hyper["l_max"], parity=hyper["parity"],
               num_features=hyper["num_features"], radial_mlp_depth=1, radial_mlp_width=hyper["radial_mlp_width"],
               num_bessels=8, polynomial_cutoff_p=6, avg_num_neighbors=hyper["avg_num_neighbors"], model_dtype="float32",
               scales=hyper["per_type_energy_scales"], shifts=hyper["per_type_energy_shifts"])
    weights = {k.replace("model.func.", ""): v.detach().cpu() for k, v in model.state_dict().items()}
    ref = omodel.energy_forces(data, cfg, weights, with_virial=True)
    fscale = float(ref["forces"].abs().max())
    torch.testing.assert_close(ref["total_energy"], out["total_energy"].cpu(), atol=5e-5 * len(pos), rtol=5e-5)
    torch.testing.assert_close(ref["forces"], out["forces"].cpu(), atol=5e-5 * max(1.0, fscale), rtol=5e-5)
    assert float((ref["forces"] - out["forces"].cpu()).abs().max()) < 1e-4 * max(1.0, fscale)
