"""Energies and forces of the HIP-backed NequIP model against the CPU oracle on identical weights and inputs.
Bars: energies/forces within 5e-5 abs/rel for float32 (nequip/utils/dtype.py:35-42), forces <= 1e-4 eV/A
(BASELINE.json north_star); float64 model 1e-9."""

import pytest
import torch

from oracle import model as omodel


def _weights(model):
    sd = model.state_dict()
    return {k.replace("model.func.", ""): v.detach().cpu() for k, v in sd.items()}


def _cfg(**kw):
    cfg = dict(r_max=4.5, num_layers=3, l_max=2, parity=False, num_features=8, radial_mlp_depth=1,
               radial_mlp_width=16, num_bessels=8, polynomial_cutoff_p=6, avg_num_neighbors=17.0,
               model_dtype="float32")
    cfg.update(kw)
    return cfg


def _build(cfg, type_names, seed=0):
    from nequip_amd.model import NequIPGNNModel

    return NequIPGNNModel(
        seed=seed, model_dtype=cfg["model_dtype"], r_max=cfg["r_max"], type_names=type_names,
        num_layers=cfg["num_layers"], l_max=cfg["l_max"], parity=cfg["parity"], num_features=cfg["num_features"],
        radial_mlp_depth=cfg["radial_mlp_depth"], radial_mlp_width=cfg["radial_mlp_width"],
        num_bessels=cfg["num_bessels"], polynomial_cutoff_p=cfg["polynomial_cutoff_p"],
        avg_num_neighbors=cfg["avg_num_neighbors"], per_type_energy_scales=cfg.get("scales"),
        per_type_energy_shifts=cfg.get("shifts"),
    )


def _run_both(cfg, data, type_names, device):
    from nequip_amd.data import AtomicDataDict

    model = _build(cfg, type_names).to(device).eval()
    out = model(AtomicDataDict.to_device(data, device))
    ref = omodel.energy_forces(data, cfg, _weights(model), with_virial=True)
    return out, ref


@pytest.mark.gpu
@pytest.mark.parametrize("parity,l_max", [(False, 2), (True, 1), (True, 2), (False, 3)])
@pytest.mark.parametrize("model_dtype", ["float32", "float64"])
def test_energy_forces_si(device, parity, l_max, model_dtype):
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.silicon_box(reps=2, seed=1)
    data = syn.make_data(pos, types, 4.5, cell)
    cfg = _cfg(parity=parity, l_max=l_max, model_dtype=model_dtype, avg_num_neighbors=20.0)
    out, ref = _run_both(cfg, data, names, device)
    tol = 5e-5 if model_dtype == "float32" else 1e-9
    fscale = float(ref["forces"].abs().max())
    torch.testing.assert_close(ref["total_energy"], out["total_energy"].cpu(), atol=tol * len(pos), rtol=tol)
    torch.testing.assert_close(ref["forces"], out["forces"].cpu(), atol=tol * max(1.0, fscale), rtol=tol)
    torch.testing.assert_close(ref["virial"], out["virial"].cpu(), atol=tol * len(pos) * max(1.0, fscale), rtol=10 * tol)
    assert float((ref["forces"] - out["forces"].cpu()).abs().max()) < (1e-4 if model_dtype == "float32" else 1e-9) * max(1.0, fscale)


@pytest.mark.gpu
def test_energy_forces_water_two_types(device):
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=4, seed=3)
    data = syn.make_data(pos, types, 4.5, cell)
    cfg = _cfg(num_features=16, radial_mlp_width=32, avg_num_neighbors=38.0, scales={"H": 1.3, "O": 0.7},
               shifts={"H": -1.0, "O": 2.0})
    out, ref = _run_both(cfg, data, names, device)
    fscale = float(ref["forces"].abs().max())
    torch.testing.assert_close(ref["total_energy"], out["total_energy"].cpu(), atol=5e-5 * len(pos), rtol=5e-5)
    torch.testing.assert_close(ref["forces"], out["forces"].cpu(), atol=5e-5 * max(1.0, fscale), rtol=5e-5)


@pytest.mark.gpu
def test_energy_forces_fused_radial_mlp_width(device):
    """radial_mlp_width 64/128 runs the fused MFMA radial kernel in eval mode; 64 features -> 64-lane chunks."""
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=3, seed=5)
    data = syn.make_data(pos, types, 4.5, cell)
    for width, nf in ((64, 32), (128, 64)):
        cfg = _cfg(num_features=nf, radial_mlp_width=width, avg_num_neighbors=38.0)
        out, ref = _run_both(cfg, data, names, device)
        fscale = float(ref["forces"].abs().max())
        torch.testing.assert_close(ref["total_energy"], out["total_energy"].cpu(), atol=5e-5 * len(pos), rtol=5e-5)
        torch.testing.assert_close(ref["forces"], out["forces"].cpu(), atol=5e-5 * max(1.0, fscale), rtol=5e-5)


@pytest.mark.gpu
def test_batched_molecules_no_cell(device):
    """cfg-1 shape: batch of 5 non-periodic 21-atom frames (configs/tutorial.yaml:74), l_max=1, parity=True."""
    from nequip_amd.data import AtomicDataDict
    from nequip_amd.utils import synthetic as syn

    frames = []
    for s in range(5):
        pos, types, _, names = syn.aspirin_like(seed=s)
        frames.append(syn.make_data(pos, types, 5.0, None, pbc=False))
    data = AtomicDataDict.batched_from_list(frames)
    cfg = _cfg(r_max=5.0, num_layers=4, l_max=1, parity=True, num_features=8, radial_mlp_depth=2, radial_mlp_width=16,
               avg_num_neighbors=15.0)
    out, ref = _run_both(cfg, data, names, device)
    fscale = float(ref["forces"].abs().max())
    torch.testing.assert_close(ref["total_energy"], out["total_energy"].cpu(), atol=2e-4, rtol=5e-5)
    torch.testing.assert_close(ref["forces"], out["forces"].cpu(), atol=5e-5 * max(1.0, fscale), rtol=5e-5)
