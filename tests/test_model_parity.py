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


@pytest.mark.gpu
def test_per_edge_type_cutoffs_and_trained_bessel_roots(device):
    """``per_edge_type_cutoff`` (asymmetric: reverse-edge pairing off) and perturbed Bessel roots as a trained
    ``bessel_trainable`` model has them (nequip/nn/embedding/_edge.py:27-52,117-120), eval mode, against the oracle."""
    from nequip_amd.data import AtomicDataDict
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.nn.embedding import cutoff_partialdict_to_tensor
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=4, seed=6)
    data = syn.make_data(pos, types, 4.5, cell)
    pt = {"H": 3.2, "O": {"H": 4.0, "O": 4.5}}
    cfg = _cfg(num_features=16, radial_mlp_width=64, avg_num_neighbors=38.0)
    model = NequIPGNNModel(seed=2, model_dtype="float32", r_max=4.5, type_names=names, num_layers=3, l_max=2, parity=False,
                           num_features=16, radial_mlp_depth=1, radial_mlp_width=64, avg_num_neighbors=38.0,
                           per_edge_type_cutoff=pt, bessel_trainable=True)
    with torch.no_grad():
        model.model.func.bessel_encode.bessel_weights.mul_(1.0 + 0.02 * torch.randn(1, 8, dtype=torch.float64))
    model = model.to(device).eval()
    out = model(AtomicDataDict.to_device(data, device))
    cfg["per_edge_type_cutoff_table"] = cutoff_partialdict_to_tensor(pt, list(names), 4.5)
    ref = omodel.energy_forces(data, cfg, _weights(model), with_virial=True)
    fscale = float(ref["forces"].abs().max())
    torch.testing.assert_close(ref["total_energy"], out["total_energy"].cpu(), atol=5e-5 * len(pos), rtol=5e-5)
    torch.testing.assert_close(ref["forces"], out["forces"].cpu(), atol=5e-5 * max(1.0, fscale), rtol=5e-5)
    assert float((ref["forces"] - out["forces"].cpu()).abs().max()) < 1e-4 * max(1.0, fscale)


# ---- the BASELINE.json model shapes themselves (the kernels bench.py times: 64 / 128 features, radial MLP 8-128-W) ----
def _baseline_case(device, data, names, cfg, what):
    """Energy within 5e-5 per atom abs/rel, forces within 1e-4 eV/A *absolute* (BASELINE.json north_star) and 5e-5
    relative to the largest force, virial within 5e-5 x atoms."""
    out, ref = _run_both(cfg, data, names, device)
    n = data["pos"].shape[0]
    f_ref, f_out = ref["forces"], out["forces"].cpu()
    fscale = float(f_ref.abs().max())
    df = float((f_ref - f_out).abs().max())
    de = float((ref["total_energy"] - out["total_energy"].cpu()).abs().max())
    print(f"[{what}] N={n} E={data['edge_index'].shape[1]} |dE|={de:.3e} max|dF|={df:.3e} eV/A (max|F|={fscale:.3e})")
    torch.testing.assert_close(ref["total_energy"], out["total_energy"].cpu(), atol=5e-5 * n, rtol=5e-5)
    assert df < 1e-4, f"forces differ from the oracle by {df:.3e} eV/A (bar 1e-4)"
    torch.testing.assert_close(f_ref, f_out, atol=5e-5 * max(1.0, fscale), rtol=5e-5)
    torch.testing.assert_close(ref["virial"], out["virial"].cpu(), atol=5e-5 * n * max(1.0, fscale), rtol=5e-4)


@pytest.mark.gpu
def test_cfg2_model_si_1000_atoms(device):
    """BASELINE config 2 at full size: 1000-atom periodic Si box, l_max=2, 64 features, r_cut 4.5 A."""
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.silicon_box(reps=5, seed=1)
    data = syn.make_data(pos, types, 4.5, cell)
    assert len(pos) == 1000
    n, e = len(pos), data["edge_index"].shape[1]
    cfg = _cfg(num_features=64, radial_mlp_width=128, avg_num_neighbors=e / n)
    _baseline_case(device, data, names, cfg, "cfg-2 si1k")


@pytest.mark.gpu
def test_cfg3_model_water_1029_atoms(device):
    """BASELINE config 3's model (l_max=2, 64 features, 3 layers, two species) on a 7^3-molecule water box (1029 atoms,
    same density / r_max as the 10 125-atom bench box, which would cost the oracle minutes)."""
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=7, seed=1)
    data = syn.make_data(pos, types, 4.5, cell)
    n, e = len(pos), data["edge_index"].shape[1]
    assert n >= 1000
    cfg = _cfg(num_features=64, radial_mlp_width=128, avg_num_neighbors=e / n)
    _baseline_case(device, data, names, cfg, "cfg-3 water1k")


@pytest.mark.gpu
def test_cfg5_model_cu_108_atoms(device):
    """BASELINE config 5's model (l_max=3, 128 features: two 64-channel chunks per node, 23-path middle layer) on a
    108-atom fcc Cu box."""
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.copper_box(reps=(3, 3, 3), seed=1)
    data = syn.make_data(pos, types, 4.5, cell)
    n, e = len(pos), data["edge_index"].shape[1]
    assert n == 108
    cfg = _cfg(l_max=3, num_features=128, radial_mlp_width=128, avg_num_neighbors=e / n)
    _baseline_case(device, data, names, cfg, "cfg-5 cu108")
