"""ASE-style calculator (nequip_amd/integrations/ase.py, mirror of nequip/integrations/ase.py:13-160): same results keys
and units as the reference's calculator, graph built on the device.  ASE is not installed here: a stand-in object with
the `Atoms` accessors the calculator uses plays its role (with ASE installed the very same calls hit a real Atoms)."""
import numpy as np
import pytest
import torch

from nequip_amd.integrations.ase import NequIPCalculator, full_3x3_to_voigt_6_stress


class FakeAtoms:
    def __init__(self, symbols, positions, cell=None, pbc=False):
        self._s, self._p = list(symbols), np.asarray(positions, dtype=np.float64)
        self._c = np.zeros((3, 3)) if cell is None else np.asarray(cell, dtype=np.float64)
        self._pbc = np.array([pbc] * 3 if isinstance(pbc, bool) else pbc, dtype=bool)

    def get_chemical_symbols(self):
        return self._s

    def get_positions(self):
        return self._p

    def get_cell(self):
        return self._c

    def get_pbc(self):
        return self._pbc

    def __len__(self):
        return len(self._s)


def test_voigt_order_and_symmetrisation():
    s = np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0], [7.0, 8.0, 9.0]])
    np.testing.assert_allclose(full_3x3_to_voigt_6_stress(s), [1.0, 5.0, 9.0, 7.0, 5.0, 3.0])


def test_calculator_is_gpu_only_and_wants_eval_mode():
    model = torch.nn.Linear(1, 1)
    model.type_names = ["H"]
    with pytest.raises(AssertionError):
        NequIPCalculator(model.train(), "cuda", r_max=4.0)
    with pytest.raises(RuntimeError):
        NequIPCalculator(model.eval(), "cpu", r_max=4.0)


@pytest.mark.gpu
@pytest.mark.parametrize("periodic", [True, False])
def test_calculator_matches_direct_model_call(device, periodic):
    from nequip_amd.data import AtomicDataDict
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=3, seed=3)
    model = NequIPGNNModel(seed=1, model_dtype="float32", r_max=4.0, type_names=names, num_layers=2, l_max=2,
                           parity=False, num_features=16, radial_mlp_depth=1, radial_mlp_width=64,
                           avg_num_neighbors=20.0).to(device).eval()
    data = syn.make_data(pos, types, 4.0, cell if periodic else None, pbc=periodic)
    ref = model(AtomicDataDict.to_device(data, device))

    atoms = FakeAtoms([names[t] for t in types], pos, cell if periodic else None, pbc=periodic)
    calc = NequIPCalculator(model, device, r_max=4.0, energy_units_to_eV=2.0, length_units_to_A=0.5)
    e = calc.get_potential_energy(atoms)
    f = calc.get_forces(atoms)
    assert set(calc.results) >= {"energy", "free_energy", "energies", "forces"}
    np.testing.assert_allclose(e, 2.0 * float(ref["total_energy"]), rtol=2e-5, atol=2e-5 * len(pos))
    fr = ref["forces"].cpu().numpy() * (2.0 / 0.5)
    np.testing.assert_allclose(f, fr, rtol=0, atol=2e-4 * max(1.0, np.abs(fr).max()))
    np.testing.assert_allclose(calc.results["energies"].sum(), e, rtol=1e-5, atol=1e-4)
    if periodic:
        s = calc.get_stress(atoms)
        sr = full_3x3_to_voigt_6_stress(ref["stress"].cpu().numpy().reshape(3, 3)) * (2.0 / 0.5**3)
        np.testing.assert_allclose(s, sr, rtol=0, atol=2e-4 * max(1e-3, np.abs(sr).max()))
    else:
        assert "stress" not in calc.results
    with pytest.raises(ValueError):
        calc.calculate(FakeAtoms(["Xx"] * 3, pos[:3]))
    # parity proper: calculator results (device neighbour list, unit conversion) against the CPU oracle on the host-built
    # neighbour list of the same atoms
    from oracle import model as omodel

    cfg = dict(r_max=4.0, num_layers=2, l_max=2, parity=False, num_features=16, radial_mlp_depth=1, radial_mlp_width=64,
               num_bessels=8, polynomial_cutoff_p=6, avg_num_neighbors=20.0, model_dtype="float32")
    weights = {k.replace("model.func.", ""): v.detach().cpu() for k, v in model.state_dict().items()}
    orc = omodel.energy_forces(data, cfg, weights, with_virial=periodic)
    np.testing.assert_allclose(e, 2.0 * float(orc["total_energy"]), rtol=5e-5, atol=5e-5 * len(pos))
    fo = orc["forces"].numpy() * (2.0 / 0.5)
    assert np.abs(f - fo).max() < 1e-4 * (2.0 / 0.5) * max(1.0, np.abs(orc["forces"].numpy()).max())
    if periodic:
        vol = abs(np.linalg.det(np.asarray(cell, dtype=np.float64)))
        so = full_3x3_to_voigt_6_stress((-orc["virial"].numpy().reshape(3, 3) / vol)) * (2.0 / 0.5**3)
        np.testing.assert_allclose(s, so, rtol=0, atol=5e-4 * max(1e-3, np.abs(so).max()))


@pytest.mark.gpu
def test_calculator_from_compiled_model(device, tmp_path):
    """`NequIPCalculator.from_compiled_model` on an AOTInductor package (the route the reference recommends,
    nequip/integrations/ase.py:16-19): same energy / forces / stress as the calculator around the eager model, on a box
    of another size than the one the package was compiled with."""
    from nequip_amd.data import AtomicDataDict
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.utils import aot
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=2, seed=3)
    model = NequIPGNNModel(seed=1, model_dtype="float32", r_max=4.0, type_names=names, num_layers=2, l_max=2,
                           parity=False, num_features=16, radial_mlp_depth=1, radial_mlp_width=64,
                           avg_num_neighbors=20.0).to(device).eval()
    example = AtomicDataDict.to_device(syn.make_data(pos, types, 4.0, cell), device)
    path = aot.aot_export_model(model, example, str(tmp_path / "m.nequip.pt2"))
    pos, types, cell, names = syn.water_box(n_side=3, seed=8)
    atoms = FakeAtoms([names[t] for t in types], pos, cell, pbc=True)
    eager = NequIPCalculator(model, device, r_max=4.0)
    comp = NequIPCalculator.from_compiled_model(path, device="cuda")
    assert comp.r_max == 4.0
    e0, f0, s0 = eager.get_potential_energy(atoms), eager.get_forces(atoms), eager.get_stress(atoms)
    e1, f1, s1 = comp.get_potential_energy(atoms), comp.get_forces(atoms), comp.get_stress(atoms)
    np.testing.assert_allclose(e1, e0, rtol=2e-5, atol=2e-5 * len(pos))
    np.testing.assert_allclose(f1, f0, rtol=0, atol=2e-5 * max(1.0, np.abs(f0).max()))
    np.testing.assert_allclose(s1, s0, rtol=0, atol=2e-5 * max(1e-3, np.abs(s0).max()))
    # parity proper: the calculator around the COMPILED package against the CPU oracle
    from oracle import model as omodel

    cfg = dict(r_max=4.0, num_layers=2, l_max=2, parity=False, num_features=16, radial_mlp_depth=1, radial_mlp_width=64,
               num_bessels=8, polynomial_cutoff_p=6, avg_num_neighbors=20.0, model_dtype="float32")
    weights = {k.replace("model.func.", ""): v.detach().cpu() for k, v in model.state_dict().items()}
    orc = omodel.energy_forces(syn.make_data(pos, types, 4.0, cell), cfg, weights, with_virial=True)
    np.testing.assert_allclose(e1, float(orc["total_energy"]), rtol=5e-5, atol=5e-5 * len(pos))
    fo = orc["forces"].numpy()
    assert np.abs(f1 - fo).max() < 1e-4 * max(1.0, np.abs(fo).max())
    vol = abs(np.linalg.det(np.asarray(cell, dtype=np.float64)))
    so = full_3x3_to_voigt_6_stress(-orc["virial"].numpy().reshape(3, 3) / vol)
    np.testing.assert_allclose(s1, so, rtol=0, atol=5e-4 * max(1e-3, np.abs(so).max()))


@pytest.mark.gpu
def test_graphed_md_calculator_follows_a_trajectory(device):
    """``graphed_md=True``: the calls of an MD run replay one hipGraph (capacity-padded neighbour list, deferred pairing
    verdict) and give what the ordinary calculator gives -- along a trajectory, after a change of the cell (written in
    place), for other atoms (captured anew) and for a molecule (no cell to pad with: the ordinary path)."""
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=4, seed=3)
    cell = np.asarray(cell, dtype=np.float64).reshape(3, 3)
    model = NequIPGNNModel(seed=1, model_dtype="float32", r_max=4.0, type_names=names, num_layers=2, l_max=2,
                           parity=False, num_features=16, radial_mlp_depth=1, radial_mlp_width=64,
                           avg_num_neighbors=20.0).to(device).eval()
    plain = NequIPCalculator(model, device, r_max=4.0)
    fast = NequIPCalculator(model, device, r_max=4.0, graphed_md=True)
    symbols = [names[t] for t in types]
    rng = np.random.default_rng(0)

    def compare(atoms):
        e, f, s = fast.get_potential_energy(atoms), fast.get_forces(atoms).copy(), fast.get_stress(atoms).copy()
        e0, f0, s0 = plain.get_potential_energy(atoms), plain.get_forces(atoms), plain.get_stress(atoms)
        np.testing.assert_allclose(e, e0, rtol=1e-6, atol=1e-5 * len(pos))
        np.testing.assert_allclose(f, f0, rtol=0, atol=1e-5 * max(1.0, np.abs(f0).max()))
        np.testing.assert_allclose(s, s0, rtol=0, atol=1e-5 * max(1e-3, np.abs(s0).max()))
        np.testing.assert_allclose(fast.results["energies"], plain.results["energies"], rtol=0, atol=1e-5)

    p = pos.copy()
    for _ in range(4):
        p = p + 0.03 * rng.normal(size=p.shape)
        compare(FakeAtoms(symbols, p, cell, pbc=True))
    step = fast._graphed[1]
    assert step.num_captures == 1 and step.num_eager_fallbacks == 0
    compare(FakeAtoms(symbols, 1.01 * p, 1.01 * cell, pbc=True))  # a new cell: same graph
    assert fast._graphed[1] is step and step.num_captures == 1
    perm = rng.permutation(len(p))
    compare(FakeAtoms([symbols[i] for i in perm], p[perm], cell, pbc=True))  # other atoms: a new graph
    assert fast._graphed[1] is not step
    mol = FakeAtoms(symbols[:9], pos[:9])
    np.testing.assert_allclose(fast.get_forces(mol), plain.get_forces(mol), rtol=0, atol=1e-6)
    with pytest.raises(ValueError):
        NequIPCalculator(model, device, r_max=4.0, graphed_md=True, transforms=[lambda d: d])
