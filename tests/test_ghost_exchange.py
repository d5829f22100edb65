"""Local / ghost evaluation as a LAMMPS ML-IAP rank sees it (nequip/nn/interaction_block.py:159-199,
nequip/nn/_ghost_exchange_lmp_mliap.py:11-64, nequip/integrations/lammps_mliap/lmp_mliap_wrapper.py:196-245): the
periodic images of a box become ghost atoms with their own indices, every layer after the first keeps the local rows
only and fetches the ghosts' features through ``lmp_data.forward_exchange`` before its tensor product.  Energies, per-atom
energies and pair forces must equal the single-domain evaluation of the same box."""
import copy

import pytest
import torch


class FakeLammpsData:
    """Stand-in for LAMMPS' ML-IAP data object on one rank: ``nlocal`` / ``ntotal`` and the two exchanges on device
    tensors (ghost g is a copy of local atom ``owner[g]``)."""

    def __init__(self, nlocal: int, owner: torch.Tensor):
        self.nlocal = int(nlocal)
        self.ntotal = int(nlocal) + int(owner.numel())
        self.owner = owner
        self.calls = {"forward": 0, "reverse": 0}

    def forward_exchange(self, src, dst, vec_len):
        assert src.shape == dst.shape == (self.ntotal, vec_len)
        dst[: self.nlocal] = src[: self.nlocal]
        dst[self.nlocal:] = src.index_select(0, self.owner)
        self.calls["forward"] += 1

    def reverse_exchange(self, gin, gout, vec_len):
        assert gin.shape == gout.shape == (self.ntotal, vec_len)
        gout.zero_()
        gout[: self.nlocal] = gin[: self.nlocal]
        gout.index_add_(0, self.owner, gin[self.nlocal:])
        self.calls["reverse"] += 1


def _ghost_representation(data):
    """(edge_index with ghost sources, atom types incl. ghosts, owner of every ghost) of a periodic neighbour list."""
    ei = data["edge_index"]
    sh = data["edge_cell_shift"].round().long()
    n = data["pos"].shape[0]
    ghosts = {}
    src_new = ei[1].clone()
    for e in range(ei.shape[1]):
        s = tuple(sh[e].tolist())
        if s != (0, 0, 0):
            key = (int(ei[1, e]), s)
            if key not in ghosts:
                ghosts[key] = n + len(ghosts)
            src_new[e] = ghosts[key]
    owner = torch.tensor([k[0] for k in ghosts], dtype=torch.long)
    types = torch.cat([data["atom_types"].view(-1), data["atom_types"].view(-1)[owner]])
    return torch.stack([ei[0], src_new]), types, owner


def _oracle_of(model, data, num_layers):
    """CPU oracle of the same periodic box: per-atom energies, total energy, forces."""
    from oracle import model as omodel

    cfg = dict(r_max=4.5, num_layers=num_layers, l_max=2, parity=False, num_features=16, radial_mlp_depth=1,
               radial_mlp_width=32, num_bessels=8, polynomial_cutoff_p=6, avg_num_neighbors=38.0, model_dtype="float32")
    weights = {k.replace("model.func.", ""): v.detach().cpu() for k, v in model.state_dict().items()}
    return omodel.energy_forces(data, cfg, weights)


def _edge_forces_to_atoms(edge_forces, edge_index, n):
    """F_i = -dE/dpos_i from dE/d(edge vector): r_e = pos[src] - pos[dst] + shift (nequip/nn/utils.py:88-114; edge_index[0]
    is the centre = dst, edge_index[1] the neighbour = src; periodic images / ghosts fold onto their owner)."""
    g = edge_forces.detach().double().cpu()
    f = torch.zeros(n, 3, dtype=torch.float64)
    f.index_add_(0, edge_index[0], g)
    f.index_add_(0, edge_index[1], -g)
    return f



@pytest.mark.gpu
@pytest.mark.parametrize("num_layers", [2, 3])
def test_lammps_style_local_ghost_evaluation_matches_periodic(device, num_layers):
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.nn import LAMMPSMLIAPGhostExchangeModule, NoOpGhostExchangeModule, with_edge_vectors_
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=3, seed=8)
    data = syn.make_data(pos, types, 4.5, cell)
    n = len(pos)
    model = NequIPGNNModel(seed=3, model_dtype="float32", r_max=4.5, type_names=names, num_layers=num_layers, l_max=2,
                           parity=False, num_features=16, radial_mlp_depth=1, radial_mlp_width=32,
                           avg_num_neighbors=38.0, per_type_energy_scales={"H": 1.2, "O": 0.8},
                           per_type_energy_shifts={"H": -0.5, "O": 1.5}).to(device).eval()
    edge_vec = with_edge_vectors_(dict(data))[K.EDGE_VECTORS_KEY]  # CPU, float64

    # single domain: every neighbour is a local atom (periodic images share the owner's row)
    a = {K.EDGE_VECTORS_KEY: edge_vec.to(device).requires_grad_(True), K.EDGE_INDEX_KEY: data["edge_index"].to(device),
         K.ATOM_TYPE_KEY: data["atom_types"].to(device)}
    out_a = model(a)

    # LAMMPS rank: images are ghost atoms
    ei_l, types_l, owner = _ghost_representation(data)
    assert owner.numel() > 0
    lmp = FakeLammpsData(n, owner.to(device))
    model_l = NoOpGhostExchangeModule.enable_LAMMPSMLIAPGhostExchange(copy.deepcopy(model))
    assert sum(isinstance(m, LAMMPSMLIAPGhostExchangeModule) for m in model_l.modules()) == num_layers
    b = {K.EDGE_VECTORS_KEY: edge_vec.to(device).requires_grad_(True), K.EDGE_INDEX_KEY: ei_l.to(device),
         K.ATOM_TYPE_KEY: types_l.to(device), K.LMP_MLIAP_DATA_KEY: lmp,
         K.NUM_LOCAL_GHOST_NODES_KEY: torch.tensor([n, owner.numel()], device=device)}
    out_b = model_l(b)
    assert lmp.calls == {"forward": num_layers - 1, "reverse": num_layers - 1}  # every layer but the first, both ways

    e_a, e_b = out_a[K.PER_ATOM_ENERGY_KEY].detach(), out_b[K.PER_ATOM_ENERGY_KEY].detach()
    assert e_b.shape[0] == n, "atomic energies of a local/ghost evaluation cover the local atoms"
    torch.testing.assert_close(e_b, e_a, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(out_b[K.TOTAL_ENERGY_KEY].detach(), out_a[K.TOTAL_ENERGY_KEY].detach(), atol=2e-5 * n,
                               rtol=1e-6)
    f_a, f_b = out_a[K.EDGE_FORCE_KEY].detach(), out_b[K.EDGE_FORCE_KEY].detach()
    torch.testing.assert_close(f_b, f_a, atol=2e-5 * max(1.0, float(f_a.abs().max())), rtol=1e-5)
    # parity proper: the local / ghost evaluation against the CPU oracle of the periodic box -- per-atom energies, total
    # energy, and the pair forces folded onto the atoms (ghost -> owner) against the oracle's forces
    orc = _oracle_of(model, data, num_layers)
    torch.testing.assert_close(e_b.double().cpu().view(-1), orc["atomic_energy"].view(-1), atol=5e-5, rtol=5e-5)
    torch.testing.assert_close(out_b[K.TOTAL_ENERGY_KEY].detach().double().cpu().view(-1), orc["total_energy"].view(-1),
                               atol=5e-5 * n, rtol=5e-5)
    f_atoms = _edge_forces_to_atoms(f_b, data["edge_index"], n)
    df = float((f_atoms - orc["forces"]).abs().max())
    assert df < 1e-4 * max(1.0, float(orc["forces"].abs().max())), f"ghost evaluation: forces differ from the oracle by {df:.3e}"


class FakeMLIAPData(FakeLammpsData):
    """The fields ``compute_forces`` reads and writes (``lmp_mliap_wrapper.py:196-257``), on device tensors as a Kokkos
    build hands them over."""

    def __init__(self, nlocal, owner, rij, pair_i, pair_j, elems):
        super().__init__(nlocal, owner)
        self.rij, self.pair_i, self.pair_j, self.elems = rij, pair_i, pair_j, elems
        self.npairs = int(pair_i.numel())
        self.eatoms = torch.zeros(nlocal, dtype=torch.float64, device=rij.device)
        self.energy = None
        self.pair_forces = None

    def update_pair_forces_gpu(self, f):
        self.pair_forces = f.clone()


@pytest.mark.gpu
def test_mliap_wrapper_drives_the_model_like_lammps(device, tmp_path):
    """``NequIPLAMMPSMLIAPWrapper.compute_forces`` on a rank's local + ghost data = the single-domain evaluation: per-atom
    energies written to ``eatoms``, the rank's energy, pair forces dE/d(r_ij) handed to ``update_pair_forces_gpu``; the
    wrapper survives the ``torch.save`` / ``torch.load`` round trip LAMMPS puts it through."""
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.integrations.lammps_mliap import NequIPLAMMPSMLIAPWrapper, create_lmp_mliap_file
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.nn import with_edge_vectors_
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=3, seed=8)
    data = syn.make_data(pos, types, 4.5, cell)
    n = len(pos)
    model = NequIPGNNModel(seed=3, model_dtype="float32", r_max=4.5, type_names=names, num_layers=3, l_max=2,
                           parity=False, num_features=16, radial_mlp_depth=1, radial_mlp_width=32,
                           avg_num_neighbors=38.0, per_type_energy_scales={"H": 1.2, "O": 0.8},
                           per_type_energy_shifts={"H": -0.5, "O": 1.5}).eval()
    edge_vec = with_edge_vectors_(dict(data))[K.EDGE_VECTORS_KEY]
    ref = copy.deepcopy(model).to(device)({K.EDGE_VECTORS_KEY: edge_vec.to(device).requires_grad_(True),
                                           K.EDGE_INDEX_KEY: data["edge_index"].to(device),
                                           K.ATOM_TYPE_KEY: data["atom_types"].to(device)})
    path = create_lmp_mliap_file(model, str(tmp_path / "water.nequip.lmp.pt"), device="cuda")
    wrapper = torch.load(path, weights_only=False)
    assert isinstance(wrapper, NequIPLAMMPSMLIAPWrapper)
    assert wrapper.element_types == names and wrapper.rcutfac == 2.25 and wrapper.nparams == 1
    ei_l, types_l, owner = _ghost_representation(data)
    lmp = FakeMLIAPData(n, owner.to(device), edge_vec.to(device), ei_l[0].to(device), ei_l[1].to(device), types_l.to(device))
    wrapper.compute_forces(lmp)
    assert lmp.calls == {"forward": 2, "reverse": 2}
    torch.testing.assert_close(lmp.eatoms.double(), ref[K.PER_ATOM_ENERGY_KEY].detach().view(-1).double(), atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(lmp.energy.double().view(()), ref[K.TOTAL_ENERGY_KEY].detach().sum().double(), atol=2e-5 * n, rtol=1e-6)
    f_ref = ref[K.EDGE_FORCE_KEY].detach()
    torch.testing.assert_close(lmp.pair_forces.double(), f_ref.double(), atol=2e-5 * max(1.0, float(f_ref.abs().max())), rtol=1e-5)
    # parity proper: what LAMMPS receives against the CPU oracle of the periodic box
    orc = _oracle_of(model, data, 3)
    torch.testing.assert_close(lmp.eatoms.double().cpu(), orc["atomic_energy"].view(-1), atol=5e-5, rtol=5e-5)
    torch.testing.assert_close(lmp.energy.double().cpu().view(-1), orc["total_energy"].view(-1), atol=5e-5 * n, rtol=5e-5)
    f_atoms = _edge_forces_to_atoms(lmp.pair_forces, data["edge_index"], n)
    df = float((f_atoms - orc["forces"]).abs().max())
    assert df < 1e-4 * max(1.0, float(orc["forces"].abs().max())), f"ML-IAP wrapper: forces differ from the oracle by {df:.3e}"
    # a rank without work returns before touching anything
    empty = FakeMLIAPData(0, owner[:0].to(device), edge_vec[:0].to(device), ei_l[0, :0].to(device), ei_l[1, :0].to(device),
                          types_l[:0].to(device))
    wrapper.compute_forces(empty)
    assert empty.energy is None and empty.pair_forces is None


def test_mliap_wrapper_attributes_and_cpu_refusal(tmp_path):
    from nequip_amd.integrations.lammps_mliap import NequIPLAMMPSMLIAPWrapper, create_lmp_mliap_file
    from nequip_amd.model import NequIPGNNModel

    model = NequIPGNNModel(seed=0, r_max=5.0, type_names=["H", "C", "O"], num_layers=2, l_max=1, parity=True,
                           num_features=8, avg_num_neighbors=10.0)
    with pytest.raises(ValueError, match="nequip.lmp.pt"):
        create_lmp_mliap_file(model, str(tmp_path / "model.pt"))
    wrapper = torch.load(create_lmp_mliap_file(model, str(tmp_path / "m.nequip.lmp.pt")), weights_only=False)
    assert isinstance(wrapper, NequIPLAMMPSMLIAPWrapper)
    assert wrapper.element_types == ["H", "C", "O"] and wrapper.rcutfac == 2.5
    assert wrapper.nparams == 1 and wrapper.ndescriptors == 1
    assert wrapper.compute_descriptors(None) is None and wrapper.compute_gradients(None) is None

    class CpuData:  # what a LAMMPS build without Kokkos passes (module name has no "kokkos")
        nlocal, ntotal, npairs = 4, 4, 6

    with pytest.raises(RuntimeError, match="GPU only"):
        wrapper.compute_forces(CpuData())


def test_noop_ghost_exchange_is_default_and_modifier_is_private():
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.nn import InteractionBlock, NoOpGhostExchangeModule
    from nequip_amd.nn.model_modifier_utils import is_model_modifier

    m = NequIPGNNModel(seed=0, model_dtype="float32", r_max=4.5, type_names=["H", "O"], num_layers=2, l_max=1,
                       parity=False, num_features=4, radial_mlp_depth=1, radial_mlp_width=8, avg_num_neighbors=10.0)
    blocks = [x for x in m.modules() if isinstance(x, InteractionBlock)]
    assert all(isinstance(b.ghost_exchange, NoOpGhostExchangeModule) for b in blocks)
    assert len(m.state_dict()) == len({k for k in m.state_dict()})  # the exchange modules own no parameters
    assert not any("ghost_exchange" in k for k in m.state_dict())
    assert is_model_modifier(NoOpGhostExchangeModule.enable_LAMMPSMLIAPGhostExchange)
