"""The reference's model property suite (nequip/utils/unittests/model_tests_basic.py:450-843) run directly on the
HIP-backed model -- evidence for the GPU path that does not go through the oracle:

* E(3) symmetries: rotations / reflections leave the energy invariant and rotate the forces (and the virial tensor),
  translations and atom permutations act trivially (equivariance bar 5e-5, tests/model/test_nequip_model.py:93);
* forces = -dE/dpos and virial = -dE/dstrain by central finite differences on a float64 model;
* batching: frames evaluated in one batch give the same per-frame results as evaluated alone, and there is no
  cross-frame coupling (model_tests_basic.py:598-629);
* smooth cutoff: an atom moved across r_max changes nothing discontinuously, edges beyond r_max contribute exactly zero.
"""

import math

import numpy as np
import pytest
import torch


def _model(device, names, dtype="float32", parity=True, l_max=2, num_layers=2, nf=8):
    from nequip_amd.model import NequIPGNNModel

    return NequIPGNNModel(seed=3, model_dtype=dtype, r_max=4.0, type_names=names, num_layers=num_layers, l_max=l_max,
                          parity=parity, num_features=nf, radial_mlp_depth=1, radial_mlp_width=64, num_bessels=8,
                          polynomial_cutoff_p=6, avg_num_neighbors=12.0,
                          per_type_energy_scales={n: 1.0 + 0.3 * i for i, n in enumerate(names)},
                          per_type_energy_shifts={n: -0.2 * i for i, n in enumerate(names)}).to(device).eval()


def _molecule(seed=0, n=14):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-2.2, 2.2, size=(n, 3))
    types = rng.integers(0, 3, size=n)
    return pos, types, ["C", "H", "O"]


def _eval(model, pos, types, device, cell=None):
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.utils import synthetic as syn

    data = syn.make_data(pos, types, 4.0, cell, pbc=cell is not None)
    out = model(K.to_device(data, device))
    return {k: out[k].detach().cpu() for k in (K.TOTAL_ENERGY_KEY, K.FORCE_KEY) + ((K.VIRIAL_KEY,) if cell is not None else ())}


def _rotation(seed, improper=False):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if (np.linalg.det(q) < 0) != improper:
        q[:, 0] = -q[:, 0]
    return q


@pytest.mark.gpu
@pytest.mark.parametrize("parity", [True, False])
@pytest.mark.parametrize("improper", [False, True])
def test_rotation_reflection_translation_permutation(device, parity, improper):
    from nequip_amd.data import AtomicDataDict as K

    pos, types, names = _molecule(1)
    model = _model(device, names, parity=parity)
    ref = _eval(model, pos, types, device)
    R = _rotation(7, improper)
    if improper and not parity:
        pytest.skip("a parity=False model is only SO(3)-equivariant")
    rot = _eval(model, pos @ R.T, types, device)
    torch.testing.assert_close(rot[K.TOTAL_ENERGY_KEY], ref[K.TOTAL_ENERGY_KEY], atol=5e-5, rtol=5e-5)
    torch.testing.assert_close(rot[K.FORCE_KEY], ref[K.FORCE_KEY] @ torch.tensor(R.T), atol=5e-5, rtol=5e-5)
    tr = _eval(model, pos + np.array([3.1, -0.7, 11.0]), types, device)
    torch.testing.assert_close(tr[K.TOTAL_ENERGY_KEY], ref[K.TOTAL_ENERGY_KEY], atol=5e-5, rtol=5e-5)
    torch.testing.assert_close(tr[K.FORCE_KEY], ref[K.FORCE_KEY], atol=5e-5, rtol=5e-5)
    perm = np.random.default_rng(5).permutation(len(pos))
    pm = _eval(model, pos[perm], types[perm], device)
    torch.testing.assert_close(pm[K.TOTAL_ENERGY_KEY], ref[K.TOTAL_ENERGY_KEY], atol=5e-5, rtol=5e-5)
    torch.testing.assert_close(pm[K.FORCE_KEY], ref[K.FORCE_KEY][perm], atol=5e-5, rtol=5e-5)
    assert float(ref[K.FORCE_KEY].abs().max()) > 1e-3, "degenerate test: forces vanish"
    torch.testing.assert_close(ref[K.FORCE_KEY].sum(0), torch.zeros(3, dtype=torch.float64), atol=1e-4, rtol=0)


@pytest.mark.gpu
def test_periodic_rotation_of_virial(device):
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(2, seed=4)
    model = _model(device, names, parity=False, l_max=2)
    ref = _eval(model, pos, types, device, cell)
    R = _rotation(11)
    rot = _eval(model, pos @ R.T, types, device, np.asarray(cell) @ R.T)
    Rt = torch.tensor(R)
    torch.testing.assert_close(rot[K.TOTAL_ENERGY_KEY], ref[K.TOTAL_ENERGY_KEY], atol=1e-4, rtol=5e-5)
    torch.testing.assert_close(rot[K.FORCE_KEY], ref[K.FORCE_KEY] @ Rt.T, atol=5e-5, rtol=5e-5)
    v = ref[K.VIRIAL_KEY].view(3, 3)
    torch.testing.assert_close(rot[K.VIRIAL_KEY].view(3, 3), Rt @ v @ Rt.T, atol=2e-4, rtol=1e-4)
    torch.testing.assert_close(v, v.T, atol=1e-6, rtol=0)


@pytest.mark.gpu
def test_finite_difference_forces_and_virial(device):
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(2, seed=9)
    cell = np.asarray(cell)
    model = _model(device, names, dtype="float64", parity=True, l_max=2)
    ref = _eval(model, pos, types, device, cell)
    h = 1e-5
    rng = np.random.default_rng(0)
    for i in rng.choice(len(pos), 4, replace=False):
        for a in range(3):
            p, m = pos.copy(), pos.copy()
            p[i, a] += h
            m[i, a] -= h
            ep = _eval(model, p, types, device, cell)[K.TOTAL_ENERGY_KEY].item()
            em = _eval(model, m, types, device, cell)[K.TOTAL_ENERGY_KEY].item()
            assert abs(-(ep - em) / (2 * h) - ref[K.FORCE_KEY][i, a].item()) < 1e-6
    # virial = -dE/d(strain), symmetric strain applied to positions and cell
    for (a, b) in [(0, 0), (0, 1), (1, 2), (2, 2)]:
        eps = np.zeros((3, 3))
        eps[a, b] += 0.5 * h
        eps[b, a] += 0.5 * h
        ep = _eval(model, pos @ (np.eye(3) + eps), types, device, cell @ (np.eye(3) + eps))[K.TOTAL_ENERGY_KEY].item()
        em = _eval(model, pos @ (np.eye(3) - eps), types, device, cell @ (np.eye(3) - eps))[K.TOTAL_ENERGY_KEY].item()
        fd = -(ep - em) / (2 * h)
        assert abs(fd - ref[K.VIRIAL_KEY].view(3, 3)[a, b].item()) < 1e-5, (a, b)


@pytest.mark.gpu
def test_batching_and_cross_frame_independence(device):
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.utils import synthetic as syn

    frames, singles = [], []
    names = ["C", "H", "O"]
    model = _model(device, names, parity=True, l_max=1)
    for s in range(3):
        pos, types, _ = _molecule(20 + s, n=9 + 2 * s)
        d = syn.make_data(pos, types, 4.0, None, pbc=False)
        frames.append(d)
        singles.append(model(K.to_device(dict(d), device)))
    batched = K.batched_from_list([dict(f) for f in frames])
    out = model(K.to_device(batched, device))
    off = 0
    for s, (f, o) in enumerate(zip(frames, singles)):
        n = f[K.POSITIONS_KEY].shape[0]
        torch.testing.assert_close(out[K.TOTAL_ENERGY_KEY][s], o[K.TOTAL_ENERGY_KEY][0], atol=5e-5, rtol=5e-5)
        torch.testing.assert_close(out[K.FORCE_KEY][off : off + n], o[K.FORCE_KEY], atol=5e-5, rtol=5e-5)
        off += n
    # cross-frame gradient: the energy of frame 0 must not depend on the positions of the other frames
    b2 = K.to_device(K.batched_from_list([dict(f) for f in frames]), device)
    n0 = frames[0][K.POSITIONS_KEY].shape[0]
    b2[K.POSITIONS_KEY] = b2[K.POSITIONS_KEY].clone()
    b2[K.POSITIONS_KEY][n0:] += 0.01  # rigid shift of frames 1, 2 (their own edges unchanged)
    out2 = model(b2)
    torch.testing.assert_close(out2[K.TOTAL_ENERGY_KEY][0], out[K.TOTAL_ENERGY_KEY][0], atol=1e-6, rtol=0)


@pytest.mark.gpu
def test_cutoff_smoothness(device):
    from nequip_amd.data import AtomicDataDict as K

    names = ["C", "H", "O"]
    model = _model(device, names, dtype="float64", parity=True, l_max=2)
    base = np.array([[0.0, 0.0, 0.0], [-1.1, 0.2, -0.3], [-0.4, 1.3, 0.5]])  # atoms 1, 2 stay > r_max from atom 3
    types = np.array([0, 1, 2])
    e = []
    for r in (3.9999, 4.0001, 4.5):  # fourth atom just inside / just outside / well outside r_max = 4.0 of atom 0
        pos = np.vstack([base, [[r, 0.0, 0.0]]])
        e.append(_eval(model, pos, np.append(types, 1), device))
    d_in_out = abs(e[0][K.TOTAL_ENERGY_KEY].item() - e[1][K.TOTAL_ENERGY_KEY].item())
    assert d_in_out < 1e-6, d_in_out  # p = 6 polynomial envelope: value and derivatives vanish at r_max
    f_jump = (e[0][K.FORCE_KEY][0] - e[1][K.FORCE_KEY][0]).abs().max().item()
    assert f_jump < 1e-4, f_jump


@pytest.mark.gpu
@pytest.mark.parametrize("periodic", [False, True])
def test_no_edges(device, periodic):
    """Atoms farther apart than r_max: energy = sum of per-type constants, forces / virial exactly zero (no edge at all
    reaches the kernels: E = 0 everywhere, first and only use of the empty-graph paths end to end)."""
    from nequip_amd.data import AtomicDataDict as K

    names = ["C", "H", "O"]
    model = _model(device, names, parity=True, l_max=2)
    pos = np.array([[0.0, 0.0, 0.0], [9.0, 0.0, 0.0], [0.0, 11.0, 0.0]])
    types = np.array([0, 1, 2])
    cell = np.eye(3) * 30.0 if periodic else None
    out = _eval(model, pos, types, device, cell)
    assert torch.isfinite(out[K.TOTAL_ENERGY_KEY]).all()
    assert float(out[K.FORCE_KEY].abs().max()) == 0.0
    if periodic:
        assert float(out[K.VIRIAL_KEY].abs().max()) == 0.0
    single = [_eval(model, pos[i : i + 1], types[i : i + 1], device, cell)[K.TOTAL_ENERGY_KEY].item() for i in range(3)]
    assert abs(sum(single) - out[K.TOTAL_ENERGY_KEY].item()) < 1e-5


@pytest.mark.gpu
def test_edge_vector_inputs_give_edge_forces(device):
    """LAMMPS ML-IAP style call (nequip/nn/grad_output.py:276-296, lmp_mliap_wrapper.py:202-233): edge vectors instead
    of positions in, `edge_forces = dE/d edge_vectors` out (no sign flip).  Consistency with the position path:
    same energy, and the edge forces scattered onto the two atoms of every edge give the forces."""
    from nequip_amd.data import AtomicDataDict as K

    from nequip_amd.utils import synthetic as syn

    pos, types, names = _molecule(seed=5, n=20)
    data = syn.make_data(pos, types, 4.0, None, pbc=False)
    model = _model(device, names)
    ref = model(K.to_device(dict(data), device))
    ei = data[K.EDGE_INDEX_KEY]
    P = data[K.POSITIONS_KEY]
    edge_vec = (P[ei[1]] - P[ei[0]]).to(device)
    out = model({K.EDGE_VECTORS_KEY: edge_vec, K.EDGE_INDEX_KEY: ei.to(device), K.ATOM_TYPE_KEY: data[K.ATOM_TYPE_KEY].to(device)})
    assert K.EDGE_FORCE_KEY in out and K.FORCE_KEY not in out
    torch.testing.assert_close(out[K.TOTAL_ENERGY_KEY].detach(), ref[K.TOTAL_ENERGY_KEY].detach(), rtol=1e-6, atol=1e-5)
    ef = out[K.EDGE_FORCE_KEY].detach().double()
    assert ef.shape == edge_vec.shape
    grad_pos = torch.zeros_like(ref[K.FORCE_KEY].detach().double())
    grad_pos.index_add_(0, ei[1].to(device), ef)
    grad_pos.index_add_(0, ei[0].to(device), -ef)
    f_ref = ref[K.FORCE_KEY].detach().double()
    torch.testing.assert_close(-grad_pos, f_ref, rtol=0, atol=2e-5 * max(1.0, float(f_ref.abs().max())))


@pytest.mark.gpu
@pytest.mark.parametrize("nf", [8, 64])
def test_spatial_node_order_is_invisible_in_the_results(device, monkeypatch, nf):
    """Round 6: in eval mode the convolution stack of a large single frame runs in a Morton order of the atoms
    (`ForceStressOutput._spatial_order`: relabelled edge list + permuted types in, node fields permuted back out).  Energy,
    forces, virial / stress, per-atom energies, node features and the edge-wise fields -- in the CALLER's atom and edge
    order -- must be what the model gives without it, on a box whose atoms arrive in a shuffled order (so that the
    permutation is far from the identity); the caller's own tensors come back untouched."""
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.nn._topology import topology_cache
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=5, seed=2)  # 375 atoms
    rng = np.random.default_rng(0)
    shuffle = rng.permutation(len(pos))
    pos, types = np.asarray(pos)[shuffle], np.asarray(types)[shuffle]
    model = _model(device, names, parity=False, nf=nf, num_layers=3)
    data0 = syn.make_data(pos, types, 4.0, cell)

    def run(flag):
        monkeypatch.setenv("NQA_SPATIAL_ORDER", flag)
        monkeypatch.setenv("NQA_SPATIAL_ORDER_MIN", "1")
        topology_cache.clear()
        data = {k: v.clone().to(device) for k, v in data0.items()}
        out = model(data)
        assert torch.equal(out[K.EDGE_INDEX_KEY], data0[K.EDGE_INDEX_KEY].to(device))       # the caller's list is handed back
        assert torch.equal(out[K.ATOM_TYPE_KEY].cpu().view(-1), data0[K.ATOM_TYPE_KEY].view(-1))
        got = {k: out[k].detach().double().cpu() for k in (K.TOTAL_ENERGY_KEY, K.FORCE_KEY, K.VIRIAL_KEY, K.STRESS_KEY,
                                                           K.PER_ATOM_ENERGY_KEY, K.EDGE_VECTORS_KEY, K.EDGE_ATTRS_KEY)}
        for k in (K.NODE_FEATURES_KEY, K.NODE_ATTRS_KEY):  # (whatever node-wise tensors the chain leaves behind)
            if k in out and torch.is_tensor(out[k]):
                got[k] = out[k].detach().double().cpu()
        return got

    plain, ordered = run("0"), run("1")
    for k in plain:
        scale = max(1.0, float(plain[k].abs().max()))
        assert float((plain[k] - ordered[k]).abs().max()) <= 2e-5 * scale, k
    # ... and the permutation really was applied (a second evaluation re-uses the cached one)
    monkeypatch.setenv("NQA_SPATIAL_ORDER", "1")
    data = {k: v.clone().to(device) for k, v in data0.items()}
    model(data)
    ei = data[K.EDGE_INDEX_KEY]
    sp = getattr(topology_cache.get(ei[0], ei[1], len(pos)), "_spatial", None)
    assert sp is not None and not torch.equal(sp.perm.cpu(), torch.arange(len(pos)))
