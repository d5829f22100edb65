"""Size-independent properties at the BASELINE sizes themselves -- where the CPU oracle would need minutes per
evaluation: the exact bench.py workloads cfg-2 (1000-atom Si), cfg-3 (10 125-atom water, the headline metric) and the
cfg-5 model on a 20 000-atom Cu box.  Checked: translation invariance (the forces of a periodic box sum to zero), SO(3)
equivariance (rotating positions and cell leaves the energy and rotates forces and virial), permutation equivariance
(relabelled atoms and shuffled edge list), and agreement of the paired-radial evaluation with the per-edge one.
Tolerances: fp32 model, 5e-5 relative to the largest force (nequip/utils/dtype.py:35-42 equivariance bar)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _setup(workload, device):
    import bench
    from nequip_amd.data import AtomicDataDict

    w = bench.WORKLOADS[workload]
    data_cpu, names = bench.build_box(w, seed=0)
    n, e = data_cpu["pos"].shape[0], data_cpu["edge_index"].shape[1]
    model = bench.build_model(bench.model_cfg(w, e / n), names, device)
    return model, AtomicDataDict.to_device(data_cpu, device), n


def _eval(model, data):
    out = model(dict(data))
    return (out["total_energy"].detach().double().cpu(), out["forces"].detach().double().cpu(),
            out["virial"].detach().double().cpu())


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["si1k", "water10k", "cu20k"])
def test_symmetries_at_baseline_size(device, workload):
    model, data, n = _setup(workload, device)
    e0, f0, v0 = _eval(model, data)
    fmax = float(f0.abs().max())
    assert fmax > 1e-4 and torch.isfinite(f0).all() and torch.isfinite(e0).all()
    tol_f = 5e-5 * max(1.0, fmax)

    # translation invariance: the adjoint of the edge-vector map is summed in float64, every edge gradient enters two
    # atoms with opposite signs
    assert float(f0.sum(0).abs().max()) < 1e-9 * n * max(1.0, fmax)

    # rotation of positions and cell (edge list and integer shifts are unchanged)
    rng = np.random.default_rng(3)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    R = torch.tensor(q, dtype=torch.float64, device=device)
    rot = dict(data)
    rot["pos"] = data["pos"] @ R.T
    rot["cell"] = data["cell"].view(-1, 3, 3) @ R.T
    e1, f1, v1 = _eval(model, rot)
    Rc = R.cpu()
    torch.testing.assert_close(e1, e0, rtol=5e-6, atol=5e-5 * n * 1e-2)
    torch.testing.assert_close(f1, f0 @ Rc.T, rtol=0, atol=tol_f)
    vscale = max(1.0, float(v0.abs().max()))
    torch.testing.assert_close(v1, Rc @ v0 @ Rc.T, rtol=0, atol=5e-5 * vscale)

    # permutation of the atoms and of the edge list
    g = torch.Generator().manual_seed(1)
    perm = torch.randperm(n, generator=g).to(device)          # new index -> old index
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n, device=device)
    E = data["edge_index"].shape[1]
    eperm = torch.randperm(E, generator=g).to(device)
    p = dict(data)
    p["pos"] = data["pos"][perm]
    p["atom_types"] = data["atom_types"][perm]
    p["edge_index"] = inv[data["edge_index"]][:, eperm].contiguous()
    p["edge_cell_shift"] = data["edge_cell_shift"][eperm].contiguous()
    e2, f2, v2 = _eval(model, p)
    torch.testing.assert_close(e2, e0, rtol=5e-6, atol=5e-5 * n * 1e-2)
    torch.testing.assert_close(f2, f0[perm.cpu()], rtol=0, atol=tol_f)


@pytest.mark.gpu
def test_paired_equals_per_edge_at_headline_size(device, monkeypatch):
    """cfg-3 box: the radial MLP once per reverse-edge pair (what bench.py times) against once per directed edge."""
    from nequip_amd.nn._topology import topology_cache

    model, data, n = _setup("water10k", device)
    monkeypatch.delenv("NQA_NO_PAIRED", raising=False)
    e0, f0, v0 = _eval(model, data)
    monkeypatch.setenv("NQA_NO_PAIRED", "1")
    topology_cache.clear()
    e1, f1, v1 = _eval(model, data)
    torch.testing.assert_close(e1, e0, rtol=1e-6, atol=1e-4)
    torch.testing.assert_close(f1, f0, rtol=0, atol=3e-6 * max(1.0, float(f0.abs().max())))


@pytest.mark.gpu
def test_pair_centric_backward_equals_per_edge_backward_at_headline_size(device, monkeypatch):
    """cfg-3 box: the pair-centric backward kernels (`nqa_tp_scatter_bwd_pairs`, what bench.py times) against the per-edge
    fused / edge backward on the same paired weights: same energy, forces and virial."""
    model, data, n = _setup("water10k", device)
    monkeypatch.delenv("NQA_NO_PAIRED", raising=False)
    monkeypatch.delenv("NQA_NO_PAIR_BWD", raising=False)
    e0, f0, v0 = _eval(model, data)
    monkeypatch.setenv("NQA_NO_PAIR_BWD", "1")
    e1, f1, v1 = _eval(model, data)
    torch.testing.assert_close(e1, e0, rtol=0, atol=0)  # the forward pass is the same code
    torch.testing.assert_close(f1, f0, rtol=0, atol=3e-6 * max(1.0, float(f0.abs().max())))
    torch.testing.assert_close(v1, v0, rtol=0, atol=3e-6 * max(1.0, float(v0.abs().max())))
