"""The CSR / pairing cache (nn/_topology.py): reuse for a static neighbour list, several live graphs at once, and the
guards against index memory that is rewritten behind PyTorch's back (robustness)."""
import pytest
import torch


def _graph(device, n=40, e=300, seed=0):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g, dtype=torch.int64).to(device)
    return ei, n


@pytest.mark.gpu
def test_alternating_graphs_do_not_rebuild(device):
    from nequip_amd.nn._topology import topology_cache

    topology_cache.clear()
    (a, n), (b, _) = _graph(device, seed=1), _graph(device, seed=2)
    ta, tb = topology_cache.get(a[0], a[1], n), topology_cache.get(b[0], b[1], n)
    assert ta is not tb
    for _ in range(3):
        assert topology_cache.get(a[0], a[1], n) is ta
        assert topology_cache.get(b[0], b[1], n) is tb
    a[0, 0] = (a[0, 0] + 1) % n  # an in-place write PyTorch sees bumps the version: new topology
    assert topology_cache.get(a[0], a[1], n) is not ta


@pytest.mark.gpu
def test_untrusted_scope_shares_within_one_evaluation_only(device):
    from nequip_amd.nn._topology import topology_cache

    topology_cache.clear()
    a, n = _graph(device, seed=3)
    with topology_cache.scope(trust_identity=False):
        t1 = topology_cache.get(a[0], a[1], n)
        assert topology_cache.get(a[0], a[1], n) is t1  # the layers of one forward share it
    with topology_cache.scope(trust_identity=False):
        assert topology_cache.get(a[0], a[1], n) is not t1  # the next evaluation rebuilds
    assert topology_cache.get(a[0], a[1], n) is not t1  # and nothing of it was left in the cross-call cache


@pytest.mark.gpu
def test_out_of_band_rewrite_is_caught_in_verify_mode(device, monkeypatch):
    from nequip_amd.nn._topology import topology_cache

    topology_cache.clear()
    monkeypatch.setenv("NQA_TOPOLOGY_VERIFY", "1")
    a, n = _graph(device, seed=4)
    t = topology_cache.get(a[0], a[1], n)
    assert topology_cache.get(a[0], a[1], n) is t
    v = a._version
    a.data[0, :5] = (a.data[0, :5] + 1) % n  # `.data` write: contents change, the version counter does not
    assert a._version == v
    with pytest.raises(RuntimeError, match="rewritten in place"):
        topology_cache.get(a[0], a[1], n)
    topology_cache.invalidate()
    assert topology_cache.get(a[0], a[1], n) is not t


@pytest.mark.gpu
def test_lammps_style_data_never_uses_the_cross_call_cache(device):
    """GraphModel.forward opens an untrusted scope when the caller is the LAMMPS ML-IAP wrapper (its index arrays are
    views of buffers it refills every step)."""
    from nequip_amd.data import AtomicDataDict
    from nequip_amd.nn._topology import topology_cache
    from nequip_amd.nn.graph_model import GraphModel

    seen = []

    class Probe(torch.nn.Module):
        irreps_in, irreps_out = {}, {}

        def forward(self, data):
            ei = data[AtomicDataDict.EDGE_INDEX_KEY]
            seen.append(topology_cache.get(ei[0], ei[1], 40))
            seen.append(topology_cache.get(ei[0], ei[1], 40))
            return data

    topology_cache.clear()
    gm = GraphModel(Probe(), type_names=["X"])
    ei, _ = _graph(device, seed=5)

    class Lmp:
        nlocal = 40

    gm({AtomicDataDict.EDGE_INDEX_KEY: ei, AtomicDataDict.LMP_MLIAP_DATA_KEY: Lmp()})
    gm({AtomicDataDict.EDGE_INDEX_KEY: ei, AtomicDataDict.LMP_MLIAP_DATA_KEY: Lmp()})
    assert seen[0] is seen[1] and seen[2] is seen[3] and seen[0] is not seen[2]
    seen.clear()
    gm({AtomicDataDict.EDGE_INDEX_KEY: ei})
    gm({AtomicDataDict.EDGE_INDEX_KEY: ei})
    assert seen[0] is seen[1] is seen[2] is seen[3]  # an ordinary caller keeps its static graph
