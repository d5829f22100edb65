"""Reverse-edge pairing (nqa_edge_pairs, csrc/edge_pairs.hip) and the paired radial-MLP / tensor-product evaluation
(nn/_paired_radial.py): the reference evaluates InteractionBlock.edge_mlp on every directed edge
(nequip/nn/interaction_block.py:190-192); evaluating it once per (i <- j, S) / (j <- i, -S) pair must give the same
energies, forces and stress."""
import numpy as np
import pytest
import torch


@pytest.fixture(autouse=True)
def default_switches(monkeypatch):
    """These tests assert that the paired path IS taken: run them with the switches that disable it at their defaults."""
    import os

    if os.environ.get("NQA_FORCE_GENERIC", "") not in ("", "0"):
        pytest.skip("NQA_FORCE_GENERIC is read once per process by the library: no specialised kernels, no pairing")
    monkeypatch.delenv("NQA_NO_PAIRED", raising=False)
    monkeypatch.delenv("NQA_MLP_EXACT_FP32", raising=False)


def _pairing_of(data, device):
    from nequip_amd.nn._topology import EdgeTopology

    ei = data["edge_index"].to(device)
    topo = EdgeTopology(ei[0].contiguous(), ei[1].contiguous(), data["pos"].shape[0])
    sh = data["edge_cell_shift"].to(device) if "edge_cell_shift" in data else None
    return topo, topo.pairing(sh)


@pytest.mark.gpu
@pytest.mark.parametrize("system", ["water", "si_small_cell", "molecule"])
def test_pairing_is_a_reverse_edge_matching(device, system):
    from nequip_amd.utils import synthetic as syn

    if system == "water":
        pos, types, cell, names = syn.water_box(n_side=3, seed=1)
        data = syn.make_data(pos, types, 4.5, cell)
    elif system == "si_small_cell":  # cell thinner than 2 r_max: several images of the same (i, j), self images
        pos, types, cell, names = syn.silicon_box(reps=1, seed=2)
        data = syn.make_data(pos, types, 4.5, cell)
    else:
        pos, types, cell, names = syn.water_box(n_side=2, seed=3)
        data = syn.make_data(pos, types, 4.5, None, pbc=False)
    topo, pr = _pairing_of(data, device)
    assert pr is not None
    E = data["edge_index"].shape[1]
    P = E // 2
    rows = pr.rows.cpu().numpy()
    rep = pr.rep_edge.cpu().numpy()
    dst, src = data["edge_index"][0].numpy(), data["edge_index"][1].numpy()
    sh = data["edge_cell_shift"].numpy() if "edge_cell_shift" in data else np.zeros((E, 3))
    assert sorted(rows.tolist()) == list(range(2 * P))  # every row of the [2P, W] gradient buffer is written once
    other = np.empty(P, dtype=np.int64)
    other[rows[rows >= P] - P] = np.nonzero(rows >= P)[0]
    assert np.array_equal(rows[rep], np.arange(P))
    assert np.array_equal(dst[rep], src[other]) and np.array_equal(src[rep], dst[other])
    assert np.array_equal(sh[rep], -sh[other])
    # slot-order copies follow the CSR permutations
    eid_d = topo.by_dst[1][:E].cpu().numpy()
    assert np.array_equal(pr.slots_dst.cpu().numpy(), rows[eid_d])
    eid_s = topo.by_src[1][:E].cpu().numpy()
    assert np.array_equal(pr.slots_src.cpu().numpy(), rows[eid_s])


@pytest.mark.gpu
@pytest.mark.parametrize("system", ["water", "si_small_cell", "molecule", "shuffled"])
def test_owner_lists_from_the_csr_match_the_host_statement_of_the_rule(device, system):
    """`nqa_pair_owner_lists` (counting passes over the dst-CSR, no sort) against `build_owner_csr` (the rule written with ATen
    sorts, itself tested on the CPU in test_host_logic.py): the same slots per owner and per other node -- as sets, the order
    within a node is the CSR's here -- and every slot consistent with the edge list."""
    from nequip_amd.nn._topology import build_owner_csr
    from nequip_amd.utils import synthetic as syn

    if system in ("water", "shuffled"):
        pos, types, cell, names = syn.water_box(n_side=3, seed=1)
        data = syn.make_data(pos, types, 4.5, cell)
        if system == "shuffled":  # an edge list in no particular order (the CSR sort has work to do)
            perm = torch.randperm(data["edge_index"].shape[1], generator=torch.Generator().manual_seed(0))
            data["edge_index"] = data["edge_index"][:, perm].contiguous()
            data["edge_cell_shift"] = data["edge_cell_shift"][perm].contiguous()
    elif system == "si_small_cell":
        pos, types, cell, names = syn.silicon_box(reps=1, seed=2)
        data = syn.make_data(pos, types, 4.5, cell)
    else:
        pos, types, cell, names = syn.water_box(n_side=2, seed=3)
        data = syn.make_data(pos, types, 4.5, None, pbc=False)
    topo, pr = _pairing_of(data, device)
    assert pr is not None
    N, P = data["pos"].shape[0], pr.num_pairs
    got = [t.cpu().long() for t in pr.owner_csr]
    ref = [t.cpu().long() for t in build_owner_csr(topo._dst, topo._src, pr.rows, P, N)]
    orow, oth, prow, ein, eout, trow, tslot = got
    assert torch.equal(orow, ref[0]) and torch.equal(trow, ref[5])  # the same number of slots per node, both ways
    dst, src = data["edge_index"][0], data["edge_index"][1]
    rows = pr.rows.cpu().long()
    assert sorted(prow.tolist()) == list(range(P)) and sorted(tslot.tolist()) == list(range(P))
    owner = torch.repeat_interleave(torch.arange(N), orow[1:] - orow[:-1])
    assert torch.equal(dst[ein], owner) and torch.equal(src[ein], oth[:P])     # edge_in: other -> owner
    assert torch.equal(dst[eout], oth[:P]) and torch.equal(src[eout], owner)   # edge_out: owner -> other
    assert torch.equal(rows[ein] % P, prow) and torch.equal(rows[eout] % P, prow) and not torch.equal(rows[ein], rows[eout])
    other_of_slot = torch.repeat_interleave(torch.arange(N), trow[1:] - trow[:-1])
    assert torch.equal(oth[:P][tslot], other_of_slot)
    # as sets per owner: (pair, edge_in, edge_out) triples equal the reference's
    def triples(lst):
        o = torch.repeat_interleave(torch.arange(N), lst[0][1:] - lst[0][:-1])
        return sorted(zip(o.tolist(), lst[2].tolist(), lst[3].tolist(), lst[4].tolist()))
    assert triples(got) == triples(ref)


@pytest.mark.gpu
def test_unpairable_lists_are_rejected(device):
    from nequip_amd.nn._topology import EdgeTopology

    dst = torch.tensor([0, 1, 2, 0], device=device)
    src = torch.tensor([1, 0, 0, 2], device=device)
    assert EdgeTopology(dst, src, 3).pairing(None) is not None
    assert EdgeTopology(dst[:3].contiguous(), src[:3].contiguous(), 3).pairing(None) is None          # odd count
    assert EdgeTopology(torch.tensor([0, 1, 2, 2], device=device), torch.tensor([1, 0, 0, 1], device=device),
                        3).pairing(None) is None                                                       # no reverse
    assert EdgeTopology(torch.tensor([0, 0, 1, 1], device=device), torch.tensor([1, 1, 0, 0], device=device),
                        2).pairing(None) is None                                                       # duplicates
    sh = torch.tensor([[1.0, 0, 0], [1.0, 0, 0]], device=device, dtype=torch.float64)
    assert EdgeTopology(torch.tensor([0, 1], device=device), torch.tensor([1, 0], device=device), 2).pairing(sh) is None
    sh[1, 0] = -1.0
    assert EdgeTopology(torch.tensor([0, 1], device=device), torch.tensor([1, 0], device=device), 2).pairing(sh) is not None


@pytest.mark.gpu
@pytest.mark.parametrize("periodic", [True, False])
def test_paired_evaluation_matches_per_edge_evaluation(device, periodic, monkeypatch):
    from nequip_amd.data import AtomicDataDict
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.nn import _paired_radial
    from nequip_amd.nn._topology import topology_cache
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=4, seed=5)
    model = NequIPGNNModel(seed=2, model_dtype="float32", r_max=4.5, type_names=names, num_layers=3, l_max=2,
                           parity=False, num_features=64, radial_mlp_depth=1, radial_mlp_width=128,
                           avg_num_neighbors=30.0).to(device).eval()
    data = AtomicDataDict.to_device(syn.make_data(pos, types, 4.5, cell if periodic else None, pbc=periodic), device)
    calls = []
    orig = _paired_radial._PairedRadialTPFn.forward
    monkeypatch.setattr(_paired_radial._PairedRadialTPFn, "forward",
                        staticmethod(lambda *a, **k: (calls.append(1), orig(*a, **k))[1]))
    out_p = model(dict(data))
    assert len(calls) == 3, "the paired path must be taken in all three layers"
    monkeypatch.setenv("NQA_NO_PAIRED", "1")
    topology_cache.clear()
    out_e = model(dict(data))
    assert len(calls) == 3
    e_p, e_e = out_p["total_energy"].detach(), out_e["total_energy"].detach()
    torch.testing.assert_close(e_p, e_e, rtol=1e-6, atol=1e-5)
    f_p, f_e = out_p["forces"].detach(), out_e["forces"].detach()
    torch.testing.assert_close(f_p, f_e, rtol=0, atol=2e-6 * max(1.0, float(f_e.abs().max())))
    if periodic:
        torch.testing.assert_close(out_p["stress"].detach(), out_e["stress"].detach(), rtol=0,
                                   atol=2e-6 * max(1e-3, float(out_e["stress"].abs().max())))


@pytest.mark.gpu
def test_paired_training_step_matches_per_edge(device, monkeypatch):
    """Force-matching loss and all parameter gradients (double backward) with and without pairing."""
    from nequip_amd.data import AtomicDataDict
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.nn._topology import topology_cache
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=3, seed=11)
    model = NequIPGNNModel(seed=4, model_dtype="float32", r_max=4.5, type_names=names, num_layers=3, l_max=2,
                           parity=False, num_features=64, radial_mlp_depth=1, radial_mlp_width=128,
                           avg_num_neighbors=30.0).to(device).train()
    data = AtomicDataDict.to_device(syn.make_data(pos, types, 4.5, cell), device)
    gen = torch.Generator().manual_seed(0)
    f_t = torch.randn(len(pos), 3, generator=gen, dtype=torch.float64).to(device)

    def run():
        model.zero_grad(set_to_none=True)
        out = model(dict(data))
        loss = (out["forces"] - f_t).square().mean() + out["total_energy"].square().mean() / len(pos) ** 2
        loss.backward()
        return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    loss_p, grads_p = run()
    monkeypatch.setenv("NQA_NO_PAIRED", "1")
    topology_cache.clear()
    loss_e, grads_e = run()
    torch.testing.assert_close(loss_p, loss_e, rtol=1e-5, atol=1e-6)
    for k in grads_e:
        scale = max(1e-6, float(grads_e[k].abs().max()))
        torch.testing.assert_close(grads_p[k], grads_e[k], rtol=0, atol=3e-4 * scale, msg=lambda m, k=k: f"{k}: {m}")


@pytest.mark.gpu
def test_paired_tp_kernels_match_expanded_weights(device):
    """nqa_tp_scatter_*_paired with w = [P, W] against the plain entry points with the rows expanded to [E, W]:
    identical outputs / feature and edge-attr gradients, weight gradient = the per-edge gradient scattered to rows
    p / p + P."""
    from nequip_amd.nn import TensorProductScatter
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=3, seed=2)
    data = syn.make_data(pos, types, 4.5, cell)
    topo, pr = _pairing_of(data, device)
    assert pr is not None
    N, E, P = len(pos), data["edge_index"].shape[1], pr.num_pairs
    from nequip_amd.model import NequIPGNNModel

    model = NequIPGNNModel(seed=2, model_dtype="float32", r_max=4.5, type_names=names, num_layers=3, l_max=2,
                           parity=False, num_features=64, radial_mlp_depth=1, radial_mlp_width=128,
                           avg_num_neighbors=30.0).to(device).eval()
    tps = [m for m in model.modules() if isinstance(m, TensorProductScatter)][1]  # the middle layer's structure
    k = tps._get_kernels()
    assert k.has_spec(torch.float32)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, k.dim_in1, generator=g).to(device)
    y = torch.randn(E, k.dim_in2, generator=g).to(device)
    w_half = torch.randn(P, k.weight_numel, generator=g).to(device)
    go = torch.randn(N, k.dim_out, generator=g).to(device)
    rows = pr.rows.long()
    w_full = w_half[rows % P].contiguous()

    out_p, out_e = k.fwd(x, y, w_half, topo, pr), k.fwd(x, y, w_full, topo)
    assert torch.equal(out_p, out_e)
    gx_p, gx_e = k.bwd_x(y, w_half, go, topo, pr), k.bwd_x(y, w_full, go, topo)
    assert torch.equal(gx_p, gx_e)
    (G, gy_p), (gw_e, gy_e) = k.bwd_edge(x, y, w_half, go, topo, True, True, pairing=pr), k.bwd_edge(x, y, w_full, go, topo, True, True)
    assert G.shape == (2 * P, k.weight_numel) and torch.equal(gy_p, gy_e) and torch.equal(G[rows], gw_e)
    fx, fG, fy = k.bwd_fused(x, y, w_half, go, topo, pairing=pr)
    ex, eG, ey = k.bwd_fused(x, y, w_full, go, topo)
    assert torch.equal(fx, ex) and torch.equal(fy, ey) and torch.equal(fG[rows], eG)


@pytest.mark.gpu
@pytest.mark.parametrize("E,H,W", [(1000, 128, 704), (77, 64, 64), (4133, 128, 192)])
def test_paired_mlp_backward_adds_the_two_streams(device, E, H, W):
    from nequip_amd.nn import mlp as M

    torch.manual_seed(E)
    mod = M.ScalarMLPFunction(input_dim=8, output_dim=W, hidden_layers_depth=1, hidden_layers_width=H).to(device).eval()
    emb = (torch.randn(E, 8) * 0.7).to(device)
    assert mod._fused_ok(emb)
    ga, gb = torch.randn(E, W, device=device), torch.randn(E, W, device=device)
    cache = M._WeightImages()
    cache.validate(mod.mlp[2].weight)
    args = (emb, mod.mlp[0].weight.detach(), mod.mlp[2].weight.detach(), mod._alphas[0], mod._alphas[1])
    got = M._launch_bwd_paired(*args, ga, gb, M._lib.NQA_MLP_BF16X6, cache)
    ref = M._launch_bwd(*args, (ga + gb).contiguous(), M._lib.NQA_MLP_BF16X6, cache)
    # (two different kernels since round 5 -- the single-stream form runs on the balanced launch of radial_mlp_pipe.h, whose
    # partial sums and SiLU' evaluation are not bit-identical to the paired kernel's: equal at the fp32 rounding level)
    torch.testing.assert_close(got, ref, rtol=0, atol=4e-6 * float(ref.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("system", ["water", "si_small_cell"])
def test_paired_edge_embedding_launch_matches_gather_and_expand(device, system):
    """`nqa_edge_embed_fwd_paired / _bwd_paired` (round 5): the per-pair radial rows of the joint embedding launch are
    bitwise `emb[rep_edge]`, and its backward with a per-pair cotangent is bitwise the per-edge backward of the expanded
    cotangent (what `pair_gather` / `pair_expand` + the plain launches computed)."""
    from nequip_amd.nn.embedding import _edge
    from nequip_amd.nn._paired_radial import _PairExpandFn
    from nequip_amd.utils import synthetic as syn

    if system == "water":
        pos, types, cell, names = syn.water_box(n_side=3, seed=4)
    else:
        pos, types, cell, names = syn.silicon_box(reps=1, seed=5)
    data = syn.make_data(pos, types, 4.5, cell)
    topo, pr = _pairing_of(data, device)
    assert pr is not None
    ei = data["edge_index"]
    vec = (data["pos"][ei[1]] - data["pos"][ei[0]] + data["edge_cell_shift"].to(torch.float64) @ data["cell"].view(3, 3)).to(device)
    vec = vec.to(torch.float64).contiguous().requires_grad_(True)
    bw = torch.linspace(1.0, 8.0, 8, dtype=torch.float64, device=device)
    cfg = dict(dtype=torch.float32, lmax=2, want_sh=True, want_emb=True, nb=8, rmax_recip=1.0 / 4.5, p=6.0, factor=0.31)
    sh, emb, emb_pairs = _edge._EdgeEmbedPairedFn.apply(vec, bw, cfg, pr)
    sh0, emb0 = _edge._EdgeEmbedFn.apply(vec, bw, cfg)
    assert torch.equal(sh, sh0) and torch.equal(emb, emb0)
    assert torch.equal(emb_pairs, emb0[pr.rep_edge])
    g_sh = torch.randn_like(sh)
    g_pairs = torch.randn_like(emb_pairs)
    g_emb = torch.randn_like(emb)
    (gv,) = torch.autograd.grad([sh, emb, emb_pairs], [vec], [g_sh, g_emb, g_pairs], retain_graph=True)
    expanded = _PairExpandFn.apply(g_pairs, pr, vec.shape[0])
    (gv0,) = torch.autograd.grad([sh0, emb0], [vec], [g_sh, g_emb + expanded], retain_graph=True)
    # (the kernel adds the two float32 cotangents in float64, `g_emb + expanded` above rounds their sum to float32 first)
    torch.testing.assert_close(gv, gv0, rtol=0, atol=2e-6 * float(gv0.abs().max()))
    (gv_p,) = torch.autograd.grad([emb_pairs], [vec], [g_pairs], retain_graph=True)  # only the per-pair cotangent
    (gv_p0,) = torch.autograd.grad([emb0], [vec], [expanded])
    assert torch.equal(gv_p, gv_p0)
