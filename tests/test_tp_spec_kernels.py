"""Op-level parity of the float32 *structure-specialised* TensorProductScatter kernels -- the ones bench.py times --
against the CPU oracle (``oracle/tp.py``), for every structure ``gen_spec.baseline_structures()`` prebuilds, at the
benchmarked channel counts mul = 32 / 64 / 128 (128 = two 64-lane chunks per node: per-chunk ``grad_y`` partials,
``tp_spec.h``), on ragged neighbourhoods (isolated nodes, degree 1 .. 13, unsorted edge list).

Recipe of the reference's own boundary test (``tests/unit/nn/test_tp_scatter_kernel.py:34-179``: instructions built
as InteractionBlock builds them, forward and the gradient w.r.t. each operand, atol = rtol = 1e-5), applied to each
native entry point separately: ``fwd``, ``bwd_x``, ``bwd_edge``, ``bwd_fused`` and their ``_paired`` forms
(one weight row per reverse-edge pair).  atol is scaled by max(1, max|ref|): sums over up to 13 edges x up to 6
paths of O(1) products reach O(30).
"""

import os
import sys

import pytest
import torch

from oracle import tp as otp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "nequip_amd", "csrc"))
import gen_spec  # noqa: E402

TOL = 1e-5


def _unique_structures():
    seen, out = set(), []
    for name, f_in, lmax, f_out in gen_spec.baseline_irreps():
        st = gen_spec.nequip_structure(f_in, lmax, f_out, name)
        if st.instr and st.key() not in seen:
            seen.add(st.key())
            out.append((name, f_in, lmax, f_out))
    return out


STRUCTS = _unique_structures()


def _graph(n_nodes, seed, symmetric):
    """Ragged random graph.  symmetric=True: every (i <- j) has its (j <- i) (a pairable list), edges shuffled."""
    g = torch.Generator().manual_seed(seed)
    pairs = set()
    deg_target = torch.randint(0, 7, (n_nodes,), generator=g).tolist()
    deg_target[0] = 0  # an isolated node
    for i in range(1, n_nodes):
        for _ in range(deg_target[i]):
            j = int(torch.randint(1, n_nodes, (1,), generator=g))
            if j != i:
                pairs.add((min(i, j), max(i, j)))
    pairs = sorted(pairs)
    if symmetric:
        dst = [a for a, b in pairs] + [b for a, b in pairs]
        src = [b for a, b in pairs] + [a for a, b in pairs]
    else:  # directed, with repeated edges (the reference's test draws dst / src independently)
        dst = [a for a, b in pairs] + [a for a, b in pairs[::3]]
        src = [b for a, b in pairs] + [b for a, b in pairs[::3]]
    perm = torch.randperm(len(dst), generator=g)
    return torch.tensor(dst)[perm].contiguous(), torch.tensor(src)[perm].contiguous()


def _module(f_in_1x, lmax, f_out_1x, mul, device):
    from nequip_amd.nn import TensorProductScatter
    from nequip_amd.o3 import Irreps

    f_in = f_in_1x.replace("1x", f"{mul}x")
    f_out = f_out_1x.replace("1x", f"{mul}x")
    e_at = str(Irreps.spherical_harmonics(lmax))
    mid, instructions = otp.build_instructions(f_in, e_at, f_out)
    mid_s = "+".join(f"{m}x{l}{'e' if p == 1 else 'o'}" for m, l, p in mid)
    tps = TensorProductScatter(Irreps(f_in), Irreps(e_at), Irreps(mid_s), instructions).to(device)
    return tps, f_in, e_at, mid_s, instructions


def _close(ref, got, what):
    scale = max(1.0, float(ref.abs().max()))
    torch.testing.assert_close(got.cpu(), ref, atol=TOL * scale, rtol=TOL, msg=lambda m: f"{what}: {m}")


def _cases():
    for name, f_in, lmax, f_out in STRUCTS:
        for mul in (32, 64, 128):
            yield pytest.param(name, f_in, lmax, f_out, mul, id=f"{name}-mul{mul}")


@pytest.mark.gpu
@pytest.mark.parametrize("name,f_in_1x,lmax,f_out_1x,mul", list(_cases()))
def test_spec_kernels_vs_oracle(device, name, f_in_1x, lmax, f_out_1x, mul):
    from nequip_amd.nn._topology import EdgeTopology

    if os.environ.get("NQA_FORCE_GENERIC", "") not in ("", "0"):
        pytest.skip("specialised kernels switched off")
    tps, f_in, e_at, mid_s, instructions = _module(f_in_1x, lmax, f_out_1x, mul, device)
    k = tps._get_kernels()
    assert k.has_spec(torch.float32), f"structure {name} has no specialised kernel"
    N = 19
    for symmetric in (False, True):
        dst, src = _graph(N, seed=1000 * lmax + mul + int(symmetric), symmetric=symmetric)
        E = dst.numel()
        g = torch.Generator().manual_seed(17 + mul)
        x = torch.randn(N, k.dim_in1, generator=g)
        y = torch.randn(E, k.dim_in2, generator=g)
        go = torch.randn(N, k.dim_out, generator=g)
        topo = EdgeTopology(dst.to(device), src.to(device), N)
        pr = topo.pairing(None) if symmetric else None
        assert (pr is not None) == symmetric
        if symmetric:
            P = pr.num_pairs
            w_rows = torch.randn(P, k.weight_numel, generator=g)
            rows = pr.rows.long().cpu()
            w = w_rows[rows % P].contiguous()  # the per-edge weights the oracle sees
        else:
            w_rows = w = torch.randn(E, k.weight_numel, generator=g)

        xr, yr, wr = (t.clone().requires_grad_(True) for t in (x, y, w))
        ref = otp.tp_scatter(xr, yr, wr, dst, src, f_in, e_at, mid_s, instructions)
        rgx, rgy, rgw = torch.autograd.grad(ref, (xr, yr, wr), go)
        ref = ref.detach()

        d = lambda t: t.to(device)  # noqa: E731
        xd, yd, wd, god = d(x), d(y), d(w_rows), d(go)
        tag = f"{name} mul={mul} {'paired' if symmetric else 'plain'}"
        _close(ref, k.fwd(xd, yd, wd, topo, pr), f"fwd {tag}")
        _close(rgx, k.bwd_x(yd, wd, god, topo, pr), f"bwd_x {tag}")
        gw, gy = k.bwd_edge(xd, yd, wd, god, topo, True, True, pairing=pr)
        if symmetric:
            gw = gw[d(rows)]
        _close(rgw, gw, f"bwd_edge gw {tag}")
        _close(rgy, gy, f"bwd_edge gy {tag}")
        # single-output instantiations of the edge backward
        gw1, none_y = k.bwd_edge(xd, yd, wd, god, topo, True, False, pairing=pr)
        none_w, gy1 = k.bwd_edge(xd, yd, wd, god, topo, False, True, pairing=pr)
        assert none_y is None and none_w is None
        _close(rgw, gw1[d(rows)] if symmetric else gw1, f"bwd_edge<gw only> {tag}")
        _close(rgy, gy1, f"bwd_edge<gy only> {tag}")
        fused = k.bwd_fused(xd, yd, wd, god, topo, pairing=pr)
        assert fused is not None
        fx, fw, fy = fused
        if symmetric:
            fw = fw[d(rows)]
        _close(rgx, fx, f"bwd_fused gx {tag}")
        _close(rgw, fw, f"bwd_fused gw {tag}")
        _close(rgy, fy, f"bwd_fused gy {tag}")
        # forward JVP (second-order backward): the three bilinear terms in one pass, and every subset of them
        xc, yc = d(torch.randn(N, k.dim_in1, generator=g)), d(torch.randn(E, k.dim_in2, generator=g))
        wc = d(torch.randn(wd.shape, generator=g))
        for cx_, cy_, cw_ in ((xc, yc, wc), (xc, yc, None), (None, yc, wc), (xc, None, None)):
            want = sum(t for t in ((k.fwd(cx_, yd, wd, topo, pr) if cx_ is not None else None),
                                   (k.fwd(xd, cy_, wd, topo, pr) if cy_ is not None else None),
                                   (k.fwd(xd, yd, cw_, topo, pr) if cw_ is not None else None)) if t is not None)
            _close(want.cpu(), k.fwd_jvp(xd, yd, wd, cx_, cy_, cw_, topo, pr), f"fwd_jvp {tag}")
        _close((k.bwd_x(yc, wd, god, topo, pr) + k.bwd_x(yd, wc, god, topo, pr)).cpu(),
               k.bwd_x_dual(yd, wd, yc, wc, god, topo, pr), f"bwd_x_dual {tag}")
        if symmetric and k.has_pairs_kernel(torch.float32):
            # pair-centric backward: grad_w comes out summed over the two directed edges of every pair
            px, pw, py = k.bwd_pairs(xd, yd, wd, god, topo, pr)
            rgw_pairs = torch.zeros(P, k.weight_numel).index_add_(0, rows % P, rgw)
            _close(rgx, px, f"bwd_pairs gx {tag}")
            _close(rgw_pairs, pw, f"bwd_pairs gw {tag}")
            _close(rgy, py, f"bwd_pairs gy {tag}")
            if k.has_dual_pairs_kernel(torch.float32):
                # dual pass of the second-order backward: Bw(x2, y, g) + Bw(x, y2, g) and By(x2, w, g)
                x2d = d(torch.randn(N, k.dim_in1, generator=g))
                y2d = d(torch.randn(E, k.dim_in2, generator=g))
                _, w_a, y_a = k.bwd_pairs(x2d, yd, wd, god, topo, pr, need_gx=False)
                _, w_b, _ = k.bwd_pairs(xd, y2d, wd, god, topo, pr, need_gx=False)
                dw, dy = k.edge_grads_dual(xd, x2d, yd, y2d, wd, god, topo, pr)
                _close((w_a + w_b).cpu(), dw, f"dual gw {tag}")
                _close(y_a.cpu(), dy, f"dual gy {tag}")
                _, y_c = k.bwd_edge(xd, yd, wc, god, topo, need_gw=False, need_gy=True, pairing=pr)
                dw3, dy3 = k.edge_grads_dual(xd, x2d, yd, y2d, wd, god, topo, pr, w_cot=wc)
                _close((w_a + w_b).cpu(), dw3, f"dual gw (with w_cot) {tag}")
                _close((y_a + y_c).cpu(), dy3, f"dual gy (with w_cot) {tag}")
            none_x, pw2, py2 = k.bwd_pairs(xd, yd, wd, god, topo, pr, need_gx=False)  # the edge gradients only
            assert none_x is None
            _close(rgw_pairs, pw2, f"bwd_pairs<no gx> gw {tag}")
            _close(rgy, py2, f"bwd_pairs<no gx> gy {tag}")


@pytest.mark.gpu
@pytest.mark.parametrize("mul", [64, 128])
def test_spec_kernels_large_degree_and_wave_split(device, mul):
    """The launch heuristics split a node's edges over four wavefronts on small graphs (LDS tree reduction) and use one
    wavefront per node on large ones; exercise both with 60-neighbour nodes (the cfg-5 Cu box has ~38-78)."""
    from nequip_amd.nn._topology import EdgeTopology

    name, f_in_1x, lmax, f_out_1x = next(s for s in STRUCTS if s[0] == ("l2n_mid" if mul == 64 else "l3n_mid"))
    tps, f_in, e_at, mid_s, instructions = _module(f_in_1x, lmax, f_out_1x, mul, device)
    k = tps._get_kernels()
    assert k.has_spec(torch.float32)
    N = 7
    g = torch.Generator().manual_seed(5)
    dst = torch.cat([torch.full((60,), 3), torch.full((1,), 5), torch.full((33,), 6)])
    src = torch.randint(0, N, (dst.numel(),), generator=g)
    E = dst.numel()
    x = torch.randn(N, k.dim_in1, generator=g)
    y = torch.randn(E, k.dim_in2, generator=g)
    w = torch.randn(E, k.weight_numel, generator=g) / 4
    go = torch.randn(N, k.dim_out, generator=g)
    xr, yr, wr = (t.clone().requires_grad_(True) for t in (x, y, w))
    ref = otp.tp_scatter(xr, yr, wr, dst, src, f_in, e_at, mid_s, instructions)
    rgx, rgy, rgw = torch.autograd.grad(ref, (xr, yr, wr), go)
    topo = EdgeTopology(dst.to(device), src.to(device), N)
    d = lambda t: t.to(device)  # noqa: E731
    _close(ref.detach(), k.fwd(d(x), d(y), d(w), topo), "fwd")
    _close(rgx, k.bwd_x(d(y), d(w), d(go), topo), "bwd_x")
    fx, fw, fy = k.bwd_fused(d(x), d(y), d(w), d(go), topo)
    _close(rgx, fx, "fused gx")
    _close(rgw, fw, "fused gw")
    _close(rgy, fy, "fused gy")


# the three forms of the pair-centric backward for multiples of 64 channels (round 6; both switches are read at every call):
# the LDS-ring kernel with the other node's grad_x in an atomically summed accumulator (default), the ring kernel with one
# row per pair + the fixed-order row sum, and the register kernel of rounds 3-5
PAIR_FORMS = {"ring_atomic": {"NQA_PAIR_RING": "1", "NQA_PAIR_GX_ATOMIC": "1"},  # (explicit: the suite may run under either switch)
              "ring_rows": {"NQA_PAIR_RING": "1", "NQA_PAIR_GX_ATOMIC": "0"},
              "registers": {"NQA_PAIR_RING": "0", "NQA_PAIR_GX_ATOMIC": "1"}}


@pytest.mark.gpu
@pytest.mark.parametrize("form", list(PAIR_FORMS))
@pytest.mark.parametrize("system", ["si_small_cell", "water", "water_two_chunks", "water_wide", "cu_l3"])
def test_pair_backward_equals_per_edge_backward(device, system, form, monkeypatch):
    """`nqa_tp_scatter_bwd_pairs` against `nqa_tp_scatter_bwd_fused_paired` on real neighbour lists: a cell thinner than
    2 r_max (several images of one (i, j), self images: owner == other), a water box large enough for one wavefront per
    node and for four, the same with 128 channels (two chunks per node, grad_y through the partial buffer) and with the
    narrow first / last layer structures, and the l_max = 3 / 128-feature structure (split kernel)."""
    from nequip_amd.nn._topology import EdgeTopology
    from nequip_amd.utils import synthetic as syn

    for k_, v_ in PAIR_FORMS[form].items():
        monkeypatch.setenv(k_, v_)
    if system == "water_two_chunks":
        pos, types, cell, names = syn.water_box(n_side=4, seed=1)
        sname, mul = "l2n_mid", 128
    elif system == "water_wide":
        pos, types, cell, names = syn.water_box(n_side=13, seed=4)  # 6591 atoms: one wavefront per node in the ring kernel
        sname, mul = "l2n_last", 64
    elif system == "si_small_cell":
        pos, types, cell, names = syn.silicon_box(reps=1, seed=2)
        sname, mul = "l2n_mid", 64
    elif system == "water":
        pos, types, cell, names = syn.water_box(n_side=4, seed=1)
        sname, mul = "l2n_mid", 64
    else:
        pos, types, cell, names = syn.copper_box(reps=(3, 3, 3), seed=3)
        sname, mul = "l3n_mid", 128
    data = syn.make_data(pos, types, 4.5, cell)
    name, f_in_1x, lmax, f_out_1x = next(s for s in STRUCTS if s[0] == sname)
    tps, f_in, e_at, mid_s, instructions = _module(f_in_1x, lmax, f_out_1x, mul, device)
    k = tps._get_kernels()
    if not k.has_pairs_kernel(torch.float32):
        pytest.skip("no pair-centric kernel generated for this structure")
    ei = data["edge_index"].to(device)
    N, E = data["pos"].shape[0], ei.shape[1]
    topo = EdgeTopology(ei[0].contiguous(), ei[1].contiguous(), N)
    pr = topo.pairing(data["edge_cell_shift"].to(device))
    assert pr is not None
    P = pr.num_pairs
    # the owner lists: a partition of the pairs, edge_in / edge_out are the two directed edges of the pair
    orow, oth, prow, ein, eout, trow, tslot = (t.cpu().long() for t in pr.owner_csr)
    assert sorted(prow.tolist()) == list(range(P)) and sorted(tslot.tolist()) == list(range(P))
    dst, src, rows = ei[0].cpu(), ei[1].cpu(), pr.rows.cpu().long()
    owner = torch.repeat_interleave(torch.arange(N), orow[1:] - orow[:-1])
    assert torch.equal(dst[ein], owner) and torch.equal(src[ein], oth)
    assert torch.equal(dst[eout], oth) and torch.equal(src[eout], owner)
    assert torch.equal(rows[ein] % P, prow) and torch.equal(rows[eout] % P, prow)
    assert torch.equal(oth[tslot], torch.repeat_interleave(torch.arange(N), trow[1:] - trow[:-1]))
    per_owner = (orow[1:] - orow[:-1]).float()
    per_node = torch.bincount(dst, minlength=N).float()
    assert float((per_owner - per_node / 2).abs().max()) <= 0.25 * float(per_node.max()) + 2  # balanced halves

    g = torch.Generator().manual_seed(11)
    d = lambda t: t.to(device)  # noqa: E731
    x, y = d(torch.randn(N, k.dim_in1, generator=g)), d(torch.randn(E, k.dim_in2, generator=g))
    w, go = d(torch.randn(P, k.weight_numel, generator=g) / 4), d(torch.randn(N, k.dim_out, generator=g))
    fx, fw, fy = k.bwd_fused(x, y, w, go, topo, pairing=pr)
    px, pw, py = k.bwd_pairs(x, y, w, go, topo, pr)
    _close(fx.cpu(), px, "gx")
    _close((fw[:P] + fw[P:]).cpu(), pw, "gw (summed over the pair)")
    _close(fy.cpu(), py, "gy")


@pytest.mark.gpu
@pytest.mark.parametrize("form,expect", [("ring_atomic", ("bwd_pair_ring_kernel", "gx_acc_finish_kernel")),
                                         ("ring_rows", ("bwd_pair_ring_kernel", "gx_rows_sum_kernel")),
                                         ("registers", ("bwd_pair_kernel", "gx_rows_sum_kernel"))])
def test_the_selected_pair_kernels_are_the_ones_that_run(device, form, expect, monkeypatch):
    """Which GPU kernels a call of `nqa_tp_scatter_bwd_pairs` launches under each of the three forms (kernel names from the
    profiler's device activity records): the default must BE the LDS-ring kernel with the accumulator, not a fallback."""
    from torch.profiler import ProfilerActivity, profile

    from nequip_amd.nn._topology import EdgeTopology
    from nequip_amd.utils import synthetic as syn

    for k_, v_ in PAIR_FORMS[form].items():
        monkeypatch.setenv(k_, v_)
    pos, types, cell, names = syn.water_box(n_side=4, seed=1)
    data = syn.make_data(pos, types, 4.5, cell)
    name, f_in_1x, lmax, f_out_1x = next(s for s in STRUCTS if s[0] == "l2n_mid")
    tps, f_in, e_at, mid_s, instructions = _module(f_in_1x, lmax, f_out_1x, 64, device)
    k = tps._get_kernels()
    ei = data["edge_index"].to(device)
    N, E = data["pos"].shape[0], ei.shape[1]
    topo = EdgeTopology(ei[0].contiguous(), ei[1].contiguous(), N)
    pr = topo.pairing(data["edge_cell_shift"].to(device))
    P = pr.num_pairs
    g = torch.Generator().manual_seed(3)
    d = lambda t: t.to(device)  # noqa: E731
    x, y = d(torch.randn(N, k.dim_in1, generator=g)), d(torch.randn(E, k.dim_in2, generator=g))
    w, go = d(torch.randn(P, k.weight_numel, generator=g) / 4), d(torch.randn(N, k.dim_out, generator=g))
    k.bwd_pairs(x, y, w, go, topo, pr)  # (lists, workspace, first-launch attributes: outside the recorded region)
    torch.cuda.synchronize()
    try:
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            k.bwd_pairs(x, y, w, go, topo, pr)
            torch.cuda.synchronize()
        names_seen = {e.key for e in prof.key_averages()}
    except Exception as exc:  # pragma: no cover
        pytest.skip(f"no device activity records here: {exc}")
    if not any("nqa" in n for n in names_seen):
        pytest.skip("the profiler returned no kernel names on this box")
    for want in expect:
        assert any(want + "<" in n or want + "(" in n for n in names_seen), (want, sorted(names_seen))
    if form != "registers":
        assert not any("bwd_pair_kernel<" in n for n in names_seen), sorted(names_seen)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["ring_atomic", "ring_rows"])
def test_ring_kernel_more_than_64_pairs_per_wavefront(device, form, monkeypatch):
    """One wavefront of the ring kernel keeps the indices of 64 of its pairs in registers and fetches the next block when it
    runs out: a random pairable graph with ~170 neighbours per node (85 owned pairs) on enough nodes for the one-wavefront-
    per-node launch shape, against the per-edge fused backward."""
    from nequip_amd.nn._topology import EdgeTopology

    for k_, v_ in PAIR_FORMS[form].items():
        monkeypatch.setenv(k_, v_)
    name, f_in_1x, lmax, f_out_1x = next(s for s in STRUCTS if s[0] == "l2n_mid")
    tps, f_in, e_at, mid_s, instructions = _module(f_in_1x, lmax, f_out_1x, 64, device)
    k = tps._get_kernels()
    N = 6400
    dst, src = _big_graph(N, N * 86, seed=5)
    dst, src = dst.to(device), src.to(device)
    E = dst.numel()
    topo = EdgeTopology(dst, src, N)
    pr = topo.pairing(None)
    assert pr is not None
    orow = pr.owner_csr[0].cpu().long()
    assert int((orow[1:] - orow[:-1]).max()) > 64, "the graph must exercise the second index block"
    P = pr.num_pairs
    g = torch.Generator().manual_seed(12)
    d = lambda t: t.to(device)  # noqa: E731
    x, y = d(torch.randn(N, k.dim_in1, generator=g)), d(torch.randn(E, k.dim_in2, generator=g))
    w, go = d(torch.randn(P, k.weight_numel, generator=g) / 4), d(torch.randn(N, k.dim_out, generator=g))
    fx, fw, fy = k.bwd_fused(x, y, w, go, topo, pairing=pr)
    px, pw, py = k.bwd_pairs(x, y, w, go, topo, pr)
    _close(fx.cpu(), px, "gx")
    _close((fw[:P] + fw[P:]).cpu(), pw, "gw (summed over the pair)")
    _close(fy.cpu(), py, "gy")


def _big_graph(n_nodes, n_pairs, seed):
    """Pairable random graph built with tensor ops (tens of thousands of nodes): every (i <- j) has its (j <- i)."""
    g = torch.Generator().manual_seed(seed)
    a = torch.randint(1, n_nodes, (n_pairs,), generator=g)
    b = torch.randint(1, n_nodes, (n_pairs,), generator=g)
    keep = a != b
    lo, hi = torch.minimum(a, b)[keep], torch.maximum(a, b)[keep]
    key = torch.unique(lo * n_nodes + hi)
    lo, hi = key // n_nodes, key % n_nodes
    dst, src = torch.cat([lo, hi]), torch.cat([hi, lo])
    perm = torch.randperm(dst.numel(), generator=g)
    return dst[perm].contiguous(), src[perm].contiguous()


@pytest.mark.gpu
@pytest.mark.parametrize("name,f_in_1x,lmax,f_out_1x,mul,n_nodes", [
    ("l2n_mid", "1x0e+1x1o+1x2e", 2, "1x0e+1x1o+1x2e", 64, 50000),
    ("l3n_mid", "1x0e+1x1o+1x2e+1x3o", 3, "1x0e+1x1o+1x2e+1x3o", 128, 25000),
])
def test_one_wavefront_per_node_launch_shape_vs_oracle(device, name, f_in_1x, lmax, f_out_1x, mul, n_nodes):
    """The launch shape of the LARGE boxes -- one wavefront per (node, 64-channel chunk) instead of four, chosen when
    nodes x chunks >= 49 152 (cfg-5 and beyond; `spec_wpn`) -- against the oracle, for the forward, the pair-centric
    backward (one wavefront per pair for l_max 2, split by input block over four for l_max 3) and the per-edge fused
    backward.  Sparse graph (about three neighbours per node) so that the oracle stays affordable."""
    from nequip_amd.nn._topology import EdgeTopology

    tps, f_in, e_at, mid_s, instructions = _module(f_in_1x, lmax, f_out_1x, mul, device)
    k = tps._get_kernels()
    assert k.has_spec(torch.float32)
    assert n_nodes * ((mul + 63) // 64) >= 49152
    dst, src = _big_graph(n_nodes, int(1.5 * n_nodes), seed=11 + lmax)
    E = dst.numel()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n_nodes, k.dim_in1, generator=g)
    y = torch.randn(E, k.dim_in2, generator=g)
    go = torch.randn(n_nodes, k.dim_out, generator=g)
    topo = EdgeTopology(dst.to(device), src.to(device), n_nodes)
    pr = topo.pairing(None)
    assert pr is not None
    P = pr.num_pairs
    w_rows = torch.randn(P, k.weight_numel, generator=g)
    rows = pr.rows.long().cpu()
    w = w_rows[rows % P].contiguous()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    xr, yr, wr = (t.clone().requires_grad_(True) for t in (x, y, w))
    ref = otp.tp_scatter(xr, yr, wr, dst, src, f_in, e_at, mid_s, instructions, edge_chunk=8192)
    rgx, rgy, rgw = torch.autograd.grad(ref, (xr, yr, wr), go)
    ref = ref.detach()
    d = lambda t: t.to(device)  # noqa: E731
    xd, yd, wd, god = d(x), d(y), d(w_rows), d(go)
    tag = f"{name} mul={mul} N={n_nodes} E={E} (one wavefront per node)"
    _close(ref, k.fwd(xd, yd, wd, topo, pr), f"fwd {tag}")
    fx, fw, fy = k.bwd_fused(xd, yd, wd, god, topo, pairing=pr)
    _close(rgx, fx, f"bwd_fused gx {tag}")
    _close(rgw, fw[d(rows)], f"bwd_fused gw {tag}")
    _close(rgy, fy, f"bwd_fused gy {tag}")
    assert k.has_pairs_kernel(torch.float32)
    px, pw, py = k.bwd_pairs(xd, yd, wd, god, topo, pr)
    rgw_pairs = torch.zeros(P, k.weight_numel).index_add_(0, rows % P, rgw)
    _close(rgx, px, f"bwd_pairs gx {tag}")
    _close(rgw_pairs, pw, f"bwd_pairs gw {tag}")
    _close(rgy, py, f"bwd_pairs gy {tag}")
