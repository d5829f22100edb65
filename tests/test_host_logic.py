"""CPU tests (no GPU): host-side irreps bookkeeping, instruction building, native plan creation and the C ABI."""

import os
import re

import pytest
import torch

from nequip_amd.o3 import Irrep, Irreps
from oracle import irreps as oir
from oracle import tp as otp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_irreps_parse_sort_simplify():
    ir = Irreps("8x0e + 8x2e + 8x1o")
    assert ir.dim == 8 + 40 + 24 and ir.num_irreps == 24 and ir.lmax == 2
    s, p, inv = ir.sort()
    assert str(s) == "8x0e+8x1o+8x2e" and p == (0, 2, 1) and inv == (0, 2, 1)
    mid = Irreps("64x0e+64x0e+64x1o+64x1o+64x2e")
    assert str(mid.simplify()) == "128x0e+128x1o+64x2e"
    # tuple order (l, p): odd before even at equal l (e3nn convention)
    assert str(Irreps("1x1e+1x1o+1x0e+1x0o").sort().irreps) == "1x0o+1x0e+1x1o+1x1e"
    assert list(Irrep("1o") * Irrep("2e")) == [Irrep("1o"), Irrep("2o"), Irrep("3o")]
    assert str(Irreps.spherical_harmonics(3)) == "1x0e+1x1o+1x2e+1x3o"
    assert Irrep("2e") in Irreps("3x2e") and Irrep("2o") not in Irreps("3x2e")
    assert Irreps("4x0e + 3x1o").randn(5, -1).shape == (5, 13)


@pytest.mark.parametrize("f_in,lmax,f_out", [
    ("64x0e", 2, "64x0e+64x1o+64x2e"),
    ("64x0e+64x1o+64x2e", 2, "192x0e+64x1o+64x2e"),
    ("64x0e+64x1o+64x2e", 2, "64x0e"),
    ("32x0e+32x0o+32x1e+32x1o", 1, "32x0e"),
    ("128x0e+128x1o+128x2e+128x3o", 3, "512x0e+128x1o+128x2e+128x3o"),
])
def test_instruction_building_matches_oracle_and_survey(f_in, lmax, f_out):
    """InteractionBlock-style instruction list (nequip/nn/interaction_block.py:89-109) built by the product module
    and, independently, by the oracle; path counts / W / D_mid against SURVEY.md Appendix B."""
    from nequip_amd.csrc.gen_spec import nequip_structure

    st = nequip_structure(f_in, lmax, f_out, "t")
    mid, instr = otp.build_instructions(f_in, oir.to_str(oir.spherical_harmonics(lmax)), f_out)
    assert [(a, b, c) for a, b, c, *_ in instr] == st.instr
    assert [l for _, l, _ in mid] == st.out_ls
    expected = {  # (paths, W, D_mid) from SURVEY.md App. B
        ("64x0e", 2): (3, 192, 576), ("64x0e+64x1o+64x2e", 2, "192x0e+64x1o+64x2e"): (11, 704, 2240),
    }
    key = (f_in, lmax, f_out) if (f_in, lmax, f_out) in expected else (f_in, lmax)
    if key in expected:
        paths, W, dmid = expected[key]
        assert len(instr) == paths and otp.weight_numel(f_in, oir.to_str(oir.spherical_harmonics(lmax)), instr) == W
        assert oir.dim(mid) == dmid


def test_capi_exports_every_declared_symbol():
    """libnequip_amd.so loads (no GPU needed) and exports every function include/nequip_amd.h declares."""
    import ctypes

    from nequip_amd import _lib

    header = open(os.path.join(ROOT, "include", "nequip_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(nqa_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in nequip_amd.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes signature table out of sync with the header"
    assert lib.nqa_abi_version() == 1 and lib.nqa_lmax() >= 3 and lib.nqa_sh_lmax() >= 3


def test_native_plan_matches_host_bookkeeping():
    from nequip_amd import _lib
    from nequip_amd.nn import TensorProductScatter

    f_in, e_at = Irreps("64x0e+64x1o+64x2e"), Irreps.spherical_harmonics(2)
    mid, instr = otp.build_instructions(str(f_in), str(e_at), "192x0e+64x1o+64x2e")
    tps = TensorProductScatter(f_in, e_at, Irreps(oir.to_str(mid)), instr)
    plan = tps._plan
    assert plan.query(_lib.NQA_PLAN_DIM_IN1) == 576 and plan.query(_lib.NQA_PLAN_DIM_IN2) == 9
    assert plan.query(_lib.NQA_PLAN_DIM_OUT) == 2240 and plan.query(_lib.NQA_PLAN_WEIGHT_NUMEL) == 704
    assert tps.tp.weight_numel == 704 and plan.query(_lib.NQA_PLAN_NUM_INSTR) == 11
    assert plan.query(_lib.NQA_PLAN_HAS_SPECIALIZED) == 1, "BASELINE mid-layer structure must have prebuilt kernels"
    assert plan.query(_lib.NQA_PLAN_OUT_NEEDS_ZERO) == 0
    # no persistent state: the module adds nothing to the state dict (reference contract)
    assert list(tps.state_dict().keys()) == []
    # irregular irreps fall back to the generic kernels
    mid2, instr2 = otp.build_instructions("4x0e + 3x1o + 2x2e", "0e + 1o", "0e + 1o + 2e")
    tps2 = TensorProductScatter(Irreps("4x0e + 3x1o + 2x2e"), Irreps("0e + 1o"), Irreps(oir.to_str(mid2)), instr2)
    assert tps2._plan.query(_lib.NQA_PLAN_HAS_SPECIALIZED) == 0
    assert plan.query(_lib.NQA_PLAN_FUSED_ROWS_OK) == 1 and tps2._plan.query(_lib.NQA_PLAN_FUSED_ROWS_OK) == 0
    # the full l_max = 4 middle layer (the XL preset's first channel segment) has specialised kernels, but its fused
    # edge-row backward would hold 364 values per channel: callers take nqa_tp_scatter_bwd_x + _bwd_edge instead
    h4 = "32x0e+32x1o+32x2e+32x3o+32x4e"
    mid4, instr4 = otp.build_instructions(h4, str(Irreps.spherical_harmonics(4)), h4)
    tps4 = TensorProductScatter(Irreps(h4), Irreps.spherical_harmonics(4), Irreps(oir.to_str(mid4)), instr4)
    assert tps4._plan.query(_lib.NQA_PLAN_HAS_SPECIALIZED) == 1 and tps4._plan.query(_lib.NQA_PLAN_FUSED_ROWS_OK) == 0
    assert tps4._get_kernels().fused_rows_ok is False


def test_native_plan_rejects_invalid_instructions():
    from nequip_amd.nn import TensorProductScatter

    with pytest.raises(RuntimeError, match="parity"):
        TensorProductScatter(Irreps("2x1o"), Irreps("1x1o"), Irreps("2x1o"), [(0, 0, 0, "uvu", True)])
    with pytest.raises(RuntimeError, match="triangle"):
        TensorProductScatter(Irreps("2x0e"), Irreps("1x1o"), Irreps("2x2e"), [(0, 0, 0, "uvu", True)])
    with pytest.raises(RuntimeError, match="l_max|l=|supported"):
        TensorProductScatter(Irreps("2x5o"), Irreps("1x0e"), Irreps("2x5o"), [(0, 0, 0, "uvu", True)])
    with pytest.raises(ValueError):
        TensorProductScatter(Irreps("2x0e"), Irreps("1x0e"), Irreps("3x0e"), [(0, 0, 0, "uvu", True)])


def test_gpu_only_modules_fail_loudly_on_cpu():
    """No CPU fallback: the kernel-backed modules raise on CPU tensors."""
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.silicon_box(reps=1, seed=0)
    data = syn.make_data(pos, types, 4.5, cell)
    model = NequIPGNNModel(r_max=4.5, type_names=names, num_layers=2, l_max=1, num_features=4, radial_mlp_width=8,
                           avg_num_neighbors=10.0)
    with pytest.raises(RuntimeError, match="GPU only|no CPU fallback"):
        model(data)


def test_model_parameter_names_and_count():
    """Parameter names follow the reference (nequip/model/param_groups.py:62-71); cfg-2/3/4 model has ~1.85 M."""
    from nequip_amd.model import NequIPGNNModel

    m = NequIPGNNModel(r_max=4.5, type_names=["H", "O"], num_layers=3, l_max=2, parity=False, num_features=64,
                       radial_mlp_depth=1, radial_mlp_width=128, avg_num_neighbors=38.0)
    names = dict(m.named_parameters())
    for k in ("layer1_convnet.conv.linear_1.weight", "layer1_convnet.conv.linear_2.weight",
              "layer1_convnet.conv.sc.weight", "layer1_convnet.conv.edge_mlp.mlp.0.weight",
              "layer1_convnet.conv.edge_mlp.mlp.2.weight", "type_embed.embed_module.weight",
              "per_atom_energy_readout.mlp_module.mlp.0.weight"):
        assert "model.func." + k in names
    assert "model.func.layer0_convnet.conv.sc.weight" not in names  # no self-connection in the first layer
    assert sum(p.numel() for p in m.parameters()) == 1_846_464
    assert names["model.func.layer1_convnet.conv.sc.weight"].numel() == 64 * 64 * 192 + 2 * 64 * 64 * 64
    assert m.nequip_custom_ops_libs == ("nequip_amd",)


def test_neighbor_list_known_answer():
    """2-atom Si primitive cell, r = 2.5 -> 8 edges (reference tests/unit/data/test_neighborlist.py:88-100)."""
    import numpy as np

    from nequip_amd.utils import synthetic as syn

    a = 5.43
    cell = np.array([[0, a / 2, a / 2], [a / 2, 0, a / 2], [a / 2, a / 2, 0]])
    pos = np.array([[0, 0, 0], [a / 4, a / 4, a / 4]])
    ei, S = syn.neighbor_list(pos, 2.5, cell)
    assert ei.shape == (2, 8) and (ei[0] == [0, 0, 0, 0, 1, 1, 1, 1]).all() and (ei[1, :4] == 1).all()
    vec = pos[ei[1]] - pos[ei[0]] + S @ cell
    assert np.allclose(np.linalg.norm(vec, axis=1), a * np.sqrt(3) / 4)


def test_modifier_registration_on_stand_in_class():
    """enable_NequipAMD attaches to a TensorProductScatter-like class the way the reference's enable_* modifiers
    do (decorated classmethod carrying the `_nequip_model_modifier_is_persistent` marker)."""
    from nequip_amd.integrations import nequip_extension as ext
    from nequip_amd.nn.model_modifier_utils import is_model_modifier

    class FakeTPS(torch.nn.Module):
        pass

    ext.register(FakeTPS)
    fn = getattr(FakeTPS, ext.MODIFIER_NAME)
    assert is_model_modifier(fn) and fn._nequip_model_modifier_is_persistent is False
    assert fn._nequip_model_modifier_unsupported_devices == ["cpu"]
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            fn(torch.nn.Sequential(FakeTPS()))


def test_force_stress_modifiers():
    """enable/disable ForceStressOutput through `modify` by name (nequip tests/unit/nn/test_grad_output.py:44-72) and the
    reference's error for an unknown modifier name."""
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.model import get_all_modifiers, modify
    from nequip_amd.nn import ForceStressOutput, GraphModel
    from nequip_amd.nn._graph_mixin import GraphModuleMixin

    class Energy(GraphModuleMixin, torch.nn.Module):
        def __init__(self):
            super().__init__()
            self._init_irreps(irreps_in={K.POSITIONS_KEY: "1o"}, irreps_out={K.TOTAL_ENERGY_KEY: "0e"})

        def forward(self, data):
            data[K.TOTAL_ENERGY_KEY] = data[K.POSITIONS_KEY].square().sum().view(1, 1)
            return data

    model = GraphModel(ForceStressOutput(func=Energy(), do_derivatives=True))
    batch = {K.POSITIONS_KEY: torch.randn(5, 3, dtype=torch.float64),
             K.EDGE_INDEX_KEY: torch.zeros(2, 0, dtype=torch.long), K.ATOM_TYPE_KEY: torch.zeros(5, dtype=torch.long)}
    out = model(dict(batch))
    assert K.FORCE_KEY in out
    torch.testing.assert_close(out[K.FORCE_KEY], -2.0 * batch[K.POSITIONS_KEY])
    assert {"enable_ForceStressOutput", "disable_ForceStressOutput"} <= set(get_all_modifiers(model))

    model = modify(model, [{"modifier": "disable_ForceStressOutput"}])
    out = model(dict(batch))
    assert K.FORCE_KEY not in out and K.TOTAL_ENERGY_KEY in out
    model = modify(model, [{"modifier": "enable_ForceStressOutput"}])
    assert K.FORCE_KEY in model(dict(batch))
    with pytest.raises(RuntimeError, match="is not a registered model modifier"):
        modify(model, [{"modifier": "enable_Nothing"}])


def test_bench_gpus_flag_is_honoured_without_a_launcher():
    """`python bench.py --gpus N` must start N ranks itself (or fail loudly) -- never run one rank and report it as N;
    a launcher-provided WORLD_SIZE that contradicts --gpus is an error too."""
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    import torch

    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
    env2 = dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env2, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


def test_model_survives_deepcopy_and_pickle():
    """nequip workflows copy and save whole modules (EMA copies, torch.save of a model): the native plan handle is
    rebuilt from its description."""
    import copy
    import io

    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.nn.interaction_block import InteractionBlock

    m = NequIPGNNModel(seed=0, model_dtype="float32", r_max=4.5, type_names=["H", "O"], num_layers=3, l_max=2,
                       parity=False, num_features=8, radial_mlp_depth=1, radial_mlp_width=16, avg_num_neighbors=10.0)
    m2 = copy.deepcopy(m)
    blocks = [x for x in m2.modules() if isinstance(x, InteractionBlock)]
    assert len(blocks) == 3
    assert m2.state_dict().keys() == m.state_dict().keys()
    assert len(list(m2.parameters())) == len(list(m.parameters()))
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    for (k, a), (_, b) in zip(m.state_dict().items(), m3.state_dict().items()):
        assert torch.equal(a, b), k


def test_owner_lists_of_the_pair_centric_backward():
    """`build_owner_csr` (host-side index arithmetic, device agnostic): the owner lists are a partition of the pairs, every
    slot carries the two directed edges of its pair with the right roles, the other-node lists index the same slots, self
    image pairs belong to their node, and the owner rule hands every node about half of its pairs."""
    import torch

    from nequip_amd.nn._topology import build_owner_csr

    g = torch.Generator().manual_seed(0)
    N = 40
    und = set()
    for _ in range(400):
        a, b = (int(v) for v in torch.randint(0, N, (2,), generator=g))
        if a != b:
            und.add((min(a, b), max(a, b)))
    und = sorted(und)
    self_pairs = [3, 17]  # an atom and its own periodic images: two directed edges i <- i with opposite shifts
    P = len(und) + len(self_pairs)
    dst = [a for a, b in und] + self_pairs + [b for a, b in und] + self_pairs
    src = [b for a, b in und] + self_pairs + [a for a, b in und] + self_pairs
    rows = torch.arange(2 * P)  # representative edge of pair p = edge p, its reverse = edge p + P
    perm = torch.randperm(2 * P, generator=g)  # edges in arbitrary order
    dst, src, rows = torch.tensor(dst)[perm], torch.tensor(src)[perm], rows[perm]

    orow, oth, prow, ein, eout, trow, tslot = (t.long() for t in build_owner_csr(dst, src, rows.to(torch.int32), P, N))
    assert orow[0] == 0 and orow[-1] == P and trow[0] == 0 and trow[-1] == P
    assert sorted(prow.tolist()) == list(range(P)) and sorted(tslot.tolist()) == list(range(P))
    owner = torch.repeat_interleave(torch.arange(N), orow[1:] - orow[:-1])
    assert torch.equal(dst[ein], owner) and torch.equal(src[ein], oth)      # edge_in:  other -> owner
    assert torch.equal(dst[eout], oth) and torch.equal(src[eout], owner)    # edge_out: owner -> other
    assert torch.equal(rows[ein] % P, prow) and torch.equal(rows[eout] % P, prow) and not torch.equal(ein, eout)
    assert torch.equal(oth[tslot], torch.repeat_interleave(torch.arange(N), trow[1:] - trow[:-1]))
    for node in self_pairs:  # owner == other
        assert int(((owner == node) & (oth == node)).sum()) == 1
    # within an owner the slots are ordered by the other node (locality of the gathered rows)
    for n in range(N):
        seg = oth[orow[n]:orow[n + 1]]
        assert torch.equal(seg, seg.sort().values)
    degree = torch.bincount(torch.cat([dst, src]), minlength=N).float() / 2  # pairs per node (self pairs count twice / 2)
    owned = (orow[1:] - orow[:-1]).float()
    assert float((owned - degree / 2).abs().max()) <= 0.3 * float(degree.max()) + 2
    assert abs(float(owned.sum()) - P) < 1e-6


def test_split_mode_workspace_sizes():
    """Workspace sizes of the radial MLP's GEMM modes and of the packed node weights (host arithmetic of the C ABI, no
    GPU): the fp16 images hold two planes plus their power-of-two exponents, the bf16 images three planes."""
    import ctypes
    import struct

    from nequip_amd import _lib
    from nequip_amd.o3._node_kernels import NodeLinearMeta

    lib = _lib.load()
    H, W = 128, 704
    ntiles = (W + 31) // 32
    assert lib.nqa_radial_mlp_workspace_bytes(_lib.NQA_MLP_BF16X6, 0, H, W) == ntiles * (H // 16) * 3 * 1024
    assert lib.nqa_radial_mlp_workspace_bytes(_lib.NQA_MLP_BF16X6, 1, H, W) == ntiles * (H // 16) * 3 * 1024
    assert lib.nqa_radial_mlp_workspace_bytes(_lib.NQA_MLP_F16X3, 0, H, W) == ntiles * (H // 16) * 2 * 1024 + 256
    assert lib.nqa_radial_mlp_workspace_bytes(_lib.NQA_MLP_F16X3, 1, H, W) == ntiles * 2 * 2 * (H // 32) * 1024 + 256
    assert lib.nqa_radial_mlp_workspace_bytes(_lib.NQA_MLP_FP32, 0, H, W) == 0
    assert lib.nqa_radial_mlp_workspace_bytes(7, 0, H, W) == -1
    meta = NodeLinearMeta(Irreps("64x0e+40x1o"), Irreps("128x0e+96x1o"), [(0, 0), (1, 1)])
    chunks, instr = meta.fwd
    cb = ctypes.create_string_buffer(b"".join(struct.pack("<8i", *c) for c in chunks))
    ib = ctypes.create_string_buffer(b"".join(struct.pack("<4i", *i) for i in instr))
    frags = (64 // 16) * (128 // 32) + ((40 + 15) // 16) * (96 // 32)  # (K blocks) x (column tiles) per instruction
    nexp = 64 // 16 + (40 + 15) // 16
    sizes = {}
    for mode in ("1", "0"):
        os.environ["NQA_NODE_F16"] = mode
        try:
            sizes[mode] = lib.nqa_node_weights_pack_bytes(ctypes.cast(cb, ctypes.c_void_p), len(chunks),
                                                          ctypes.cast(ib, ctypes.c_void_p), len(instr), 2)
        finally:
            os.environ.pop("NQA_NODE_F16", None)
    assert sizes["0"] == frags * 3 * 1024 * 2
    assert sizes["1"] == frags * 2 * 1024 * 2 + ((nexp * 2 * 4 + 255) // 256) * 256


def test_mode_switches_and_volatile_weight_copies(monkeypatch):
    """Host-side choices that the GPU tests only exercise in their default setting: the environment switches of the split
    modes, and the mark that keeps per-step copies of trainable weights out of the packed (prepass) path."""
    from nequip_amd import _lib
    from nequip_amd.nn import mlp
    from nequip_amd.o3._node_kernels import NodeLinearMeta, meta_transposed_weights

    monkeypatch.delenv("NQA_MLP_FWD_F16", raising=False)
    monkeypatch.delenv("NQA_MLP_BWD_F16", raising=False)
    assert mlp.forward_mode(_lib.NQA_MLP_BF16X6) == _lib.NQA_MLP_F16X3 == mlp.backward_mode(_lib.NQA_MLP_BF16X6)
    assert mlp.forward_mode(_lib.NQA_MLP_FP32) == _lib.NQA_MLP_FP32 == mlp.backward_mode(_lib.NQA_MLP_FP32)
    monkeypatch.setenv("NQA_MLP_FWD_F16", "0")
    monkeypatch.setenv("NQA_MLP_BWD_F16", "0")
    assert mlp.forward_mode(_lib.NQA_MLP_BF16X6) == _lib.NQA_MLP_BF16X6 == mlp.backward_mode(_lib.NQA_MLP_BF16X6)

    meta = NodeLinearMeta(Irreps("8x0e+4x1o"), Irreps("6x0e+4x1o"), [(0, 0), (1, 1)])
    wp = torch.randn(2, meta.wstride)
    wt = meta_transposed_weights(meta, wp)            # constant weights: cached on the tensor, packable
    assert meta_transposed_weights(meta, wp) is wt and not getattr(wt, "_nqa_volatile", False)
    wg = wp.clone().requires_grad_(True)
    with torch.no_grad():                              # what a backward pass without create_graph looks like
        vt = meta_transposed_weights(meta, wg)
    assert vt.requires_grad is False and vt._nqa_volatile is True
    assert torch.equal(vt, wt)


def test_skinny_product_gradients():
    """`o3.modules._skinny_mm` (the per-type contraction of the self-connection weights in training): the split-K formulation
    of the gradient w.r.t. the small operand is the gradient (first and second order), and equals torch.mm's at the real
    size."""
    from nequip_amd.o3.modules import _skinny_mm

    g = torch.Generator().manual_seed(0)
    for K in (1024, 1536, 100):
        a = torch.randn(3, 4, dtype=torch.float64, generator=g, requires_grad=True)
        b = torch.randn(4, K, dtype=torch.float64, generator=g, requires_grad=True)
        assert torch.autograd.gradcheck(_skinny_mm, (a, b))
        if K <= 200:
            assert torch.autograd.gradgradcheck(_skinny_mm, (a, b))
    a = torch.randn(5, 64, generator=g, requires_grad=True)
    b = torch.randn(64, 20480, generator=g, requires_grad=True)
    go = torch.randn(5, 20480, generator=g)
    ref = torch.autograd.grad(torch.mm(a, b), (a, b), go)
    got = torch.autograd.grad(_skinny_mm(a, b), (a, b), go)
    assert float((ref[0] - got[0]).abs().max()) < 2e-6 * float(ref[0].abs().max())
    assert torch.equal(ref[1], got[1])


def test_per_edge_type_cutoff_host_guards():
    """The reference's guards around ``per_edge_type_cutoff`` (nequip/nn/embedding/utils.py: positive cutoffs;
    nequip/nn/utils.py:121-133: ``with_edge_type_`` accepts any edge_index layout) and the early failure of the compile
    path, which has no dispatcher-op form for it."""
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.nn.embedding import EdgeLengthNormalizer, cutoff_partialdict_to_tensor
    from nequip_amd.utils.aot import aot_export_model

    with pytest.raises(AssertionError):
        cutoff_partialdict_to_tensor({"A": 0.0}, ["A", "B"], 4.0)
    with pytest.raises(AssertionError):
        cutoff_partialdict_to_tensor({"A": {"B": -1.0}}, ["A", "B"], 4.0)
    norm = EdgeLengthNormalizer(r_max=4.0, type_names=["A", "B"], per_edge_type_cutoff={"A": 3.0, "B": {"A": 3.5, "B": 2.5}})
    types = torch.tensor([0, 1, 1, 0])
    ei_t = torch.tensor([[0, 1], [1, 2], [2, 3], [3, 0], [1, 0]])  # [E, 2]: its transpose is a non-contiguous [2, E]
    vec = torch.randn(5, 3, dtype=torch.float64)
    out = norm({K.EDGE_VECTORS_KEY: vec, K.ATOM_TYPE_KEY: types.view(-1, 1), K.EDGE_INDEX_KEY: ei_t.t()})
    want = torch.tensor([[3.0, 3.0], [3.5, 2.5]], dtype=torch.float64).reciprocal()
    torch.testing.assert_close(out["_nqa_rmax_recip_edge"], want[types[ei_t[:, 0]], types[ei_t[:, 1]]])

    model = NequIPGNNModel(seed=0, model_dtype="float32", r_max=4.0, type_names=["A", "B"], num_layers=2, l_max=1,
                           num_features=8, radial_mlp_width=64, avg_num_neighbors=10.0, per_edge_type_cutoff={"A": 3.0})
    with pytest.raises(NotImplementedError, match="per_edge_type_cutoff"):
        aot_export_model(model, {}, "/tmp/x.nequip.pt2")


def test_graphed_step_and_padded_list_are_gpu_only():
    """The MD-step runner and its neighbour list have no CPU path: they say so instead of falling back."""
    import pytest
    import torch

    from nequip_amd.data._nl import PaddedNeighborList
    from nequip_amd.integrations.graphed_step import GraphedStep

    model = torch.nn.Linear(1, 1).eval()
    with pytest.raises(RuntimeError, match="GPU"):
        GraphedStep(model, torch.zeros(4, dtype=torch.long), torch.eye(3), True, 4.0)
    with pytest.raises(RuntimeError, match="GPU"):
        PaddedNeighborList(4, 4.0, torch.eye(3), True, 64)
    with pytest.raises(ValueError, match="cell"):
        PaddedNeighborList(4, 4.0, None, True, 64)


def test_deferred_parameter_gradients_block_is_inert_without_the_gpu_kernels():
    """``deferred_parameter_gradients`` only reroutes gradients that the HIP kernels produce; around a backward pass that has
    none (CPU tensors) it changes nothing, is not re-entrant-sensitive and leaves no state behind."""
    import torch

    from nequip_amd.utils import wgrad as wg

    lin = torch.nn.Linear(3, 2)
    x = torch.randn(5, 3)
    lin(x).square().sum().backward()
    ref = [p.grad.clone() for p in lin.parameters()]
    lin.zero_grad(set_to_none=True)
    with wg.deferred_parameter_gradients():
        assert wg._deferred is not None and not wg.deferring()  # (grad mode is on outside a backward node)
        with wg.deferred_parameter_gradients():  # inner block: no-op
            lin(x).square().sum().backward()
        assert wg._deferred is not None
    assert wg._deferred is None
    for p, r in zip(lin.parameters(), ref):
        assert torch.equal(p.grad, r)


def test_energy_seed_shortcut_is_decided_by_the_chain_structure():
    """the backward may be seeded at the per-atom energies only when total_energy IS their plain sum by
    construction (last module of a sequential chain = sum AtomwiseReduce of PER_ATOM_ENERGY -> TOTAL_ENERGY); any other
    ``func`` keeps the reference's autograd.grad(total_energy.sum()) (nequip/nn/grad_output.py:217-221)."""
    from nequip_amd.data import AtomicDataDict as K
    from nequip_amd.nn import AtomwiseReduce, ForceStressOutput
    from nequip_amd.nn._graph_mixin import SequentialGraphNetwork

    class Tail(torch.nn.Module):  # a module behind the reduce (e.g. a pair-potential term added to the total energy)
        def forward(self, data):
            return data

    irr = {K.PER_ATOM_ENERGY_KEY: "1x0e"}
    plain = AtomwiseReduce(irreps_in=irr, reduce="sum", field=K.PER_ATOM_ENERGY_KEY, out_field=K.TOTAL_ENERGY_KEY)
    seq = torch.nn.Sequential(torch.nn.Identity(), plain)
    fso = ForceStressOutput.__new__(ForceStressOutput)
    torch.nn.Module.__init__(fso)
    fso.func = seq
    assert fso._energy_seed_allowed()
    fso.func = torch.nn.Sequential(torch.nn.Identity(), plain, Tail())
    assert not fso._energy_seed_allowed()  # (re-decided: the chain changed)
    other = AtomwiseReduce(irreps_in={"x": "1x0e"}, reduce="sum", field="x", out_field=K.TOTAL_ENERGY_KEY)
    fso.func = torch.nn.Sequential(other)
    assert not fso._energy_seed_allowed()
    scaled = AtomwiseReduce(irreps_in=irr, reduce="sum", field=K.PER_ATOM_ENERGY_KEY, out_field=K.TOTAL_ENERGY_KEY)
    scaled.constant = 0.5  # the reference's normalised reduce (nequip/nn/atomwise.py: constant = 1/sqrt(avg_num_atoms))
    fso.func = torch.nn.Sequential(scaled)
    assert not fso._energy_seed_allowed()
    fso.func = Tail()  # not a chain at all
    assert not fso._energy_seed_allowed()


def test_deferred_parameter_gradients_refuse_hooks_and_drop_on_exceptions():
    """a deferred parameter bypasses AccumulateGrad, so gradient hooks on it are an error; a block that exits
    with an exception writes no ``.grad``."""
    from nequip_amd.utils import wgrad

    p = torch.nn.Parameter(torch.zeros(3))
    p.register_hook(lambda g: g)
    with pytest.raises(RuntimeError, match="gradient hooks"):
        wgrad._check_no_grad_hooks(p)
    q = torch.nn.Parameter(torch.zeros(3))
    q.register_post_accumulate_grad_hook(lambda t: None)
    with pytest.raises(RuntimeError, match="gradient hooks"):
        wgrad._check_no_grad_hooks(q)
    wgrad._check_no_grad_hooks(torch.nn.Parameter(torch.zeros(3)))
    with pytest.raises(ValueError):
        with wgrad.deferred_parameter_gradients():
            assert wgrad._deferred is not None
            raise ValueError("backward failed")
    assert wgrad._deferred is None  # (the block is closed, nothing was flushed)
