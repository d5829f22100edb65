"""libnequip_amd_torch.so: the `torch.ops.nequip_amd.*` inference ops registered from C++ (TORCH_LIBRARY), for runtimes
that load an AOTInductor package without a Python interpreter (the reference ships its kernels to
LAMMPS that way: `nequip_custom_ops_libs`, nequip/utils/aoti_metadata.py:24-54).

CPU: the host tables the C++ side derives from the op's text arguments (node_linear chunk / instruction tables and weight
transposition, gate column tables, tensor-product plan dimensions) are byte-identical to the Python host's; the schemas
registered from C++ (in a process that never imports nequip_amd) are the Python registrations' schemas; CPU tensors raise.
GPU: every C++ op reproduces the Python-registered op bit for bit on the same inputs, and an AOTInductor package of the
whole model runs in a process where ONLY the C++ library defines the ops (Python is just the host of the test there),
and through the stand-alone C++ runner (no interpreter at all)."""
import ctypes
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
LIB = os.path.join(ROOT, "nequip_amd", "csrc", "libnequip_amd_torch.so")
RUNNER = os.path.join(ROOT, "nequip_amd", "csrc", "nequip_amd_aoti_run")

OPS = ["tp_scatter_fwd", "tp_scatter_bwd", "edge_vectors", "edge_vectors_adj", "edge_embed_fwd", "edge_embed_bwd",
       "radial_mlp_fwd", "radial_mlp_bwd", "node_linear", "gate", "gate_bwd", "radial_tp_fwd", "radial_tp_bwd",
       "node_stage_fwd", "node_stage_bwd", "energy_head_fwd", "energy_head_bwd", "force_virial"]


@pytest.fixture(autouse=True)
def _fixed_order_pair_backward(monkeypatch):
    """The tests of this module compare two evaluations BIT FOR BIT (C++ ops vs Python ops, cached vs fresh topology): the
    pair-centric backward has to sum the other node's grad_x in a fixed order for that (rows + row sum instead of the
    default accumulator of round 6, whose atomics add in arrival order).  Inherited by the child processes."""
    monkeypatch.setenv("NQA_PAIR_GX_ATOMIC", "0")


@pytest.fixture(scope="module")
def cpp():
    if not os.path.exists(LIB) or not os.path.exists(RUNNER):
        from nequip_amd.csrc import build as _build  # host C++ only (g++ against this interpreter's libtorch)

        _build.build(force=False, verbose=False)
        _build.build_torch_ops(force=False, verbose=False)
    import nequip_amd  # noqa: F401  (the Python registrations come first in this process: the C++ library must yield)

    lib = ctypes.CDLL(LIB)
    lib.nqa_torch_linear_tables.restype = ctypes.c_int
    lib.nqa_torch_linear_transpose_perm.restype = ctypes.c_int64
    lib.nqa_torch_gate_table.restype = ctypes.c_int64
    lib.nqa_torch_gate_blocks.restype = ctypes.c_int64
    lib.nqa_torch_plan_dims.restype = ctypes.c_int
    return lib


def test_library_exports_what_its_header_declares(cpp):
    import re

    header = open(os.path.join(ROOT, "include", "nequip_amd_torch.h")).read()
    declared = set(re.findall(r"\b(nqa_torch_[a-z0-9_]+)\s*\(", header))
    assert len(declared) == 9, declared
    for name in declared:
        assert hasattr(cpp, name), name
    assert cpp.nqa_torch_ops_registered_here() == 1
    # the topology-cache contract: per-evaluation reuse is the default, cross-evaluation reuse is opt-in
    if os.environ.get("NQA_TOPOLOGY_CACHE", "") == "":
        assert cpp.nqa_torch_topology_cache_mode(-1) == 1
        assert cpp.nqa_torch_topology_cache_mode(2) == 1 and cpp.nqa_torch_topology_cache_mode(-1) == 2
        assert cpp.nqa_torch_topology_cache_mode(1) == 2
    cpp.nqa_torch_topology_invalidate()
    cpp.nqa_torch_begin_evaluation()
    for op in OPS:  # every op the library registers is documented in the header with its schema
        assert re.search(rf"\b{op}\(", header), op


LINEAR_KEYS = [
    ("64x0e+64x1o+64x2e", "64x0e+64x1o+64x2e", [(0, 0), (1, 1), (2, 2)]),
    ("64x0e+64x0e+64x0e+64x1o+64x1o+64x1o+64x1o+64x2e+64x2e+64x2e+64x2e", "192x0e+64x1o+64x2e",
     [(0, 0), (1, 0), (2, 0), (3, 1), (4, 1), (5, 1), (6, 1), (7, 2), (8, 2), (9, 2), (10, 2)]),
    ("128x0e+64x1o+32x2e", "128x0e+64x1o+32x2e", [(0, 0), (1, 1), (2, 2)]),
    ("32x0e", "16x0e+8x1o", [(0, 0)]),
    ("5x0e+3x1o+3x1e+200x2e", "70x0e+130x1o+2x1e+64x2e", [(0, 0), (1, 1), (2, 2), (3, 3)]),
]


def test_linear_tables_and_weight_transposition_match_the_python_host(cpp):
    import struct

    from nequip_amd.o3._node_kernels import NodeLinearMeta, _transposed
    from nequip_amd.o3._node_ops import linear_key
    from nequip_amd.o3.irreps import Irreps

    for s_in, s_out, ins in LINEAR_KEYS:
        key = linear_key(Irreps(s_in), Irreps(s_out), ins).encode()
        meta = NodeLinearMeta(Irreps(s_in), Irreps(s_out), ins)
        for transposed, m in ((0, meta), (1, _transposed(meta))):
            chunks, instr = m.fwd
            cb = b"".join(struct.pack("<8i", *c) for c in chunks)
            ib = b"".join(struct.pack("<4i", *i) for i in instr)
            cbuf = (ctypes.c_int32 * (len(cb) // 4 + 8))()
            ibuf = (ctypes.c_int32 * (len(ib) // 4 + 8))()
            dims = (ctypes.c_int64 * 3)()
            rc = cpp.nqa_torch_linear_tables(key, transposed, cbuf, len(cbuf), ibuf, len(ibuf), dims)
            assert rc >= 0 and (rc >> 16) == len(cb) // 4 and (rc & 0xFFFF) == len(ib) // 4, (s_in, transposed)
            assert bytes(cbuf)[: len(cb)] == cb and bytes(ibuf)[: len(ib)] == ib, (s_in, transposed)
            assert list(dims) == [m.din, m.dout, m.wstride]
        perm = (ctypes.c_int64 * max(meta.wstride, 1))()
        n = cpp.nqa_torch_linear_transpose_perm(key, perm, len(perm))
        assert n == meta.wstride
        w = torch.arange(meta.wstride, dtype=torch.float32).view(1, -1)
        assert torch.equal(meta.transpose_weights(w).view(-1).long(), torch.tensor(list(perm)[:n]))


def test_gate_tables_match_the_python_host(cpp):
    from nequip_amd.o3._node_kernels import GateMeta
    from nequip_amd.o3._node_ops import _gate_meta, gate_key
    from nequip_amd.o3.irreps import Irreps

    cases = [
        ("64x0e", [("silu", 1.6790)], "128x0e", [("silu", 1.6790)], "64x1o+64x2e"),
        ("16x0e+4x0o", [("silu", 1.679), ("tanh", 1.5925)], "8x0e+8x0e+4x0o", [("silu", 1.679), ("silu", 1.679), ("tanh", 1.59)],
         "8x1o+8x2e+4x3o"),
        ("32x0e", [("silu", 1.0)], "", [], ""),
    ]
    for s_s, a_s, s_g, a_g, s_d in cases:
        key = gate_key(Irreps(s_s), a_s, Irreps(s_g), a_g, Irreps(s_d))
        meta: GateMeta = _gate_meta(key)
        for which, ref in ((0, meta._fwd), (1, meta._bwd)):
            buf = (ctypes.c_uint8 * (len(ref) + 64))()
            dims = (ctypes.c_int64 * 2)()
            n = cpp.nqa_torch_gate_table(key.encode(), which, buf, len(buf), dims)
            assert n == len(ref) and bytes(buf)[:n] == ref, (key, which)
            assert list(dims) == [meta.din, meta.dout]


def test_gate_blocks_of_the_fused_node_stage_match_the_python_host(cpp):
    """`node_stage_*` in C++ derive the `nqa_gate_block` array of `nqa_node_fused` from the gate key: byte-identical to
    `GateMeta.blocks_c()` of the Python host (cfg-3's gate, one with odd scalars / several activations, scalars only)."""
    from nequip_amd.o3._node_ops import _gate_meta, gate_key
    from nequip_amd.o3.irreps import Irreps

    cases = [
        ("64x0e", [("silu", 1.6790)], "64x0e+64x0e", [("silu", 1.6790), ("silu", 1.6790)], "64x1o+64x2e"),
        ("16x0e+4x0o", [("silu", 1.679), ("tanh", 1.5925)], "8x0e+8x0e+4x0o", [("silu", 1.679), ("silu", 1.679), ("tanh", 1.59)],
         "8x1o+8x2e+4x3o"),
        ("32x0e", [("silu", 1.0)], "", [], ""),
    ]
    for s_s, a_s, s_g, a_g, s_d in cases:
        key = gate_key(Irreps(s_s), a_s, Irreps(s_g), a_g, Irreps(s_d))
        meta = _gate_meta(key)
        arr, n = meta.blocks_c()
        ref = bytes(arr)[: n * ctypes.sizeof(arr._type_)]
        buf = (ctypes.c_uint8 * (len(ref) + 64))()
        dims = (ctypes.c_int64 * 2)()
        got = cpp.nqa_torch_gate_blocks(key.encode(), buf, len(buf), dims)
        assert got == len(ref) and bytes(buf)[:got] == ref, key
        assert list(dims) == [meta.din, meta.dout]
    bad = gate_key(Irreps("64x0e"), [("silu", 1.0)], Irreps("128x0e"), [("silu", 1.0)], Irreps("64x1o+64x2e"))
    assert cpp.nqa_torch_gate_blocks(bad.encode(), None, 0, None) == -1  # two gated irreps, one gate activation


def test_plan_dimensions_match_the_python_host(cpp):
    from nequip_amd.nn._tp_scatter_ops import _kernels, plan_dims, plan_key
    from nequip_amd.o3.irreps import Irreps
    from oracle import irreps as oir
    from oracle import tp as otp

    for f_in, lmax, f_out in (("64x0e+64x1o+64x2e", 2, "192x0e+64x1o+64x2e"), ("32x0e+32x1o+32x2e+32x3o+32x4e", 4,
                                                                              "32x0e+32x1o+32x2e+32x3o+32x4e"),
                              ("4x0e+3x1o+2x2e", 1, "0e+1o+2e")):
        sh = Irreps.spherical_harmonics(lmax)
        mid, instr = otp.build_instructions(f_in, str(sh), f_out)
        key = plan_key(Irreps(f_in), sh, Irreps(oir.to_str(mid)), instr)
        out = (ctypes.c_int64 * 7)()
        assert cpp.nqa_torch_plan_dims(key.encode(), out) == 0
        assert tuple(out[:4]) == plan_dims(key)
        k = _kernels(key, torch.device("cpu"))
        assert (bool(out[4]), bool(out[5]), bool(out[6])) == (k.out_needs_zero, k.prefer_fused_bwd, k.fused_rows_ok)


_SCHEMA_PROBE = """
import json, sys, torch
torch.ops.load_library(sys.argv[1])
assert "nequip_amd" not in sys.modules
out = {}
for name in sys.argv[2:]:
    out[name] = str(getattr(torch.ops.nequip_amd, name).default._schema)
try:
    torch.ops.nequip_amd.gate(torch.zeros(2, 3), "3x0e|silu:1.0|||")
    out["cpu_raises"] = False
except (NotImplementedError, RuntimeError) as e:
    out["cpu_raises"] = True
print(json.dumps(out))
"""


def test_cpp_registration_has_the_python_schemas_and_no_cpu_kernel(cpp):
    """In a process that never imports nequip_amd the C++ library alone defines the ops."""
    r = subprocess.run([sys.executable, "-c", _SCHEMA_PROBE, LIB] + OPS, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=""), cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got.pop("cpu_raises") is True
    for name in OPS:
        assert got[name] == str(getattr(torch.ops.nequip_amd, name).default._schema), name


# ---- GPU: bitwise agreement with the Python-registered ops, and a package run without them ---------------------------------
_OP_REPLAY = """
import sys, torch
torch.ops.load_library(sys.argv[1])
assert "nequip_amd" not in sys.modules
rec = torch.load(sys.argv[2])
bad = []
for name, args, ref in rec:
    out = getattr(torch.ops.nequip_amd, name)(*[a.cuda() if isinstance(a, torch.Tensor) else a for a in args])
    outs = list(out) if isinstance(out, (tuple, list)) else [out]
    refs = list(ref) if isinstance(ref, (tuple, list)) else [ref]
    for i, (o, r) in enumerate(zip(outs, refs)):
        if r is None:  # (an output whose contents are unspecified)
            continue
        if o.shape != r.shape or o.dtype != r.dtype or not torch.equal(o.cpu(), r):
            bad.append((name, i, tuple(o.shape), tuple(r.shape), float((o.cpu().double() - r.double()).abs().max()) if o.shape == r.shape and o.numel() else -1.0))
print("BAD", bad)
sys.exit(1 if bad else 0)
"""


@pytest.mark.gpu
def test_cpp_ops_reproduce_the_python_ops_bitwise(device, tmp_path, cpp):
    import nequip_amd  # noqa: F401
    from nequip_amd.nn._tp_scatter_ops import plan_key
    from nequip_amd.o3._node_ops import gate_key, linear_key
    from nequip_amd.o3.irreps import Irreps
    from nequip_amd.utils import synthetic as syn
    from oracle import irreps as oir
    from oracle import tp as otp

    torch.manual_seed(0)
    pos, types, cell, names = syn.water_box(n_side=3, seed=4)
    data = syn.make_data(pos, types, 4.5, cell)
    ei = data["edge_index"].to(device)
    N, E = len(pos), ei.shape[1]
    ops = torch.ops.nequip_amd
    rec = []

    def run(name, *args, unspecified=()):
        out = getattr(ops, name)(*[a.to(device) if isinstance(a, torch.Tensor) else a for a in args])
        cpu = lambda t: t.cpu()  # noqa: E731
        rec.append((name, [cpu(a) if isinstance(a, torch.Tensor) else a for a in args],
                    tuple(None if i in unspecified else cpu(o) for i, o in enumerate(out))
                    if isinstance(out, (tuple, list)) else cpu(out)))
        return out

    posd = data["pos"].double()
    shift = data["edge_cell_shift"].double()
    celld = data["cell"].double().view(1, 3, 3)
    vec = run("edge_vectors", posd, celld, ei.cpu(), shift, None)
    run("edge_vectors_adj", torch.randn(E, 3, dtype=torch.float64), ei.cpu(), shift, None, N, 1, True)
    run("edge_vectors_adj", torch.randn(E, 3, dtype=torch.float64), ei.cpu(), None, None, N, 1, False)
    bw = torch.arange(1, 9, dtype=torch.float64) * torch.pi
    cfg_sh = (2, True, False, 0, 1.0, 6.0, 1.0, True)
    cfg_emb = (0, False, True, 8, 1.0 / 4.5, 6.0, 0.31, True)
    run("edge_embed_fwd", vec.cpu(), bw, *cfg_sh)
    run("edge_embed_fwd", vec.cpu(), bw, *cfg_emb)
    run("edge_embed_bwd", vec.cpu(), bw, torch.randn(E, 9), torch.zeros(0), *cfg_sh)
    run("edge_embed_bwd", vec.cpu(), bw, torch.zeros(0), torch.randn(E, 8), *cfg_emb)
    emb = torch.randn(E, 8) * 0.5
    w0, w1 = torch.randn(8, 128), torch.randn(128, 704)
    run("radial_mlp_fwd", emb, w0, w1, 0.35, 0.125)
    run("radial_mlp_bwd", emb, w0, w1, torch.randn(E, 704), 0.35, 0.125)
    for f_in, lmax, f_out in (("64x0e+64x1o+64x2e", 2, "192x0e+64x1o+64x2e"), ("4x0e+3x1o+2x2e", 1, "5x0e+1x1o+2x2e"),
                              ("32x0e+32x1o+32x2e+32x3o+32x4e", 4, "32x0e+32x1o+32x2e+32x3o+32x4e")):
        sh = Irreps.spherical_harmonics(lmax)
        mid, instr = otp.build_instructions(f_in, str(sh), f_out)
        mid_ir = Irreps(oir.to_str(mid))
        key = plan_key(Irreps(f_in), sh, mid_ir, instr)
        wn = sum(Irreps(f_in)[i[0]].mul for i in instr)
        x, y, w = torch.randn(N, Irreps(f_in).dim), torch.randn(E, sh.dim), torch.randn(E, wn)
        run("tp_scatter_fwd", x, y, w, ei[0].cpu(), ei[1].cpu(), key)
        g = torch.randn(N, mid_ir.dim)
        run("tp_scatter_bwd", g, x, y, w, ei[0].cpu(), ei[1].cpu(), key, True, True, True)
        run("tp_scatter_bwd", g, x, y, w, ei[0].cpu(), ei[1].cpu(), key, True, False, False)
        run("tp_scatter_bwd", g, x, y, w, ei[0].cpu(), ei[1].cpu(), key, False, True, True)
    t = torch.tensor(types, dtype=torch.int64)
    for s_in, s_out, ins in LINEAR_KEYS:
        key = linear_key(Irreps(s_in), Irreps(s_out), ins)
        ws = sum(Irreps(s_in)[i].mul * Irreps(s_out)[o].mul for i, o in ins)
        x = torch.randn(N, Irreps(s_in).dim)
        run("node_linear", x, torch.randn(1, ws), None, None, key, 1.0, False)
        run("node_linear", x, torch.randn(2, ws), torch.randn(N, Irreps(s_out).dim), t, key, 0.7, False)
        run("node_linear", torch.randn(N, Irreps(s_out).dim), torch.randn(2, ws), None, t, key, 0.7, True)
    gk = gate_key(Irreps("64x0e"), [("silu", 1.679)], Irreps("64x0e+64x0e"), [("silu", 1.679), ("silu", 1.679)],
                  Irreps("64x1o+64x2e"))
    xg = torch.randn(N, 64 + 128 + 64 * 8)
    run("gate", xg, gk)
    run("gate_bwd", xg, torch.randn(N, 64 + 64 * 8), gk)
    # the fused forms of an exported graph: radial MLP + tensor product (paired: the box's list is symmetric; unpaired: the
    # same list with one edge dropped), layer boundary, readout, force / virial tail
    f_in, f_out = "64x0e+64x1o+64x2e", "192x0e+64x1o+64x2e"
    sh = Irreps.spherical_harmonics(2)
    mid, instr = otp.build_instructions(f_in, str(sh), f_out)
    mid_ir = Irreps(oir.to_str(mid))
    key = plan_key(Irreps(f_in), sh, mid_ir, instr)
    wn = sum(Irreps(f_in)[i[0]].mul for i in instr)
    w1t = torch.randn(128, wn)
    x, y, g = torch.randn(N, Irreps(f_in).dim), torch.randn(E, sh.dim), torch.randn(N, mid_ir.dim)
    lengths = vec.cpu().norm(dim=1, keepdim=True).float()
    emb_sym = torch.cat([torch.sin(lengths * k) for k in range(1, 9)], dim=1) * 0.5  # (a function of the edge length)
    shift32 = data["edge_cell_shift"].float()
    out, rows = run("radial_tp_fwd", emb_sym, x, y, w0, w1t, 0.35, 0.125, ei[0].cpu(), ei[1].cpu(), shift32, key)
    assert rows.numel() == (E // 2) * wn
    for need in ((True, True, True), (True, False, True), (False, True, False)):
        run("radial_tp_bwd", g, emb_sym, x, y, rows.cpu(), w0, w1t, 0.35, 0.125, ei[0].cpu(), ei[1].cpu(), shift32, key, *need)
    eo = ei[:, :-1].cpu().contiguous()  # an odd number of edges: no pairing
    out_o, rows_o = run("radial_tp_fwd", emb_sym[:-1], x, y[:-1], w0, w1t, 0.35, 0.125, eo[0], eo[1], shift32[:-1], key,
                        unspecified=(1,))  # (no pairing: the second result is an uninitialised placeholder)
    run("radial_tp_bwd", g, emb_sym[:-1], x, y[:-1], rows_o.cpu(), w0, w1t, 0.35, 0.125, eo[0], eo[1], shift32[:-1], key,
        True, True, True)
    lk = linear_key(Irreps("64x0e+64x1o+64x2e"), Irreps("64x0e+64x1o+64x2e"), [(0, 0), (1, 1), (2, 2)])
    sk = linear_key(Irreps("64x0e+64x1o+64x2e"), Irreps("192x0e+64x1o+64x2e"), [(0, 0), (1, 1), (2, 2)])
    wp1, wps = torch.randn(1, 3 * 64 * 64) * 0.1, torch.randn(2, 64 * 192 + 2 * 64 * 64) * 0.1
    x1, sc = run("node_stage_fwd", xg, t, wp1, wps, gk, lk, sk, 0.23)
    run("node_stage_bwd", torch.randn_like(x1.cpu()), torch.randn_like(sc.cpu()), xg, t, wp1, wps, gk, lk, sk, 0.23)
    hh, wr = torch.randn(N, 64), torch.randn(64)
    scales, shifts = torch.tensor([1.5, 0.5], dtype=torch.float64), torch.tensor([-1.0, 2.0], dtype=torch.float64)
    run("energy_head_fwd", hh, wr, scales, shifts, t, 1, 1.679)
    run("energy_head_fwd", hh, wr, None, None, t, 1, 1.679)
    run("energy_head_bwd", torch.randn(N, 1, dtype=torch.float64), hh, wr, scales, t, 1, 1.679)
    run("force_virial", torch.randn(E, 3, dtype=torch.float64), vec.cpu(), ei.cpu(), None, celld, N, 1)
    run("force_virial", torch.randn(E, 3, dtype=torch.float64), vec.cpu(), ei.cpu(), None, None, N, 1)
    path = tmp_path / "ops.pt"
    torch.save(rec, path)
    r = subprocess.run([sys.executable, "-c", _OP_REPLAY, LIB, str(path)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, PYTHONPATH=""), cwd="/tmp")
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])


_PKG_RUN = """
import sys, torch
torch.ops.load_library(sys.argv[1])
assert "nequip_amd" not in sys.modules
compiled = torch._inductor.aoti_load_package(sys.argv[2])
io = torch.load(sys.argv[3])
ok = True
for inputs, ref in io:
    out = compiled([t.cuda() for t in inputs])
    assert len(out) == len(ref)
    for name, o, r in zip(sys.argv[4:], out, ref):
        err = float((o.cpu().double() - r.double()).abs().max())
        scale = max(1.0, float(r.abs().max()))
        print(name, err, scale)
        ok = ok and err <= 2e-5 * scale
sys.exit(0 if ok else 1)
"""


@pytest.fixture(scope="module")
def exported(tmp_path_factory):
    """(package path, [(inputs, reference outputs)] for two box sizes, output keys) -- one AOTInductor compile per module."""
    return _export(torch.device("cuda:0"), tmp_path_factory.mktemp("aoti"))


def _export(device, tmp_path):
    from nequip_amd.data import AtomicDataDict
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.utils import aot
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.water_box(n_side=3, seed=5)
    data = syn.make_data(pos, types, 4.5, cell)
    model = NequIPGNNModel(seed=3, model_dtype="float32", r_max=4.5, type_names=names, num_layers=3, l_max=2,
                           parity=False, num_features=64, radial_mlp_depth=1, radial_mlp_width=128,
                           avg_num_neighbors=float(data["edge_index"].shape[1] / data["pos"].shape[0]),
                           per_type_energy_scales=1.0, per_type_energy_shifts=0.0).to(device).eval()
    path = aot.aot_export_model(model, AtomicDataDict.to_device(data, device), str(tmp_path / "water.nequip.pt2"))
    compiled, md = aot.load_aotinductor_model(path, device="cuda")
    in_keys, out_keys = md[aot.NEQUIP_AOTI_INPUTS_KEY].split(), md[aot.NEQUIP_AOTI_OUTPUTS_KEY].split()
    io = []
    for n_side, seed in ((3, 5), (4, 9)):  # the box it was compiled on and a larger one (dynamic shapes)
        p2, t2, c2, _ = syn.water_box(n_side=n_side, seed=seed)
        d2 = AtomicDataDict.to_device(syn.make_data(p2, t2, 4.5, c2), device)
        ref = model(dict(d2))
        io.append(([d2[k].cpu() for k in in_keys], [ref[k].detach().cpu() for k in out_keys]))
    return path, io, out_keys


@pytest.mark.gpu
def test_aoti_package_runs_on_the_cpp_ops_alone(device, tmp_path, exported, cpp):
    path, io, out_keys = exported
    assert {"total_energy", "forces", "virial"} <= set(out_keys), out_keys
    torch.save(io, tmp_path / "io.pt")
    r = subprocess.run([sys.executable, "-c", _PKG_RUN, LIB, path, str(tmp_path / "io.pt")] + out_keys, capture_output=True,
                       text=True, timeout=900, env=dict(os.environ, PYTHONPATH=""), cwd="/tmp")
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.gpu
def test_standalone_cpp_runner(device, tmp_path, exported, cpp):
    """No interpreter: nequip_amd_aoti_run loads the package with AOTIModelPackageLoader, the ops come from the C++ library
    it links, tensors travel as raw files (manifest: `name dtype ndim dims...`)."""
    import numpy as np

    assert os.path.exists(RUNNER), f"{RUNNER} is missing: run python -m nequip_amd.csrc.build"
    path, io, out_keys = exported
    names = {torch.float32: "f32", torch.float64: "f64", torch.int64: "i64"}
    for case, (inputs, ref) in enumerate(io):
        d = tmp_path / f"case{case}"
        d.mkdir()
        with open(d / "inputs.txt", "w") as f:
            for i, t in enumerate(inputs):
                t = t.contiguous()
                t.numpy().tofile(d / f"in{i}.bin")
                f.write(f"in{i}.bin {names[t.dtype]} {t.dim()} " + " ".join(str(s) for s in t.shape) + "\n")
        r = subprocess.run([RUNNER, path, str(d)], capture_output=True, text=True, timeout=900, cwd="/tmp",
                           env={k: v for k, v in os.environ.items() if not k.startswith("PYTHON")})
        assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
        lines = open(d / "outputs.txt").read().strip().splitlines()
        assert len(lines) == len(ref)
        for line, rt in zip(lines, ref):
            fn, dt, nd, *dims = line.split()
            arr = np.fromfile(d / fn, dtype={"f32": np.float32, "f64": np.float64, "i64": np.int64}[dt]).reshape([int(s) for s in dims])
            err = float(np.abs(arr.astype(np.float64) - rt.double().numpy()).max())
            assert err <= 2e-5 * max(1.0, float(rt.abs().max())), (fn, err)


_STALE = """
import ctypes, sys, torch
torch.ops.load_library(sys.argv[1])
assert "nequip_amd" not in sys.modules
lib = ctypes.CDLL(sys.argv[1])
hip = ctypes.CDLL("libamdhip64.so")
rec = torch.load(sys.argv[2])
ops = torch.ops.nequip_amd
x, y, w, key, ei_a, ei_b, vec, bw, cfg = [t.cuda() if isinstance(t, torch.Tensor) else t for t in rec]
buf = ei_a.clone()                      # the host's ONE persistent index buffer
def evaluate(new_evaluation):
    if new_evaluation:
        ops.edge_embed_fwd(vec, bw, *cfg)   # every exported energy graph starts with this op
    return ops.tp_scatter_fwd(x, y, w, buf[0], buf[1], key)
out_a = evaluate(True)
# refill the buffer behind torch's back (no version bump), as a Kokkos / raw-HIP host would
assert hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(ei_b.data_ptr()), ctypes.c_size_t(buf.nbytes), 3) == 0
torch.cuda.synchronize()
ref_b = ops.tp_scatter_fwd(x, y, w, ei_b[0].clone(), ei_b[1].clone(), key)
assert not torch.equal(ref_b, out_a)
mode = lib.nqa_torch_topology_cache_mode(-1)
assert mode == 1, mode
out_b = evaluate(True)                  # default: a new evaluation never sees the CSRs of the previous one
assert torch.equal(out_b, ref_b), "stale CSR reused across evaluations"
# the opt-in persistent mode keeps entries across evaluations: this is the documented hazard, and invalidate() the way out
lib.nqa_torch_topology_cache_mode(2)
out_a2 = ops.tp_scatter_fwd(x, y, w, buf[0], buf[1], key)   # (entry built for the buffer's current content = graph B)
assert hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(ei_a.data_ptr()), ctypes.c_size_t(buf.nbytes), 3) == 0
torch.cuda.synchronize()
stale = evaluate(True)
assert torch.equal(stale, ref_b), "mode 2 is expected to reuse the entry (that is its contract)"
lib.nqa_torch_topology_invalidate()
fresh = evaluate(True)
assert torch.equal(fresh, out_a)
print("ok")
"""


@pytest.mark.gpu
def test_cpp_topology_cache_is_per_evaluation_by_default(device, tmp_path, cpp):
    """a host without Python refills one persistent edge-index buffer out of band (no version bump).
    The C++ registration must not serve the previous neighbour list's CSRs to the next evaluation."""
    from nequip_amd.nn._tp_scatter_ops import plan_key
    from nequip_amd.o3.irreps import Irreps
    from nequip_amd.utils import synthetic as syn
    from oracle import irreps as oir
    from oracle import tp as otp

    torch.manual_seed(1)
    pos, types, cell, names = syn.water_box(n_side=3, seed=4)
    data = syn.make_data(pos, types, 4.5, cell)
    ei_a = data["edge_index"]
    N, E = len(pos), ei_a.shape[1]
    perm = torch.randperm(N)
    ei_b = perm[ei_a]  # another graph with the same edge count, in the same buffer
    f_in, f_out = "8x0e+8x1o+8x2e", "8x0e+8x1o+8x2e"
    sh = Irreps.spherical_harmonics(2)
    mid, instr = otp.build_instructions(f_in, str(sh), f_out)
    key = plan_key(Irreps(f_in), sh, Irreps(oir.to_str(mid)), instr)
    wn = sum(Irreps(f_in)[i[0]].mul for i in instr)
    x, y, w = torch.randn(N, Irreps(f_in).dim), torch.randn(E, sh.dim), torch.randn(E, wn)
    vec = torch.randn(E, 3, dtype=torch.float64)
    bw = torch.arange(1, 9, dtype=torch.float64) * torch.pi
    cfg = (2, True, False, 0, 1.0, 6.0, 1.0, True)
    path = tmp_path / "stale.pt"
    torch.save([x, y, w, key, ei_a, ei_b, vec, bw, cfg], path)
    env = {k: v for k, v in os.environ.items() if k not in ("NQA_TOPOLOGY_CACHE", "NQA_TOPOLOGY_VERIFY")}
    r = subprocess.run([sys.executable, "-c", _STALE, LIB, str(path)], capture_output=True, text=True, timeout=900,
                       env=dict(env, PYTHONPATH=""), cwd="/tmp")
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
