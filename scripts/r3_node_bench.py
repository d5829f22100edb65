#!/usr/bin/env python3
"""Per-shape timing of the node-side launches of one model (default: cfg-3 shapes, 10 125 atoms): every
`nqa_node_linear` call of an energy + forces evaluation (forward and transposed), HIP events, 50 repetitions each.
Also checks each launch against the ATen formulation.  NQA_NODE_V1=1 selects the round-2 kernel (read once per process).

    python scripts/r3_node_bench.py [--atoms N] [--lmax L] [--features F] [--types T]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from nequip_amd.o3.irreps import Irreps  # noqa: E402
from nequip_amd.o3.modules import FullyConnectedTensorProduct, Linear  # noqa: E402
from nequip_amd.nn.interaction_block import uvu_paths  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--atoms", type=int, default=10125)
ap.add_argument("--lmax", type=int, default=2)
ap.add_argument("--features", type=int, default=64)
ap.add_argument("--types", type=int, default=2)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--only", default="")
args = ap.parse_args()
dev = torch.device("cuda:0")
F, L, N, T = args.features, args.lmax, args.atoms, args.types
hidden = Irreps([(F, (l, 1 if l % 2 == 0 else -1)) for l in range(L + 1)])
sh = Irreps([(1, (l, 1 if l % 2 == 0 else -1)) for l in range(L + 1)])
emb = Irreps([(F, (0, 1))])
gates = Irreps([(F, (0, 1))] * L) if L > 0 else Irreps([])
conv_out = (Irreps([(F, (0, 1))]) + gates + Irreps(list(hidden)[1:])).simplify()


def timed(fn, reps, inner=20):
    """us per call, from replays of a hipGraph holding `inner` back-to-back calls (no host launch overhead)."""
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(inner):
                fn()
        graph.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(max(1, reps // inner * 2)):
            graph.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / (max(1, reps // inner * 2) * inner) * 1e3


torch.manual_seed(0)
types = torch.randint(0, T, (N,), device=dev)
table = torch.randn(T, F, device=dev)
total = 0.0
rows = []
for name, irr_in in (("layer0", emb), ("layer1", hidden)):
    mid, _ = uvu_paths(irr_in, sh, conv_out)
    cases = [("linear_1", Linear(irr_in, irr_in), None), ("linear_2", Linear(mid.simplify(), conv_out), None)]
    if name != "layer0":
        cases.append(("sc", FullyConnectedTensorProduct(irr_in, emb, conv_out), types))
    for cname, mod, ty in cases:
        if args.only and args.only not in f"{name}.{cname}":
            continue
        mod = mod.to(dev).eval()
        irr = mod.irreps_in if ty is None else mod.irreps_in1
        x = torch.randn(N, irr.dim, device=dev, requires_grad=True)
        f = (lambda: mod(x)) if ty is None else (lambda: mod.forward_typed(x, ty, table))
        out = f()
        ref = mod._forward_reference(x) if ty is None else mod.forward(x, table[ty])
        err = float((out - ref).detach().abs().max()) / max(1.0, float(ref.detach().abs().max()))
        g = torch.randn_like(out)
        (gx,) = torch.autograd.grad(out, x, g, retain_graph=True)
        (gr,) = torch.autograd.grad(ref, x, g)
        errb = float((gx - gr).abs().max()) / max(1.0, float(gr.abs().max()))
        from nequip_amd.o3._node_kernels import _transposed, meta_transposed_weights, node_linear
        tf = timed(f, args.reps)
        wp = mod.eval_weights(dev, torch.float32) if ty is None else mod.eval_weights_typed(table, torch.float32)
        wt, mt = meta_transposed_weights(mod._meta, wp), _transposed(mod._meta)
        tb = timed(lambda: node_linear(g, wt, ty, mt), args.reps)
        rows.append((f"{name}.{cname}", irr.dim, out.shape[1], tf, tb, err, errb))
        total += tf + tb
print(f"N={N} l_max={L} features={F} types={T}  kernel={'v1' if os.environ.get('NQA_NODE_V1', '') not in ('', '0') else 'wave'}")
for r in rows:
    print(f"{r[0]:18s} {r[1]:6d} -> {r[2]:6d}  fwd {r[3]:7.1f} us  bwd(x) {r[4]:7.1f} us  relerr {r[5]:.1e} / {r[6]:.1e}")
print(f"sum {total:.1f} us")
