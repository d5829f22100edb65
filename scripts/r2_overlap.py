"""Feasibility probe: do the MFMA-bound radial-MLP kernels and the vector-ALU-bound tensor-product kernels of the cfg-3
middle layer overlap when they are launched on two HIP streams?  Prints sequential vs concurrent time per combination."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nequip_amd.data import AtomicDataDict
from nequip_amd.nn import TensorProductScatter, mlp as M
from nequip_amd.nn._topology import EdgeTopology
from nequip_amd import _lib

dev = torch.device("cuda:0")
w = bench.WORKLOADS["water10k"]
data_cpu, names = bench.build_box(w, seed=0)
cfg = bench.model_cfg(w, data_cpu["edge_index"].shape[1] / data_cpu["pos"].shape[0])
model = bench.build_model(cfg, names, dev)
data = AtomicDataDict.to_device(data_cpu, dev)
ei = data["edge_index"]
N, E = data["pos"].shape[0], ei.shape[1]
topo = EdgeTopology(ei[0].contiguous(), ei[1].contiguous(), N)
pr = topo.pairing(data["edge_cell_shift"])
P = pr.num_pairs
tps = [m for m in model.modules() if isinstance(m, TensorProductScatter)][1]
k = tps._get_kernels()
mlp = [m for m in model.modules() if isinstance(m, M.ScalarMLPFunction) and m.dims[-1] == k.weight_numel][0]
emb = (torch.randn(P, 8, device=dev) * 0.5)
assert mlp._fused_ok(emb)
cache = M._WeightImages(); cache.validate(mlp.mlp[2].weight)
margs = (emb, mlp.mlp[0].weight.detach(), mlp.mlp[2].weight.detach(), mlp._alphas[0], mlp._alphas[1])
mode = _lib.NQA_MLP_BF16X6
x = torch.randn(N, k.dim_in1, device=dev); y = torch.randn(E, k.dim_in2, device=dev)
wh = torch.randn(P, k.weight_numel, device=dev); go = torch.randn(N, k.dim_out, device=dev)
G = torch.randn(2 * P, k.weight_numel, device=dev)

ops = {
    "mlp_fwd": lambda: M._launch_fwd(*margs, mode, cache),
    "mlp_bwd": lambda: M._launch_bwd_paired(*margs, G[:P], G[P:], mode, cache),
    "tp_fwd": lambda: k.fwd(x, y, wh, topo, pr),
    "tp_bwd_fused": lambda: k.bwd_fused(x, y, wh, go, topo, pairing=pr),
    "tp_bwd_edge": lambda: k.bwd_edge(x, y, wh, go, topo, True, True, pairing=pr),
}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def timed(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

def both(f, g, order):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    first, second = (f, g) if order == 0 else (g, f)
    with torch.cuda.stream(s1): first()
    with torch.cuda.stream(s2): second()
    cur.wait_stream(s1); cur.wait_stream(s2)

single = {n: timed(f) for n, f in ops.items()}
print({n: round(v, 1) for n, v in single.items()})
for a, b in [("mlp_fwd", "tp_fwd"), ("mlp_bwd", "tp_bwd_fused"), ("mlp_bwd", "tp_fwd"), ("mlp_fwd", "tp_bwd_edge"), ("mlp_bwd", "tp_bwd_edge")]:
    for order in (0, 1):
        t = timed(lambda: both(ops[a], ops[b], order))
        print(f"{a} || {b} (order {order}): sequential {single[a] + single[b]:.0f} us, concurrent {t:.0f} us")
