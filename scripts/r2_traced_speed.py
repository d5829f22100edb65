"""Traced graph (dispatcher ops, no pairing / side streams) against the eager fused model on the cfg-3 box, both without
hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nequip_amd.data import AtomicDataDict
from nequip_amd.utils.tracing import trace_model

dev = torch.device("cuda:0")
w = bench.WORKLOADS["water10k"]
data, names = bench.build_box(w)
data = AtomicDataDict.to_device(data, dev)
n, e = data["pos"].shape[0], data["edge_index"].shape[1]
model = bench.build_model(bench.model_cfg(w, e / n), names, dev)
inputs = {k: data[k] for k in ("pos", "cell", "edge_index", "edge_cell_shift", "atom_types")}

def timeit(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3

t_eager = timeit(lambda: model(dict(inputs))["forces"])
gm, params, buffers = trace_model(model, inputs, tracing_mode="real")
ref = model(dict(inputs)); out = gm(params, buffers, inputs)
err = float((out["forces"] - ref["forces"]).abs().max())
t_graph = timeit(lambda: gm(params, buffers, inputs)["forces"])
ours = sorted({str(nd.target) for nd in gm.graph.nodes if nd.op == "call_function" and str(nd.target).startswith("nequip_amd.")})
print(f"eager (fused, paired, no hipGraph) {t_eager:.2f} ms | traced graph {t_graph:.2f} ms | max |dF| {err:.2e} | graph nodes {len(list(gm.graph.nodes))}")
print("ops in the graph:", ", ".join(o.replace('nequip_amd.', '').replace('.default', '') for o in ours))
