"""Which ATen kernels run in the cfg-3 inference step besides ours?  torch.profiler with shapes, eager (no hipGraph)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from nequip_amd.data import AtomicDataDict
from nequip_amd.model import NequIPGNNModel

dev = torch.device("cuda:0")
w = bench.WORKLOADS[os.environ.get("WL", "water10k")]
data, names = bench.build_box(w)
data = AtomicDataDict.to_device(data, dev)
n_atoms, n_edges = data["pos"].shape[0], data["edge_index"].shape[1]
model = bench.build_model(bench.model_cfg(w, n_edges / n_atoms), names, dev)

def step():
    out = model(dict(data))
    return out["forces"]

for _ in range(3):
    step()
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    t = getattr(ev, "self_device_time_total", 0) or 0
    if t <= 0 or not ev.name.startswith("aten::"):
        continue
    key = (ev.name, str(ev.input_shapes)[:120])
    agg[key][0] += t
    agg[key][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"atoms {n_atoms} edges {n_edges}; ATen device time per step {tot / N:.1f} us")
for (name, shp), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{t / N:8.1f} us/step {n / N:5.1f} calls  {name[:40]:40s} {shp}")
