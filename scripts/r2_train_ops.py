"""Which ATen kernels run in the training step besides ours (gradient folds, reductions)?  torch.profiler with shapes over
the train256 workload of bench.py; prints device time per (op, input shapes)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from nequip_amd.data import AtomicDataDict
from nequip_amd.model import NequIPGNNModel
from nequip_amd.train import SimpleDDPStrategy
from nequip_amd.utils import synthetic as syn

dev = torch.device("cuda:0")
w = bench.TRAIN_WORKLOADS[os.environ.get("WL", "train256")]
frames = []
for f in range(w["batch"]):
    pos, types, cell, names = syn.random_frame(w["n_atoms"], w["n_species"], seed=f)
    frames.append(syn.make_data(pos, types, 4.5, cell))
data = AtomicDataDict.to_device(AtomicDataDict.batched_from_list(frames), dev)
n_atoms, n_edges = data["pos"].shape[0], data["edge_index"].shape[1]
gen = torch.Generator().manual_seed(0)
f_target = torch.randn(n_atoms, 3, generator=gen, dtype=torch.float64).to(dev)
e_target = torch.randn(w["batch"], 1, generator=gen, dtype=torch.float64).to(dev)
model = NequIPGNNModel(seed=0, model_dtype="float32", r_max=4.5, type_names=names, num_layers=w["num_layers"],
                       l_max=w["l_max"], parity=False, num_features=w["num_features"], radial_mlp_depth=1,
                       radial_mlp_width=128, avg_num_neighbors=n_edges / n_atoms, per_type_energy_scales=1.0,
                       per_type_energy_shifts=0.0).to(dev).train()
strategy = SimpleDDPStrategy(model)
opt = torch.optim.Adam(model.parameters(), lr=1e-2)

def step():
    opt.zero_grad(set_to_none=True)
    out = model(dict(data))
    loss = (out["forces"] - f_target).square().mean() + (out["total_energy"] - e_target).square().mean()
    (loss * strategy.world_size).backward()
    strategy.post_backward(loss)
    opt.step()

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
    if t <= 0:
        continue
    key = (ev.name, str(ev.input_shapes)[:110])
    agg[key][0] += t
    agg[key][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"atoms {n_atoms} edges {n_edges}; device time per step {tot / N / 1e3:.2f} ms")
for (name, shp), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{t / N:9.1f} us/step {n / N:6.1f} calls  {name[:60]:60s} {shp}")
