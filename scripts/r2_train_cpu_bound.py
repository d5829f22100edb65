"""Is the training step bound by the host (Python / launch overhead) or by the GPU?  Host time to enqueue a step against the
synchronised step time (train256 workload of bench.py)."""
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


import time
for _ in range(5):
    step()
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    step()
t_enq = (time.perf_counter() - t0) / N * 1e3
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / N * 1e3
print(f"host time to enqueue one step {t_enq:.2f} ms; synchronised step {t_all:.2f} ms")
