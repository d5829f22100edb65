#!/usr/bin/env python3
"""Op-level view of the cfg-4 training step (torch.profiler, grouped by input shape): which ATen ops / custom
Functions the GPU time of a force-matching step goes to.  Usage (GPU box): python scripts/train_ops.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from nequip_amd.data import AtomicDataDict  # noqa: E402
from nequip_amd.model import NequIPGNNModel  # noqa: E402
from nequip_amd.utils import synthetic as syn  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
device = torch.device("cuda:0")
w = bench.TRAIN_WORKLOADS["train256"]
frames = []
for f in range(w["batch"]):
    pos, types, cell, names = syn.random_frame(w["n_atoms"], w["n_species"], seed=f)
    frames.append(syn.make_data(pos, types, 4.5, cell))
data = AtomicDataDict.to_device(AtomicDataDict.batched_from_list(frames), device)
n_atoms, n_edges = data["pos"].shape[0], data["edge_index"].shape[1]
gen = torch.Generator().manual_seed(0)
f_target = torch.randn(n_atoms, 3, generator=gen, dtype=torch.float64).to(device)
e_target = torch.randn(w["batch"], 1, generator=gen, dtype=torch.float64).to(device)
model = NequIPGNNModel(
    seed=0, model_dtype="float32", r_max=4.5, type_names=names, num_layers=w["num_layers"], l_max=w["l_max"],
    parity=False, num_features=w["num_features"], radial_mlp_depth=1, radial_mlp_width=128,
    avg_num_neighbors=n_edges / n_atoms, per_type_energy_scales=1.0, per_type_energy_shifts=0.0,
).to(device).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-2)


def step():
    opt.zero_grad(set_to_none=True)
    out = model(dict(data))
    loss = (out["forces"] - f_target).square().mean() + (out["total_energy"] - e_target).square().mean()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
print(f"atoms {n_atoms} edges {n_edges} steps {steps}")
ka = prof.key_averages(group_by_input_shape=True)
key = "self_device_time_total" if hasattr(ka[0], "self_device_time_total") else "self_cuda_time_total"
rows = sorted(ka, key=lambda e: -getattr(e, key))
tot = sum(getattr(e, key) for e in rows)
print(f"total self device time per step: {tot / steps / 1e3:.3f} ms")
for e in rows[:70]:
    t = getattr(e, key)
    print(f"{t / steps / 1e3:8.3f} ms  n={e.count / steps:6.1f}  {e.key[:48]:48s} {str(e.input_shapes)[:110]}")
