"""One molecular-dynamics-like step on the device: positions move -> device neighbour list -> CSRs / pairing -> energy + forces.

  eager   : a new (exact-size) list every step, kernels launched one by one (the edge count changes from step to step)
  graphed : nequip_amd.integrations.graphed_step.GraphedStep -- capacity-padded list, deferred pairing verdict, the whole step
            replayed as one hipGraph; the timed loop includes the per-step read of the step's flags (a synchronisation)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nequip_amd.data import AtomicDataDict as K
from nequip_amd.data._nl import compute_neighborlist_
from nequip_amd.integrations.graphed_step import GraphedStep
dev = torch.device("cuda:0")
w = bench.WORKLOADS[os.environ.get("NQA_MD_WORKLOAD", "water10k")]
data_cpu, names = bench.build_box(w, seed=0)
n = data_cpu["pos"].shape[0]
cfg = bench.model_cfg(w, data_cpu["edge_index"].shape[1] / n)
model = bench.build_model(cfg, names, dev)
pos0 = data_cpu[K.POSITIONS_KEY].to(dev)
types = data_cpu[K.ATOM_TYPE_KEY].to(dev)
cell = data_cpu[K.CELL_KEY].to(dev)
pbc = torch.tensor([[True, True, True]], device=dev)
r_max = float(cfg["r_max"])
K_STEPS = int(os.environ.get("NQA_MD_STEPS", "20"))
gen = torch.Generator(device=dev).manual_seed(0)
noise = [0.02 * torch.randn(pos0.shape, generator=gen, device=dev, dtype=pos0.dtype) for _ in range(K_STEPS)]

def eager(i):
    d = {K.POSITIONS_KEY: pos0 + noise[i], K.ATOM_TYPE_KEY: types, K.CELL_KEY: cell, K.PBC_KEY: pbc}
    d = compute_neighborlist_(d, r_max)
    out = model(d)
    return out[K.FORCE_KEY].detach(), d[K.EDGE_INDEX_KEY].shape[1]

for i in range(3): eager(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(K_STEPS): f, e = eager(i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K_STEPS
print(f"MD-like step, eager   (NL + CSR + energy/forces): {dt*1e3:.3f} ms, {n/dt:.0f} atom-steps/s, last E={e}")
f_eager = [eager(i)[0].clone() for i in range(K_STEPS)]

step = GraphedStep(model, types, cell.view(3, 3), True, r_max, headroom=float(os.environ.get("NQA_MD_HEADROOM", "1.02")))
for i in range(3): step(pos0 + noise[i])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(K_STEPS): out = step(pos0 + noise[i])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K_STEPS
print(f"MD-like step, graphed (one hipGraph, flags read every step): {dt*1e3:.3f} ms, {n/dt:.0f} atom-steps/s, "
      f"last E={step.last_num_edges} of {step.edge_capacity} slots, captures={step.num_captures}, "
      f"eager fallbacks={step.num_eager_fallbacks}")
worst = 0.0
for i in range(K_STEPS):
    worst = max(worst, float((step(pos0 + noise[i])[K.FORCE_KEY] - f_eager[i]).abs().max()))
print(f"max |F_graphed - F_eager| over {K_STEPS} steps: {worst:.3e} (force scale {float(f_eager[0].abs().max()):.3f})")
