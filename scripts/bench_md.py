"""One molecular-dynamics-like step on the device: positions move -> device neighbour list -> CSRs -> energy + forces
(eager launches: the edge count changes from step to step, so the step is not graph-captured)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nequip_amd.data import AtomicDataDict as K
from nequip_amd.data._nl import compute_neighborlist_
dev = torch.device("cuda:0")
w = bench.WORKLOADS["water10k"]
data_cpu, names = bench.build_box(w, seed=0)
n = data_cpu["pos"].shape[0]
cfg = bench.model_cfg(w, data_cpu["edge_index"].shape[1] / n)
model = bench.build_model(cfg, names, dev)
pos0 = data_cpu[K.POSITIONS_KEY].to(dev)
types = data_cpu[K.ATOM_TYPE_KEY].to(dev)
cell = data_cpu[K.CELL_KEY].to(dev)
pbc = torch.tensor([[True, True, True]], device=dev)
gen = torch.Generator(device=dev).manual_seed(0)
def step(i):
    pos = pos0 + 0.02 * torch.randn(pos0.shape, generator=gen, device=dev, dtype=pos0.dtype)
    d = {K.POSITIONS_KEY: pos, K.ATOM_TYPE_KEY: types, K.CELL_KEY: cell, K.PBC_KEY: pbc}
    d = compute_neighborlist_(d, 4.5)
    out = model(d)
    return out[K.FORCE_KEY], d[K.EDGE_INDEX_KEY].shape[1]
for i in range(3): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
K_STEPS = 20
for i in range(K_STEPS): f, e = step(i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K_STEPS
print(f"MD-like step (NL + CSR + energy/forces, eager): {dt*1e3:.2f} ms, {n/dt:.0f} atom-steps/s, last E={e}")
