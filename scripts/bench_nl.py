"""Timing of the device neighbour list (cfg-3 water box and a 100k-atom copper box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nequip_amd.data._nl import _compute_neighborlist_single_frame
from nequip_amd.utils import synthetic as syn
dev = torch.device("cuda:0")
for name, (pos, types, cell, names) in {"water10k": syn.water_box(15, seed=0), "cu100k": syn.copper_box((25, 25, 40), seed=0)}.items():
    p = torch.tensor(pos, device=dev); c = torch.tensor(cell, device=dev)
    for _ in range(2): ei, sh = _compute_neighborlist_single_frame(p, 4.5, cell=c, pbc=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ei, sh = _compute_neighborlist_single_frame(p, 4.5, cell=c, pbc=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    t1 = time.perf_counter(); ref_ei, _ = syn.neighbor_list(pos, 4.5, cell); t_host = time.perf_counter() - t1
    print(f"{name}: N={len(pos)} E={ei.shape[1]} device NL {dt*1e3:.2f} ms (host scipy stand-in {t_host*1e3:.0f} ms, E={ref_ei.shape[1]})", flush=True)
