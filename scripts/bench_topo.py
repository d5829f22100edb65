"""Cost of (re)building the graph for one cfg-3 frame on the device: neighbour list + the two CSRs (what an MD step pays
when positions move), next to one energy+forces evaluation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nequip_amd.data._nl import _compute_neighborlist_single_frame
from nequip_amd.nn._topology import EdgeTopology
from nequip_amd.utils import synthetic as syn
dev = torch.device("cuda:0")
pos, types, cell, names = syn.water_box(15, seed=0)
p = torch.tensor(pos, device=dev); c = torch.tensor(cell, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
ei, sh = _compute_neighborlist_single_frame(p, 4.5, cell=c, pbc=True)
def topo():
    t = EdgeTopology(ei[0], ei[1], len(pos)); t.by_dst; t.by_src
print(f"neighbour list {timeit(lambda: _compute_neighborlist_single_frame(p, 4.5, cell=c, pbc=True)):.3f} ms, "
      f"dst+src CSR {timeit(topo):.3f} ms (N={len(pos)}, E={ei.shape[1]})")
