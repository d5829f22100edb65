"""Cost of (re)building the graph for one cfg-3 frame on the device -- what an MD step pays when positions move: neighbour
list, CSRs, reverse-edge pairing (with its verdict read-back), owner lists, slot-order rows -- next to each other."""
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
N, E = len(pos), ei.shape[1]
def topo(dst=True, src=True, pair=False, owner=False, slots=False):
    t = EdgeTopology(ei[0], ei[1], N)
    if dst: t.by_dst
    if src: t.by_src
    if pair:
        pr = t.pairing(sh)
        if owner: pr.owner_csr
        if slots: pr.slots_dst; pr.slots_src
print(f"N={N} E={E}")
print(f"neighbour list               {timeit(lambda: _compute_neighborlist_single_frame(p, 4.5, cell=c, pbc=True)):.3f} ms")
print(f"dst CSR                      {timeit(lambda: topo(True, False)):.3f} ms")
print(f"dst + src CSR                {timeit(lambda: topo()):.3f} ms")
print(f"pairing alone (with verdict) {timeit(lambda: topo(False, False, True)):.3f} ms")
print(f"dst CSR + pairing + owner    {timeit(lambda: topo(True, False, True, True)):.3f} ms")
print(f"everything (+ src CSR, slots){timeit(lambda: topo(True, True, True, True, True)):.3f} ms")
