"""Micro-benchmark of nqa_node_linear alone (back-to-back launches) for the cfg-3 shapes and for scaled atom counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nequip_amd.o3 import Irreps
from nequip_amd.o3.modules import Linear
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
mid = "192x0e+256x1o+256x2e"
hid = "64x0e+64x1o+64x2e"
gat = "192x0e+64x1o+64x2e"
for Z in (10125, 40500):
    for a, b in [(hid, hid), (mid, gat), (gat, mid), ("64x0e", "64x0e"), ("64x0e", hid)]:
        lin = Linear(Irreps(a), Irreps(b)).to(dev).eval()
        x = torch.randn(Z, Irreps(a).dim, device=dev)
        with torch.no_grad():
            t = timeit(lambda: lin(x))
        nbytes = 4.0 * Z * (Irreps(a).dim + Irreps(b).dim)
        flops = 2.0 * Z * sum(lin.irreps_in[i].mul * lin.irreps_out[o].mul * lin.irreps_in[i].ir.dim for i, o in lin.instructions)
        print(f"Z={Z} {a} -> {b}: {t:.1f} us  {nbytes/t/1e3:.0f} GB/s  {flops/t/1e6:.1f} TF", flush=True)
