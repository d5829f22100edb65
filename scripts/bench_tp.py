"""Micro-benchmark of the TP-scatter kernels alone on the cfg-3 middle layer (real water-box topology)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nequip_amd.nn import TensorProductScatter
from nequip_amd.nn._topology import EdgeTopology
from nequip_amd.o3 import Irreps
from nequip_amd.utils import synthetic as syn
from oracle import tp as otp, irreps as oir
dev = torch.device("cuda:0")
pos, types, cell, names = syn.water_box(15, seed=0)
data = syn.make_data(pos, types, 4.5, cell)
ei = data["edge_index"].to(dev)
N, E = len(pos), ei.shape[1]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for f_in, f_out in [("64x0e+64x1o+64x2e", "192x0e+64x1o+64x2e"), ("64x0e", "64x0e+64x1o+64x2e"), ("64x0e+64x1o+64x2e", "64x0e")]:
    mid, instr = otp.build_instructions(f_in, "1x0e+1x1o+1x2e", f_out)
    tps = TensorProductScatter(Irreps(f_in), Irreps.spherical_harmonics(2), Irreps(oir.to_str(mid)), instr).to(dev)
    k = tps._get_kernels(); topo = EdgeTopology(ei[0], ei[1], N)
    x = torch.randn(N, k.dim_in1, device=dev); y = torch.randn(E, 9, device=dev); w = torch.randn(E, k.weight_numel, device=dev)
    g = torch.randn(N, k.dim_out, device=dev)
    topo.by_dst; topo.by_src
    res = dict(fwd=timeit(lambda: k.fwd(x, y, w, topo)), bwd_x=timeit(lambda: k.bwd_x(y, w, g, topo)),
               edge_gw=timeit(lambda: k.bwd_edge(x, y, w, g, topo, True, False)),
               edge_gy=timeit(lambda: k.bwd_edge(x, y, w, g, topo, False, True)),
               edge_both=timeit(lambda: k.bwd_edge(x, y, w, g, topo, True, True)),
               fused=timeit(lambda: k.bwd_fused(x, y, w, g, topo, True, True)))
    print(f"{f_in} -> {f_out} W={k.weight_numel}: " + " ".join(f"{a}={b:.0f}us" for a, b in res.items()), flush=True)
