"""Per-structure timing of the TP-scatter entry points on the cfg-3 water box (real topology): specialised kernels by
default, the any-irreps kernels under NQA_FORCE_GENERIC=1.

    python scripts/bench_tp_struct.py l4n_mid:32 l4n_mid_k3:32 l4n_mid_k2:32 l4n_first:224
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nequip_amd", "csrc"))
import torch
import gen_spec
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
BY_NAME = {r[0]: r for r in gen_spec.baseline_irreps()}


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for spec in sys.argv[1:]:
    name, mul = spec.split(":")
    _, f_in, lmax, f_out = BY_NAME[name]
    f_in, f_out = f_in.replace("1x", f"{mul}x"), f_out.replace("1x", f"{mul}x")
    sh = Irreps.spherical_harmonics(lmax)
    mid, instr = otp.build_instructions(f_in, str(sh), f_out)
    tps = TensorProductScatter(Irreps(f_in), sh, Irreps(oir.to_str(mid)), instr).to(dev)
    k = tps._get_kernels()
    topo = EdgeTopology(ei[0], ei[1], N)
    x = torch.randn(N, k.dim_in1, device=dev)
    y = torch.randn(E, sh.dim, device=dev)
    w = torch.randn(E, k.weight_numel, device=dev)
    g = torch.randn(N, k.dim_out, device=dev)
    topo.by_dst, topo.by_src
    res = dict(fwd=timeit(lambda: k.fwd(x, y, w, topo)), bwd_x=timeit(lambda: k.bwd_x(y, w, g, topo)),
               bwd_edge=timeit(lambda: k.bwd_edge(x, y, w, g, topo, True, True)))
    if k.has_spec(torch.float32):
        res["bwd_fused"] = timeit(lambda: k.bwd_fused(x, y, w, g, topo, True, True))
    print(f"{name} mul={mul} W={k.weight_numel} spec={k.has_spec(torch.float32)}: "
          + " ".join(f"{a}={b:.2f}ms" for a, b in res.items()), flush=True)
