"""Would the l_max = 3 kernels gain from splitting a node's work over wavefronts by INPUT BLOCK?  Timing probe: the cfg-5
middle-layer structure (in 0e+1o+2e+3o) against the four single-input-block structures whose union it is, same graph
(cu20k), same multiplicity.  The single-block kernels keep a quarter of the registers (-> more wavefronts per SIMD); their
summed time is what a split kernel would take (minus the y / index loads it would share).
Needs a library built with NQA_GEN_EXTRA="p1:1x1o:3:H;p2:1x2e:3:H;p3:1x3o:3:H" (H = 1x0e+1x1o+1x2e+1x3o)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nequip_amd.nn import TensorProductScatter
from nequip_amd.nn._topology import EdgeTopology
from nequip_amd.o3 import Irreps
from nequip_amd.utils import synthetic as syn
from oracle import tp as otp

dev = torch.device("cuda:0")
mul = int(os.environ.get("MUL", 128))
reps = tuple(int(v) for v in os.environ.get("REPS", "25,25,8").split(","))
pos, types, cell, names = syn.copper_box(reps=reps, seed=0)
data = syn.make_data(pos, types, 4.5, cell)
ei = data["edge_index"].to(dev)
N, E = pos.shape[0], ei.shape[1]
topo = EdgeTopology(ei[0].contiguous(), ei[1].contiguous(), N)
H = "1x0e+1x1o+1x2e+1x3o"

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

tot = {"fwd": 0.0, "bwd_fused": 0.0, "bwd_edge": 0.0, "bwd_x": 0.0}
for f_in_1x in [H, "1x0e", "1x1o", "1x2e", "1x3o"]:
    f_in = f_in_1x.replace("1x", f"{mul}x"); f_out = H.replace("1x", f"{mul}x")
    e_at = str(Irreps.spherical_harmonics(3))
    mid, instructions = otp.build_instructions(f_in, e_at, f_out)
    mid_s = "+".join(f"{m}x{l}{'e' if p == 1 else 'o'}" for m, l, p in mid)
    tps = TensorProductScatter(Irreps(f_in), Irreps(e_at), Irreps(mid_s), instructions).to(dev)
    k = tps._get_kernels()
    assert k.has_spec(torch.float32), f_in_1x
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, k.dim_in1, device=dev, generator=g)
    y = torch.randn(E, k.dim_in2, device=dev, generator=g)
    w = torch.randn(E, k.weight_numel, device=dev, generator=g)
    go = torch.randn(N, k.dim_out, device=dev, generator=g)
    t = {"fwd": timeit(lambda: k.fwd(x, y, w, topo)), "bwd_fused": timeit(lambda: k.bwd_fused(x, y, w, go, topo)),
         "bwd_edge": timeit(lambda: k.bwd_edge(x, y, w, go, topo, True, True)), "bwd_x": timeit(lambda: k.bwd_x(y, w, go, topo))}
    print(f"in={f_in_1x:24s} paths={len(instructions):2d} dout/mul={k.dim_out // mul:3d}  " + "  ".join(f"{a} {b:7.3f} ms" for a, b in t.items()), flush=True)
    if f_in_1x != H:
        for a in tot: tot[a] += t[a]
    del x, y, w, go
print("sum of the four single-block structures:   " + "  ".join(f"{a} {b:7.3f} ms" for a, b in tot.items()))
