#!/usr/bin/env python3
"""Time one o3.Linear of arbitrary irreps on N atoms (hipGraph of 20 calls).  usage: r3_node_shape.py N "256x1o" "64x1o" """
import os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from nequip_amd.o3.modules import Linear
N = int(sys.argv[1]); dev = torch.device("cuda:0")
for a, b in zip(sys.argv[2::2], sys.argv[3::2]):
    mod = Linear(a, b).to(dev).eval()
    x = torch.randn(N, mod.irreps_in.dim, device=dev)
    with torch.no_grad():
        for _ in range(3): mod(x)
        torch.cuda.synchronize()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side): mod(x)
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): mod(x)
        g.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): g.replay()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 100 * 1e3
    mb = N * (mod.irreps_in.dim + mod.irreps_out.dim) * 4 / 1e6
    fl = 2.0 * N * sum(mod.irreps_in[i].mul * mod.irreps_out[o].mul * mod.irreps_in[i].ir.dim for i, o in mod.instructions) / 1e9
    print(f"{a:>24s} -> {b:<20s} {us:7.1f} us  {mb:6.1f} MB {mb / us / 1e3:5.2f} TB/s  {fl:5.2f} GFLOP {fl / us * 1e3:6.1f} TF")
