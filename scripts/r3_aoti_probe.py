#!/usr/bin/env python3
"""Probe: can the traced energy+forces graph (dispatcher ops of the fused kernels) be packaged with AOTInductor and run
from Python on the GPU box?  (nequip/scripts/compile.py:248-344 does make_fx -> torch.export -> aoti_compile_and_package.)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from nequip_amd.data import AtomicDataDict  # noqa: E402
from nequip_amd.model import NequIPGNNModel  # noqa: E402
from nequip_amd.utils import synthetic as syn  # noqa: E402
from nequip_amd.utils.tracing import trace_model  # noqa: E402

dev = torch.device("cuda:0")
FIELDS = ("pos", "cell", "edge_index", "edge_cell_shift", "atom_types")
pos, types, cell, names = syn.water_box(n_side=3, seed=5)
data = syn.make_data(pos, types, 4.5, cell)
model = NequIPGNNModel(seed=3, model_dtype="float32", r_max=4.5, type_names=names, num_layers=3, l_max=2, parity=False,
                       num_features=64, radial_mlp_depth=1, radial_mlp_width=128,
                       avg_num_neighbors=float(data["edge_index"].shape[1] / data["pos"].shape[0]),
                       per_type_energy_scales=1.0, per_type_energy_shifts=0.0).to(dev).eval()
data = AtomicDataDict.to_device(data, dev)
inputs = {k: data[k] for k in FIELDS}
ref = model(dict(inputs))
t0 = time.time()
gm, params, buffers = trace_model(model, inputs, tracing_mode="symbolic")
for nd in list(gm.graph.nodes):  # unused lifted constants trip torch.export's lift_constants_pass
    if nd.op == "get_attr" and len(nd.users) == 0:
        gm.graph.erase_node(nd)
gm.graph.eliminate_dead_code()
gm.recompile()
print(f"traced in {time.time() - t0:.1f} s, {len(list(gm.graph.nodes))} nodes")


class Wrapped(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.gm = gm
        self._pn, self._bn = list(params), list(buffers)
        for i, k in enumerate(self._pn):
            self.register_buffer(f"p{i}", params[k].detach().clone())
        for i, k in enumerate(self._bn):
            self.register_buffer(f"b{i}", buffers[k].detach().clone())

    def forward(self, pos, cell, edge_index, edge_cell_shift, atom_types):
        p = {k: getattr(self, f"p{i}") for i, k in enumerate(self._pn)}
        b = {k: getattr(self, f"b{i}") for i, k in enumerate(self._bn)}
        out = self.gm(p, b, {"pos": pos, "cell": cell, "edge_index": edge_index,
                             "edge_cell_shift": edge_cell_shift, "atom_types": atom_types})
        return out["total_energy"], out["forces"]


w = Wrapped()
args = tuple(inputs[k] for k in FIELDS)
e, f = w(*args)
print("graph vs eager:", float((f - ref["forces"]).abs().max()))
N = torch.export.Dim("num_atoms", min=2, max=1 << 24)
E = torch.export.Dim("num_edges", min=2, max=1 << 28)
dyn = ({0: N}, None, {1: E}, {0: E}, {0: N})
t0 = time.time()
ep = torch.export.export(w, args, dynamic_shapes=dyn, strict=False)
print(f"exported in {time.time() - t0:.1f} s")
t0 = time.time()
path = torch._inductor.aoti_compile_and_package(ep, package_path="/tmp/nqa_model.pt2")
print(f"AOTI package {path} in {time.time() - t0:.1f} s, {os.path.getsize(path) / 1e6:.1f} MB")
runner = torch._inductor.aoti_load_package(path)
e2, f2 = runner(*args)
print("AOTI vs eager: dE", float((e2 - ref["total_energy"]).abs().max()), "dF", float((f2 - ref["forces"]).abs().max()))
# another size: dynamic shapes
pos, types, cell, names = syn.water_box(n_side=4, seed=7)
d2 = AtomicDataDict.to_device(syn.make_data(pos, types, 4.5, cell), dev)
r2 = model({k: d2[k] for k in FIELDS})
e3, f3 = runner(*(d2[k] for k in FIELDS))
print("AOTI vs eager (other box): dE", float((e3 - r2["total_energy"]).abs().max()), "dF", float((f3 - r2["forces"]).abs().max()))
print("AOTI_OK")
