"""Per-shape timing of the node kernels inside one cfg-3 energy+forces step (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nequip_amd.utils import ktimer
from nequip_amd.o3 import _node_kernels as nk
import bench
orig = nk._launch_linear
def patched(x, wp, addend, types, meta, which, scale):
    din, dout = (meta.din, meta.dout) if which == "fwd" else (meta.dout, meta.din)
    with ktimer.region(f"NL {which} {din}->{dout} T={wp.shape[0]} add={addend is not None}"):
        return orig(x, wp, addend, types, meta, which, scale)
nk._launch_linear = patched
dev = torch.device("cuda:0")
from nequip_amd.data import AtomicDataDict
w = bench.WORKLOADS["water10k"]
data_cpu, names = bench.build_box(w, seed=0)
cfg = bench.model_cfg(w, data_cpu["edge_index"].shape[1] / data_cpu["pos"].shape[0])
model = bench.build_model(cfg, names, dev)
data = AtomicDataDict.to_device(data_cpu, dev)
def step():
    out = model(dict(data))
    return out["forces"].detach()
for _ in range(3): step()
torch.cuda.synchronize()
ktimer.reset(); ktimer.enable(True)
for _ in range(3): step()
torch.cuda.synchronize(); ktimer.enable(False)
for k, v in sorted(ktimer.summary().items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"{k:50s} calls/step {v['calls']/3:.0f} avg {v['avg_ms']*1e3:.0f} us total/step {v['total_ms']/3*1e3:.0f} us")
