import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nequip_amd.data import AtomicDataDict
from nequip_amd.model import NequIPGNNModel
from nequip_amd.utils import synthetic as syn
dev = torch.device("cuda:0")
pos, types, cell, names = syn.water_box(n_side=int(os.environ.get("NSIDE","4")), seed=0)
data = AtomicDataDict.to_device(syn.make_data(pos, types, 4.5, cell), dev)
model = NequIPGNNModel(seed=0, model_dtype="float32", r_max=4.5, type_names=names, num_layers=3, l_max=2, parity=False, num_features=int(os.environ.get("NF","16")), radial_mlp_depth=1, radial_mlp_width=int(os.environ.get("RW","32")), avg_num_neighbors=38.0).to(dev).eval()
def step(do_forces=True):
    d = dict(data)
    model.model.do_derivatives = do_forces
    out = model(d)
    return out["total_energy"], out.get("forces")
for stage in ["energy", "forces"]:
    do_f = stage == "forces"
    for _ in range(3): step(do_f)
    torch.cuda.synchronize()
    print("stage", stage, "eager ok", flush=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step(do_f)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    print("side-stream ok", flush=True)
    with torch.cuda.graph(g):
        e, f = step(do_f)
    print("captured", flush=True)
    g.replay(); torch.cuda.synchronize()
    print("replayed", e.item(), flush=True)
