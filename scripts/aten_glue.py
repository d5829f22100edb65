"""Which ATen ops (with shapes and calling line) still run in one cfg-3 energy+forces step (TorchDispatchMode log)."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from nequip_amd.data import AtomicDataDict
dev = torch.device("cuda:0")
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
SKIP = ("view", "reshape", "expand", "detach", "alias", "t.default", "transpose", "unsqueeze", "squeeze", "select", "slice",
        "as_strided", "empty", "_unsafe_view", "is_same_size", "split", "unbind", "permute", "_local_scalar", "stride", "size")
log = []
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in SKIP):
            shp = [tuple(a.shape) if isinstance(a, torch.Tensor) else None for a in args]
            dt = [str(a.dtype).replace("torch.", "") for a in args if isinstance(a, torch.Tensor)]
            fr = [f for f in traceback.extract_stack() if "nequip_amd" in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "(autograd engine)"
            log.append((name.replace("aten.", ""), [s for s in shp if s is not None][:3], dt[:2], where))
        return func(*args, **(kwargs or {}))
with Log():
    step()
for e in log:
    print(f"{e[0]:28s} {str(e[1]):55s} {str(e[2]):22s} {e[3]}")
print(len(log), "ops")
