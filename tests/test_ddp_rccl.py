"""The RCCL leg of SimpleDDPStrategy on hardware (the CPU suite covers world size 2 over gloo, tests/test_ddp_gloo.py).
The GPU boxes of this build have one device, so this is a one-rank process group over backend "nccl" (= RCCL on ROCm):
it proves that the group initialises on the device, that the flat gradient all-reduce with ReduceOp.AVG -- the call
`post_backward` makes on RCCL (nequip/train/simple_ddp.py:38-49) -- and the parameter broadcast run on the GPU, and that
one rank's gradients come back unchanged.  Multi-GPU throughput is the driver's `bench.py --gpus N` run."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

_WORKER = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
t0 = time.time()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda:0")
dist.all_reduce(t)
torch.cuda.synchronize()
print("INIT_OK %.1f s" % (time.time() - t0), flush=True)
from nequip_amd.model import NequIPGNNModel
from nequip_amd.data import AtomicDataDict
from nequip_amd.train import SimpleDDPStrategy
from nequip_amd.utils import synthetic as syn
dev = torch.device("cuda", 0)
pos, types, cell, names = syn.water_box(n_side=2, seed=0)
data = AtomicDataDict.to_device(syn.make_data(pos, types, 4.5, cell), dev)
model = NequIPGNNModel(seed=0, model_dtype="float32", r_max=4.5, type_names=names, num_layers=2, l_max=1, parity=False,
                       num_features=16, radial_mlp_depth=1, radial_mlp_width=32, avg_num_neighbors=20.0).to(dev).train()
strategy = SimpleDDPStrategy(model)          # broadcast over RCCL
assert strategy.world_size == 1
out = model(dict(data))
loss = out["forces"].square().mean() + out["total_energy"].square().mean()
(loss * strategy.world_size).backward()
before = torch.cat([p.grad.detach().view(-1).clone() for p in model.parameters() if p.grad is not None])
strategy.post_backward(loss)                 # flat all-reduce (AVG) over RCCL
after = torch.cat([p.grad.detach().view(-1) for p in model.parameters() if p.grad is not None])
assert before.numel() > 100 and torch.equal(before, after), "one-rank AVG all-reduce must return the gradients unchanged"
t = torch.ones(4, device=dev)
dist.all_reduce(t)
assert float(t.sum()) == 4.0
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK", dist.is_nccl_available())
"""


@pytest.mark.gpu
def test_simple_ddp_over_rccl_one_rank(device):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        r = subprocess.run([sys.executable, "-c", _WORKER.format(root=ROOT)], env=env, capture_output=True, text=True,
                           timeout=420)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL did not come up within 7 minutes on this box (communicator initialisation)")
    if "INIT_OK" not in r.stdout:
        # the box cannot create an RCCL communicator (driver / IPC set-up): nothing of this package has run yet
        pytest.skip("RCCL communicator initialisation failed on this box: " + r.stderr[-500:])
    print(r.stdout[-300:])
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
