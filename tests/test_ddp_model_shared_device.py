"""Model-level DDP check that runs on a ONE-GPU box: two processes share device 0, the process
group runs over gloo (RCCL refuses two ranks on one device), and the REAL data path -- batched frames -> HIP model
forward -> double backward (force-matching loss) -> ``SimpleDDPStrategy.post_backward`` flat all-reduce -> Adam -- runs
with world size 2.  Frames are sharded ``rank::2`` (the reference's ``DistributedSampler`` split), the loss is
multiplied by ``world_size`` before ``backward`` (nequip/train/lightning.py:259-266) and the gradients are averaged
(nequip/train/simple_ddp.py:26-59).  Rank 0 then repeats the step single-process on the full batch: the averaged
gradients and the parameters after one Adam step must agree.

The second parametrisation runs the backward through ``SimpleDDPStrategy.backward`` = ``with deferred_parameter_gradients():
loss.backward()``: the deferred gradients must be in ``.grad`` before ``post_backward`` reads it, on
both ranks, and the averaged gradients must still equal the single-process full batch (which uses plain autograd).

What this does NOT cover: RCCL itself with more than one rank (needs two physical GPUs; ``tests/test_ddp_rccl.py``
covers the one-rank RCCL calls, the driver's ``bench.py --gpus N`` the throughput)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from nequip_amd.model import NequIPGNNModel
from nequip_amd.data import AtomicDataDict
from nequip_amd.train import SimpleDDPStrategy
from nequip_amd.utils import synthetic as syn

NF, NA = 4, 64                      # frames, atoms per frame
frames = []
for f in range(NF):
    pos, types, cell, names = syn.random_frame(NA, 3, seed=10 + f)
    frames.append(syn.make_data(pos, types, 4.5, cell))
gen = torch.Generator().manual_seed(0)
f_t = torch.randn(NF, NA, 3, generator=gen, dtype=torch.float64)
e_t = torch.randn(NF, 1, generator=gen, dtype=torch.float64)


def build(seed):
    return NequIPGNNModel(seed=seed, model_dtype="float32", r_max=4.5, type_names=names, num_layers=3, l_max=2,
                          parity=False, num_features=64, radial_mlp_depth=1, radial_mlp_width=128,
                          avg_num_neighbors=19.0, per_type_energy_scales=1.0, per_type_energy_shifts=0.0).to(dev).train()


def loss_on(model, ids):
    data = AtomicDataDict.to_device(AtomicDataDict.batched_from_list([frames[i] for i in ids]), dev)
    out = model(data)
    ft = torch.cat([f_t[i] for i in ids]).to(dev)
    et = torch.stack([e_t[i] for i in ids]).to(dev)
    return (out["forces"] - ft).square().mean() + (out["total_energy"] - et).square().mean()


model = build(seed=100 + rank)       # different weights per rank: the strategy's broadcast must fix that
strategy = SimpleDDPStrategy(model)
assert strategy.world_size == 2
w0 = torch.cat([p.detach().view(-1) for p in model.parameters()])
both = [torch.empty_like(w0) for _ in range(world)]
dist.all_gather(both, w0)
assert torch.equal(both[0], both[1]), "parameters not broadcast"
start = {{k: v.detach().clone() for k, v in model.state_dict().items()}}

opt = torch.optim.Adam(model.parameters(), lr=1e-2)
opt.zero_grad(set_to_none=True)
loss = loss_on(model, list(range(rank, NF, world)))            # frames sharded rank::world
if {deferred!r}:
    # the shipped training step (bench.py --workload train256): parameter gradients of the HIP modules leave the data chain
    # on a side stream and land in .grad when the block exits -- BEFORE post_backward all-reduces .grad (utils/wgrad.py)
    from nequip_amd.utils import wgrad
    seen = []
    orig_flush = wgrad._flush_deferred
    wgrad._flush_deferred = lambda bucket: (seen.append(len([k for k in bucket if k != "_pending"]) + len(bucket.get("_pending", []))), orig_flush(bucket))[1]
    strategy.backward(loss * strategy.world_size)
    wgrad._flush_deferred = orig_flush
    assert seen and seen[0] > 0, "no parameter gradient took the deferred route: the test would not cover the ordering"
else:
    (loss * strategy.world_size).backward()                     # lightning.py:259-266
local = torch.cat([p.grad.detach().view(-1).clone() for p in model.parameters()])
strategy.post_backward(loss)                                    # flat all-reduce, averaged
avg = torch.cat([p.grad.detach().view(-1).clone() for p in model.parameters()])
opt.step()
torch.cuda.synchronize()
after = torch.cat([p.detach().view(-1) for p in model.parameters()])
both = [torch.empty_like(after) for _ in range(world)]
dist.all_gather(both, after)
assert torch.equal(both[0], both[1]), "ranks diverged after the optimizer step"
locs = [torch.empty_like(local) for _ in range(world)]
dist.all_gather(locs, local)
assert not torch.equal(locs[0], locs[1]), "the two shards gave identical gradients: nothing was sharded"
torch.testing.assert_close(avg, (locs[0] + locs[1]) / 2, atol=1e-7, rtol=1e-6)

if rank == 0:
    ref = build(seed=0)
    ref.load_state_dict(start)
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    ropt.zero_grad(set_to_none=True)
    full = loss_on(ref, list(range(NF)))
    (full * world).backward()
    # mean over ranks of grad(world * shard-mean loss) == grad(world * full-batch mean loss) for equal shard sizes
    g_ref = torch.cat([p.grad.detach().view(-1) for p in ref.parameters()])
    scale = float(g_ref.abs().max())
    err = float((avg - g_ref).abs().max())
    print("GRAD max|d| = %.3e of max|g| = %.3e" % (err, scale), flush=True)
    assert err <= 2e-4 * scale, "DDP-averaged gradients differ from the single-process full-batch gradients"
    ropt.step()
    torch.cuda.synchronize()
    p_ref = torch.cat([p.detach().view(-1) for p in ref.parameters()])
    # Adam's first step moves a weight by lr * g / (|g| + eps): compare the updates where the gradient is not noise
    w_start = torch.cat([start[k].view(-1) for k, _ in model.named_parameters()])
    du, du_ref = after - w_start, p_ref - w_start
    mask = g_ref.abs() > 1e-2 * scale
    worst = float((du - du_ref)[mask].abs().max())
    print("ADAM %d weights compared, worst update difference %.2e (lr 1e-2)" % (int(mask.sum()), worst), flush=True)
    assert int(mask.sum()) > 1000 and worst < 2e-4
    print("DDP_MODEL_OK", flush=True)
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.gpu
@pytest.mark.parametrize("deferred", [False, True], ids=["autograd", "deferred_parameter_gradients"])
def test_model_level_ddp_two_ranks_on_one_device(device, tmp_path, deferred):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0",
                OMP_NUM_THREADS="8")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, deferred=deferred))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(base, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{so[-2000:]}\n{se[-4000:]}"
    print(outs[0][0][-400:])
    assert "DDP_MODEL_OK" in outs[0][0]
