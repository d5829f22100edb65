"""Multi-rank path on CPU: world_size-2 gloo test of the flat gradient all-reduce
(mirror of nequip/train/simple_ddp.py:26-59; the reference itself has no distributed test)."""

import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nequip_amd.o3.modules import FullyConnectedTensorProduct, Gate, Linear
    from nequip_amd.train import SimpleDDPStrategy

    torch.manual_seed(100 + rank)  # different initial weights per rank: broadcast must fix that
    net = torch.nn.ModuleDict({
        "lin": Linear("4x0e+4x1o", "6x0e+2x1o"),
        "sc": FullyConnectedTensorProduct("4x0e+4x1o", "3x0e", "6x0e+2x1o"),
    })
    gate = Gate("4x0e", [torch.nn.functional.silu], "2x0e", [torch.nn.functional.silu], "2x1o")
    strategy = SimpleDDPStrategy(net)
    w0 = torch.cat([p.detach().view(-1) for p in net.parameters()])
    gathered = [torch.empty_like(w0) for _ in range(world)]
    dist.all_gather(gathered, w0)
    assert all(torch.equal(gathered[0], g) for g in gathered), "weights not broadcast"

    # frames sharded across ranks: rank r gets frames {r, r + R, ...}
    g = torch.Generator().manual_seed(0)
    x_all = torch.randn(8, 16, generator=g)
    a_all = torch.randn(8, 3, generator=g)
    xs, as_ = x_all[rank::world], a_all[rank::world]
    out = gate(net["lin"](xs) + net["sc"](xs, as_))
    loss = out.square().mean() * strategy.world_size  # nequip/train/lightning.py:259-266
    loss.backward()
    strategy.post_backward(loss)
    grads = torch.cat([p.grad.view(-1) for p in net.parameters()])
    if rank == 0:
        # single-process reference on all frames
        net.zero_grad()
        out = gate(net["lin"](x_all) + net["sc"](x_all, a_all))
        (out.square().mean() * world).backward()
        ref = torch.cat([p.grad.view(-1) for p in net.parameters()])
        torch.save({"ddp": grads, "ref": ref}, os.path.join(out_dir, "grads.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_simple_ddp_gradient_allreduce_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = torch.load(os.path.join(str(tmp_path), "grads.pt"))
    # mean over ranks of (world * per-shard mean loss) gradients == gradient of world * full-batch mean loss / 1
    torch.testing.assert_close(res["ddp"], res["ref"], atol=1e-6, rtol=1e-5)
