"""Data-parallel gradient synchronisation: one flat all-reduce per optimizer step.

Mirror of ``nequip.train.SimpleDDPStrategy.post_backward`` (``nequip/train/simple_ddp.py:26-59``) without the
Lightning dependency: concatenate every parameter gradient into a single fp32 buffer, ``all_reduce`` it (``AVG`` on
RCCL -- ``backend="nccl"`` on ROCm -- ``SUM`` + divide on gloo, which lacks ``AVG``), and copy the reduced slices
back.  Frames are sharded across ranks (one process per GPU); this is the only collective on the data path
(SURVEY.md 8(e)): at ~1.85 M parameters the message is 7.4 MB, latency-bound on xGMI, so a single fused buffer is
the right shape.  The caller multiplies the loss by ``world_size`` to undo the averaging
(``nequip/train/lightning.py:259-266``).
"""

from __future__ import annotations

from typing import Iterable

import torch
import torch.distributed as dist


def all_reduce_gradients(parameters: Iterable[torch.nn.Parameter], group=None) -> None:
    """In-place average of ``.grad`` over the process group (no-op when not initialised / world size 1)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    params = [p for p in parameters if p.requires_grad and p.grad is not None]
    if not params:
        return
    flat_grads = torch.cat([p.grad.data.view(-1) for p in params])
    if dist.get_backend(group) == "gloo":
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
        flat_grads /= dist.get_world_size(group)
    else:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.AVG, group=group)
    offset = 0
    for p in params:
        numel = p.grad.numel()
        p.grad.data.copy_(flat_grads[offset : offset + numel].view_as(p.grad.data))
        offset += numel


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank ``src``'s weights (what DDP does at construction)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    # detach() shares the version counter with the parameter (".data" does not): the in-place broadcast then invalidates
    # the eval-mode weight caches of the fused modules, which are keyed on it
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.detach(), src=src, group=group)
    for m in module.modules():
        inv = getattr(m, "invalidate_weight_cache", None)
        if inv is not None:
            inv()


class SimpleDDPStrategy:
    """Framework-free stand-in for the Lightning strategy: call ``post_backward()`` after ``loss.backward()``."""

    def __init__(self, model: torch.nn.Module, group=None):
        self.model = model
        self.group = group
        broadcast_parameters(model, 0, group)

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def backward(self, loss: torch.Tensor, defer_parameter_gradients: bool = True, **kwargs) -> None:
        """``loss.backward()`` the way the shipped training step runs it (``bench.py --workload train256``): parameter
        gradients of the HIP modules leave the data chain on a side stream (``utils/wgrad.py::deferred_parameter_gradients``;
        inert on CPU tensors and with ``NQA_DEFER_PARAM_GRADS=0``) and are joined when the block exits -- BEFORE
        ``post_backward`` reads ``.grad``, which is the order the deferral requires.  Call ``post_backward()`` next."""
        from ..utils.wgrad import deferred_parameter_gradients

        if defer_parameter_gradients:
            with deferred_parameter_gradients():
                loss.backward(**kwargs)
        else:
            loss.backward(**kwargs)

    def post_backward(self, closure_loss: torch.Tensor = None) -> None:
        all_reduce_gradients(self.model.parameters(), self.group)
