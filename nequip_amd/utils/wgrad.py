"""Host side of ``nqa_wgrad`` (``nequip_amd/csrc/wgrad.hip``): parameter gradients ``dW = A^T B`` of the dense maps on the
path (ScalarMLP layers, ``o3.Linear``, the self-connection), reduced over the edge / atom rows in one split-K fp32-MFMA
launch plus one sum over the partial tiles.  Training only; first order (the result is not differentiable -- callers
that are asked for a differentiable parameter gradient, i.e. ``create_graph=True``, use their ATen formulation).

``param_grads_wanted()`` / ``inputs_only_backward()``: ``torch.autograd.grad(energy, [pos], create_graph=True)`` (the force
pass of force-matching training, ``nequip/nn/grad_output.py:216-221``) runs every custom ``Function.backward`` in full,
including parameter-gradient outputs that the engine then throws away (built-in nodes skip them through
``task_should_compute_output``; Python Functions cannot see that).  ``ForceStressOutput`` therefore wraps its
``autograd.grad`` call in ``inputs_only_backward()``, and the Functions on the path consult ``param_grads_wanted()``."""

import contextlib
import ctypes
import struct
import threading
from typing import Optional, Sequence, Tuple

import torch

from .. import _lib
from . import ktimer

# Process-wide, not thread-local: the autograd engine runs Function.backward on its own per-device worker threads, so
# a thread-local set by the caller of autograd.grad would never be seen there.  autograd.grad is synchronous, and the
# depth counter makes nesting safe; concurrent training steps from several Python threads of one process would only
# lose the optimisation's precision (a parameter gradient computed that nobody reads), never correctness, because the
# flag is only ever raised around calls that do not ask for parameter gradients -- and those who do ask for them
# (loss.backward()) must not run concurrently with such a call in the same process.
_inputs_only_depth = 0
_lock = threading.Lock()


def param_grads_wanted() -> bool:
    return _inputs_only_depth == 0


@contextlib.contextmanager
def inputs_only_backward():
    """Inside: backward passes are known to be asked for gradients of data inputs only (no parameters)."""
    global _inputs_only_depth
    with _lock:
        _inputs_only_depth += 1
    try:
        yield
    finally:
        with _lock:
            _inputs_only_depth -= 1


_eval_param_grads = False


def eval_parameter_gradients(enabled: Optional[bool] = None) -> bool:
    """Eval-mode modules (radial MLP, ``o3.Linear``, self-connection) treat their weights as constants: packed / split
    weight images are cached per parameter version and no parameter gradient is produced -- the inference fast path.  The
    reference produces parameter gradients in eval mode as well; ``eval_parameter_gradients(True)`` makes these modules use
    their training formulation whenever grad mode is on and the weights require grad.  Returns the current setting."""
    global _eval_param_grads
    if enabled is not None:
        _eval_param_grads = bool(enabled)
    return _eval_param_grads


def differentiable_parameters(module_training: bool, *params: torch.Tensor) -> bool:
    """Should a module run its parameter-differentiable formulation?  (training mode, or eval with the switch above)"""
    if module_training:
        return True
    return _eval_param_grads and torch.is_grad_enabled() and any(p.requires_grad for p in params)


class _WeightCacheMixin:
    """Modules that cache derived weights in eval mode: the cache goes when the mode changes or a state dict is loaded
    (``p.data.copy_()`` style writes bypass the version counter the cache is keyed on -- call
    ``invalidate_weight_cache()`` after such a write)."""

    _weight_cache_attrs = ("_eval_wp", "_weight_images", "_deep_images", "_head_w")

    def invalidate_weight_cache(self) -> None:
        for name in self._weight_cache_attrs:
            if name in self.__dict__:
                del self.__dict__[name]

    def train(self, mode: bool = True):
        self.invalidate_weight_cache()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_weight_cache()
        return super()._load_from_state_dict(*args, **kwargs)


class WgradTable:
    """Records ``(a_off, b_off, M, N, d, out_off)`` of one launch, kept as a host buffer (kernel arguments)."""

    def __init__(self, records: Sequence[Tuple[int, int, int, int, int, int]], out_stride: int):
        self.records = [tuple(int(v) for v in r) for r in records]
        self.out_stride = int(out_stride)
        raw = b"".join(struct.pack("<6i", *r) for r in self.records)
        self.buf = ctypes.create_string_buffer(raw, max(len(raw), 1))
        self.covered = sum(r[2] * r[3] for r in self.records) == self.out_stride
        self.flops_per_row = 2.0 * sum(r[2] * r[3] * r[4] for r in self.records)

    def __len__(self):
        return len(self.records)


def supported(a: torch.Tensor, b: torch.Tensor) -> bool:
    return a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32


def wgrad(a: torch.Tensor, b: torch.Tensor, table: WgradTable, types: Optional[torch.Tensor] = None,
          n_types: int = 1) -> torch.Tensor:
    """``out[t, out_off + i*N + j] = sum_{z: type z = t} sum_m a[z, a_off + i*d + m] b[z, b_off + j*d + m]`` ->
    ``[n_types, out_stride]`` (float32, CUDA)."""
    if not supported(a, b):
        raise RuntimeError("nqa_wgrad needs float32 CUDA operands (there is no CPU path)")
    assert a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0]
    a, b = a.contiguous(), b.contiguous()
    Z = a.shape[0]
    lib = _lib.load()
    tab = ctypes.cast(table.buf, ctypes.c_void_p)
    S = lib.nqa_wgrad_splits(tab, len(table), n_types, Z)
    if S < 1:
        _lib.check(S, "nqa_wgrad_splits")
    alloc = torch.empty if table.covered else torch.zeros
    partials = alloc((S, n_types, table.out_stride), dtype=torch.float32, device=a.device)
    tptr = ctypes.c_void_p(types.data_ptr()) if (types is not None and n_types > 1) else ctypes.c_void_p()
    nbytes = 4.0 * (a.numel() + b.numel())
    with torch.cuda.device(a.device), ktimer.region("wgrad", nbytes, table.flops_per_row * Z):
        rc = lib.nqa_wgrad(_lib.NQA_F32, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), tptr, tab,
                           len(table), a.shape[1], b.shape[1], Z, n_types, table.out_stride, S,
                           ctypes.c_void_p(partials.data_ptr()),
                           ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream))
    _lib.check(rc, "nqa_wgrad")
    return partials.sum(0) if S > 1 else partials[0]
